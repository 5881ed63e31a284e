"""The evaluation tail of the reference engine (engine/Multi_GCN_Multihead_Att_engine.py:828-838) on the device:
softmax + arg-max of the logits and an integer confusion matrix accumulated with atomics; accuracy and the micro /
macro / weighted F1 the engine reports (sklearn.metrics.f1_score semantics) are computed from that matrix on the host,
so per batch nothing but NL*NL integers has to be read back -- and only when a score is asked for."""
import numpy as np
import torch

from . import _lib, ops


def predict(logits, target=None, confusion=None, want_probs=True):
    """logits [B, NL] fp32 on the GPU -> (probs [B, NL] or None, pred int32 [B]); with `target` (int64 [B]) and
    `confusion` (int32 [NL, NL]) the batch is also counted into the confusion matrix (rows = target)."""
    ops._chk(logits, "logits", ndim=2)
    B, NL = logits.shape
    probs = torch.empty_like(logits) if want_probs else None
    pred = torch.empty(B, dtype=torch.int32, device=logits.device)
    if (target is None) != (confusion is None):
        raise ValueError("pass target and confusion together")
    if target is not None:
        ops._chk(target, "target", torch.int64, 1)
        ops._chk(confusion, "confusion", torch.int32, 2)
        if target.shape[0] != B or confusion.shape != (NL, NL):
            raise ValueError("target %s / confusion %s do not match logits %s" % (tuple(target.shape), tuple(confusion.shape), (B, NL)))
    L = _lib.lib()
    _lib.check(L.mgnns_softmax_argmax_fwd(ops._p(logits), B, NL, ops._p(probs), ops._p(pred), ops._p(target), ops._p(confusion),
                                          ops._stream()), "mgnns_softmax_argmax_fwd")
    return probs, pred


def scores_from_confusion(conf):
    """accuracy, micro / macro / weighted F1 from a confusion matrix (rows = target, columns = prediction), with the
    conventions of sklearn.metrics.f1_score: labels = classes that occur in the targets or the predictions; a class
    without predictions and targets is left out, a class with zero precision + recall scores 0."""
    c = np.asarray(conf, dtype=np.float64)
    n = c.sum()
    tp = np.diag(c)
    support, predicted = c.sum(axis=1), c.sum(axis=0)
    present = (support + predicted) > 0
    denom = support + predicted                       # 2 tp + fp + fn
    f1 = np.where(denom > 0, 2.0 * tp / np.where(denom > 0, denom, 1.0), 0.0)
    acc = float(tp.sum() / n) if n else 0.0
    micro = float(2.0 * tp.sum() / denom.sum()) if denom.sum() else 0.0
    macro = float(f1[present].mean()) if present.any() else 0.0
    weighted = float((f1 * support).sum() / support.sum()) if support.sum() else 0.0
    return {"acc": acc, "micro_f1": micro, "macro_f1": macro, "weighted_f1": weighted}


class Metrics:
    """Accumulates the confusion matrix of an evaluation pass on the GPU; `update` launches one small kernel and never
    synchronises, `result` reads NL*NL integers back."""

    def __init__(self, num_labels, device):
        self.conf = torch.zeros(num_labels, num_labels, dtype=torch.int32, device=device)

    def reset(self):
        self.conf.zero_()

    def update(self, logits, target, want_probs=False):
        return predict(logits, target.to(device=logits.device, dtype=torch.int64).contiguous(), self.conf, want_probs)

    def result(self):
        return scores_from_confusion(self.conf.cpu().numpy())
