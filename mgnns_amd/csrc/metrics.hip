// f4 (metrics half): the evaluation tail after the logits (engine/Multi_GCN_Multihead_Att_engine.py:828-838):
//   output = softmax(logits, dim=1);  pred = output.argmax(dim=1);  accuracy / micro / macro / weighted F1(target, pred)
// On the device: one thread per sample computes the row softmax (max-subtracted, like torch) and the first arg-max,
// and adds the sample to an integer confusion matrix [NL, NL] (rows = target, columns = prediction) with atomics --
// every score the engine reports is a function of that matrix, so only NL*NL ints ever leave the GPU.
#include "common.hpp"

namespace {

constexpr int MAXNL = 64;

__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* __restrict__ logits, int B, int NL,
                                                             float* __restrict__ probs, int32_t* __restrict__ pred,
                                                             const int64_t* __restrict__ target, int32_t* __restrict__ conf) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* x = logits + (size_t)b * NL;
    float m = x[0];
    for (int j = 1; j < NL; ++j) m = fmaxf(m, x[j]);
    float e[MAXNL], z = 0.f;
    for (int j = 0; j < NL; ++j) {
        e[j] = expf(x[j] - m);
        z += e[j];
    }
    int best = 0;
    float pbest = -1.f;
    for (int j = 0; j < NL; ++j) {
        const float p = e[j] / z;
        if (probs) probs[(size_t)b * NL + j] = p;
        if (p > pbest) {          // strict: the first maximum wins, as torch.argmax on the CPU
            pbest = p;
            best = j;
        }
    }
    if (pred) pred[b] = best;
    if (conf && target) {
        const long long t = target[b];
        if (t >= 0 && t < NL) atomicAdd(conf + t * NL + best, 1);
    }
}

}  // namespace

extern "C" int mgnns_softmax_argmax_fwd(const float* logits, int B, int NL, float* probs, int32_t* pred,
                                        const int64_t* target, int32_t* confusion, mgnns_stream_t stream) {
    MG_REQUIRE(logits, "mgnns_softmax_argmax_fwd: null logits");
    MG_REQUIRE(B >= 0 && NL > 0 && NL <= MAXNL, "mgnns_softmax_argmax_fwd: NL=%d unsupported (1..%d)", NL, MAXNL);
    MG_REQUIRE(!confusion || target, "mgnns_softmax_argmax_fwd: a confusion matrix needs the targets");
    if (B == 0) return 0;
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, B, NL, probs,
                       pred, target, confusion);
    MG_CHECK_LAUNCH("mgnns_softmax_argmax_fwd");
    return 0;
}
