"""ctypes binding of libmgnns_hip.so (include/mgnns_hip.h).

The library is required: there is no CPU or PyTorch fallback for any operator.  `lib()`
raises if the shared object has not been built (python -m mgnns_amd.build).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGNNS_LIB") or os.path.join(_HERE, "libmgnns_hip.so")   # MGNNS_LIB: an instrumented build (tools/)
ABI_VERSION = 18

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float
_SZ = _c.c_size_t
_PP = _c.POINTER(_c.c_void_p)

# name -> argtypes, in the order of include/mgnns_hip.h
SIGNATURES = {
    "mgnns_textgcn_fwd": [_P, _I, _I, _P, _I, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P],
    "mgnns_bilstm_fwd": [_P, _P, _I, _I, _P, _I, _I, _I, _I, _PP, _PP, _PP, _PP, _P, _SZ, _P, _P, _I, _P],
    "mgnns_bilstm_bf16_fwd": [_P, _P, _I, _I, _P, _I, _I, _I, _I, _PP, _PP, _PP, _PP, _P, _SZ, _P, _P, _I, _P, _P, _P, _P],
    "mgnns_bilstm_bf16_prepack": [_PP, _PP, _I, _I, _I, _P, _P],
    "mgnns_bilstm_bf16_fold_embedding": [_P, _I, _I, _I, _P, _P, _P, _SZ, _P, _P],
    "mgnns_bilstm_bf16_table_fwd": [_P, _P, _I, _I, _P, _I, _I, _I, _I, _PP, _PP, _PP, _PP, _P, _SZ, _P, _P, _I, _P, _P, _P, _P, _P],
    "mgnns_embedding_fwd": [_P, _L, _P, _I, _I, _P, _P],
    "mgnns_gen_adj": [_P, _I, _P, _P, _P, _P, _P, _P],
    "mgnns_dense_to_csr": [_P, _I, _P, _P, _P, _P],
    "mgnns_matmul_fwd": [_P, _I, _I, _P, _I, _P, _I, _P, _SZ, _P],
    "mgnns_spmm_csr_fwd": [_P, _P, _P, _I, _P, _I, _P, _I, _P],
    "mgnns_spmm_csr_bias_fwd": [_P, _P, _P, _I, _P, _I, _P, _P, _I, _P],
    "mgnns_cast_bf16": [_P, _c.c_longlong, _P, _P],
    "mgnns_spmm_csr_bf16_fwd": [_P, _P, _P, _I, _I, _P, _I, _P, _I, _I, _I, _P, _P],
    "mgnns_spmm_tiled_bf16_fwd": [_P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P],
    "mgnns_linear_fwd": [_P, _I, _I, _P, _P, _I, _P, _P, _I, _P, _SZ, _P],
    "mgnns_imgbank_pool_fwd": [_P, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P],
    "mgnns_imgbank_pack_weights_bf16": [_P, _I, _I, _P, _P],
    "mgnns_imgbank_pool_bf16_fwd": [_P, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P],
    "mgnns_imgbank_set_form": [_I],
    "mgnns_textgcn_set_form": [_I],
    "mgnns_head_diff_fwd": [_P, _I, _I, _I, _P, _P],
    "mgnns_classifier_part_fwd": [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P],
    "mgnns_imgbank_pool_split_fwd": [_P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "mgnns_transpose_pad": [_P, _I, _I, _P, _I, _P],
    "mgnns_label_attn_core_fwd": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "mgnns_label_attn_core_masked_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "mgnns_label_gcn_fwd": [_P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _SZ, _I, _P],
    "mgnns_classifier_head_fwd": [_P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P],
    "mgnns_label_tail_bf16_fwd": [_P, _I, _I, _I, _I, _I, _PP, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P],
    "mgnns_label_tail_fwd": [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P,
                             _I, _P, _P],
    "mgnns_sq_mha_core_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "mgnns_sq_mha_pack_weights_bf16": [_P, _P, _I, _I, _I, _P, _P],
    "mgnns_cast_pad_bf16": [_P, _L, _I, _I, _P, _P],
    "mgnns_sq_mha_core_bf16_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgnns_sq_mha32_pack_weights_bf16": [_P, _P, _I, _I, _I, _P, _P],
    "mgnns_sq_mha32_plan": [_P, _I, _I, _P, _P],
    "mgnns_sq_mha32_core_bf16_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "mgnns_sq_mha_pack_weights_split": [_P, _P, _I, _I, _I, _P, _P],
    "mgnns_split_pad_bf16": [_P, _L, _I, _I, _P, _P, _P],
    "mgnns_sq_mha_core_split_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "mgnns_sq_mha_split_plan": [_P, _I, _I, _P, _P],
    "mgnns_sq_mha_layer_bf16_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _PP, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P,
                                    _I, _P, _P, _P],
    "mgnns_sq_mha_folded_fwd": [_P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _SZ, _P, _P, _P],
    "mgnns_pack_weight_f32": [_P, _I, _I, _P, _P],
    "mgnns_mha_tail_fwd": [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P, _P],
    "mgnns_pack_weight_bf16_split": [_P, _I, _I, _P, _P, _P],
    "mgnns_mha_tail_bf16_fwd": [_P, _I, _P, _I, _I, _I, _PP, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _P, _I, _P, _P, _P],
    "mgnns_mha_tail_c16_fwd": [_P, _I, _P, _I, _I, _PP, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _P, _I, _P, _P, _P],
    "mgnns_sq_mha_folded_bf16_fwd": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _I, _P, _P],
    "mgnns_transpose_cast_bf16": [_P, _I, _I, _I, _P, _P],
    "mgnns_gemm_bf16_nt_fwd": [_P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _P, _SZ, _P],
    "mgnns_gemm_bf16_set_form": [_I],
    "mgnns_gemm_bf16_pick_form": [_I, _I, _I, _I, _I],
    "mgnns_softmax_argmax_fwd": [_P, _I, _I, _P, _P, _P, _P, _P],
    "mgnns_conv_fold_bn_bf16": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P],
    "mgnns_stem_conv7_fwd": [_P, _I, _I, _I, _P, _P, _P, _P],
    "mgnns_maxpool3x3s2_nhwc_fwd": [_P, _I, _I, _I, _I, _P, _P],
    "mgnns_conv_bf16_nhwc_fwd": [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    "mgnns_label_gcn_supported": [_I, _I, _I, _I, _I],
    "mgnns_label_tail_supported": [_I, _I, _I, _I, _I, _I, _I, _I],
    "mgnns_label_tail_bf16_supported": [_I, _I, _I, _I, _I, _I, _I, _I, _I],
    "mgnns_xcd_probe": [_P],
    "mgnns_set_status_word": [_P],
    "mgnns_take_status": [],
    "mgnns_debug_slabcopy": [_P, _P, _I, _I, _I, _I, _I, _P],
    "mgnns_debug_stamp": [_P, _I, _P],
    "mgnns_debug_spin": [_I, _P, _I, _P],
    "mgnns_layernorm_fwd": [_P, _I, _I, _P, _P, _F, _P, _P],
    "mgnns_comm_unique_id": [_P, _SZ],
    "mgnns_comm_init_rank": [_I, _I, _P, _SZ, _PP],
    "mgnns_comm_init_all": [_I, _P, _PP],
    "mgnns_comm_info": [_P, _c.POINTER(_I), _c.POINTER(_I)],
    "mgnns_allgather_logits": [_P, _P, _I, _I, _P, _P],
    "mgnns_comm_group_start": [],
    "mgnns_comm_group_end": [],
    "mgnns_comm_destroy": [_P],
}

# size_t-returning helpers (buffer sizes the caller allocates)
SIZE_GETTERS = {
    "mgnns_packed_bf16_weight_bytes": [_I, _I],
    "mgnns_packed_f32_weight_bytes": [_I, _I],
    "mgnns_gemm_workspace_bytes": [],
    "mgnns_gemm_bf16_workspace_bytes": [],
    "mgnns_imgbank_packed_weight_bytes": [_I],
    "mgnns_sq_mha_packed_weight_bytes": [_I],
    "mgnns_sq_mha32_packed_weight_bytes": [_I],
    "mgnns_sq_mha32_plan_ints": [_I],
    "mgnns_sq_mha_split_packed_weight_bytes": [_I],
    "mgnns_sq_mha_folded_workspace_bytes": [_I, _I, _I],
    "mgnns_bilstm_workspace_bytes": [_I, _I, _I, _I],
    "mgnns_bilstm_bf16_prepack_bytes": [_I, _I],
    "mgnns_bilstm_bf16_table_bytes": [_I, _I],
    "mgnns_bilstm_bf16_fold_workspace_bytes": [_I],
    "mgnns_label_gcn_scratch_bytes": [_I, _I, _I],
    "mgnns_mha_tail_c16_scratch_floats": [_I, _I],
}

_lib = None


class MgnnsLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MgnnsLibraryError(
            "%s is missing: the HIP extension is mandatory (no CPU/PyTorch fallback exists). "
            "Build it with `python -m mgnns_amd.build`." % LIB_PATH)
    import torch  # noqa: F401  -- loads libamdhip64 first so the kernels share PyTorch's HIP runtime
    L = ctypes.CDLL(LIB_PATH)
    L.mgnns_last_error.restype = ctypes.c_char_p
    L.mgnns_last_error.argtypes = []
    L.mgnns_abi_version.restype = _I
    L.mgnns_abi_version.argtypes = []
    L.mgnns_source_fingerprint.restype = ctypes.c_char_p
    L.mgnns_source_fingerprint.argtypes = []
    if L.mgnns_abi_version() != ABI_VERSION:
        raise MgnnsLibraryError("libmgnns_hip.so ABI %d != binding ABI %d; rebuild" % (L.mgnns_abi_version(), ABI_VERSION))
    for name, args in SIZE_GETTERS.items():
        fn = getattr(L, name)
        fn.restype = _SZ
        fn.argtypes = args
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)      # AttributeError if the symbol is missing
        fn.restype = _I
        fn.argtypes = args
    _lib = L
    _register_status_word(L)
    return L


def source_state():
    """(fingerprint compiled into the loaded library, fingerprint of the sources next to it, equal?) -- mgnns_amd/build.py."""
    from . import build
    built, tree = lib().mgnns_source_fingerprint().decode(), build.source_fingerprint()
    return built, tree, built == tree


def check_sources(who="this measurement"):
    """Raise unless the loaded library was built from exactly the sources in the tree (an instrumented MGNNS_LIB build is named
    as such by the caller): a profile must not be filed under sources newer than the kernels that ran."""
    built, tree, same = source_state()
    if not same:
        raise MgnnsLibraryError("%s: %s was built from sources %s, the tree holds %s -- rebuild (python -m mgnns_amd.build) first"
                                % (who, LIB_PATH, built, tree))
    return built


_status_word = None


def _register_status_word(L):
    """Hand the library 4 bytes of host-pinned memory for the status of its persistent launches (include/mgnns_hip.h):
    a bounded wait that runs out is then REPORTED by the next such launch instead of being lost.  Needs a GPU."""
    global _status_word
    import torch
    if _status_word is not None or not torch.cuda.is_available():
        return
    try:
        w = torch.zeros(16, dtype=torch.int32).pin_memory()
    except RuntimeError:
        return
    L.mgnns_set_status_word(w.data_ptr())
    _status_word = w                     # keeps the allocation alive for the life of the process


def xcd_probe():
    """-> (ok, [XCC_ID seen for block indices b & 7 == k]): whether `blockIdx.x & 7` selects the XCD here (the placement several
    kernels use for SPEED; HIP promises nothing).  Synchronises the device: a set-up time measurement."""
    import ctypes
    out = (ctypes.c_int32 * 9)()
    check(lib().mgnns_xcd_probe(ctypes.addressof(out)), "mgnns_xcd_probe")
    return bool(out[0]), [int(out[1 + k]) for k in range(8)]


def take_status():
    """Read-and-clear the persistent launches' status word (0 = fine); raise if it is set.  Meaningful after a stream
    synchronise; the next persistent launch reports a raised word on its own."""
    code = lib().mgnns_take_status()
    if code:
        raise RuntimeError("a launch raised the library status word (status %d: 1 / 2 = a persistent launch gave up a bounded wait, "
                           "3 = a masked attention launch refused its plan): results since then are invalid" % code)


def check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib().mgnns_last_error().decode()))
