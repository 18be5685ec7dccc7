"""ctypes binding of the C ABI declared in include/lvdhip.h.

The library is built in-tree (``llm-groundedvideodiffusion_amd/liblvdhip.so``) by
``__graft_entry__.build()`` / ``make -C llm-groundedvideodiffusion_amd/csrc``.  There is NO
fallback: if the shared object is missing, or a kernel reports an error, a RuntimeError is raised
(the reference's generate.py:343-348 error containment keeps working on RuntimeError).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LVD_LIB", os.path.join(_HERE, "liblvdhip.so"))  # LVD_LIB: developer A/B of two builds of the library

c_bf16_p = C.c_void_p
c_f32_p = C.c_void_p
c_i32_p = C.c_void_p
i32 = C.c_int32
f32 = C.c_float


class GemmParams(C.Structure):
    _fields_ = [
        ("a1", C.c_void_p), ("a2", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("res", C.c_void_p), ("out", C.c_void_p),
        ("M", i32), ("N", i32), ("K", i32),
        ("lda1", i32), ("lda2", i32), ("c1", i32), ("cin", i32), ("mode", i32),
        ("hin", i32), ("win", i32), ("hout", i32), ("wout", i32), ("stride", i32), ("upsample", i32),
        ("frames", i32), ("hw", i32), ("rows_per_sample", i32), ("ldres", i32), ("ldc", i32),
        ("act", i32), ("out_fp32", i32), ("alpha", f32), ("accumulate", i32), ("variant", i32),
        ("ksplit", i32), ("ws", C.c_void_p), ("ws_bytes", C.c_int64), ("m_begin", i32), ("ldrowbias", i32),
        ("ln_mean_rstd", C.c_void_p), ("ln_colsum", C.c_void_p),
    ]


class GnStatsParams(C.Structure):
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p), ("ld1", i32), ("ld2", i32), ("c1", i32), ("c", i32),
        ("rows", i32), ("rows_per_sample", i32), ("groups", i32), ("eps", f32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("partial", C.c_void_p), ("chunks", i32),
        ("scale_shift", C.c_void_p), ("mean_rstd", C.c_void_p),
    ]


class GnApplyParams(C.Structure):
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p), ("ld1", i32), ("ld2", i32), ("c1", i32), ("c", i32),
        ("rows", i32), ("rows_per_sample", i32), ("partial", C.c_void_p), ("silu", i32),
        ("y", C.c_void_p), ("ldy", i32), ("chunks", i32), ("groups", i32), ("eps", f32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean_rstd", C.c_void_p),
    ]


class GnBwdStatsParams(C.Structure):
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p), ("ld1", i32), ("ld2", i32), ("c1", i32), ("c", i32),
        ("dy", C.c_void_p), ("lddy", i32),
        ("rows", i32), ("rows_per_sample", i32), ("groups", i32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean_rstd", C.c_void_p),
        ("partial", C.c_void_p), ("chunks", i32), ("gsum", C.c_void_p), ("silu", i32),
    ]


class GnBwdApplyParams(C.Structure):
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p), ("ld1", i32), ("ld2", i32), ("c1", i32), ("c", i32),
        ("dy", C.c_void_p), ("lddy", i32),
        ("rows", i32), ("rows_per_sample", i32), ("groups", i32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean_rstd", C.c_void_p), ("partial", C.c_void_p),
        ("silu", i32),
        ("dx1", C.c_void_p), ("dx2", C.c_void_p), ("lddx1", i32), ("lddx2", i32), ("accumulate", i32), ("chunks", i32),
    ]


class LnParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", i32), ("rows", i32), ("c", i32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", f32),
        ("y", C.c_void_p), ("ldy", i32), ("mean_rstd", C.c_void_p),
    ]


class LnBwdParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", i32), ("dy", C.c_void_p), ("lddy", i32), ("rows", i32), ("c", i32),
        ("gamma", C.c_void_p), ("mean_rstd", C.c_void_p), ("dx", C.c_void_p), ("lddx", i32), ("accumulate", i32),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", i32), ("k", C.c_void_p), ("ldk", i32), ("v", C.c_void_p), ("ldv", i32),
        ("k2", C.c_void_p), ("v2", C.c_void_p), ("ldk2", i32), ("ldv2", i32),
        ("o", C.c_void_p), ("ldo", i32), ("lse", C.c_void_p),
        ("samples", i32), ("heads", i32), ("sq", i32), ("skv", i32), ("skv2", i32),
        ("q_ninner", i32), ("q_os", i32), ("q_is", i32), ("q_step", i32),
        ("kv_ninner", i32), ("kv_os", i32), ("kv_is", i32), ("kv_step", i32),
        ("kv2_ninner", i32), ("kv2_os", i32), ("kv2_is", i32), ("kv2_step", i32),
        ("scale", f32), ("causal", i32),
    ]


class AttnBwdParams(C.Structure):
    _fields_ = [
        ("f", AttnParams),
        ("d_o", C.c_void_p), ("lddo", i32), ("dq", C.c_void_p), ("lddq", i32),
        ("dk", C.c_void_p), ("lddk", i32), ("dv", C.c_void_p), ("lddv", i32), ("delta", C.c_void_p),
    ]


class CaProbsParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", i32), ("k", C.c_void_p), ("ldk", i32),
        ("frames", i32), ("heads", i32), ("P", i32), ("ntext", i32), ("scale", f32),
        ("tok_ids", C.c_void_p), ("ntok", i32), ("probs", C.c_void_p), ("lse", C.c_void_p),
    ]


class CaProbsFullParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", i32), ("k", C.c_void_p), ("ldk", i32),
        ("samples", i32), ("heads", i32), ("P", i32), ("ntext", i32), ("samples_per_key", i32), ("scale", f32),
        ("probs", C.c_void_p), ("key_bias", C.c_void_p), ("ld_key_bias", i32),
    ]


class CaApplyProbsParams(C.Structure):
    _fields_ = [
        ("probs", C.c_void_p), ("v", C.c_void_p), ("ldv", i32),
        ("samples", i32), ("heads", i32), ("P", i32), ("ntext", i32), ("samples_per_key", i32),
        ("out", C.c_void_p), ("ldo", i32),
    ]


class CaSelectParams(C.Structure):
    _fields_ = [
        ("probs", C.c_void_p), ("dprobs", C.c_void_p),
        ("frames", i32), ("heads", i32), ("P", i32), ("ntok", i32), ("H", i32), ("W", i32),
        ("tok_obj", C.c_void_p), ("boxes", C.c_void_p), ("tok_weight", C.c_void_p), ("nobj", i32),
        ("fg_weight", f32), ("bg_weight", f32), ("com_loss_scale", f32),
        ("grad_scale", f32), ("loss_partial", C.c_void_p), ("com_ws", C.c_void_p),
        ("use_ratio_loss", i32), ("ratio_eps", f32), ("attn_sync_weight", f32), ("boxdiff_loss_scale", f32),
        ("boxdiff_normed", i32), ("boxdiff_L", i32),
    ]


class CaDqParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", i32), ("k", C.c_void_p), ("ldk", i32),
        ("frames", i32), ("heads", i32), ("P", i32), ("ntext", i32), ("scale", f32),
        ("tok_ids", C.c_void_p), ("ntok", i32),
        ("probs", C.c_void_p), ("dprobs", C.c_void_p), ("lse", C.c_void_p),
        ("dq", C.c_void_p), ("lddq", i32),
        ("acc32", C.c_void_p), ("ldacc", i32), ("acc_mode", i32),
    ]


# name -> (argtypes) for every symbol include/lvdhip.h declares; restype is int unless noted
_P = C.POINTER
SYMBOLS = {
    "lvdhip_last_error": None,
    "lvdhip_version": [],
    "lvdhip_gemm": [_P(GemmParams), C.c_void_p],
    "lvdhip_groupnorm_stats": [_P(GnStatsParams), C.c_void_p],
    "lvdhip_groupnorm_apply": [_P(GnApplyParams), C.c_void_p],
    "lvdhip_groupnorm_bwd_stats": [_P(GnBwdStatsParams), C.c_void_p],
    "lvdhip_groupnorm_bwd_apply": [_P(GnBwdApplyParams), C.c_void_p],
    "lvdhip_groupnorm_fused": [_P(GnStatsParams), _P(GnApplyParams), C.c_void_p],
    "lvdhip_groupnorm_bwd_fused": [_P(GnBwdApplyParams), C.c_void_p],
    "lvdhip_groupnorm_slab_loads": [C.c_int32, C.c_int32, C.c_int32, C.c_int32],
    "lvdhip_groupnorm_slab": [_P(GnStatsParams), _P(GnApplyParams), C.c_void_p],
    "lvdhip_groupnorm_bwd_slab_loads": [C.c_int32, C.c_int32, C.c_int32, C.c_int32],
    "lvdhip_groupnorm_bwd_slab": [_P(GnBwdApplyParams), C.c_void_p],
    "lvdhip_layernorm": [_P(LnParams), C.c_void_p],
    "lvdhip_layernorm_bwd": [_P(LnBwdParams), C.c_void_p],
    "lvdhip_attention_fwd": [_P(AttnParams), C.c_void_p],
    "lvdhip_attention_bwd": [_P(AttnBwdParams), C.c_void_p],
    "lvdhip_ca_probs": [_P(CaProbsParams), C.c_void_p],
    "lvdhip_ca_probs_full": [_P(CaProbsFullParams), C.c_void_p],
    "lvdhip_ca_apply_probs": [_P(CaApplyProbsParams), C.c_void_p],
    "lvdhip_ca_map_smooth": [C.c_void_p, C.c_void_p, C.c_int64, i32, i32, C.c_void_p, i32, C.c_void_p],
    "lvdhip_ca_map_renorm": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, i32, i32, i32, i32, f32, i32, C.c_void_p],
    "lvdhip_ca_map_gather_cols": [C.c_void_p, C.c_void_p, i32, C.c_void_p, C.c_int64, i32, i32, C.c_void_p],
    "lvdhip_ca_map_scatter_cols": [C.c_void_p, C.c_void_p, i32, C.c_void_p, C.c_int64, i32, i32, C.c_void_p],
    "lvdhip_ca_map_softmax_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, i32, i32, f32, C.c_void_p],
    "lvdhip_ca_select": [_P(CaSelectParams), C.c_void_p],
    "lvdhip_ca_dq": [_P(CaDqParams), C.c_void_p],
    "lvdhip_ca_probs_multi": [_P(CaProbsParams), i32, C.c_void_p],
    "lvdhip_ca_select_multi": [_P(CaSelectParams), i32, C.c_void_p],
    "lvdhip_ca_dq_multi": [_P(CaDqParams), i32, C.c_void_p],
    "lvdhip_latents_to_tokens": [C.c_void_p, C.c_void_p, i32, i32, i32, i32, i32, f32, C.c_void_p],
    "lvdhip_tokens_to_latents": [C.c_void_p, i32, C.c_void_p, i32, i32, i32, i32, C.c_void_p],
    "lvdhip_tokens_grad_to_latents": [C.c_void_p, i32, C.c_void_p, i32, i32, i32, i32, f32, C.c_void_p],
    "lvdhip_add": [C.c_void_p, i32, C.c_void_p, i32, C.c_void_p, i32, i32, i32, C.c_void_p],
    "lvdhip_tconv_combine": [C.c_void_p, i32, C.c_void_p, C.c_void_p, i32, C.c_void_p, i32, i32, i32, i32, i32, i32, C.c_void_p],
    "lvdhip_geglu_fwd": [C.c_void_p, i32, C.c_void_p, i32, i32, i32, C.c_void_p],
    "lvdhip_geglu_bwd": [C.c_void_p, i32, C.c_void_p, i32, C.c_void_p, i32, i32, i32, C.c_void_p],
    "lvdhip_upsample2x_bwd": [C.c_void_p, C.c_void_p, i32, i32, i32, i32, i32, C.c_void_p],
    "lvdhip_timestep_embedding": [C.c_void_p, C.c_void_p, i32, i32, C.c_void_p],
    "lvdhip_silu": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "lvdhip_gelu": [C.c_void_p, C.c_void_p, C.c_int64, i32, C.c_void_p],
    "lvdhip_cfg_dpm_step": [C.c_void_p, C.c_void_p, f32, C.c_void_p, C.c_void_p, f32, f32, f32, f32, f32, C.c_int64, C.c_void_p],
    "lvdhip_axpy": [C.c_void_p, C.c_void_p, f32, C.c_int64, C.c_void_p],
    "lvdhip_reduce_sum": [C.c_void_p, C.c_int64, f32, C.c_void_p, C.c_void_p],
    "lvdhip_softmax_rows": [C.c_void_p, i32, C.c_void_p, i32, i32, i32, C.c_void_p],
    "lvdhip_gemm_workspace_bytes": [_P(GemmParams), C.POINTER(C.c_int64)],
    "lvdhip_tokens_to_video": [C.c_void_p, i32, C.c_void_p, C.c_int64, C.c_void_p],
    "lvdhip_frames_to_patches": [C.c_void_p, i32, i32, i32, i32, i32, i32, C.c_void_p, C.c_void_p, i32, C.c_void_p, C.c_void_p, i32,
                                 C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, i32, C.c_void_p, C.c_void_p],
    "lvdhip_owl_detect_rows": [C.c_void_p, i32, i32, C.c_void_p, i32, C.c_void_p, C.c_void_p, i32, C.c_void_p, i32, C.c_void_p, i32, f32, f32,
                               C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
}

_lib = None
# The ctypes structs above mirror include/lvdhip.h at exactly this lvdhip_version(): a stale liblvdhip.so would silently ignore fields
# added since (ldrowbias, acc_mode, ...) and compute something else, so lib() refuses any other version.
ABI_VERSION = 108
CA_MAX_KEYS = 8  # LVD_CA_MAX_KEYS


def lib():
    """Load liblvdhip.so once; raise loudly when it is absent (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no fallback path."
            )
        # torch bundles its own libamdhip64; it must be the HIP runtime already resident when liblvdhip.so is
        # loaded, otherwise two runtimes coexist and the streams torch hands us belong to the other one.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, argtypes in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the library lacks a declared symbol
            if name == "lvdhip_last_error":
                fn.restype = C.c_char_p
                fn.argtypes = []
            else:
                fn.restype = C.c_int
                fn.argtypes = argtypes
        got = l.lvdhip_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports lvdhip_version() = {got}, the Python side was written for {ABI_VERSION}: rebuild the library "
                               "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        _lib = l
    return _lib


calls = 0  # launching C-ABI entries made by this process (bench.py reports them per step; one entry is one to three kernel launches)


def check(rc, what):
    global calls
    calls += what != "gemm_workspace_bytes"  # a size query launches nothing
    if rc != 0:
        msg = lib().lvdhip_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"lvdhip {what} failed (rc={rc}): {msg}")
