"""Tensor-level wrappers over the C ABI (include/lvdhip.h).

PyTorch is used only as the device-memory container and stream provider: every function here takes
2-D "token matrices" (rows = (batch, frame, y, x), channels contiguous, bf16) and launches the
hand-written HIP kernels on torch's current stream.  Nothing in this file computes with torch ops.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import hip

A_PLAIN, A_CONV3X3, A_TCONV3, A_CONV3X3_T2 = 0, 1, 2, 3
ACT_NONE, ACT_GEGLU = 0, 1


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def use_device(dev):
    """Make `dev` the CURRENT HIP device of this process.  The C ABI launches on the stream it is handed and hipLaunchKernelGGL
    targets the current device, so an engine built on cuda:N must run with device N current — one process per GPU selects its own
    (generate.py, scripts/*, bench.py under torchrun).  Called by every engine constructor and forward entry; a no-op when already set."""
    dev = torch.device(dev)
    if dev.type == "cuda":
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if torch.cuda.current_device() != idx:
            torch.cuda.set_device(idx)
    return dev


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, f"expected row-major 2-D tensor, got {tuple(t.shape)} / {t.stride()}"
    return t.stride(0)


def _chk_bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.bfloat16 and t.is_cuda, "bf16 CUDA tensor expected"


def _chk_f32(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), "contiguous fp32 CUDA tensor expected"


def _chk_f32_rows(t):
    if t is not None:
        assert t.dtype == torch.float32 and t.is_cuda and t.dim() == 2 and t.stride(1) == 1, "fp32 CUDA matrix with contiguous rows expected"


@dataclass
class ConvGeom:
    hin: int
    win: int
    hout: int
    wout: int
    stride: int = 1
    upsample: int = 0


# ------------------------------------------------------------------ GEMM autotuner
# The GEMM library ships several tile geometries (include/lvdhip.h LVD_GEMM_V_*).  Which one wins depends on
# (M, N, K, loader): short-K linears favour many small workgroups, long-K convs the LDS-DMA ring, and the
# 1-workgroup-per-CU 256-wide tiles only pay when the grid quantises well.  The first call of a new shape times the
# candidates on a scratch output (HIP events on the launch stream) and pins the winner for the process.
GEMM_CANDIDATES = (10, 1, 5, 9, 11, 14, 17, 105, 109, 111, 117, 161, 211, 205, 209, 217)  # 100 + v: asm-DMA instantiation of ring variant v; 200 + v: 64-deep K tiles; 161: persistent walker (gemm_stream.hip)
SPLITK_VARIANT = 20
SPLITK_WIDE_VARIANT = 25
TAIL_VARIANTS = (31, 37, 120, 125, 131, 137, 225, 231, 220)  # whole rounds on the wide geometry + split-K remainder (gemm.hip run_with_tail); need the workspace
HALO_VARIANTS = (41, 45, 47)  # conv_halo.hip: whole grid / channel-chunk split-K / whole rounds + split-K tail
_splitk_ws = {}


SPLITK_WS_BYTES = int(os.environ.get("LVD_SPLITK_WS_MB", "1024")) << 20


def _splitk_workspace(dev, nbytes):
    """ONE fp32 workspace per device for the K-split slabs, of a FIXED size (stream-ordered reuse is safe: every user is a GEMM+reduce
    pair on the same stream).  The K-split plans of the library are functions of the workspace size they are offered (a plan that does
    not fit takes fewer slices or falls back to the unsplit launch), so the size must not depend on what a process happened to run
    before: a grow-only buffer made a sharded run (table loaded, nothing tuned, small buffer) pick other plans — other fp32 summation
    orders, other bf16 bits — than the single-process run that had grown it while tuning."""
    ws = _splitk_ws.get(dev)
    if ws is None:
        ws = _splitk_ws[dev] = torch.empty(SPLITK_WS_BYTES // 4, dtype=torch.float32, device=dev)
    return ws
_debug_ws = None  # developer probes (tools/stream_trace.py): a tensor handed to every launch as p.ws whatever the product's own need ...
_DEBUG_HOOKS = os.environ.get("LVD_DEBUG_HOOKS") == "1"  # ... and only when the process was started with LVD_DEBUG_HOOKS=1
_gemm_choice = {}
_autotune = {"enabled": True, "min_flops": 2e9}
_EXCLUDE = tuple(int(v) for v in os.environ.get("LVD_GEMM_EXCLUDE", "").split(",") if v)  # developer knob: variants the tuner must not try


def set_gemm_autotune(enabled: bool):
    _autotune["enabled"] = bool(enabled)


def gemm_autotune_table():
    return dict(_gemm_choice)


def save_gemm_autotune_table(path):
    """Persist the per-shape geometry choices.  Variants differ in fp32 summation order (K-split slices, tile K order), so two
    processes that tuned independently can produce different bf16 roundings for the same (prompt, seed); loading one table in
    every rank / run (generate.py --gemm_autotune_table) makes sharded runs reproduce the single-process one bit for bit."""
    import json
    with open(path, "w") as f:
        json.dump([[list(k), v] for k, v in _gemm_choice.items()], f)


def load_gemm_autotune_table(path):
    import json
    with open(path) as f:
        for k, v in json.load(f):
            k[7] = tuple(k[7]) if k[7] is not None else None
            _gemm_choice[tuple(k)] = int(v)


def _launch_gemm(p):
    hip.check(hip.lib().lvdhip_gemm(C.byref(p), _stream()), "gemm")


def _tune_gemm(p, key, out):
    scratch = torch.empty((out.shape[0], p.ldc), dtype=out.dtype, device=out.device)  # same row stride as the real output
    saved_out, saved_acc = p.out, p.accumulate
    p.out, p.accumulate = scratch.data_ptr(), 0
    best, best_t = 0, float("inf")
    cands = GEMM_CANDIDATES + ((SPLITK_VARIANT, SPLITK_WIDE_VARIANT) + TAIL_VARIANTS if p.ws else ())
    if p.ln_mean_rstd:  # LayerNorm-folded product: asm-DMA ring kernels and K-split plans only (include/lvdhip.h)
        cands = tuple(v for v in cands if v >= 100 or v in (SPLITK_VARIANT, SPLITK_WIDE_VARIANT))
    halo = (p.mode == A_CONV3X3 and p.stride == 1 and p.win <= 87) or (p.mode == A_TCONV3 and 2 <= p.frames <= 256)
    if halo and not p.a2 and p.cin % 32 == 0:
        cands += (HALO_VARIANTS if p.ws else HALO_VARIANTS[:1])  # LDS-resident im2col (conv_halo.hip)
    # the persistent walker (gemm_stream.hip) takes plain-loader products only; anything else would silently run the one-shot ring under its
    # number (timed twice, and possibly pinned in a saved table for a product that never reaches the walker)
    stream_ok = p.mode == A_PLAIN and not p.a2 and not p.rowbias and not p.accumulate and not p.out_fp32 and p.K % 32 == 0 and p.K >= 128
    for v in cands:
        if v in _EXCLUDE or (v == 161 and not stream_ok):
            continue
        p.variant = v
        try:
            _launch_gemm(p)  # warm-up (also instruction-cache / L2)
        except RuntimeError:
            continue  # a plan this product cannot take (e.g. a K-split that degenerates to a kernel without the needed epilogue)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        evs[0].record()
        for i in range(5):
            _launch_gemm(p)
            evs[i + 1].record()
        evs[-1].synchronize()
        t = min(evs[i].elapsed_time(evs[i + 1]) for i in range(5))  # min of 5: robust against a neighbour's tail / clock ramps
        if t < best_t:
            best, best_t = v, t
    p.out, p.accumulate = saved_out, saved_acc
    _gemm_choice[key] = best
    if os.environ.get("LVD_GEMM_LOG"):  # developer knob: which shapes a process had to tune (none, when a complete table was loaded)
        import sys
        print(f"[lvd gemm autotune] {key} -> {best}", file=sys.stderr, flush=True)
    return best


def gemm(a1, w, *, n=None, k=None, a2=None, bias=None, rowbias=None, rows_per_sample=0, res=None, out=None,
         mode=A_PLAIN, conv: Optional[ConvGeom] = None, frames=0, hw=0, cin=None, act=ACT_NONE, out_fp32=False,
         alpha=1.0, accumulate=False, m=None, variant=0, m_begin=0, ln_stats=None, ln_colsum=None, ksplit=0):
    """OUT[M,N] = epi(Aload · W^T).  `w` is [N,K] bf16.  Returns `out`.  `m_begin` > 0 produces rows [m_begin, M) only.
    ln_stats ([M,2] fp32 mean/rstd from layernorm_stats) + ln_colsum ([N] fp32): LayerNorm(a1) · W^T with the norm folded into the product —
    `w` must then be gamma (.) W and `bias` b + W beta (engine._pack builds them)."""
    _chk_bf16(a1, a2, w, res)
    _chk_f32(ln_stats, ln_colsum)
    assert (ln_stats is None) == (ln_colsum is None)
    assert ln_stats is None or bias is not None, "a LayerNorm-folded product needs its bias row (b + W beta): the folded epilogue always reads it"
    _chk_f32(bias)
    _chk_f32_rows(rowbias)  # may be a column range of a wider matrix (engine: all temb projections of a forward are one product)
    N, K = w.shape if (n is None or k is None) else (n, k)
    if rowbias is not None:  # the kernels read it with 16-byte loads at rowbias + sample * stride + n
        assert rowbias.data_ptr() % 16 == 0 and rowbias.stride(0) % 4 == 0 and rowbias.shape[1] == N, "rowbias: 16-byte aligned rows of N floats expected"
    assert w.is_contiguous()
    c1 = a1.shape[1]
    c2 = a2.shape[1] if a2 is not None else 0
    if cin is None:
        cin = c1 + c2
    if m is None:
        m = a1.shape[0]
        if mode in (A_CONV3X3, A_CONV3X3_T2):
            nimg = a1.shape[0] // ((conv.hin >> conv.upsample) * (conv.win >> conv.upsample)) if mode == A_CONV3X3 else a1.shape[0] // (conv.hin * conv.win)
            m = nimg * conv.hout * conv.wout
    n_out = N // 2 if act == ACT_GEGLU else N
    if out is None:
        out = torch.empty((m, n_out), dtype=torch.float32 if out_fp32 else torch.bfloat16, device=a1.device)
    assert out.shape[0] == m and out.shape[1] >= n_out
    assert rowbias is None or (rows_per_sample > 0 and rowbias.shape[0] * rows_per_sample >= m), "rowbias: one row per sample of rows_per_sample token rows"
    p = hip.GemmParams()
    p.a1, p.a2, p.w, p.bias, p.rowbias, p.res, p.out = _p(a1), _p(a2), _p(w), _p(bias), _p(rowbias), _p(res), _p(out)
    p.M, p.N, p.K = m, N, K
    p.lda1, p.lda2, p.c1, p.cin, p.mode = _ld(a1), (_ld(a2) if a2 is not None else 0), c1, cin, mode
    if conv is not None:
        p.hin, p.win, p.hout, p.wout, p.stride, p.upsample = conv.hin, conv.win, conv.hout, conv.wout, conv.stride, conv.upsample
    p.frames, p.hw = frames, hw
    p.rows_per_sample = rows_per_sample
    p.ldrowbias = rowbias.stride(0) if rowbias is not None else 0
    p.ldres = _ld(res) if res is not None else 0
    p.ldc = _ld(out)
    p.act, p.out_fp32, p.alpha, p.accumulate = act, int(out_fp32), float(alpha), int(accumulate)
    p.m_begin = m_begin
    if ln_stats is not None:
        assert mode == A_PLAIN and a2 is None and ln_stats.shape == (m, 2) and ln_colsum.shape == (N,)
        p.ln_mean_rstd, p.ln_colsum = _p(ln_stats), _p(ln_colsum)
    p.ksplit = ksplit  # 0: the library's plan (K-split variants only); set before the workspace query, which depends on it
    need = C.c_int64(0)
    hip.check(hip.lib().lvdhip_gemm_workspace_bytes(C.byref(p), C.byref(need)), "gemm_workspace_bytes")
    if need.value:
        ws = _splitk_workspace(a1.device, need.value)  # one fixed-size buffer per device, shared by all launches of the stream
        p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
    if _debug_ws is not None and _DEBUG_HOOKS:
        p.ws, p.ws_bytes = _debug_ws.data_ptr(), _debug_ws.numel() * 4
    if variant == 0 and m_begin == 0 and _autotune["enabled"] and 2.0 * m * N * K >= _autotune["min_flops"]:
        # everything a candidate's eligibility or cost depends on: the conv image (the LDS-resident tap GEMM needs W <= 87), the temporal
        # geometry (its tile is pixels x all frames) and a temb row-bias (the asm-DMA forms do not take one)
        key = (mode, m, N, K, act, c1, cin, (conv.stride, conv.upsample, conv.hin, conv.win) if conv is not None else None, int(out_fp32),
               res is not None, bool(accumulate), frames, hw, rowbias is not None) + ((True,) if ln_stats is not None else ())
        variant = _gemm_choice.get(key) or 0
        if variant == 0 and not torch.cuda.is_current_stream_capturing():  # a capture replays what a warm-up run has tuned
            variant = _tune_gemm(p, key, out)
    p.variant = variant
    _launch_gemm(p)
    return out


GN_MAX_CHUNKS = int(os.environ.get("LVD_GN_MAX_CHUNKS", "256"))  # developer knobs (tools/small_ops_bench.py sweeps): chunks per sample ...
GN_STATS_WGS = int(os.environ.get("LVD_GN_WGS", "512"))           # ... and workgroups per statistics launch


def _gn_chunks(samples, rows_per_sample, min_rows=64):
    # two workgroups per CU over the whole launch (measured best: 512 against 768 / 1024, tools/small_ops_bench.py), but >= min_rows rows per chunk, and at most GN_MAX_CHUNKS per sample: every
    # workgroup of the apply pass folds the [chunks, groups] partials of its sample itself (no finalize launch), 8 bytes * groups per chunk
    want = max(1, min(GN_STATS_WGS // max(samples, 1), GN_MAX_CHUNKS))
    return max(1, min(want, rows_per_sample // min_rows if rows_per_sample >= min_rows else 1))


def _gn_min_rows(c):
    """Rows per statistics chunk, at least: a workgroup has 512 // (c / 8) row lanes (norm.hip gn_geometry) and a lane walks its rows eight at
    a time.  64 rows is three trips at c <= 1280; the wide two-source norms of the up blocks (c = 1920 / 2560: two row lanes / one) would walk
    32 / 64 rows per lane — the level-2 norm over [x, skip] (c = 2560, 180 rows per sample) spent 26 us in two chunks per sample
    (profiles/r04_gn_cooperative_experiment.txt)."""
    row_lanes = max(1, 512 // max(1, c // 8))
    return min(64, 24 * row_lanes)


@dataclass
class GnStats:
    """Statistics partials of a GroupNorm ([samples, chunks, groups, 2] sums) plus what the apply launch needs to finish them."""
    partial: torch.Tensor
    chunks: int
    groups: int
    eps: float
    gamma: torch.Tensor
    beta: torch.Tensor


def groupnorm_stats(x1, gamma, beta, rows_per_sample, *, groups=32, eps=1e-5, x2=None):
    """First of the two launches: per-(row chunk, group) sums.  Returns GnStats for groupnorm_apply."""
    _chk_bf16(x1, x2)
    _chk_f32(gamma, beta)
    rows = x1.shape[0]
    c1 = x1.shape[1]
    c = c1 + (x2.shape[1] if x2 is not None else 0)
    samples = rows // rows_per_sample
    chunks = _gn_chunks(samples, rows_per_sample, _gn_min_rows(c))
    dev = x1.device
    partial = torch.empty((samples, chunks, groups, 2), dtype=torch.float32, device=dev)
    p = hip.GnStatsParams()
    p.x1, p.x2, p.ld1, p.ld2, p.c1, p.c = _p(x1), _p(x2), _ld(x1), (_ld(x2) if x2 is not None else 0), c1, c
    p.rows, p.rows_per_sample, p.groups, p.eps = rows, rows_per_sample, groups, eps
    p.gamma, p.beta, p.partial, p.chunks, p.scale_shift, p.mean_rstd = _p(gamma), _p(beta), _p(partial), chunks, None, None
    hip.check(hip.lib().lvdhip_groupnorm_stats(C.byref(p), _stream()), "groupnorm_stats")
    return GnStats(partial, chunks, groups, eps, gamma, beta)


def groupnorm_apply(x1, stats: GnStats, rows_per_sample, *, silu=False, x2=None, out=None):
    """Second launch: folds the partials, normalises (+SiLU).  Returns (y, mean_rstd [S,G,2])."""
    _chk_bf16(x1, x2)
    rows = x1.shape[0]
    c1 = x1.shape[1]
    c = c1 + (x2.shape[1] if x2 is not None else 0)
    if out is None:
        out = torch.empty((rows, c), dtype=torch.bfloat16, device=x1.device)
    mr = torch.empty((rows // rows_per_sample, stats.groups, 2), dtype=torch.float32, device=x1.device)
    p = hip.GnApplyParams()
    p.x1, p.x2, p.ld1, p.ld2, p.c1, p.c = _p(x1), _p(x2), _ld(x1), (_ld(x2) if x2 is not None else 0), c1, c
    p.rows, p.rows_per_sample, p.partial, p.silu, p.y, p.ldy = rows, rows_per_sample, _p(stats.partial), int(silu), _p(out), _ld(out)
    p.chunks, p.groups, p.eps, p.gamma, p.beta, p.mean_rstd = stats.chunks, stats.groups, stats.eps, _p(stats.gamma), _p(stats.beta), _p(mr)
    hip.check(hip.lib().lvdhip_groupnorm_apply(C.byref(p), _stream()), "groupnorm_apply")
    return out, mr


# Small samples take the single-launch kernels (norm_small.hip): a (sample, group) slab of at most 64 Ki elements (128 KB: its
# second pass is an L2 hit) and a tensor small enough that the three launches of the two-stage path are launch-bound.
_gn_fused = {"max_slab": 65536, "max_bytes": int(os.environ.get("LVD_GN_FUSED_MAX_MB", "24")) << 20,
             "max_rows_per_thread": int(os.environ.get("LVD_GN_FUSED_MAX_ROWS", "16")),
             "max_bytes_slab": int(os.environ.get("LVD_GN_SLAB_MAX_MB", "128")) << 20}  # 0: the slab-in-registers kernel is off


def groupnorm_fused_ok(rows, c, rows_per_sample, groups):
    """A workgroup walks its slab with 256 // (cpg/2) row lanes: beyond ~16 rows per lane (the 5-D norms of the temporal layers:
    1080+ rows per sample, only samples*groups workgroups) the walk is latency-bound and the two-stage kernels win."""
    cpg = c // groups
    if cpg % 2 or cpg > 512 or rows_per_sample * cpg > _gn_fused["max_slab"] or rows * c * 2 > _gn_fused["max_bytes"]:
        return False
    row_lanes = 256 // (cpg // 2)
    return -(-rows_per_sample // row_lanes) <= _gn_fused["max_rows_per_thread"]


def groupnorm_fused(x1, gamma, beta, rows_per_sample, *, groups=32, eps=1e-5, silu=False, x2=None, out=None, slab=False):
    """GroupNorm(+SiLU) in one launch (small samples; slab=True: the slab-in-registers kernel).  Returns (y, mean_rstd [S,G,2])."""
    _chk_bf16(x1, x2)
    _chk_f32(gamma, beta)
    rows, c1 = x1.shape
    c = c1 + (x2.shape[1] if x2 is not None else 0)
    if out is None:
        out = torch.empty((rows, c), dtype=torch.bfloat16, device=x1.device)
    mr = torch.empty((rows // rows_per_sample, groups, 2), dtype=torch.float32, device=x1.device)
    s, a = hip.GnStatsParams(), hip.GnApplyParams()
    for q in (s, a):
        q.x1, q.x2, q.ld1, q.ld2, q.c1, q.c = _p(x1), _p(x2), _ld(x1), (_ld(x2) if x2 is not None else 0), c1, c
        q.rows, q.rows_per_sample = rows, rows_per_sample
    s.groups, s.eps, s.gamma, s.beta, s.mean_rstd = groups, eps, _p(gamma), _p(beta), _p(mr)
    a.silu, a.y, a.ldy = int(silu), _p(out), _ld(out)
    fn = hip.lib().lvdhip_groupnorm_slab if slab else hip.lib().lvdhip_groupnorm_fused
    hip.check(fn(C.byref(s), C.byref(a), _stream()), "groupnorm_slab" if slab else "groupnorm_fused")
    return out, mr


_slab_loads = {}


def groupnorm_slab_ok(rows, c, rows_per_sample, groups, c1=None, backward=False):
    """The 1024-thread kernels that hold a whole (sample, group) slab in registers (norm_small.hip gn_slab_kernel / gn_bwd_slab_kernel) take
    this shape: lvdhip_groupnorm_slab_loads / lvdhip_groupnorm_bwd_slab_loads (an even number of channels per group, a group inside one
    source, few enough loads per thread)."""
    if rows * c * 2 > _gn_fused["max_bytes_slab"]:
        return False
    key = (c, c if c1 is None else c1, groups, rows_per_sample, backward)
    n = _slab_loads.get(key)
    if n is None:
        fn = hip.lib().lvdhip_groupnorm_bwd_slab_loads if backward else hip.lib().lvdhip_groupnorm_slab_loads
        n = _slab_loads[key] = int(fn(*key[:4]))
    return n > 0


def groupnorm_auto(x1, gamma, beta, rows_per_sample, *, groups=32, eps=1e-5, silu=False, x2=None, out=None):
    """(y, mean_rstd): one launch when a (sample, group) slab is small enough for one workgroup — the 256-thread walk for short slabs, the
    1024-thread slab-in-registers kernel for the ones in between —, the two-stage kernels otherwise."""
    c = x1.shape[1] + (x2.shape[1] if x2 is not None else 0)
    if groupnorm_fused_ok(x1.shape[0], c, rows_per_sample, groups):
        return groupnorm_fused(x1, gamma, beta, rows_per_sample, groups=groups, eps=eps, silu=silu, x2=x2, out=out)
    if groupnorm_slab_ok(x1.shape[0], c, rows_per_sample, groups, x1.shape[1]):
        return groupnorm_fused(x1, gamma, beta, rows_per_sample, groups=groups, eps=eps, silu=silu, x2=x2, out=out, slab=True)
    st = groupnorm_stats(x1, gamma, beta, rows_per_sample, groups=groups, eps=eps, x2=x2)
    return groupnorm_apply(x1, st, rows_per_sample, silu=silu, x2=x2, out=out)


def groupnorm(x1, gamma, beta, rows_per_sample, *, groups=32, eps=1e-5, silu=False, x2=None, out=None, return_stats=False):
    y, mr = groupnorm_auto(x1, gamma, beta, rows_per_sample, groups=groups, eps=eps, silu=silu, x2=x2, out=out)
    return (y, mr) if return_stats else y


def groupnorm_bwd(x1, dy, gamma, beta, mean_rstd, rows_per_sample, *, groups=32, silu=False, x2=None,
                  dx1=None, dx2=None, accumulate=False):
    """Input-gradient of groupnorm(+silu).  Returns (dx1, dx2)."""
    _chk_bf16(x1, x2, dy)
    rows = x1.shape[0]
    c1 = x1.shape[1]
    c = c1 + (x2.shape[1] if x2 is not None else 0)
    samples = rows // rows_per_sample
    small = groupnorm_fused_ok(rows, c, rows_per_sample, groups)
    if small or groupnorm_slab_ok(rows, c, rows_per_sample, groups, c1, backward=True):
        if dx1 is None:
            dx1 = torch.empty_like(x1)
            assert not accumulate
        if x2 is not None and dx2 is None:
            dx2 = torch.empty_like(x2)
        q = hip.GnBwdApplyParams()
        q.x1, q.x2, q.ld1, q.ld2, q.c1, q.c = _p(x1), _p(x2), _ld(x1), (_ld(x2) if x2 is not None else 0), c1, c
        q.dy, q.lddy = _p(dy), _ld(dy)
        q.rows, q.rows_per_sample, q.groups = rows, rows_per_sample, groups
        q.gamma, q.beta, q.mean_rstd, q.silu = _p(gamma), _p(beta), _p(mean_rstd), int(silu)
        q.dx1, q.dx2 = _p(dx1), _p(dx2)
        q.lddx1, q.lddx2 = _ld(dx1), (_ld(dx2) if dx2 is not None else 0)
        q.accumulate = int(accumulate)
        fn = hip.lib().lvdhip_groupnorm_bwd_fused if small else hip.lib().lvdhip_groupnorm_bwd_slab
        hip.check(fn(C.byref(q), _stream()), "groupnorm_bwd_fused" if small else "groupnorm_bwd_slab")
        return dx1, dx2
    # the 5-D norms of the deep levels (1-2 samples of 1080 / 4320 rows): 64-row chunks leave the backward statistics pass on 16-67
    # workgroups (34 -> 25 us at 1080 rows, 39 -> 32 us at 4320 with 32-row chunks; larger samples and the forward pass do not gain)
    chunks = _gn_chunks(samples, rows_per_sample, 32 if rows_per_sample <= 8192 else 64)
    dev = x1.device
    partial = torch.empty((samples, chunks, groups, 2), dtype=torch.float32, device=dev)
    p = hip.GnBwdStatsParams()
    p.x1, p.x2, p.ld1, p.ld2, p.c1, p.c = _p(x1), _p(x2), _ld(x1), (_ld(x2) if x2 is not None else 0), c1, c
    p.dy, p.lddy = _p(dy), _ld(dy)
    p.rows, p.rows_per_sample, p.groups = rows, rows_per_sample, groups
    p.gamma, p.beta, p.mean_rstd, p.partial, p.chunks, p.gsum, p.silu = _p(gamma), _p(beta), _p(mean_rstd), _p(partial), chunks, None, int(silu)
    hip.check(hip.lib().lvdhip_groupnorm_bwd_stats(C.byref(p), _stream()), "groupnorm_bwd_stats")
    if dx1 is None:
        dx1 = torch.empty_like(x1)
        assert not accumulate
    if x2 is not None and dx2 is None:
        dx2 = torch.empty_like(x2)
    q = hip.GnBwdApplyParams()
    q.x1, q.x2, q.ld1, q.ld2, q.c1, q.c = p.x1, p.x2, p.ld1, p.ld2, c1, c
    q.dy, q.lddy = p.dy, p.lddy
    q.rows, q.rows_per_sample, q.groups = rows, rows_per_sample, groups
    q.gamma, q.beta, q.mean_rstd, q.partial, q.chunks, q.silu = p.gamma, p.beta, p.mean_rstd, _p(partial), chunks, int(silu)
    q.dx1, q.dx2 = _p(dx1), _p(dx2)
    q.lddx1, q.lddx2 = _ld(dx1), (_ld(dx2) if dx2 is not None else 0)
    q.accumulate = int(accumulate)
    hip.check(hip.lib().lvdhip_groupnorm_bwd_apply(C.byref(q), _stream()), "groupnorm_bwd_apply")
    return dx1, dx2


def layernorm(x, gamma, beta, *, eps=1e-5, out=None, return_stats=False):
    _chk_bf16(x)
    _chk_f32(gamma, beta)
    rows, c = x.shape
    if out is None:
        out = torch.empty((rows, c), dtype=torch.bfloat16, device=x.device)
    mr = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if return_stats else None
    p = hip.LnParams()
    p.x, p.ldx, p.rows, p.c, p.gamma, p.beta, p.eps = _p(x), _ld(x), rows, c, _p(gamma), _p(beta), eps
    p.y, p.ldy, p.mean_rstd = _p(out), _ld(out), _p(mr)
    hip.check(hip.lib().lvdhip_layernorm(C.byref(p), _stream()), "layernorm")
    return (out, mr) if return_stats else out


def tconv_expand_weight(wt, taps=3):
    """[N, taps * Cin] tap-major temporal-conv weight -> [taps * N, Cin]: row t * N + n holds tap t of output channel n (tconv_expanded)."""
    n, k = wt.shape
    return wt.view(n, taps, k // taps).permute(1, 0, 2).reshape(taps * n, k // taps).contiguous()


def tconv_expanded(x, w_exp, *, frames, hw, bias=None, res=None, out=None, accumulate=False):
    """Temporal (3,1,1) conv as ONE plain product with 3N output columns (fp32) + a combine pass: for small M (deep UNet levels) three times
    the tiles and no K split beat the tap GEMM's K-split slabs.  `w_exp` = tconv_expand_weight(w)."""
    _chk_bf16(x, w_exp, res)
    m, n = x.shape[0], w_exp.shape[0] // 3
    y = gemm(x, w_exp, out_fp32=True)
    if out is None:
        out = torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
        assert not accumulate
    hip.check(hip.lib().lvdhip_tconv_combine(_p(y), _ld(y), _p(bias), _p(res), _ld(res) if res is not None else 0, _p(out), _ld(out), m, n, frames, hw,
                                             int(accumulate), _stream()), "tconv_combine")
    return out


def layernorm_stats(x, *, eps=1e-5):
    """(mean, rstd) [rows, 2] fp32 of every row — what a LayerNorm-folded GEMM (gemm(..., ln_stats=)) and layernorm_bwd read; x is only read."""
    _chk_bf16(x)
    rows, c = x.shape
    mr = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    p = hip.LnParams()
    p.x, p.ldx, p.rows, p.c, p.gamma, p.beta, p.eps = _p(x), _ld(x), rows, c, None, None, eps
    p.y, p.ldy, p.mean_rstd = None, 0, _p(mr)
    hip.check(hip.lib().lvdhip_layernorm(C.byref(p), _stream()), "layernorm_stats")
    return mr


def layernorm_bwd(x, dy, gamma, mean_rstd, *, dx=None, accumulate=False):
    _chk_bf16(x, dy)
    rows, c = x.shape
    if dx is None:
        dx = torch.empty((rows, c), dtype=torch.bfloat16, device=x.device)
        assert not accumulate
    p = hip.LnBwdParams()
    p.x, p.ldx, p.dy, p.lddy, p.rows, p.c = _p(x), _ld(x), _p(dy), _ld(dy), rows, c
    p.gamma, p.mean_rstd, p.dx, p.lddx, p.accumulate = _p(gamma), _p(mean_rstd), _p(dx), _ld(dx), int(accumulate)
    hip.check(hip.lib().lvdhip_layernorm_bwd(C.byref(p), _stream()), "layernorm_bwd")
    return dx


@dataclass
class RowMap:
    """Token row of element i of sample s = os*(s // ninner) + is_*(s % ninner) + step*i."""
    ninner: int = 1
    os: int = 0
    is_: int = 0
    step: int = 1


def attn_params(q, k, v, o, *, samples, heads, sq, skv, qmap: RowMap, kvmap: RowMap, scale, lse=None,
                k2=None, v2=None, skv2=0, kv2map: Optional[RowMap] = None, causal=False):
    _chk_bf16(q, k, v, o, k2, v2)
    p = hip.AttnParams()
    p.q, p.ldq, p.k, p.ldk, p.v, p.ldv = _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v)
    p.k2, p.v2 = _p(k2), _p(v2)
    p.ldk2, p.ldv2 = (_ld(k2) if k2 is not None else 0), (_ld(v2) if v2 is not None else 0)
    p.o, p.ldo, p.lse = _p(o), _ld(o), _p(lse)
    p.samples, p.heads, p.sq, p.skv, p.skv2 = samples, heads, sq, skv, skv2
    p.q_ninner, p.q_os, p.q_is, p.q_step = qmap.ninner, qmap.os, qmap.is_, qmap.step
    p.kv_ninner, p.kv_os, p.kv_is, p.kv_step = kvmap.ninner, kvmap.os, kvmap.is_, kvmap.step
    m2 = kv2map or RowMap()
    p.kv2_ninner, p.kv2_os, p.kv2_is, p.kv2_step = m2.ninner, m2.os, m2.is_, m2.step
    p.scale = scale
    p.causal = int(causal)
    return p


def attention_fwd(q, k, v, o, **kw):
    p = attn_params(q, k, v, o, **kw)
    hip.check(hip.lib().lvdhip_attention_fwd(C.byref(p), _stream()), "attention_fwd")
    return o


def attention_bwd(q, k, v, o, lse, d_o, dq, dk, dv, **kw):
    """dq always; dk/dv when given (None for text cross-attention)."""
    _chk_bf16(d_o, dq, dk, dv)
    b = hip.AttnBwdParams()
    b.f = attn_params(q, k, v, o, lse=lse, **kw)
    b.d_o, b.lddo, b.dq, b.lddq = _p(d_o), _ld(d_o), _p(dq), _ld(dq)
    b.dk, b.lddk = _p(dk), (_ld(dk) if dk is not None else 0)
    b.dv, b.lddv = _p(dv), (_ld(dv) if dv is not None else 0)
    delta = torch.empty((kw["samples"], kw["heads"], kw["sq"]), dtype=torch.float32, device=q.device)
    b.delta = _p(delta)
    hip.check(hip.lib().lvdhip_attention_bwd(C.byref(b), _stream()), "attention_bwd")
    return dq, dk, dv


# ------------------------------------------------------------------ elementwise
def latents_to_tokens(latents, cpad=8, scale=1.0, out=None):
    _chk_f32(latents)
    B, Cc, F, H, W = latents.shape
    if out is None:
        out = torch.empty((B * F * H * W, cpad), dtype=torch.bfloat16, device=latents.device)
    hip.check(hip.lib().lvdhip_latents_to_tokens(_p(latents), _p(out), B, Cc, F, H * W, cpad, scale, _stream()), "latents_to_tokens")
    return out


def tokens_to_latents(tokens, B, Cc, F, H, W, out=None):
    _chk_f32(tokens)
    if out is None:
        out = torch.empty((B, Cc, F, H, W), dtype=torch.float32, device=tokens.device)
    hip.check(hip.lib().lvdhip_tokens_to_latents(_p(tokens), tokens.stride(0), _p(out), B, Cc, F, H * W, _stream()), "tokens_to_latents")
    return out


def tokens_grad_to_latents(tokens, B, Cc, F, H, W, scale=1.0, out=None):
    _chk_bf16(tokens)
    if out is None:
        out = torch.empty((B, Cc, F, H, W), dtype=torch.float32, device=tokens.device)
    hip.check(hip.lib().lvdhip_tokens_grad_to_latents(_p(tokens), _ld(tokens), _p(out), B, Cc, F, H * W, scale, _stream()), "tokens_grad_to_latents")
    return out


def add(a, b, out=None):
    _chk_bf16(a, b)
    if out is None:
        out = torch.empty((a.shape[0], a.shape[1]), dtype=torch.bfloat16, device=a.device)
    hip.check(hip.lib().lvdhip_add(_p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out), a.shape[0], a.shape[1], _stream()), "add")
    return out


def geglu_fwd(pre, out=None):
    _chk_bf16(pre)
    rows, n2 = pre.shape
    if out is None:
        out = torch.empty((rows, n2 // 2), dtype=torch.bfloat16, device=pre.device)
    hip.check(hip.lib().lvdhip_geglu_fwd(_p(pre), _ld(pre), _p(out), _ld(out), rows, n2 // 2, _stream()), "geglu_fwd")
    return out


def geglu_bwd(pre, dy, out=None):
    _chk_bf16(pre, dy)
    rows, n2 = pre.shape
    if out is None:
        out = torch.empty_like(pre)
    hip.check(hip.lib().lvdhip_geglu_bwd(_p(pre), _ld(pre), _p(dy), _ld(dy), _p(out), _ld(out), rows, n2 // 2, _stream()), "geglu_bwd")
    return out


def upsample2x_bwd(dy, nimg, h, w, c, dx=None, accumulate=False):
    _chk_bf16(dy)
    assert dy.is_contiguous()
    if dx is None:
        dx = torch.empty((nimg * h * w, c), dtype=torch.bfloat16, device=dy.device)
    assert dx.is_contiguous()
    hip.check(hip.lib().lvdhip_upsample2x_bwd(_p(dy), _p(dx), nimg, h, w, c, int(accumulate), _stream()), "upsample2x_bwd")
    return dx


def timestep_embedding(t, dim):
    _chk_f32(t)
    out = torch.empty((t.numel(), dim), dtype=torch.bfloat16, device=t.device)
    hip.check(hip.lib().lvdhip_timestep_embedding(_p(t), _p(out), t.numel(), dim, _stream()), "timestep_embedding")
    return out


def silu(x):
    _chk_bf16(x)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    hip.check(hip.lib().lvdhip_silu(_p(x), _p(y), x.numel(), _stream()), "silu")
    return y


def cfg_dpm_step(eps_uncond, eps_cond, guidance_scale, x, x0_prev, alpha_t, sigma_t, c_x, c_0, c_1):
    _chk_f32(eps_uncond, eps_cond, x, x0_prev)
    hip.check(hip.lib().lvdhip_cfg_dpm_step(_p(eps_uncond), _p(eps_cond), guidance_scale, _p(x), _p(x0_prev), alpha_t, sigma_t,
                                            c_x, c_0, c_1, x.numel(), _stream()), "cfg_dpm_step")
    return x


def axpy_(x, g, scale):
    _chk_f32(x, g)
    hip.check(hip.lib().lvdhip_axpy(_p(x), _p(g), scale, x.numel(), _stream()), "axpy")
    return x


def reduce_sum(x, scale=1.0, out=None):
    _chk_f32(x)
    if out is None:
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().lvdhip_reduce_sum(_p(x), x.numel(), scale, _p(out), _stream()), "reduce_sum")
    return out


def softmax_rows(x, out=None):
    """Row softmax of an fp32 score matrix -> bf16 probabilities (VAE mid-block attention)."""
    _chk_f32(x)
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    hip.check(hip.lib().lvdhip_softmax_rows(_p(x), _ld(x), _p(out), _ld(out), rows, cols, _stream()), "softmax_rows")
    return out


def tokens_to_video(tokens, frames, height, width):
    """Decoded image tokens [(f,y,x), >=4] bf16 -> (frames, height, width, 3) fp32 in [0,1]."""
    _chk_bf16(tokens)
    rows = frames * height * width
    assert tokens.shape[0] == rows
    video = torch.empty((frames, height, width, 3), dtype=torch.float32, device=tokens.device)
    hip.check(hip.lib().lvdhip_tokens_to_video(_p(tokens), _ld(tokens), _p(video), rows, _stream()), "tokens_to_video")
    return video


def gelu(x, mode="gelu", out=None):
    """Elementwise GELU on a bf16 matrix: "gelu" (exact erf) or "quick_gelu" (x * sigmoid(1.702 x))."""
    _chk_bf16(x)
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    hip.check(hip.lib().lvdhip_gelu(_p(x), _p(out), x.numel(), {"gelu": 0, "quick_gelu": 1}[mode], _stream()), "gelu")
    return out


# ---- OWL-ViT scoring ends (detect.hip) ----------------------------------------------------------------------------------

def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _lanczos(x):
    import math
    sinc = lambda v: 1.0 if v == 0.0 else math.sin(v * math.pi) / (v * math.pi)
    return sinc(x) * sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}
_resample_tables = {}


def pil_resample_table(in_size, out_size, kind="bicubic"):
    """Pillow's 8-bit resampling coefficients for one axis (Resample.c precompute_coeffs + normalize_coeffs_8bpc; bicubic with
    a = -0.5 / support 2, Lanczos with support 3, both widened by the downscale factor): bounds int32 [out, 2] = (first tap,
    taps), coef int32 [out, ksize] with 22 fractional bits.  Evaluated in double precision in the library's operation order, so
    the integers are its own."""
    key = (in_size, out_size, kind)
    if key not in _resample_tables:
        import math
        import numpy as np
        fn, base_support = _FILTERS[kind]
        scale = in_size / out_size
        filterscale = max(scale, 1.0)
        support = base_support * filterscale
        ksize = int(math.ceil(support)) * 2 + 1
        bounds = np.zeros((out_size, 2), dtype=np.int32)
        coef = np.zeros((out_size, ksize), dtype=np.int32)
        inv = 1.0 / filterscale
        for o in range(out_size):
            center = (o + 0.5) * scale
            lo = max(int(center - support + 0.5), 0)
            hi = min(int(center + support + 0.5), in_size)
            w = [fn((t + lo - center + 0.5) * inv) for t in range(hi - lo)]
            total = 0.0
            for v in w:
                total += v
            if total != 0.0:
                w = [v / total for v in w]
            bounds[o] = (lo, hi - lo)
            coef[o, :hi - lo] = [int(v * (1 << 22) + (0.5 if v >= 0 else -0.5)) for v in w]
        _resample_tables[key] = (bounds, coef)
    return _resample_tables[key]


def pil_bicubic_table(in_size, out_size):
    return pil_resample_table(in_size, out_size, "bicubic")


def frames_to_patches(frames, size, patch, mean, std, return_resized=False, kind="bicubic", width=None):
    """uint8 frames (B,H,W,3) on the GPU -> bf16 patch matrix [B*(SH/patch)*(SW/patch), width >= 3*patch^2] (PIL-exact resize to
    `size` = S or (SH, SW), rescale, normalise).  Columns beyond 3*patch^2 (`width`, e.g. 8 for a conv_in token matrix) are zero."""
    assert frames.dtype == torch.uint8 and frames.is_cuda and frames.is_contiguous() and frames.ndim == 4 and frames.shape[-1] == 3
    B, H, W, _ = frames.shape
    SH, SW = (size, size) if isinstance(size, int) else size
    dev = frames.device
    tabs = []
    for n, s in ((W, SW), (H, SH)):
        key = (n, s, kind, dev)
        if key not in _resample_tables:
            b, k = pil_resample_table(n, s, kind)
            _resample_tables[key] = (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev))
        tabs.append(_resample_tables[key])
    (xb, xk), (yb, yk) = tabs
    cols = 3 * patch * patch
    width = cols if width is None else width
    alloc = torch.empty if width == cols else torch.zeros
    patches = alloc((B * (SH // patch) * (SW // patch), width), dtype=torch.bfloat16, device=dev)
    resized = torch.empty((B, SH, SW, 3), dtype=torch.uint8, device=dev) if return_resized else None
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    hip.check(hip.lib().lvdhip_frames_to_patches(_p(frames), B, H, W, SH, SW, patch, _p(xb), _p(xk), xk.shape[1], _p(yb), _p(yk), yk.shape[1],
                                                 m3, s3, _p(patches), _ld(patches), _p(resized), _stream()), "frames_to_patches")
    return (patches, resized) if return_resized else patches


def owl_detect_rows(class_embeds, queries, shift_scale, box_raw, box_bias, tokens_per_image, img_w, img_h, query_mask=None):
    """Fused tail of the OWL-ViT heads + post-process (see lvdhip_owl_detect_rows).  Returns (logits [rows,Q], scores [rows],
    labels int64 [rows], boxes xyxy pixels [rows,4])."""
    _chk_f32(class_embeds, queries, shift_scale, box_raw, box_bias)
    rows, D = class_embeds.shape
    Q = queries.shape[0]
    assert queries.is_contiguous() and queries.shape[1] == D and box_bias.is_contiguous() and box_bias.shape == (tokens_per_image, 4)
    dev = class_embeds.device
    logits = torch.empty((rows, Q), dtype=torch.float32, device=dev)
    scores = torch.empty((rows,), dtype=torch.float32, device=dev)
    labels = torch.empty((rows,), dtype=torch.int64, device=dev)
    boxes = torch.empty((rows, 4), dtype=torch.float32, device=dev)
    hip.check(hip.lib().lvdhip_owl_detect_rows(_p(class_embeds), _ld(class_embeds), D, _p(queries), Q, _p(query_mask), _p(shift_scale), _ld(shift_scale),
                                               _p(box_raw), _ld(box_raw), _p(box_bias), tokens_per_image, float(img_w), float(img_h), rows,
                                               _p(logits), _p(scores), _p(labels), _p(boxes), _stream()), "owl_detect_rows")
    return logits, scores, labels, boxes
