"""Backward guidance on the HIP engine: drop-in for the reference's ``latent_backward_guidance``.

Reference interfaces mirrored here
  models/pipelines.py:21-150                latent_backward_guidance(scheduler, unet, cond_embeddings, index, bboxes,
                                            object_positions, t, latents, loss, loss_scale, loss_threshold, max_iter,
                                            max_index_step, cross_attention_kwargs, guidance_attn_keys, verbose,
                                            return_saved_attn, clear_cache, **kwargs) -> (latents, loss[, saved_attn])
  utils/guidance.py:529-574,160-526         compute_ca_lossv3 / add_ca_loss_per_attn_map_to_loss (max-based top-k
                                            energy + centre-of-mass terms)
  models/controllable_pipeline_text_to_video_synth.py:572,827-831   the ``custom_latent_backward_guidance`` hook

What runs on the GPU per guidance iteration: a recorded forward of the cond branch that stops right after the query
projection of the last guidance key, three fused loss kernels per key (csrc/guidance_loss.hip) that produce the loss
and d(loss)/dQ without ever materialising an attention map, and the hand-scheduled input-gradient tape back to the
latents.  No torch autograd graph exists at any point.
"""
import math
import warnings
import ctypes as C

import numpy as np
import torch

from . import hip, ops
from .engine import HipUNet3D, Tape, Geom

DEFAULT_GUIDANCE_ATTN_KEYS = [("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0)]  # models/pipelines.py:13-18

# loss options of utils/guidance.py:160-526 that are NOT built: they raise (upsample_scale != 1 raises RuntimeError in the reference's own max-based /
# CE forms; exclude_bg_heads is an assertion failure there)
_UNSUPPORTED = dict(exclude_bg_heads=False, upsample_scale=1)
# optional terms that ARE built into the fused loss kernel (generate.py:78-106, generation/lvd.py:85-106 forward them)
_LOSS_OPTIONS = ("fg_top_p", "bg_top_p", "fg_weight", "bg_weight", "com_loss_scale", "use_ratio_based_loss", "attn_sync_weight",
                 "boxdiff_loss_scale", "boxdiff_normed", "boxdiff_L", "use_max_based_loss", "use_ce_based_loss",
                 # map-level options (utils/guidance.py:209-226): whole-map path, csrc/guidance_maps.hip
                 "smooth_attn", "kernel_size", "sigma", "attn_renorm", "num_tokens", "renorm_scale")
_MAP_OPTIONS = ("smooth_attn", "kernel_size", "sigma", "attn_renorm", "num_tokens", "renorm_scale")


def _energy_form(use_ratio_based_loss, use_max_based_loss, use_ce_based_loss):
    """The if / elif chain of utils/guidance.py:312,346,363,398 -> lvd_ca_select_params.use_ratio_loss (0 max, 1 ratio, 2 CE)."""
    if use_ratio_based_loss:
        return 1
    if use_max_based_loss:
        return 0
    if use_ce_based_loss:
        return 2
    raise ValueError("Unknown loss: no loss selected")


def scale_proportion(box, H, W):
    """utils/utils.py:82-103 (non-legacy branch; Python banker's rounding is part of the contract)."""
    x_min, y_min = round(box[0] * W), round(box[1] * H)
    bw, bh = round((box[2] - box[0]) * W), round((box[3] - box[1]) * H)
    x_max, y_max = x_min + bw, y_min + bh
    return max(x_min, 0), max(y_min, 0), min(x_max, W), min(y_max, H)


def _topk_count(n, p):
    # (mask.sum() * top_p).long().clamp_(min=1) with a float32 mask sum (utils/guidance.py:328-337)
    return max(1, int(np.float32(n) * np.float32(p)))


def ca_probability_maps(q, k, *, samples, heads, positions, ntext, samples_per_key=1, scale=0.125, key_bias=None):
    """softmax(scale * Q K^T) over all text positions as the reference's slow path materialises and saves it
    (models/attention_processor.py:515-552 -> :553-586): q [samples*positions, heads*64] bf16, k [(samples / samples_per_key) * ntext,
    heads*64] bf16 -> (samples, heads, positions, ntext) fp32, one launch of `lvdhip_ca_probs_full`."""
    ops._chk_bf16(q, k)
    if q.shape[0] != samples * positions or q.shape[1] != heads * 64 or k.shape[1] != heads * 64 or samples % samples_per_key \
            or k.shape[0] != (samples // samples_per_key) * ntext:
        raise ValueError(f"ca_probability_maps: q {tuple(q.shape)} / k {tuple(k.shape)} do not match samples={samples}, heads={heads}, "
                         f"positions={positions}, ntext={ntext}, samples_per_key={samples_per_key}")
    probs = torch.empty((samples, heads, positions, ntext), dtype=torch.float32, device=q.device)
    a = hip.CaProbsFullParams(q=q.data_ptr(), ldq=q.stride(0), k=k.data_ptr(), ldk=k.stride(0), samples=samples, heads=heads, P=positions,
                              ntext=ntext, samples_per_key=samples_per_key, scale=float(scale), probs=probs.data_ptr())
    if key_bias is not None:  # (samples, ntext) fp32 additive score bias per text position: the cross-attention attention_mask
        if key_bias.dtype != torch.float32 or tuple(key_bias.shape) != (samples, ntext) or key_bias.stride(1) != 1 or key_bias.device != q.device:
            raise ValueError(f"ca_probability_maps: key_bias must be fp32 ({samples}, {ntext}) on the queries' device, got {tuple(key_bias.shape)} {key_bias.dtype}")
        a.key_bias, a.ld_key_bias = key_bias.data_ptr(), key_bias.stride(0)
    ops.use_device(q.device)
    hip.check(hip.lib().lvdhip_ca_probs_full(C.byref(a), torch.cuda.current_stream().cuda_stream), "ca_probs_full")
    return probs


def ca_apply_probabilities(probs, v, *, samples, heads, positions, ntext, samples_per_key=1, out=None):
    """bmm(attention_probs, value) of the slow path (models/attention_processor.py:549) for probabilities held in memory (after an
    `attn_process_fn`): probs (samples, heads, positions, ntext) fp32, v [(samples / samples_per_key) * ntext, heads*64] bf16 ->
    [samples*positions, heads*64] bf16 (`lvdhip_ca_apply_probs`)."""
    ops._chk_bf16(v)
    probs = probs.reshape(samples, heads, positions, ntext)
    if probs.dtype != torch.float32 or not probs.is_contiguous():
        probs = probs.float().contiguous()
    if v.shape[1] != heads * 64 or samples % samples_per_key or v.shape[0] != (samples // samples_per_key) * ntext or probs.device != v.device:
        raise ValueError(f"ca_apply_probabilities: probs {tuple(probs.shape)} / v {tuple(v.shape)} do not match samples_per_key={samples_per_key}")
    if out is None:
        out = torch.empty((samples * positions, heads * 64), dtype=torch.bfloat16, device=v.device)
    a = hip.CaApplyProbsParams(probs=probs.data_ptr(), v=v.data_ptr(), ldv=v.stride(0), samples=samples, heads=heads, P=positions, ntext=ntext,
                               samples_per_key=samples_per_key, out=out.data_ptr(), ldo=out.stride(0))
    ops.use_device(v.device)
    hip.check(hip.lib().lvdhip_ca_apply_probs(C.byref(a), torch.cuda.current_stream().cuda_stream), "ca_apply_probs")
    return out


class GuidanceLayout:
    """Device-side description of the boxes / object tokens for one attention resolution (H, W)."""

    def __init__(self, bboxes, object_positions, frames, H, W, fg_top_p, bg_top_p, device):
        nobj = len(bboxes)
        arr = np.zeros((nobj, frames, 6), dtype=np.int32)
        for o, obj_boxes in enumerate(bboxes):
            assert len(obj_boxes) == frames, f"Number of frames {frames} mismatches with number of frames in box condition {len(obj_boxes)}"
            for f, box in enumerate(obj_boxes):
                x0, y0, x1, y1 = scale_proportion(box, H, W)
                n = max(0, x1 - x0) * max(0, y1 - y0)
                arr[o, f] = (x0, y0, x1, y1, _topk_count(n, fg_top_p), _topk_count(H * W - n, bg_top_p))
        tok_ids, tok_obj, tok_w = [], [], []
        for o, positions in enumerate(object_positions):
            for pos in positions:
                tok_ids.append(int(pos))
                tok_obj.append(o)
                tok_w.append(1.0 / len(positions))
        self.nobj, self.ntok, self.H, self.W = nobj, len(tok_ids), H, W
        self.boxes = torch.from_numpy(arr).to(device)
        self.tok_ids_host = np.asarray(tok_ids, dtype=np.int64)
        self.tok_ids = torch.tensor(tok_ids, dtype=torch.int32, device=device)
        self.tok_obj = torch.tensor(tok_obj, dtype=torch.int32, device=device)
        self.tok_weight = torch.tensor(tok_w, dtype=torch.float32, device=device)

    def token_slice(self, a, b):
        """The same boxes with object tokens [a, b) only (views: nothing is copied)."""
        sub = object.__new__(GuidanceLayout)
        sub.nobj, sub.ntok, sub.H, sub.W, sub.boxes = self.nobj, b - a, self.H, self.W, self.boxes
        sub.tok_ids_host, sub.tok_ids, sub.tok_obj, sub.tok_weight = self.tok_ids_host[a:b], self.tok_ids[a:b], self.tok_obj[a:b], self.tok_weight[a:b]
        return sub


loss_kernel_events = None  # measurement hook (bench.py sets a list): (start, end, algorithmic bytes) per fused-loss launch set
MAX_TOKENS_PER_LAUNCH = 16  # csrc/guidance_loss.hip MAXTOK: object-token columns one launch keeps per query


def ca_energy_loss_and_dq(q, k, heads, frames, layout: GuidanceLayout, *, ntext, grad_scale, fg_weight, bg_weight,
                          com_loss_scale, loss_partial, want_dq=True, use_ratio_based_loss=False, attn_sync_weight=0.0,
                          boxdiff_loss_scale=0.0, boxdiff_normed=True, boxdiff_L=1, use_max_based_loss=True, use_ce_based_loss=False, _acc=None,
                          **map_options):
    """One guidance key: q [frames*P, heads*64] bf16, k [ntext, heads*64] bf16 (strided ok).

    Writes the per-(frame, head, token) loss terms into ``loss_partial`` [frames*heads*ntok] and returns dQ.  Layouts with
    more object tokens than one launch holds run in chunks of tokens: the loss terms are per token and dQ is linear in them.
    `smooth_attn` / `attn_renorm` (and their parameters) take the whole-map path, ca_energy_loss_and_dq_maps."""
    unknown = set(map_options) - set(_MAP_OPTIONS)
    if unknown:
        raise TypeError(f"ca_energy_loss_and_dq: unexpected options {sorted(unknown)}")
    if map_options.get("smooth_attn") or map_options.get("attn_renorm"):
        assert _acc is None and want_dq
        return ca_energy_loss_and_dq_maps(q, k, heads, frames, layout, ntext=ntext, grad_scale=grad_scale, fg_weight=fg_weight, bg_weight=bg_weight,
                                          com_loss_scale=com_loss_scale, loss_partial=loss_partial, use_ratio_based_loss=use_ratio_based_loss,
                                          attn_sync_weight=attn_sync_weight, boxdiff_loss_scale=boxdiff_loss_scale, boxdiff_normed=boxdiff_normed,
                                          boxdiff_L=boxdiff_L, use_max_based_loss=use_max_based_loss, use_ce_based_loss=use_ce_based_loss, **map_options)
    if layout.ntok and int(layout.tok_ids_host.max()) >= ntext:  # the reference indexes attn[..., pos] and raises the same way
        raise IndexError(f"object token position {int(layout.tok_ids_host.max())} is out of bounds for {ntext} text tokens")
    if layout.ntok > MAX_TOKENS_PER_LAUNCH and _acc is None:
        # dQ is linear in the tokens: the chunks are summed in an fp32 accumulator by the dq kernel itself (first chunk stores, middle
        # chunks add, the last one adds and rounds to bf16 once)
        starts = list(range(0, layout.ntok, MAX_TOKENS_PER_LAUNCH))
        acc32 = torch.empty((q.shape[0], q.shape[1]), dtype=torch.float32, device=q.device) if want_dq else None
        dq, off = None, 0
        for ci, c0 in enumerate(starts):
            sub = layout.token_slice(c0, min(layout.ntok, c0 + MAX_TOKENS_PER_LAUNCH))
            n = frames * heads * sub.ntok
            mode = 1 if ci == 0 else (3 if ci == len(starts) - 1 else 2)
            dq = ca_energy_loss_and_dq(q, k, heads, frames, sub, ntext=ntext, grad_scale=grad_scale, fg_weight=fg_weight, bg_weight=bg_weight,
                                       com_loss_scale=com_loss_scale, loss_partial=loss_partial[off:off + n], want_dq=want_dq,
                                       use_ratio_based_loss=use_ratio_based_loss, attn_sync_weight=attn_sync_weight,
                                       boxdiff_loss_scale=boxdiff_loss_scale, boxdiff_normed=boxdiff_normed, boxdiff_L=boxdiff_L,
                                       use_max_based_loss=use_max_based_loss, use_ce_based_loss=use_ce_based_loss, _acc=(acc32, mode))
            off += n
        return dq
    a, b, c, keep = _ca_params(q, k, heads, frames, layout, ntext=ntext, grad_scale=grad_scale, fg_weight=fg_weight, bg_weight=bg_weight,
                               com_loss_scale=com_loss_scale, loss_partial=loss_partial, want_dq=want_dq, use_ratio_based_loss=use_ratio_based_loss,
                               attn_sync_weight=attn_sync_weight, boxdiff_loss_scale=boxdiff_loss_scale, boxdiff_normed=boxdiff_normed,
                               boxdiff_L=boxdiff_L, use_max_based_loss=use_max_based_loss, use_ce_based_loss=use_ce_based_loss, _acc=_acc)
    st = torch.cuda.current_stream().cuda_stream
    hip.check(hip.lib().lvdhip_ca_probs(C.byref(a), st), "ca_probs")
    hip.check(hip.lib().lvdhip_ca_select(C.byref(b), st), "ca_select")
    if not want_dq:
        return None
    hip.check(hip.lib().lvdhip_ca_dq(C.byref(c), st), "ca_dq")
    return keep["dq"]


def _ca_params(q, k, heads, frames, layout, *, ntext, grad_scale, fg_weight, bg_weight, com_loss_scale, loss_partial, want_dq,
               use_ratio_based_loss, attn_sync_weight, boxdiff_loss_scale, boxdiff_normed, boxdiff_L, use_max_based_loss=True,
               use_ce_based_loss=False, _acc=None, dq_out=None):
    """The three parameter blocks of one key (probabilities, selection / loss, dQ) and the buffers they point into (`keep`)."""
    dev = q.device
    P = layout.H * layout.W
    assert q.shape[0] == frames * P, (q.shape, frames, P)
    probs = torch.empty((frames, heads, layout.ntok, P), dtype=torch.float32, device=dev)
    lse = torch.empty((frames, heads, P), dtype=torch.float32, device=dev)
    a = hip.CaProbsParams()
    a.q, a.ldq, a.k, a.ldk = q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0)
    a.frames, a.heads, a.P, a.ntext, a.scale = frames, heads, P, ntext, 0.125
    a.tok_ids, a.ntok, a.probs, a.lse = layout.tok_ids.data_ptr(), layout.ntok, probs.data_ptr(), lse.data_ptr()

    dprobs = torch.empty_like(probs)
    com_ws = torch.empty((frames, heads, layout.ntok, 4), dtype=torch.float32, device=dev)
    b = hip.CaSelectParams()
    b.probs, b.dprobs = probs.data_ptr(), dprobs.data_ptr()
    b.frames, b.heads, b.P, b.ntok, b.H, b.W = frames, heads, P, layout.ntok, layout.H, layout.W
    b.tok_obj, b.boxes, b.tok_weight, b.nobj = layout.tok_obj.data_ptr(), layout.boxes.data_ptr(), layout.tok_weight.data_ptr(), layout.nobj
    b.fg_weight, b.bg_weight, b.com_loss_scale, b.grad_scale = fg_weight, bg_weight, com_loss_scale, grad_scale
    b.loss_partial, b.com_ws = loss_partial.data_ptr(), com_ws.data_ptr()
    b.use_ratio_loss, b.ratio_eps, b.attn_sync_weight = _energy_form(use_ratio_based_loss, use_max_based_loss, use_ce_based_loss), 1.0e-2, float(attn_sync_weight)
    b.boxdiff_loss_scale, b.boxdiff_normed, b.boxdiff_L = float(boxdiff_loss_scale), int(bool(boxdiff_normed)), int(boxdiff_L)
    if use_ratio_based_loss:
        warnings.warn("Using ratio-based loss, which is deprecated. Max-based loss is recommended. The scale may be different.")
    keep = dict(probs=probs, lse=lse, dprobs=dprobs, com_ws=com_ws, dq=None)
    c = None
    if want_dq:
        acc32, mode = _acc if _acc is not None else (None, 0)
        if dq_out is not None:  # the caller's rows of a larger dQ matrix (several samples recorded in one forward)
            assert mode == 0 and dq_out.shape == q.shape and dq_out.dtype == torch.bfloat16 and dq_out.stride(1) == 1
            dq = dq_out
        else:
            dq = torch.empty((q.shape[0], q.shape[1]), dtype=torch.bfloat16, device=dev) if mode in (0, 3) else None
        c = hip.CaDqParams()
        c.q, c.ldq, c.k, c.ldk = a.q, a.ldq, a.k, a.ldk
        c.frames, c.heads, c.P, c.ntext, c.scale = frames, heads, P, ntext, 0.125
        c.tok_ids, c.ntok = a.tok_ids, a.ntok
        c.probs, c.dprobs, c.lse = probs.data_ptr(), dprobs.data_ptr(), lse.data_ptr()
        if dq is not None:
            c.dq, c.lddq = dq.data_ptr(), dq.stride(0)
        else:  # accumulate-only chunk: the bf16 output is not written (any valid pointer)
            c.dq, c.lddq = acc32.data_ptr(), 0
        if acc32 is not None:
            c.acc32, c.ldacc, c.acc_mode = acc32.data_ptr(), acc32.stride(0), mode
        keep["dq"] = dq
    return a, b, c, keep


def gaussian_kernel_3x3(sigma):
    """utils/attn.py GaussianSmoothing(kernel_size=3, sigma): per dimension 1 / (s sqrt(2 pi)) * exp(-((x - mean) / (2 s))^2) — the reference's
    own exponent, not the textbook one —, outer product, normalised to sum 1; row-major [position tap][token tap]."""
    x = torch.arange(3, dtype=torch.float32)
    g = 1.0 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-(((x - 1.0) / (2 * sigma)) ** 2))
    k = g[:, None] * g[None, :]
    return (k / k.sum()).reshape(9).contiguous()


def ca_maps_loss_and_grad(a0, heads, frames, layout: GuidanceLayout, *, grad_scale, fg_weight, bg_weight, com_loss_scale, loss_partial,
                          smooth_attn=False, kernel_size=3, sigma=0.5, attn_renorm=False, num_tokens=None, renorm_scale=2.0, **term_options):
    """The energy on WHOLE probability maps a0 [frames, heads, P, T] fp32 with the map-level options of add_ca_loss_per_attn_map_to_loss
    (utils/guidance.py:209-226): [3x3 Gaussian, reflect-padded] -> [softmax of renorm_scale * tokens 1 .. num_tokens-2] -> object columns -> the
    unchanged selection / loss kernel (loss terms into `loss_partial`), and the same chain backwards.  Returns d(loss)/d(a0), same shape
    (already scaled by grad_scale like the selection kernel's dA).  csrc/guidance_maps.hip."""
    dev = a0.device
    P = layout.H * layout.W
    rows, T = frames * heads, a0.shape[-1]
    assert a0.dtype == torch.float32 and a0.is_contiguous() and tuple(a0.shape) == (frames, heads, P, T), a0.shape
    ops.use_device(dev)
    st = torch.cuda.current_stream().cuda_stream
    lib = hip.lib()
    m = a0
    w9 = None
    if smooth_attn:
        if int(kernel_size) != 3:  # the reference pads by one position / token whatever the kernel size and asserts the shape is kept
            raise NotImplementedError(f"smooth_attn with kernel_size={kernel_size}: only the 3x3 kernel keeps the map's shape (utils/guidance.py:213-220)")
        w9 = gaussian_kernel_3x3(float(sigma)).to(dev)
        sm = torch.empty_like(m)
        hip.check(lib.lvdhip_ca_map_smooth(m.data_ptr(), sm.data_ptr(), rows, P, T, w9.data_ptr(), 0, st), "ca_map_smooth")
        m = sm
    cols_host = layout.tok_ids_host.astype(np.int64)
    renorm_in = None
    limit = T
    if attn_renorm:
        if term_options.get("attn_sync_weight", 0.0) != 0.0:
            raise AssertionError("attn_sync with attn_renorm not implemented together")  # the reference's own assertion, :404-406
        if num_tokens is None:
            raise TypeError("attn_renorm needs num_tokens (the reference slices attn_map[..., 1:num_tokens - 1])")
        limit = int(num_tokens) - 2
        if limit < 1 or limit + 1 > T:
            raise IndexError(f"num_tokens={num_tokens} leaves no token range inside the map's {T} text positions")
        cols_host = cols_host - 1  # "Since we removed SOT, we subtract 1 from obj_position", :291-294
        rn = torch.empty_like(m)
        hip.check(lib.lvdhip_ca_map_renorm(m.data_ptr(), None, rn.data_ptr(), rows, P, T, 1, limit, float(renorm_scale), 0, st), "ca_map_renorm")
        renorm_in, m = m, rn
    if layout.ntok and (int(cols_host.min()) < 0 or int(cols_host.max()) >= limit):  # the reference indexes attn_map[..., pos]
        raise IndexError(f"object token position {int(layout.tok_ids_host.max())} is out of bounds for the {limit} tokens of the map")
    cols = torch.from_numpy(cols_host.astype(np.int32)).to(dev)
    probs = torch.empty((frames, heads, layout.ntok, P), dtype=torch.float32, device=dev)
    dprobs = torch.empty_like(probs)
    com_ws = torch.empty((frames, heads, layout.ntok, 4), dtype=torch.float32, device=dev)
    b = hip.CaSelectParams()
    b.probs, b.dprobs = probs.data_ptr(), dprobs.data_ptr()
    b.frames, b.heads, b.P, b.ntok, b.H, b.W = frames, heads, P, layout.ntok, layout.H, layout.W
    b.tok_obj, b.boxes, b.tok_weight, b.nobj = layout.tok_obj.data_ptr(), layout.boxes.data_ptr(), layout.tok_weight.data_ptr(), layout.nobj
    b.fg_weight, b.bg_weight, b.com_loss_scale, b.grad_scale = fg_weight, bg_weight, com_loss_scale, grad_scale
    b.loss_partial, b.com_ws = loss_partial.data_ptr(), com_ws.data_ptr()
    b.use_ratio_loss = _energy_form(term_options.get("use_ratio_based_loss", False), term_options.get("use_max_based_loss", True),
                                    term_options.get("use_ce_based_loss", False))
    b.ratio_eps, b.attn_sync_weight = 1.0e-2, float(term_options.get("attn_sync_weight", 0.0))
    b.boxdiff_loss_scale, b.boxdiff_normed, b.boxdiff_L = (float(term_options.get("boxdiff_loss_scale", 0.0)), int(bool(term_options.get("boxdiff_normed", True))),
                                                           int(term_options.get("boxdiff_L", 1)))
    hip.check(lib.lvdhip_ca_map_gather_cols(m.data_ptr(), cols.data_ptr(), layout.ntok, probs.data_ptr(), rows, P, T, st), "ca_map_gather_cols")
    hip.check(lib.lvdhip_ca_select(C.byref(b), st), "ca_select")
    dm = torch.empty_like(a0)
    hip.check(lib.lvdhip_ca_map_scatter_cols(dprobs.data_ptr(), cols.data_ptr(), layout.ntok, dm.data_ptr(), rows, P, T, st), "ca_map_scatter_cols")
    if attn_renorm:
        back = torch.empty_like(a0)
        hip.check(lib.lvdhip_ca_map_renorm(renorm_in.data_ptr(), dm.data_ptr(), back.data_ptr(), rows, P, T, 1, limit, float(renorm_scale), 1, st), "ca_map_renorm")
        dm = back
    if smooth_attn:
        back = torch.empty_like(a0)
        hip.check(lib.lvdhip_ca_map_smooth(dm.data_ptr(), back.data_ptr(), rows, P, T, w9.data_ptr(), 1, st), "ca_map_smooth")
        dm = back
    return dm


def ca_energy_loss_and_dq_maps(q, k, heads, frames, layout: GuidanceLayout, *, ntext, dq_out=None, **options):
    """One guidance key with `smooth_attn` and / or `attn_renorm`: probabilities of ALL text positions (lvdhip_ca_probs_full), the map chain of
    ca_maps_loss_and_grad, the softmax backward on the whole map and dQ = dS . K (lvdhip_ca_apply_probs).  Same contract as ca_energy_loss_and_dq
    (loss terms into `loss_partial`, returns dQ).  Not a fast path: no entry point of the reference switches these options on."""
    P = layout.H * layout.W
    a0 = ca_probability_maps(q, k, samples=frames, heads=heads, positions=P, ntext=ntext, samples_per_key=frames)  # [frames, heads, P, T] fp32
    dm = ca_maps_loss_and_grad(a0, heads, frames, layout, **options)
    ds = torch.empty_like(a0)
    hip.check(hip.lib().lvdhip_ca_map_softmax_bwd(a0.data_ptr(), dm.data_ptr(), ds.data_ptr(), frames * heads, P, ntext, 0.125,
                                                  torch.cuda.current_stream().cuda_stream), "ca_map_softmax_bwd")
    return ca_apply_probabilities(ds, k, samples=frames, heads=heads, positions=P, ntext=ntext, samples_per_key=frames, out=dq_out)


def ca_energy_loss_and_dq_all_keys(items, frames, *, ntext, grad_scale, fg_weight, bg_weight, com_loss_scale, **loss_options):
    """Every key of a guidance iteration in ONE launch per stage (probabilities / selection + loss / dQ: 3 launches instead of 3 per key;
    csrc/guidance_loss.hip *_multi).  `items`: (q, k, heads, layout, loss_partial slice[, dq rows to write]) per key.  Same bits as the
    key-by-key calls.  Layouts with more object tokens than a launch holds (chunked dQ accumulation) take the key-by-key path."""
    items = [it if len(it) == 6 else (*it, None) for it in items]
    map_opts = {o: loss_options.pop(o) for o in _MAP_OPTIONS if o in loss_options}
    if map_opts.get("smooth_attn") or map_opts.get("attn_renorm"):  # whole-map path, key by key (csrc/guidance_maps.hip)
        return [ca_energy_loss_and_dq_maps(q, k, heads, frames, lay, ntext=ntext, grad_scale=grad_scale, fg_weight=fg_weight, bg_weight=bg_weight,
                                           com_loss_scale=com_loss_scale, loss_partial=part, dq_out=dq_out, **map_opts, **loss_options)
                for q, k, heads, lay, part, dq_out in items]
    if len(items) > hip.CA_MAX_KEYS or any(lay.ntok > MAX_TOKENS_PER_LAUNCH for _, _, _, lay, _, _ in items):
        outs = []
        for q, k, heads, lay, part, dq_out in items:
            dq = ca_energy_loss_and_dq(q, k, heads, frames, lay, ntext=ntext, grad_scale=grad_scale, fg_weight=fg_weight, bg_weight=bg_weight,
                                       com_loss_scale=com_loss_scale, loss_partial=part, **loss_options)
            if dq_out is not None:
                dq_out.copy_(dq)
                dq = dq_out
            outs.append(dq)
        return outs
    opts = dict(use_ratio_based_loss=False, attn_sync_weight=0.0, boxdiff_loss_scale=0.0, boxdiff_normed=True, boxdiff_L=1)
    opts.update(loss_options)
    n = len(items)
    A, B, Cq = (hip.CaProbsParams * n)(), (hip.CaSelectParams * n)(), (hip.CaDqParams * n)()
    keeps = []
    for i, (q, k, heads, lay, part, dq_out) in enumerate(items):
        if lay.ntok and int(lay.tok_ids_host.max()) >= ntext:  # the reference indexes attn[..., pos] and raises the same way
            raise IndexError(f"object token position {int(lay.tok_ids_host.max())} is out of bounds for {ntext} text tokens")
        a, b, c, keep = _ca_params(q, k, heads, frames, lay, ntext=ntext, grad_scale=grad_scale, fg_weight=fg_weight, bg_weight=bg_weight,
                                   com_loss_scale=com_loss_scale, loss_partial=part, want_dq=True, dq_out=dq_out, **opts)
        A[i], B[i], Cq[i] = a, b, c
        keeps.append(keep)
    st = torch.cuda.current_stream().cuda_stream
    ev = None
    if loss_kernel_events is not None:  # bench.py: the three launches of an iteration bracketed by events on the launch stream
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    hip.check(hip.lib().lvdhip_ca_probs_multi(A, n, st), "ca_probs_multi")
    hip.check(hip.lib().lvdhip_ca_select_multi(B, n, st), "ca_select_multi")
    hip.check(hip.lib().lvdhip_ca_dq_multi(Cq, n, st), "ca_dq_multi")
    if ev is not None:
        ev[1].record()
        loss_kernel_events.append((ev[0], ev[1], sum(2 * q.shape[0] * q.shape[1] * 2 for q, *_ in items)))  # algorithmic bytes: read Q + write dQ
    return [kp["dq"] for kp in keeps]


def guidance_loss_and_grad(engine: HipUNet3D, latents, t, text, bboxes, object_positions, guidance_attn_keys, *, loss_scale,
                           fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=1.0, com_loss_scale=0.0, latent_scale=1.0, saved_attn=None,
                           **loss_options):
    """One recorded forward + fused loss + hand-scheduled backward.

    Returns (loss [1] fp32 device tensor, already multiplied by loss_scale; grad (1,4,F,h,w) fp32)."""
    assert latents.shape[0] == 1, "guidance runs on the cond branch only (batch 1), models/pipelines.py:78"
    keys = [tuple(k) for k in guidance_attn_keys]
    tape = Tape()
    collect = {"keys": set(keys), "q": {}, "stop_after": _last_key_in_order(engine, keys)}
    engine.forward(latents, t, text=text, tape=tape, collect=collect)
    missing = [k for k in keys if k not in collect["q"]]
    if missing:
        raise KeyError(f"guidance keys not produced by this UNet: {missing}")
    nobj = len(bboxes)
    frames = latents.shape[2]
    grad_scale = loss_scale / (nobj * len(keys))
    layouts = {}
    sizes = []
    for key in keys:
        q, k, heads, g = collect["q"][key]
        sizes.append(frames * heads * sum(len(p) for p in object_positions))
    partial = torch.empty((sum(sizes),), dtype=torch.float32, device=latents.device)
    off = 0
    items = []
    for key, n in zip(keys, sizes):
        q, k, heads, g = collect["q"][key]
        lay = layouts.get((g.H, g.W))
        if lay is None:
            lay = layouts[(g.H, g.W)] = GuidanceLayout(bboxes, object_positions, frames, g.H, g.W, fg_top_p, bg_top_p, latents.device)
        items.append((q, k, heads, lay, partial[off:off + n]))
        off += n
    dqs = ca_energy_loss_and_dq_all_keys(items, frames, ntext=text.ntext, grad_scale=grad_scale, fg_weight=fg_weight, bg_weight=bg_weight,
                                         com_loss_scale=com_loss_scale, **loss_options)
    for (q, _, _, _, _), dq in zip(items, dqs):
        tape.accumulate(q, dq)
    if saved_attn is not None:  # visualisation only (return_saved_attn): the full maps, as AttnProcessor.__call__ would have saved them
        for key in keys:
            q, k, heads, g = collect["q"][key]
            saved_attn[key] = ca_probability_maps(q, k, samples=g.B * g.F, heads=heads, positions=g.HW, ntext=text.ntext,
                                                  samples_per_key=g.F).cpu()
    loss = ops.reduce_sum(partial, grad_scale)
    tape.backward()
    g0 = Geom(1, frames, latents.shape[3], latents.shape[4])
    grad = engine.input_gradient(tape, g0, scale=latent_scale)
    return loss, grad


def guidance_loss_and_grad_many(engine: HipUNet3D, latents, t, text, bboxes_list, object_positions_list, guidance_attn_keys, *, loss_scale,
                                fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=1.0, com_loss_scale=0.0, latent_scale=1.0, **loss_options):
    """`guidance_loss_and_grad` for V independent samples in ONE recorded forward / backward of batch V (throughput mode: the UNet levels whose
    grids do not fill 256 CUs at batch 1 — most of a guidance pass — run at twice the rows).  latents (V,4,F,h,w); `text` a TextCache of the V
    cond embeddings; one (bboxes, object_positions) layout per sample.  The energies stay per sample: every key's query rows and text keys are
    sliced per sample, each sample's loss kernels see exactly what a batch-1 call hands them (own boxes, own token positions, own
    grad_scale = loss_scale / (objects x keys)), and write dQ into their rows of one [V*F*P, C] matrix per key.
    Returns ([V] list of loss tensors, grad (V,4,F,h,w) fp32)."""
    V = latents.shape[0]
    assert V == len(bboxes_list) == len(object_positions_list) == text.B
    keys = [tuple(k) for k in guidance_attn_keys]
    tape = Tape()
    collect = {"keys": set(keys), "q": {}, "stop_after": _last_key_in_order(engine, keys)}
    engine.forward(latents, t, text=text, tape=tape, collect=collect)
    missing = [k for k in keys if k not in collect["q"]]
    if missing:
        raise KeyError(f"guidance keys not produced by this UNet: {missing}")
    frames = latents.shape[2]
    dq_full = {key: torch.empty_like(collect["q"][key][0]) for key in keys}
    losses = []
    for v in range(V):
        bboxes, object_positions = bboxes_list[v], object_positions_list[v]
        grad_scale = loss_scale / (len(bboxes) * len(keys))
        ntok = sum(len(p) for p in object_positions)
        sizes = [frames * collect["q"][key][2] * ntok for key in keys]
        partial = torch.empty((sum(sizes),), dtype=torch.float32, device=latents.device)
        layouts, items, off = {}, [], 0
        for key, n in zip(keys, sizes):
            q, k, heads, g = collect["q"][key]
            rows = frames * g.HW
            lay = layouts.get((g.H, g.W))
            if lay is None:
                lay = layouts[(g.H, g.W)] = GuidanceLayout(bboxes, object_positions, frames, g.H, g.W, fg_top_p, bg_top_p, latents.device)
            items.append((q[v * rows:(v + 1) * rows], k[v * text.ntext:(v + 1) * text.ntext], heads, lay, partial[off:off + n],
                          dq_full[key][v * rows:(v + 1) * rows]))
            off += n
        ca_energy_loss_and_dq_all_keys(items, frames, ntext=text.ntext, grad_scale=grad_scale, fg_weight=fg_weight, bg_weight=bg_weight,
                                       com_loss_scale=com_loss_scale, **loss_options)
        losses.append(ops.reduce_sum(partial, grad_scale))
    for key in keys:
        tape.accumulate(collect["q"][key][0], dq_full[key])
    tape.backward()
    grad = engine.input_gradient(tape, Geom(V, frames, latents.shape[3], latents.shape[4]), scale=latent_scale)
    return losses, grad


def _key_order(engine):
    cfg = engine.cfg
    order = []
    for i, bt in enumerate(cfg.down_block_types):
        if bt == "CrossAttnDownBlock3D":
            order += [("down", i, j, 0) for j in range(cfg.layers_per_block)]
    order.append(("mid", 0, 0, 0))
    for i, bt in enumerate(cfg.up_block_types):
        if bt == "CrossAttnUpBlock3D":
            order += [("up", i, j, 0) for j in range(cfg.layers_per_block + 1)]
    return order


def _last_key_in_order(engine, keys):
    order = _key_order(engine)
    unknown = [k for k in keys if k not in order]
    if unknown:
        raise KeyError(f"unknown guidance attention keys {unknown}")
    return max(keys, key=order.index)


def hip_latent_backward_guidance(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss,
                                 loss_scale=30, loss_threshold=0.2, max_iter=5, max_index_step=10, cross_attention_kwargs=None,
                                 guidance_attn_keys=None, verbose=False, return_saved_attn=False, clear_cache=False, **kwargs):
    """Same call contract as models/pipelines.py:21-150; plugs into ``custom_latent_backward_guidance``.

    ``unet`` is anything exposing ``.engine`` (HipUNet3D) or the engine itself; ``cond_embeddings`` (1,77,D) or a
    TextCache.  ``loss`` is the carried loss (tensor or float).  Unsupported loss variants raise instead of
    silently computing something else."""
    engine = getattr(unet, "engine", unet)
    for k, default in _UNSUPPORTED.items():
        if k in kwargs and kwargs[k] != default:
            raise NotImplementedError(f"guidance option {k}={kwargs[k]!r} is outside the hot path built here")
    if return_saved_attn not in (False, None, "first", "last"):
        raise ValueError(return_saved_attn)
    saved_attn_to_return = None
    if guidance_attn_keys is None:
        guidance_attn_keys = DEFAULT_GUIDANCE_ATTN_KEYS
    loss_kw = {k: kwargs[k] for k in _LOSS_OPTIONS if k in kwargs}
    text = cond_embeddings if hasattr(cond_embeddings, "kv") else engine.encode_text(cond_embeddings)
    iteration = 0
    host = getattr(loss, "_host_copy", None)  # a loss this function returned earlier: its value is already in pinned host memory
    if host is not None:
        host[1].synchronize()
        loss_val = float(host[0])
    else:
        loss_val = float(loss)
    if index < max_index_step:
        if isinstance(max_iter, list):
            max_iter = max_iter[index]
        if verbose:
            print(f"time index {index}, loss: {loss_val / loss_scale:.3f} (de-scaled with scale {loss_scale:.1f}), loss threshold: {loss_threshold:.3f}")
        if len(bboxes) == 0:
            # the reference crashes in autograd here and generate.py skips the prompt; no boxes = no guidance
            z = torch.zeros((), device=latents.device)
            return (latents, z, None) if return_saved_attn else (latents, z)
        while loss_val / loss_scale > loss_threshold and iteration < max_iter and index < max_index_step:
            lat_in = scheduler.scale_model_input(latents, t) if hasattr(scheduler, "scale_model_input") else latents
            # models/pipelines.py:85-97: "first" keeps the maps of iteration 0, "last" those of iteration max_iter - 1 (not saved if
            # the loop returns earlier); the maps are materialised with torch only then — the loss kernels never need them
            want_maps = (return_saved_attn == "first" and iteration == 0) or (return_saved_attn == "last" and iteration == max_iter - 1)
            maps = {} if want_maps else None
            loss_t, grad = guidance_loss_and_grad(engine, lat_in, t, text, bboxes, object_positions, guidance_attn_keys,
                                                  loss_scale=loss_scale, saved_attn=maps, **loss_kw)
            if want_maps:
                saved_attn_to_return = maps
            if hasattr(scheduler, "alphas_cumprod"):
                scale = float((1 - scheduler.alphas_cumprod[int(t)]) ** 0.5)  # classifier-guidance scaling, pipelines.py:124-132
            else:
                warnings.warn("No scaling in guidance is performed")
                scale = 1.0
            latents = ops.axpy_(latents.to(torch.float32).contiguous().clone(), grad, scale)
            loss = loss_t
            iteration += 1
            # The reference reads loss.item() after every iteration (models/pipelines.py:134).  The value is needed on the host only to
            # decide whether ANOTHER iteration follows (or to print it).  When this was the last iteration the device tensor is handed
            # back with an asynchronous pinned-memory copy attached (`_host_copy`: the copy sits on the stream right behind the loss
            # reduction, i.e. in front of nothing the caller still queues); the entry check of the next guided step reads that copy
            # after its event — same control flow, no pipeline stall between the backward and the CFG forward.
            if not (iteration < max_iter or verbose):
                pinned = torch.empty(1, dtype=torch.float32, pin_memory=True)
                pinned.copy_(loss_t.reshape(1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                loss_t._host_copy = (pinned, ev)
            if iteration < max_iter or verbose:
                loss_val = float(loss_t.item())
                if math.isnan(loss_val):
                    print("**Loss is NaN**")
                if verbose:
                    print(f"time index {index}, loss: {loss_val / loss_scale:.3f}, loss threshold: {loss_threshold:.3f}, iteration: {iteration}")
    if return_saved_attn:
        return latents, loss, saved_attn_to_return
    return latents, loss


def hip_latent_backward_guidance_many(scheduler, unet, text, index, bboxes_list, object_positions_list, t, latents_list, loss_list,
                                      loss_scale=30, loss_threshold=0.2, max_iter=5, max_index_step=10, guidance_attn_keys=None, verbose=False,
                                      **kwargs):
    """`hip_latent_backward_guidance` for V samples that share the schedule and the guidance hyper-parameters (pipeline.sample_many): every
    iteration runs the samples whose carried loss is still above the threshold through ONE recorded forward / backward
    (`guidance_loss_and_grad_many`); a sample leaves the batch as soon as its own loop condition (models/pipelines.py:81) fails, exactly when
    a batch-1 loop over it would have stopped.  `text`: TextCache of the V cond embeddings.  Returns ([latents_v], [loss_v])."""
    engine = getattr(unet, "engine", unet)
    for k, default in _UNSUPPORTED.items():
        if k in kwargs and kwargs[k] != default:
            raise NotImplementedError(f"guidance option {k}={kwargs[k]!r} is outside the hot path built here")
    if guidance_attn_keys is None:
        guidance_attn_keys = DEFAULT_GUIDANCE_ATTN_KEYS
    loss_kw = {k: kwargs[k] for k in _LOSS_OPTIONS if k in kwargs}
    V = len(latents_list)
    latents_list, loss_list = list(latents_list), list(loss_list)
    if isinstance(max_iter, list):
        max_iter = max_iter[index] if index < len(max_iter) else 0
    vals = []
    for loss in loss_list:
        host = getattr(loss, "_host_copy", None)
        if host is not None:
            host[1].synchronize()
            vals.append(float(host[0]))
        else:
            vals.append(float(loss))
    if index >= max_index_step:
        return latents_list, loss_list
    for v in range(V):  # no boxes = no guidance (see hip_latent_backward_guidance)
        if len(bboxes_list[v]) == 0:
            loss_list[v], vals[v] = torch.zeros((), device=latents_list[v].device), 0.0
    if hasattr(scheduler, "alphas_cumprod"):
        scale = float((1 - scheduler.alphas_cumprod[int(t)]) ** 0.5)
    else:
        warnings.warn("No scaling in guidance is performed")
        scale = 1.0
    iteration = 0
    while iteration < max_iter:
        active = [v for v in range(V) if vals[v] / loss_scale > loss_threshold and len(bboxes_list[v])]
        if not active:
            break
        lat = torch.cat([latents_list[v] for v in active]).to(torch.float32).contiguous()
        lat_in = scheduler.scale_model_input(lat, t) if hasattr(scheduler, "scale_model_input") else lat
        sub_text = text if len(active) == V else engine.text_subset(text, active)
        losses, grad = guidance_loss_and_grad_many(engine, lat_in, t, sub_text, [bboxes_list[v] for v in active],
                                                   [object_positions_list[v] for v in active], guidance_attn_keys, loss_scale=loss_scale, **loss_kw)
        iteration += 1
        more = iteration < max_iter or verbose
        for i, v in enumerate(active):
            latents_list[v] = ops.axpy_(lat[i:i + 1].contiguous().clone(), grad[i:i + 1].contiguous(), scale)
            loss_list[v] = losses[i]
        # the V losses reach the host together: ONE pinned copy and ONE event (last iteration: read by the next step's entry check, no stall
        # here), or one synchronous read when the loop condition needs them now
        stacked = torch.stack([l.reshape(()) for l in losses])
        if not more:
            pinned = torch.empty(len(active), dtype=torch.float32, pin_memory=True)
            pinned.copy_(stacked, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            for i in range(len(active)):
                losses[i]._host_copy = (pinned[i:i + 1], ev)
        else:
            host_vals = stacked.tolist()
            for i, v in enumerate(active):
                vals[v] = float(host_vals[i])
                if math.isnan(vals[v]):
                    print("**Loss is NaN**")
                if verbose:
                    print(f"sample {v}, time index {index}, loss: {vals[v] / loss_scale:.3f}, loss threshold: {loss_threshold:.3f}, iteration: {iteration}")
    return latents_list, loss_list
