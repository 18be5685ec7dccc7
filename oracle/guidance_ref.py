"""ORACLE (test infrastructure only) — restatement of the cross-attention-energy guidance.

Follows, on any device and in fp32:
  scale_proportion                    /root/reference/utils/utils.py:82-103   (python banker's round)
  get_hw_from_attn_dim                /root/reference/utils/utils.py:253-256
  add_ca_loss_per_attn_map_to_loss    /root/reference/utils/guidance.py:160-526 (max-based top-k terms :328-353,
                                      centre-of-mass + velocity terms :468-522, per-object token normalisation :524)
  compute_ca_lossv3                   /root/reference/utils/guidance.py:529-574
  latent_backward_guidance            /root/reference/models/pipelines.py:21-150
Restated: max-based, ratio-based and CE energies, attention sync, BoxDiff, centre of mass, `smooth_attn`, `attn_renorm`;
not restated: upsample_scale != 1 (the reference's own max-based / CE forms raise on it).  Pinned by tests/golden/guidance_*.npz generated from the shimmed
reference import (oracle/make_golden.py).
"""
import math

import torch


def scale_proportion(box, H, W):
    x_min, y_min = round(box[0] * W), round(box[1] * H)
    bw, bh = round((box[2] - box[0]) * W), round((box[3] - box[1]) * H)
    x_max, y_max = x_min + bw, y_min + bh
    return max(x_min, 0), max(y_min, 0), min(x_max, W), min(y_max, H)


def get_hw_from_attn_dim(attn_dim, base_attn_dim):
    scale = int(math.sqrt((base_attn_dim[0] * base_attn_dim[1]) / attn_dim))
    return base_attn_dim[0] // scale, base_attn_dim[1] // scale


def _com(x, h_range, w_range):
    tot = x.sum(dim=(1, 2))
    return (x.sum(dim=2) * h_range).sum(-1) / tot, (x.sum(dim=1) * w_range).sum(-1) / tot


def ca_loss_per_map(attn_map, bboxes, object_positions, base_attn_dim, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0,
                    bg_weight=1.0, com_loss_scale=0.0, use_ratio_based_loss=False, eps=1.0e-2, attn_sync_weight=0.0,
                    boxdiff_loss_scale=0.0, boxdiff_normed=True, boxdiff_L=1, use_max_based_loss=True, use_ce_based_loss=False,
                    smooth_attn=False, kernel_size=3, sigma=0.5, attn_renorm=False, num_tokens=None, renorm_scale=2.0):
    """attn_map: (frames, heads, P, tokens) probabilities.  Returns the un-normalised loss contribution.
    utils/guidance.py:160-526 (add_ca_loss_per_attn_map_to_loss) with upsample_scale = 1, no smoothing / renorm / CE:
    max-based top-k energy (default), the deprecated ratio-based energy (:312-323) or the CE / NLL form (:363-399), attention sync between consecutive frames
    (:401-430), BoxDiff corner constraint (:240-287, 433-465), centre-of-mass position / velocity terms (:467-522)."""
    n_f, heads, P, _ = attn_map.shape
    if smooth_attn:
        # :209-220 — F.pad(attn_map, (1, 1, 1, 1), "reflect") pads the (position, token) plane of every (frame, head); GaussianSmoothing
        # (utils/attn.py:88-160, dim=2, channels=heads) is a depthwise conv2d with the normalised outer product of
        # g[x] = 1 / (s sqrt(2 pi)) * exp(-((x - mean) / (2 s))^2)   (the reference's own exponent)
        x = torch.arange(kernel_size, dtype=torch.float32, device=attn_map.device)
        g = 1.0 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-(((x - (kernel_size - 1) / 2) / (2 * sigma)) ** 2))
        k2 = g[:, None] * g[None, :]
        k2 = (k2 / k2.sum()).view(1, 1, kernel_size, kernel_size).repeat(heads, 1, 1, 1)
        attn_map = torch.nn.functional.conv2d(torch.nn.functional.pad(attn_map, (1, 1, 1, 1), mode="reflect"), k2.to(attn_map.dtype), groups=heads)
        assert attn_map.shape[:3] == (n_f, heads, P)
    if attn_renorm:
        # :222-226 — tokens 1 .. num_tokens-2 (start / end-of-text dropped), scaled, re-normalised; object positions shift by one (:291-294)
        attn_map = torch.softmax(attn_map[..., 1:num_tokens - 1] * renorm_scale, dim=-1)
        object_positions = [[p - 1 for p in ps] for ps in object_positions]
        assert attn_sync_weight == 0.0, "attn_sync with attn_renorm not implemented together"
    H, W = get_hw_from_attn_dim(P, base_attn_dim)
    dev = attn_map.device
    h_range = torch.arange(H, device=dev, dtype=torch.float32)[None]
    w_range = torch.arange(W, device=dev, dtype=torch.float32)[None]
    loss = torch.zeros((), device=dev)
    for obj_idx, obj_boxes in enumerate(bboxes):
        assert len(obj_boxes) == n_f
        obj_loss = 0
        for f, box in enumerate(obj_boxes):
            f1 = min(f + 1, n_f - 1)
            mask = torch.zeros(H, W, device=dev)
            x0, y0, x1, y1 = scale_proportion(box, H, W)
            mask[y0:y1, x0:x1] = 1
            if boxdiff_loss_scale > 0:
                corner_x, corner_y = torch.zeros(1, W, device=dev), torch.zeros(1, H, device=dev)
                L = boxdiff_L
                corner_x[:, max(x0 - L, 0):min(x0 + L + 1, W)] = 1.0
                corner_x[:, max(x1 - L, 0):min(x1 + L + 1, W)] = 1.0
                corner_y[:, max(y0 - L, 0):min(y0 + L + 1, H)] = 1.0
                corner_y[:, max(y1 - L, 0):min(y1 + L + 1, H)] = 1.0
            mask_t1 = torch.zeros(H, W, device=dev)
            # NB: the reference re-uses the names x_min..y_max for the NEXT frame's box here, so the attention-sync crop below is
            # taken with the box of frame f1 (utils/guidance.py:273-276 overwrite :250-252 before :418-423 read them)
            x0, y0, x1, y1 = scale_proportion(obj_boxes[f1], H, W)
            mask_t1[y0:y1, x0:x1] = 1
            k_fg = int((mask.sum() * fg_top_p).long().clamp_(min=1))
            k_bg = int(((1 - mask).sum() * bg_top_p).long().clamp_(min=1))
            m1 = mask.view(1, -1)
            for pos in object_positions[obj_idx]:
                a = attn_map[f, :, :, pos].float()      # (heads, P)
                a1 = attn_map[f1, :, :, pos].float()
                if use_ratio_based_loss:
                    act = (a.view(heads, H, W) * mask).reshape(heads, -1).sum(-1) / (a.sum(-1) + eps)
                    obj_loss = obj_loss + torch.mean((1 - act) ** 2)
                elif use_max_based_loss:
                    obj_loss = obj_loss + fg_weight * (1 - (a * m1).topk(k_fg).values.mean(1)).sum(0)
                    obj_loss = obj_loss + bg_weight * (a * (1 - m1)).topk(k_bg).values.mean(1).sum(0)
                elif use_ce_based_loss:
                    ac = torch.clamp(a, min=eps, max=1 - eps)
                    obj_loss = obj_loss + fg_weight * (-torch.log(torch.clamp((m1 * ac).topk(k_fg).values, min=eps))).mean(1).sum(0)
                    obj_loss = obj_loss + bg_weight * (-torch.log(1 - ((1 - m1) * ac).topk(k_bg).values.mean(1))).sum(0)
                else:
                    raise ValueError("Unknown loss: no loss selected")
                if attn_sync_weight != 0.0 and f != n_f - 1:
                    a2 = attn_map[f + 1, :, :, pos].float()
                    d = a.view(heads, H, W)[:, y0:y1, x0:x1] - a2.view(heads, H, W)[:, y0:y1, x0:x1]
                    obj_loss = obj_loss + (d ** 2).mean(dim=(1, 2)).sum(0) * attn_sync_weight
                if boxdiff_loss_scale > 0:
                    a2d = a.view(heads, H, W)
                    mx, my = a2d.max(dim=1).values, a2d.max(dim=2).values          # (heads, W), (heads, H)
                    mmx, mmy = mask[None].max(dim=1).values, mask[None].max(dim=2).values
                    if boxdiff_normed:
                        cc = ((mx - mmx).abs() * corner_x).mean() + ((my - mmy).abs() * corner_y).mean()
                    else:
                        cc = ((mx - mmx).abs() * corner_x).sum() + ((my - mmy).abs() * corner_y).sum()
                    obj_loss = obj_loss + cc * boxdiff_loss_scale
                if com_loss_scale > 0 and mask.sum() > 0:
                    ch, cw = _com(a.view(heads, H, W), h_range, w_range)
                    mh, mw = _com(mask[None], h_range, w_range)
                    obj_loss = obj_loss + com_loss_scale * (((ch - mh) ** 2).mean() + ((cw - mw) ** 2).mean())
                    if mask_t1.sum() > 0:
                        ch1, cw1 = _com(a1.view(heads, H, W), h_range, w_range)
                        mh1, mw1 = _com(mask_t1[None], h_range, w_range)
                        obj_loss = obj_loss + com_loss_scale * ((((ch1 - ch) - (mh1 - mh)) ** 2).mean() + (((cw1 - cw) - (mw1 - mw)) ** 2).mean())
        loss = loss + obj_loss / len(object_positions[obj_idx])
    return loss


def compute_ca_loss(saved_attn, bboxes, object_positions, guidance_attn_keys, base_attn_dim, **kw):
    """compute_ca_lossv3: mean over objects and keys of the per-map energies."""
    dev = next(iter(saved_attn.values())).device if saved_attn else "cpu"
    loss = torch.zeros((), device=dev)
    n_obj = len(bboxes)
    if n_obj == 0:
        return loss
    for key in guidance_attn_keys:
        loss = loss + ca_loss_per_map(saved_attn[key], bboxes, object_positions, base_attn_dim, **kw)
    if len(guidance_attn_keys) > 0:
        loss = loss / (n_obj * len(guidance_attn_keys))
    return loss


def latent_backward_guidance(unet_fn, alphas_cumprod, cond_embeddings, index, bboxes, object_positions, t, latents, loss,
                             loss_scale=30.0, loss_threshold=0.2, max_iter=5, max_index_step=10, guidance_attn_keys=None,
                             base_attn_dim=(40, 72), **loss_kw):
    """models/pipelines.py:21-150.  `unet_fn(latents, t, cond, save_dict, save_keys)` runs the cond-branch forward and
    fills save_dict; `loss` is the carried (scaled) loss tensor/float.  Returns (latents, loss)."""
    iteration = 0
    loss_val = float(loss)
    if index < max_index_step:
        if isinstance(max_iter, list):
            max_iter = max_iter[index]
        while loss_val / loss_scale > loss_threshold and iteration < max_iter:
            saved = {}
            lat = latents.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                unet_fn(lat, t, cond_embeddings, saved, guidance_attn_keys)
                l = compute_ca_loss(saved, bboxes, object_positions, guidance_attn_keys, base_attn_dim, **loss_kw) * loss_scale
                (grad,) = torch.autograd.grad(l, [lat])
            scale = (1 - float(alphas_cumprod[int(t)])) ** 0.5
            latents = latents.detach() - scale * grad
            loss_val = float(l.detach())
            iteration += 1
    return latents, loss_val
