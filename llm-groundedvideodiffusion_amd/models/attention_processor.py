"""`HipAttnProcessor`: the reference's per-op plug-in (`Attention.set_processor`,
/root/reference/models/attention_processor.py:167-180) backed by the HIP kernels.

Same call contract as AttnProcessor.__call__ (:432-449, including the `return_attntion_probs` spelling): given an
`Attention`-like module (`.to_q/.to_k/.to_v/.to_out[0]` Linear layers, `.heads`), hidden states (B, L, C) and optional
encoder states (B, T, D) it returns the attended hidden states; when the key is listed in `save_keys` the probabilities
(B, heads, L, T) are stored in `save_attn_to_dict[tuple(attn_key)]`; an `attn_process_fn` rewrites the cross-attention
probabilities before they meet V (:537-549).  Every branch runs on the C ABI: the fused attention (`lvdhip_attention_fwd`), the
materialised map of the slow path (`lvdhip_ca_probs_full`) and the product of processed probabilities with V
(`lvdhip_ca_apply_probs`).  Constraints of the kernels: CUDA tensors, head dim 64; `attention_mask` only as the per-text-position
additive bias of cross-attention (what Transformer2DModel builds from `encoder_attention_mask`; the reference never passes one on
this path, unet_3d_blocks.py:406 TODO) — any other mask raises."""
import torch

from .. import ops


def _w(linear):
    cache = getattr(linear, "_lvd_packed", None)
    if cache is None or cache[0] is not linear.weight:
        w = linear.weight.detach().to(torch.bfloat16).contiguous()
        b = None if linear.bias is None else linear.bias.detach().to(torch.float32).contiguous()
        cache = (linear.weight, w, b)
        linear._lvd_packed = cache
    return cache[1], cache[2]


class HipAttnProcessor:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, return_attntion_probs=False,
                 attn_key=None, attn_process_fn=None, return_cond_ca_only=False, return_token_ca_only=None,
                 offload_cross_attn_to_cpu=False, save_attn_to_dict=None, save_keys=None, enable_flash_attn=True,
                 cross_attn_save_hidden_states=False):
        key_bias = None
        if attention_mask is not None:
            # the additive bias Transformer2DModel derives from encoder_attention_mask ((1 - mask) * -10000, one singleton query dimension:
            # models/transformer_2d.py:303-307) and get_attention_scores adds to the scores (:222-258): one value per (sample, text position).
            # Cross-attention only; a bias that varies over the queries or the heads, or any self-attention mask, has no kernel here.
            m = attention_mask
            if encoder_hidden_states is None or not torch.is_floating_point(m) or m.shape[-1] != encoder_hidden_states.shape[1] \
                    or m.dim() not in (2, 3) or (m.dim() == 3 and m.shape[1] != 1) or m.shape[0] != hidden_states.shape[0]:
                raise NotImplementedError("attention_mask: only a floating-point additive bias of shape (batch, text) or (batch, 1, text) on cross-attention is "
                                          f"supported by the HIP kernels (got {tuple(m.shape)} {m.dtype}, cross-attention: {encoder_hidden_states is not None})")
            key_bias = m.reshape(m.shape[0], m.shape[-1]).to(torch.float32).contiguous()
        if hidden_states.dim() != 3 or not hidden_states.is_cuda:
            raise ValueError("HipAttnProcessor expects CUDA hidden states of shape (batch, tokens, channels)")
        B, L, Cc = hidden_states.shape
        heads = attn.heads
        if Cc != heads * 64:
            raise NotImplementedError(f"head_dim must be 64 (got channels={Cc}, heads={heads})")
        cross = encoder_hidden_states is not None
        x = hidden_states.reshape(B * L, Cc).to(torch.bfloat16).contiguous()
        ctx = x if not cross else encoder_hidden_states.reshape(-1, encoder_hidden_states.shape[-1]).to(torch.bfloat16).contiguous()
        T = L if not cross else encoder_hidden_states.shape[1]
        wq, _ = _w(attn.to_q)
        wk, _ = _w(attn.to_k)
        wv, _ = _w(attn.to_v)
        wo, bo = _w(attn.to_out[0])
        q, k, v = ops.gemm(x, wq), ops.gemm(ctx, wk), ops.gemm(ctx, wv)
        scale = float(getattr(attn, "scale", 0.125))
        if cross and cross_attn_save_hidden_states:
            self.hidden_states = hidden_states  # models/attention_processor.py:456-457
        want = return_attntion_probs or (save_attn_to_dict is not None and (save_keys is None or tuple(attn_key) in save_keys))
        probs = None
        if cross and (want or attn_process_fn is not None or key_bias is not None):
            # probabilities are only materialised on request (visualisation / external losses / a caller's rewrite / a masked prompt); fp32 like the loss maths
            from ..guidance import ca_apply_probabilities, ca_probability_maps
            probs = ca_probability_maps(q, k, samples=B, heads=heads, positions=L, ntext=T, scale=scale, key_bias=key_bias)
        if cross and attn_process_fn is not None:
            # :537-549 — the callback sees (batch*heads, L, T) probabilities and head-batched q / k / v, and returns what meets V
            hb = lambda t, n: t.reshape(B, n, heads, 64).permute(0, 2, 1, 3).reshape(B * heads, n, 64)
            processed = attn_process_fn(probs.reshape(B * heads, L, T).clone(), hb(q, L), hb(k, T), hb(v, T), attn_key=attn_key, cross_attn=cross,
                                        batch_size=B, heads=heads)
            o = ca_apply_probabilities(processed, v, samples=B, heads=heads, positions=L, ntext=T)
        elif key_bias is not None:  # masked cross-attention: the biased map times V (the fused kernel takes no bias)
            o = ca_apply_probabilities(probs, v, samples=B, heads=heads, positions=L, ntext=T)
        else:
            o = torch.empty_like(q)
            ops.attention_fwd(q, k, v, o, samples=B, heads=heads, sq=L, skv=T, qmap=ops.RowMap(1, L, 0, 1), kvmap=ops.RowMap(1, T, 0, 1), scale=scale)
        out = ops.gemm(o, wo, bias=bo).reshape(B, L, Cc).to(hidden_states.dtype)
        if probs is not None and (return_attntion_probs or save_attn_to_dict is not None):  # :553-589, on the probabilities BEFORE the rewrite
            if return_token_ca_only is not None:
                probs = probs[..., return_token_ca_only:return_token_ca_only + 1] if isinstance(return_token_ca_only, int) else probs[..., return_token_ca_only]
            if return_cond_ca_only:
                assert B % 2 == 0, f"Samples are not in pairs: {B} samples"
                probs = probs[B // 2:]
            if offload_cross_attn_to_cpu:
                probs = probs.cpu()
            if save_attn_to_dict is not None and save_keys is not None and tuple(attn_key) in save_keys:
                save_attn_to_dict[tuple(attn_key)] = probs
            if return_attntion_probs:
                return out, probs
        return out

    def free(self):
        if hasattr(self, "hidden_states"):
            del self.hidden_states


AttentionProcessor = HipAttnProcessor
