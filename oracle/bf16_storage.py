"""ORACLE (test infrastructure only) — the fp32 oracle with bf16 STORAGE of activations and activation gradients.

What a bf16 trunk must cost.  The product stores every activation (and, in the guidance pass, every activation gradient) as bf16 and
accumulates in fp32; the reference runs fp16 storage (/root/reference/generation/lvd.py:39-44).  `BF16Storage` is a TorchFunctionMode
that makes the oracle (oracle/unet_ref.py, unchanged) round exactly where the product stores a tensor:
  * outputs of F.linear / F.conv2d / F.conv3d / F.layer_norm (GEMM and norm outputs);
  * F.group_norm outputs, except when F.silu consumes them (GroupNorm+SiLU is one kernel, one store: SiLU is then applied to the
    unrounded normalisation and its output is rounded);
  * F.silu outputs; products h * gelu(g) (GEGLU: one store); sums of two equal-shaped activations (the residual adds, fused into GEMM
    epilogues in the product: the stored tensor is the rounded sum);
  * attention: softmax output is rounded as the operand of P·V (bf16 P feeds the MFMA) and the P·V result is rounded; the fp32
    probabilities handed to the guidance loss are NOT rounded (the product's loss kernel recomputes them in fp32 from bf16 Q and K).
Every rounding point rounds the gradient flowing back through it as well (the product's backward stores bf16 gradients at the same
places).  Scores Q·K^T, softmax statistics, norm statistics and everything outside the UNet (the loss arithmetic) stay fp32.
The distance between this run and the plain fp32 oracle is the noise floor of bf16 storage; tests assert that the HIP path stays
within a stated factor of it (tests/test_noise_floor.py, tests/test_guidance_gpu.py).
"""
import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def round_bf16(x):
    """Identity up to bf16 rounding, in both directions of autograd."""
    return _Round.apply(x)


_ADD = {torch.add, torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__}
_MUL = {torch.mul, torch.Tensor.mul, torch.Tensor.__mul__, torch.Tensor.__rmul__}
_MATMUL = {torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__}
_SOFTMAX = {torch.softmax, torch.Tensor.softmax, F.softmax}
_STORE = {F.linear, F.conv2d, F.conv3d, F.layer_norm}


def _big_pair(args):
    return (len(args) >= 2 and torch.is_tensor(args[0]) and torch.is_tensor(args[1]) and args[0].shape == args[1].shape
            and args[0].dim() >= 3 and args[0].is_floating_point())


class BF16Storage(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.silu:
            raw = getattr(args[0], "_gn_raw", None)
            return round_bf16(func(raw if raw is not None else args[0], *args[1:], **kwargs))
        if func in _MATMUL and getattr(args[0], "_is_probs", False):
            return round_bf16(func(round_bf16(args[0]), *args[1:], **kwargs))
        out = func(*args, **kwargs)
        if func in _STORE:
            return round_bf16(out)
        if func is F.group_norm:
            r = round_bf16(out)
            r._gn_raw = out  # GroupNorm+SiLU is one store: F.silu picks the unrounded normalisation up
            return r
        if func in _SOFTMAX:
            out._is_probs = True
            return out
        if (func in _ADD or func in _MUL) and _big_pair(args):
            return round_bf16(out)
        return out


def unet_forward_bf16_storage(unet_forward, sd, cfg, sample, *a, **kw):
    """`unet_ref.unet_forward` under bf16 storage; the latents enter as bf16 tokens (lvdhip_latents_to_tokens) and their gradient leaves
    as bf16 tokens (lvdhip_tokens_grad_to_latents)."""
    with BF16Storage():
        return unet_forward(sd, cfg, round_bf16(sample), *a, **kw)


def oracle_guidance_update(cfg, sd, latents, cond, boxes, positions, t, storage, keys, base_attn_dim=None, **hp):
    """(latent update, scaled loss) of one `latent_backward_guidance` iteration of the oracle; storage = "fp32" | "bf16".
    The distance between the two is the bf16-storage noise floor the HIP path is measured against."""
    from . import guidance_ref, scheduler_ref, unet_ref
    sched = scheduler_ref.DPMSolverPP2M()

    def unet_fn(x, tt, c, save, save_keys):
        kw = dict(save_attn_to_dict=save, save_keys=save_keys, stop_after_key=keys[-1])
        if storage == "bf16":
            unet_forward_bf16_storage(unet_ref.unet_forward, sd, cfg, x, int(tt), c, **kw)
        else:
            unet_ref.unet_forward(sd, cfg, x, int(tt), c, **kw)

    new, loss = guidance_ref.latent_backward_guidance(unet_fn, sched.alphas_cumprod, cond, 0, boxes, positions, t, latents.clone(), 10000.0,
                                                      base_attn_dim=base_attn_dim or tuple(latents.shape[-2:]), guidance_attn_keys=keys, **hp)
    return new - latents, loss
