"""Token-matrix executor of the UNet3D denoiser on MI355X.

Design (MI355X-first, not a module-by-module port of models/unet_3d_condition.py):
  * every activation is ONE row-major bf16 matrix [rows, C] with rows = (batch, frame, y, x).  Linear layers,
    3x3 convs (implicit GEMM), temporal convs (3 shifted taps of the same matrix) and the skip-connection
    concat (two-source A operand) are all the same MFMA GEMM kernel; the reference's NCHW<->NLC permutes
    (transformer_2d.py:317-326,362-370), (B·F,C,H,W)<->(B,C,F,H,W) reshapes (unet_3d_condition.py:726-728,
    transformer_temporal.py:148-156,175-182, TemporalConvLayer) and torch.cat (unet_3d_blocks.py:641,736) never
    touch HBM — spatial and temporal attention address the same matrix with different row strides.
  * text keys/values of all cross-attention layers are projected once per prompt (TextCache): the reference
    recomputes to_k/to_v on the frame-repeated text for every frame, layer and step
    (unet_3d_condition.py:721-723 + attention_processor.py:387-399).
  * the guidance pass records a static tape of kernel launches; backward() replays hand-written
    input-gradient kernels in reverse (no autograd graph, parameter gradients are never formed), and the
    forward stops right after the query projection of the last guidance key (the reference runs the whole UNet,
    models/pipelines.py:99 TODO).
"""
import math
from dataclasses import dataclass

import torch

from . import ops
from .weights import UNetConfig, interleave_geglu, pack_conv3x3, pack_tconv3


class StopForward(Exception):
    pass


class Tape:
    """Reverse-mode tape over token matrices.  Gradients are bf16 matrices keyed by tensor identity."""

    def __init__(self):
        self.fns = []
        self.g = {}

    def push(self, fn):
        self.fns.append(fn)

    def pop(self, t):
        e = self.g.pop(id(t), None)
        return None if e is None else e[1]

    def peek(self, t):
        e = self.g.get(id(t))
        return None if e is None else e[1]

    def accumulate(self, t, g):
        e = self.g.get(id(t))
        if e is None:
            self.g[id(t)] = (t, g)
        else:
            ops.add(e[1], g, out=e[1])

    def target(self, t):
        """Buffer that receives d(t) and whether the producer must accumulate into it."""
        e = self.g.get(id(t))
        if e is not None:
            return e[1], True
        buf = torch.empty((t.shape[0], t.shape[1]), dtype=torch.bfloat16, device=t.device)
        self.g[id(t)] = (t, buf)
        return buf, False

    def backward(self):
        for fn in reversed(self.fns):
            fn()
        self.fns = []


@dataclass
class Geom:
    B: int
    F: int
    H: int
    W: int

    @property
    def HW(self):
        return self.H * self.W

    @property
    def rows(self):
        return self.B * self.F * self.H * self.W


class LnRef:
    """A LayerNorm that is never materialised: the rows x, their (mean, rstd) and the norm's parameter name.  The consuming linear
    layers (to_qkv / to_q / ff.net.0.proj) run on x with the norm folded into the product (ops.gemm ln_stats=); `virt` stands in
    for the normalised activation on the guidance tape (its gradient buffer is keyed by this object)."""

    class _Virtual:
        def __init__(self, shape, device):
            self.shape, self.device = shape, device

    def __init__(self, x, mr, name):
        self.x, self.mr, self.name = x, mr, name
        self.virt = LnRef._Virtual(tuple(x.shape), x.device)
        self.shape = x.shape


class TextCache:
    """Per cross-attention layer [B*77, 2C] key/value projections of the prompt embeddings."""

    def __init__(self, tokens, kv, B, ntext):
        self.tokens = tokens  # [B*ntext, cross_dim] bf16
        self.kv = kv
        self.B = B
        self.ntext = ntext


class HipUNet3D:
    def __init__(self, cfg: UNetConfig, state_dict, device="cuda"):
        self.cfg = cfg
        self.dev = ops.use_device(device)
        self.w = {}
        self.alpha = {}
        self._dgrad = {}
        # BasicTransformerBlock.norm1/2/3 folded into the products that consume them; LVD_LN_FOLD=0 restores the
        # separate LayerNorm launches (A/B knob)
        import os
        self.ln_fold = os.environ.get("LVD_LN_FOLD", "1") != "0"
        self.ln_fold_all = os.environ.get("LVD_LN_FOLD", "1") != "narrow"
        # classifier-free guidance: the part of the network in front of the first text-dependent layer once per sample (forward(cfg_pairs=True));
        # LVD_CFG_SHARED_PREFIX=0 makes the callers feed the duplicated batch like the reference does (A/B knob)
        self.cfg_shared_prefix = os.environ.get("LVD_CFG_SHARED_PREFIX", "1") != "0"
        self.tconv_expand_max_rows = int(os.environ.get("LVD_TCONV_EXPAND_MAX_ROWS", "4320"))  # 0: every temporal conv on the tap GEMM
        self._pack(state_dict)

    # ------------------------------------------------------------------ weights
    def _pack(self, sd):
        dev = self.dev
        w = self.w
        for name, t in sd.items():
            t = t.detach().to(torch.float32)
            if t.dim() == 0:
                self.alpha[name] = math.tanh(float(t))
            elif t.dim() == 4:
                if t.shape[2] == 3:
                    if t.shape[1] % 8:  # conv_in: pad the 4 latent channels to 8
                        pad = 8 - t.shape[1] % 8
                        t = torch.cat([t, torch.zeros(t.shape[0], pad, 3, 3, device=t.device)], 1)
                    w[name] = pack_conv3x3(t).to(dev)
                else:
                    w[name] = t.reshape(t.shape[0], t.shape[1]).to(torch.bfloat16).contiguous().to(dev)
            elif t.dim() == 5:
                w[name] = pack_tconv3(t).to(dev)
            elif t.dim() == 2:
                w[name] = t.to(torch.bfloat16).contiguous().to(dev)
            else:
                w[name] = t.contiguous().to(dev)  # norm affine / bias / null features stay fp32
        # derived layouts
        for name in [n for n in list(w.keys()) if n.endswith(".to_q.weight")]:
            p = name[: -len(".to_q.weight")]
            is_cross = ".attentions." in p and p.endswith(".attn2")
            if is_cross:
                w[p + ".to_kv.weight"] = torch.cat([w.pop(p + ".to_k.weight"), w.pop(p + ".to_v.weight")], 0).contiguous()
            else:
                w[p + ".to_qkv.weight"] = torch.cat([w.pop(p + ".to_q.weight"), w.pop(p + ".to_k.weight"), w.pop(p + ".to_v.weight")], 0).contiguous()
        for name in [n for n in list(w.keys()) if n.endswith(".ff.net.0.proj.weight")]:
            p = name[: -len(".weight")]
            w[p + ".weight"], w[p + ".bias"] = interleave_geglu(w[p + ".weight"], w[p + ".bias"])
        # LayerNorm-folded copies of the products that read a transformer block's norm1/2/3: W' = gamma (.) W (bf16), colsum_n = sum_k W'_nk
        # (of the ROUNDED W', so that a constant row cancels exactly), bias' = b + W beta.  The GLIGEN fuser keeps its materialised norms.
        if self.ln_fold:
            for name in [n for n in list(w.keys()) if n.endswith(".norm1.weight") and ".transformer_blocks." in n and ".fuser." not in n]:
                blk = name[: -len(".norm1.weight")]
                pairs = [("norm1", "attn1.to_qkv"), ("norm2", "attn2.to_qkv" if (blk + ".attn2.to_qkv.weight") in w else "attn2.to_q"),
                         ("norm3", "ff.net.0.proj")]
                for nrm, lin in pairs:
                    W32 = w[f"{blk}.{lin}.weight"].float()
                    gamma, beta = w[f"{blk}.{nrm}.weight"].float(), w[f"{blk}.{nrm}.bias"].float()
                    Wp = (W32 * gamma[None, :]).to(torch.bfloat16).contiguous()
                    b = w.get(f"{blk}.{lin}.bias")
                    w[f"{blk}.{lin}.lnw"] = Wp
                    w[f"{blk}.{lin}.lncolsum"] = Wp.float().sum(1).contiguous()
                    w[f"{blk}.{lin}.lnbias"] = ((b.float() if b is not None else 0) + W32 @ beta).contiguous()
        self.cross_layers = sorted(n[: -len(".to_kv.weight")] for n in w if n.endswith(".to_kv.weight"))
        # the time-embedding projections of all ResnetBlock2Ds read the same [B, 1280] input: ONE product [B, sum of Cout] per forward
        # instead of 22 two-row GEMM launches; each conv1 reads its column range (lvd_gemm_params.ldrowbias)
        tnames = sorted(n[: -len(".time_emb_proj.weight")] for n in w if n.endswith(".time_emb_proj.weight"))
        self.temb_slices, off = {}, 0
        for n in tnames:
            co = w[n + ".time_emb_proj.weight"].shape[0]
            self.temb_slices[n] = (off, off + co)
            off += co
        if tnames:
            w["time_emb_proj_all.weight"] = torch.cat([w.pop(n + ".time_emb_proj.weight") for n in tnames], 0).contiguous()
            w["time_emb_proj_all.bias"] = torch.cat([w.pop(n + ".time_emb_proj.bias") for n in tnames], 0).contiguous()

    def wt(self, name, kind="linear"):
        """Lazily built input-gradient layout of a packed weight (kept resident: 288 GB of HBM)."""
        key = (name, kind)
        t = self._dgrad.get(key)
        if t is None:
            W = self.w[name]
            if kind == "linear":
                t = W.t().contiguous()
            elif kind == "conv":  # [cout, 9*cin] -> [cin, 9*cout], taps flipped
                co = W.shape[0]
                ci = W.shape[1] // 9
                t = W.reshape(co, 3, 3, ci).flip(1, 2).permute(3, 1, 2, 0).reshape(ci, 9 * co).contiguous()
            elif kind == "conv_t2":
                co = W.shape[0]
                ci = W.shape[1] // 9
                t = W.reshape(co, 3, 3, ci).permute(3, 1, 2, 0).reshape(ci, 9 * co).contiguous()
            elif kind == "tconv":
                co = W.shape[0]
                ci = W.shape[1] // 3
                t = W.reshape(co, 3, ci).flip(1).permute(2, 1, 0).reshape(ci, 3 * co).contiguous()
            else:
                raise ValueError(kind)
            self._dgrad[key] = t
        return t

    # ------------------------------------------------------------------ text
    def encode_text(self, ehs):
        """ehs: [B, 77, cross_dim] (any float dtype) -> TextCache with K|V of every cross-attention layer."""
        B, nt, dc = ehs.shape
        x = ehs.reshape(B * nt, dc).to(self.dev, torch.bfloat16).contiguous()
        kv = {p: ops.gemm(x, self.w[p + ".to_kv.weight"]) for p in self.cross_layers}
        return TextCache(x, kv, B, nt)

    @staticmethod
    def text_subset(text: TextCache, items):
        """TextCache of the listed batch items (copies of their rows: a guidance batch that lost a sample to its loss threshold)."""
        idx = torch.cat([torch.arange(i * text.ntext, (i + 1) * text.ntext) for i in items]).to(text.tokens.device)
        return TextCache(text.tokens.index_select(0, idx), {p: kv.index_select(0, idx) for p, kv in text.kv.items()}, len(items), text.ntext)

    # ------------------------------------------------------------------ primitive ops with tape hooks
    def _linear(self, x, name, *, tape, res=None, bias=True, x2=None, alpha=1.0, act=ops.ACT_NONE, out_fp32=False):
        """x: token matrix, or an LnRef (the product then runs on the raw rows with the LayerNorm folded in)."""
        ln = x if isinstance(x, LnRef) else None
        if ln is not None:
            assert x2 is None and not out_fp32 and res is None
            out = ops.gemm(ln.x, self.w[name + ".lnw"], bias=self.w[name + ".lnbias"], alpha=alpha, act=act,
                           ln_stats=ln.mr, ln_colsum=self.w[name + ".lncolsum"])
            x = ln.virt  # the tape's handle of the (never written) normalised rows
        else:
            W = self.w[name + ".weight"]
            b = self.w.get(name + ".bias") if bias else None
            out = ops.gemm(x, W, a2=x2, bias=b, res=res, alpha=alpha, act=act, out_fp32=out_fp32)
        if tape is not None:
            assert act == ops.ACT_NONE and not out_fp32

            def bw():
                dy = tape.pop(out)
                if dy is None:
                    return
                Wt = self.wt(name + ".weight")  # d(LN out) = dy . W with the ORIGINAL W: the fold changes how the forward is computed, not what
                if x2 is None:
                    buf, acc = tape.target(x)
                    ops.gemm(dy, Wt, out=buf, accumulate=acc, alpha=alpha)
                else:
                    tmp = ops.gemm(dy, Wt, alpha=alpha)
                    c1 = x.shape[1]
                    tape.accumulate(x, tmp[:, :c1])
                    tape.accumulate(x2, tmp[:, c1:])
                if res is not None:
                    tape.accumulate(res, dy)
            tape.push(bw)
        return out

    def _conv3x3(self, x, name, g_in: Geom, *, tape, stride=1, upsample=0, res=None, rowbias=None, x2=None, out_fp32=False, up_to=None):
        if upsample and up_to is not None and tuple(up_to) != (2 * g_in.H, 2 * g_in.W):
            return self._conv3x3_resized(x, name, g_in, up_to, tape=tape)
        W = self.w[name + ".weight"]
        hin, win = (g_in.H * 2, g_in.W * 2) if upsample else (g_in.H, g_in.W)
        hout, wout = ((hin - 1) // 2 + 1, (win - 1) // 2 + 1) if stride == 2 else (hin, win)
        geo = ops.ConvGeom(hin, win, hout, wout, stride, upsample)
        rps = g_in.F * hout * wout
        out = ops.gemm(x, W, a2=x2, bias=self.w[name + ".bias"], rowbias=rowbias, rows_per_sample=rps if rowbias is not None else 0,
                       res=res, mode=ops.A_CONV3X3, conv=geo, out_fp32=out_fp32)
        if tape is not None:
            assert x2 is None and not out_fp32

            def bw():
                dy = tape.pop(out)
                if dy is None:
                    return
                nimg = g_in.B * g_in.F
                if stride == 2:
                    buf, acc = tape.target(x)
                    ops.gemm(dy, self.wt(name + ".weight", "conv_t2"), out=buf, accumulate=acc, mode=ops.A_CONV3X3_T2,
                             conv=ops.ConvGeom(hout, wout, hin, win), m=nimg * hin * win)
                elif upsample:
                    big = ops.gemm(dy, self.wt(name + ".weight", "conv"), mode=ops.A_CONV3X3, conv=ops.ConvGeom(hin, win, hin, win))
                    buf, acc = tape.target(x)
                    ops.upsample2x_bwd(big, nimg, g_in.H, g_in.W, x.shape[1], dx=buf, accumulate=acc)
                else:
                    buf, acc = tape.target(x)
                    ops.gemm(dy, self.wt(name + ".weight", "conv"), out=buf, accumulate=acc, mode=ops.A_CONV3X3, conv=ops.ConvGeom(hin, win, hin, win))
                if res is not None:
                    tape.accumulate(res, dy)
            tape.push(bw)
        return out, Geom(g_in.B, g_in.F, hout, wout)

    def _conv3x3_resized(self, x, name, g_in: Geom, size, *, tape):
        """Upsample2D with an explicit output size (latent H or W not divisible by 8: the reference passes the skip
        connection's size as `upsample_size`, unet_3d_condition.py:711-730, and diffusers interpolates nearest to it).  Rare
        and small (e.g. 256x144 -> latent 18x32 -> 9 -> 5 -> 3): the rows are gathered with the nearest index maps, then the
        plain 3x3 conv kernel runs; the exact-x2 case stays fused in the conv loader."""
        H2, W2 = size
        n, C = g_in.B * g_in.F, x.shape[1]
        idx = lambda src, dst: (torch.arange(dst, device=x.device, dtype=torch.float32) * (src / dst)).floor().long().clamp_(max=src - 1)
        yi, xi = idx(g_in.H, H2), idx(g_in.W, W2)
        big = x.view(n, g_in.H, g_in.W, C)[:, yi][:, :, xi].reshape(n * H2 * W2, C).contiguous()
        geo = ops.ConvGeom(H2, W2, H2, W2, 1, 0)
        out = ops.gemm(big, self.w[name + ".weight"], bias=self.w[name + ".bias"], mode=ops.A_CONV3X3, conv=geo)
        if tape is not None:
            def bw():
                dy = tape.pop(out)
                if dy is None:
                    return
                dbig = ops.gemm(dy, self.wt(name + ".weight", "conv"), mode=ops.A_CONV3X3, conv=geo).view(n, H2, W2, C).float()
                d = torch.zeros((n, g_in.H, W2, C), dtype=torch.float32, device=x.device).index_add_(1, yi, dbig)
                d = torch.zeros((n, g_in.H, g_in.W, C), dtype=torch.float32, device=x.device).index_add_(2, xi, d)
                tape.accumulate(x, d.reshape(n * g_in.HW, C).to(torch.bfloat16))
            tape.push(bw)
        return out, Geom(g_in.B, g_in.F, H2, W2)

    def _tconv_weight_expanded(self, name, backward):
        """[3N, Cin] form of a temporal-conv weight (ops.tconv_expand_weight) for the small-M path; built once, kept resident."""
        key = (name, "tconv_exp_bwd" if backward else "tconv_exp")
        t = self._dgrad.get(key)
        if t is None:
            t = self._dgrad[key] = ops.tconv_expand_weight(self.wt(name, "tconv") if backward else self.w[name])
        return t

    def _tconv(self, x, name, g: Geom, *, tape, res=None):
        # Deep levels (<= tconv_expand_max_rows token rows): the tap GEMM has 3 - 9 tiles of 512 rows there and runs as a K-split launch whose
        # fp32 slabs (padded to whole tiles) outweigh the product; ONE plain product with 3N columns + a combine pass is faster
        # (tools/deep_probe.py: 42.0 -> 34.6 us at 1080 rows, 53.4 -> 41.7 at 2160, 68.8 -> 65.5 at 4320; slower from 8640 rows on).
        small = x.shape[0] <= self.tconv_expand_max_rows
        if small:
            out = ops.tconv_expanded(x, self._tconv_weight_expanded(name + ".weight", False), frames=g.F, hw=g.HW, bias=self.w[name + ".bias"], res=res)
        else:
            out = ops.gemm(x, self.w[name + ".weight"], bias=self.w[name + ".bias"], res=res, mode=ops.A_TCONV3, frames=g.F, hw=g.HW)
        if tape is not None:
            def bw():
                dy = tape.pop(out)
                if dy is None:
                    return
                buf, acc = tape.target(x)
                if small:
                    ops.tconv_expanded(dy, self._tconv_weight_expanded(name + ".weight", True), frames=g.F, hw=g.HW, out=buf, accumulate=acc)
                else:
                    ops.gemm(dy, self.wt(name + ".weight", "tconv"), out=buf, accumulate=acc, mode=ops.A_TCONV3, frames=g.F, hw=g.HW)
                if res is not None:
                    tape.accumulate(res, dy)
            tape.push(bw)
        return out

    def _groupnorm(self, x, name, rps, *, tape, eps, silu, x2=None):
        gamma, beta = self.w[name + ".weight"], self.w[name + ".bias"]
        G = self.cfg.norm_num_groups
        out, mr = ops.groupnorm_auto(x, gamma, beta, rps, groups=G, eps=eps, silu=silu, x2=x2)
        if tape is not None:
            def bw():
                dy = tape.pop(out)
                if dy is None:
                    return
                if x2 is None:
                    buf, acc = tape.target(x)
                    ops.groupnorm_bwd(x, dy, gamma, beta, mr, rps, groups=G, silu=silu, dx1=buf, accumulate=acc)
                else:
                    d1, d2 = ops.groupnorm_bwd(x, dy, gamma, beta, mr, rps, groups=G, silu=silu, x2=x2)
                    tape.accumulate(x, d1)
                    tape.accumulate(x2, d2)
            tape.push(bw)
        return out

    def _layernorm(self, x, name, *, tape, fold=False):
        gamma, beta = self.w[name + ".weight"], self.w[name + ".bias"]
        if fold and self.ln_fold:
            # statistics only; the consumers fold the norm into their products (LnRef).  The backward is the ordinary LayerNorm backward:
            # it needs x, (mean, rstd) and the gradient of the normalised rows, which the consumers' dgrad GEMMs deliver under ref.virt.
            ref = LnRef(x, ops.layernorm_stats(x), name)
            if tape is not None:
                def bw():
                    dy = tape.pop(ref.virt)
                    if dy is None:
                        return
                    buf, acc = tape.target(x)
                    ops.layernorm_bwd(x, dy, gamma, ref.mr, dx=buf, accumulate=acc)
                tape.push(bw)
            return ref
        if tape is None:
            return ops.layernorm(x, gamma, beta)
        out, mr = ops.layernorm(x, gamma, beta, return_stats=True)

        def bw():
            dy = tape.pop(out)
            if dy is None:
                return
            buf, acc = tape.target(x)
            ops.layernorm_bwd(x, dy, gamma, mr, dx=buf, accumulate=acc)
        tape.push(bw)
        return out

    def _self_attention(self, x, name, heads, *, samples, seq, rowmap, tape, x_extra=None, extra_map=None, seq_extra=0):
        """LN'd tokens -> fused QKV GEMM -> flash attention.  x_extra: second key/value segment (GLIGEN objs)."""
        C = heads * 64
        ln = x if isinstance(x, LnRef) else None
        if ln is not None:
            qkv = ops.gemm(ln.x, self.w[name + ".to_qkv.lnw"], bias=self.w[name + ".to_qkv.lnbias"], ln_stats=ln.mr, ln_colsum=self.w[name + ".to_qkv.lncolsum"])
            x = ln.virt
        else:
            qkv = ops.gemm(x, self.w[name + ".to_qkv.weight"])
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        o = torch.empty((x.shape[0], C), dtype=torch.bfloat16, device=qkv.device)
        kw = dict(samples=samples, heads=heads, sq=seq, skv=seq, qmap=rowmap, kvmap=rowmap, scale=0.125)
        if x_extra is not None:
            assert tape is None, "the guidance pass runs without GLIGEN conditioning (reference: models/pipelines.py:66-72)"
            qkv2 = ops.gemm(x_extra, self.w[name + ".to_qkv.weight"])
            kw.update(k2=qkv2[:, C:2 * C], v2=qkv2[:, 2 * C:], skv2=seq_extra, kv2map=extra_map)
        lse = torch.empty((samples, heads, seq), dtype=torch.float32, device=qkv.device) if tape is not None else None
        ops.attention_fwd(q, k, v, o, lse=lse, **kw)
        if tape is not None:
            def bw():
                do = tape.pop(o)
                if do is None:
                    return
                dqkv = torch.empty_like(qkv)
                ops.attention_bwd(q, k, v, o, lse, do, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], **kw)
                buf, acc = tape.target(x)
                ops.gemm(dqkv, self.wt(name + ".to_qkv.weight"), out=buf, accumulate=acc)
            tape.push(bw)
        return o

    def _cross_attention(self, x, name, heads, text: TextCache, g: Geom, *, tape, key=None, collect=None):
        C = heads * 64
        q = self._linear(x, name + ".to_q", tape=tape, bias=False)
        kv = text.kv[name]
        k, v = kv[:, :C], kv[:, C:]
        if collect is not None and key in collect["keys"]:
            collect["q"][key] = (q, k, heads, g)
            if key == collect.get("stop_after"):
                raise StopForward()
        o = torch.empty((q.shape[0], C), dtype=torch.bfloat16, device=q.device)
        samples = g.B * g.F
        kw = dict(samples=samples, heads=heads, sq=g.HW, skv=text.ntext, qmap=ops.RowMap(1, g.HW, 0, 1),
                  kvmap=ops.RowMap(g.F, text.ntext, 0, 1), scale=0.125)
        lse = torch.empty((samples, heads, g.HW), dtype=torch.float32, device=q.device) if tape is not None else None
        ops.attention_fwd(q, k, v, o, lse=lse, **kw)
        if tape is not None:
            def bw():
                do = tape.pop(o)
                if do is None:
                    return
                buf, acc = tape.target(q)
                if acc:  # q already carries the loss gradient: add the attention-path gradient to it
                    dq = torch.empty_like(q)
                    ops.attention_bwd(q, k, v, o, lse, do, dq, None, None, **kw)
                    ops.add(buf, dq, out=buf)
                else:
                    ops.attention_bwd(q, k, v, o, lse, do, buf, None, None, **kw)
            tape.push(bw)
        return o

    def _feed_forward(self, x, name, res, *, tape, alpha=1.0):
        if tape is None:
            h = self._linear(x, name + ".net.0.proj", tape=None, act=ops.ACT_GEGLU)
        else:
            pre = self._linear(x, name + ".net.0.proj", tape=tape)
            h = ops.geglu_fwd(pre)

            def bw():
                dh = tape.pop(h)
                if dh is None:
                    return
                tape.accumulate(pre, ops.geglu_bwd(pre, dh))
            tape.push(bw)
        return self._linear(h, name + ".net.2", tape=tape, res=res, alpha=alpha)

    @staticmethod
    def _pair_rows(x, g: Geom):
        """[sample_0, sample_1, ...] -> [sample_0, sample_0, sample_1, sample_1, ...] (rows of a sample stay together): the point where the
        (uncond, cond) halves of a classifier-free-guidance pair, identical so far, start to differ.  One device copy."""
        rps = g.F * g.HW
        out = torch.empty((2 * x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
        out.view(g.B, 2, rps, x.shape[1]).copy_(x.view(g.B, 1, rps, x.shape[1]).expand(-1, 2, -1, -1))
        return out

    # ------------------------------------------------------------------ composite blocks
    def _transformer_block(self, hs, name, heads, *, tape, spatial, g: Geom, text=None, objs=None, key=None, collect=None, pair_split=False):
        if spatial:
            samples, seq, rmap = g.B * g.F, g.HW, ops.RowMap(1, g.HW, 0, 1)
        else:
            samples, seq, rmap = g.B * g.HW, g.F, ops.RowMap(g.HW, g.F * g.HW, 1, g.HW)
        # LayerNorm fold (statistics launch + row scaling in the consumer's epilogue).  The scaling costs the epilogue two FMAs and two LDS
        # reads per output element, i.e. it grows with N, while the saved LayerNorm pass does not: at level 0, batch 2, isolated launches
        # (tools/step_gemm_profile.py) to_q (N = C) +0 us for 14 us saved, to_qkv (N = 3C) +13 us, ff.net.0.proj (N = 8C, GEGLU) +36 us.  In the
        # step itself folding every norm still measures best (same box: 116.4 ms all / 117.1 ms narrow consumers only / 117.4 ms none): the
        # consumer no longer waits for a full LayerNorm pass to drain.  LVD_LN_FOLD=narrow folds norm2 -> to_q of the spatial blocks only.
        fold_all = self.ln_fold_all
        n1 = self._layernorm(hs, name + ".norm1", tape=tape, fold=fold_all)
        o = self._self_attention(n1, name + ".attn1", heads, samples=samples, seq=seq, rowmap=rmap, tape=tape)
        hs = self._linear(o, name + ".attn1.to_out.0", tape=tape, res=hs)
        if pair_split:  # first text- (or grounding-) dependent layer of the network: from here on each sample is an (uncond, cond) pair
            assert spatial and tape is None
            hs = self._pair_rows(hs, g)
            g = Geom(2 * g.B, g.F, g.H, g.W)
            samples = g.B * g.F
        if objs is not None and (name + ".fuser.linear.weight") in self.w:
            hs = self._fuser(hs, name + ".fuser", heads, objs, g)
        n2 = self._layernorm(hs, name + ".norm2", tape=tape, fold=spatial or fold_all)
        if spatial:
            o = self._cross_attention(n2, name + ".attn2", heads, text, g, tape=tape, key=key, collect=collect)
        else:
            o = self._self_attention(n2, name + ".attn2", heads, samples=samples, seq=seq, rowmap=rmap, tape=tape)
        hs = self._linear(o, name + ".attn2.to_out.0", tape=tape, res=hs)
        n3 = self._layernorm(hs, name + ".norm3", tape=tape, fold=fold_all)
        return self._feed_forward(n3, name + ".ff", hs, tape=tape)

    def _fuser(self, hs, name, heads, objs, g: Geom):
        """GatedSelfAttentionDense (models/attention.py:44-60); objs: [B*F*30, cross_dim] bf16."""
        nobj = objs.shape[0] // (g.B * g.F)
        o_tok = self._linear(objs, name + ".linear", tape=None)
        n_vis = self._layernorm(hs, name + ".norm1", tape=None)
        n_obj = self._layernorm(o_tok, name + ".norm1", tape=None)
        o = self._self_attention(n_vis, name + ".attn", heads, samples=g.B * g.F, seq=g.HW, rowmap=ops.RowMap(1, g.HW, 0, 1), tape=None,
                                 x_extra=n_obj, extra_map=ops.RowMap(1, nobj, 0, 1), seq_extra=nobj)
        hs = self._linear(o, name + ".attn.to_out.0", tape=None, res=hs, alpha=self.alpha[name + ".alpha_attn"])
        n2 = self._layernorm(hs, name + ".norm2", tape=None)
        return self._feed_forward(n2, name + ".ff", hs, tape=None, alpha=self.alpha[name + ".alpha_dense"])

    def _transformer2d(self, x, name, heads, g: Geom, text, *, tape, objs=None, key=None, collect=None, pair_split=False):
        h = self._groupnorm(x, name + ".norm", g.HW, tape=tape, eps=1e-6, silu=False)
        hs = self._linear(h, name + ".proj_in", tape=tape)
        hs = self._transformer_block(hs, name + ".transformer_blocks.0", heads, tape=tape, spatial=True, g=g, text=text, objs=objs, key=key, collect=collect,
                                     pair_split=pair_split)
        return self._linear(hs, name + ".proj_out", tape=tape, res=self._pair_rows(x, g) if pair_split else x)

    def _transformer_temporal(self, x, name, heads, g: Geom, *, tape):
        h = self._groupnorm(x, name + ".norm", g.F * g.HW, tape=tape, eps=1e-6, silu=False)
        hs = self._linear(h, name + ".proj_in", tape=tape)
        hs = self._transformer_block(hs, name + ".transformer_blocks.0", heads, tape=tape, spatial=False, g=g)
        return self._linear(hs, name + ".proj_out", tape=tape, res=x)

    def _resnet(self, x, name, g: Geom, temb_rows, *, tape, skip=None):
        eps = self.cfg.norm_eps
        h = self._groupnorm(x, name + ".norm1", g.HW, tape=tape, eps=eps, silu=True, x2=skip)
        a, b = self.temb_slices[name]
        h, _ = self._conv3x3(h, name + ".conv1", g, tape=tape, rowbias=temb_rows[:, a:b])
        h = self._groupnorm(h, name + ".norm2", g.HW, tape=tape, eps=eps, silu=True)
        if (name + ".conv_shortcut.weight") in self.w:
            sc = self._linear(x, name + ".conv_shortcut", tape=tape, x2=skip)
        else:
            assert skip is None
            sc = x
        out, _ = self._conv3x3(h, name + ".conv2", g, tape=tape, res=sc)
        return out

    def _temporal_conv(self, x, name, g: Geom, *, tape):
        rps = g.F * g.HW
        h = x
        for k, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
            h = self._groupnorm(h, f"{name}.{k}.0", rps, tape=tape, eps=1e-5, silu=True)
            h = self._tconv(h, f"{name}.{k}.{ci}", g, tape=tape, res=x if k == "conv4" else None)
        return h

    def position_net(self, boxes, masks, positive_embeddings):
        """PositionNet (unet_3d_condition.py:119-179) on [N,30,·] tensors -> objs token matrix [N*30, cross_dim] bf16.
        The Fourier features / null-feature blend are tiny host-side tensor prep; the MLP runs on the GEMM kernel."""
        m = masks.to(self.dev, torch.float32).unsqueeze(-1)
        boxes = boxes.to(self.dev, torch.float32)
        freq = 100.0 ** (torch.arange(8, dtype=torch.float32, device=self.dev) / 8)
        e = freq[None, None, None] * boxes.unsqueeze(-1)
        xyxy = torch.stack((e.sin(), e.cos()), -1).permute(0, 1, 3, 4, 2).reshape(*boxes.shape[:2], -1)
        xyxy = xyxy * m + (1 - m) * self.w["position_net.null_position_feature"].view(1, 1, -1)
        pos = positive_embeddings.to(self.dev, torch.float32) * m + (1 - m) * self.w["position_net.null_positive_feature"].view(1, 1, -1)
        h = torch.cat([pos, xyxy], -1).reshape(-1, pos.shape[-1] + xyxy.shape[-1]).to(torch.bfloat16).contiguous()
        h = ops.silu(self._linear(h, "position_net.linears.0", tape=None))
        h = ops.silu(self._linear(h, "position_net.linears.2", tape=None))
        return self._linear(h, "position_net.linears.4", tape=None)

    # ------------------------------------------------------------------ whole model
    def forward(self, sample, timestep, encoder_hidden_states=None, *, text: TextCache = None, gligen=None, fuser_enabled=True,
                tape: Tape = None, collect=None, cfg_pairs=False):
        """sample (B,4,F,h,w) fp32 CUDA -> noise prediction (B,4,F,h,w) fp32.

        cfg_pairs=True: classifier-free guidance on V samples.  `sample` holds the V latents ONCE; `text` (and `gligen`) hold 2V items
        ordered [uncond_0, cond_0, uncond_1, cond_1, ...]; the result has 2V items in that order.  The reference feeds the UNet
        torch.cat([latents] * 2) (controllable_pipeline…py:908-912), i.e. two IDENTICAL inputs that only start to differ at the first
        layer that reads the text (attn2 of the first spatial transformer) or the grounding tokens (its fuser): everything before that
        — conv_in, transformer_in, the first ResnetBlock2D / TemporalConvLayer, the first spatial self-attention — is computed once per
        sample and its rows are duplicated there (`_pair_rows`), as is the first skip connection when the up path reads it.  Same
        function, half the work on that prefix (level 0: one of the most expensive stretches of the network).

        collect = {"keys": [...], "q": {}, "stop_after": key}: store (q, k_text, heads, geom) of the listed
        cross-attention layers (attn_key tuples as in the reference) and optionally stop there.
        Returns None when stopped early.
        """
        cfg = self.cfg
        ops.use_device(self.dev)
        boc = cfg.block_out_channels
        dh = cfg.attention_head_dim
        assert dh == 64, "kernels are specialised for attention_head_dim 64"
        sample = sample.to(self.dev, torch.float32).contiguous()
        B, Cin, Fr, H, W = sample.shape
        g = Geom(B, Fr, H, W)
        if text is None:
            text = self.encode_text(encoder_hidden_states)
        pending = bool(cfg_pairs)  # the (uncond, cond) halves have not diverged yet
        if pending:
            assert tape is None and collect is None, "cfg_pairs is a plain (unrecorded) forward"
            B = 2 * B  # batch of the result and of everything behind the split
        assert text.B == B
        if torch.is_tensor(timestep):
            tt = timestep.to(self.dev, torch.float32).reshape(-1)
            # the shared CFG prefix runs on V samples and reads temb row v for sample v — item v's row, not item 2v's: only the same
            # timestep for every item (what every caller passes; the reference feeds ONE t per step) makes those rows interchangeable
            assert not pending or tt.numel() == 1 or bool((tt == tt[0]).all()), "cfg_pairs: one timestep for all items (scalar or all entries equal)"
            t = tt.expand(B).contiguous()
        else:
            t = torch.full((B,), float(timestep), dtype=torch.float32, device=self.dev)
        temb = ops.timestep_embedding(t, boc[0])
        emb = self._linear(ops.silu(self._linear(temb, "time_embedding.linear_1", tape=None)), "time_embedding.linear_2", tape=None)
        # every ResnetBlock2D applies SiLU to temb first, then its own projection: all projections in one product [B, sum of Cout] fp32
        temb_rows = ops.gemm(ops.silu(emb), self.w["time_emb_proj_all.weight"], bias=self.w["time_emb_proj_all.bias"], out_fp32=True)

        tokens = ops.latents_to_tokens(sample, cpad=8)
        x, _ = self._conv3x3(tokens, "conv_in", g, tape=tape)
        x = self._transformer_temporal(x, "transformer_in", cfg.transformer_in_heads, g, tape=tape)
        objs = None
        if gligen is not None and fuser_enabled and cfg.gated:
            objs = self.position_net(gligen["boxes"], gligen["masks"], gligen["positive_embeddings"])

        def layer(prefix, j, key, x, g, has_attn, c, skip=None):
            nonlocal pending
            x = self._resnet(x, f"{prefix}.resnets.{j}", g, temb_rows, tape=tape, skip=skip)
            x = self._temporal_conv(x, f"{prefix}.temp_convs.{j}", g, tape=tape)
            if has_attn:
                x = self._transformer2d(x, f"{prefix}.attentions.{j}", c // dh, g, text, tape=tape, objs=objs, key=key, collect=collect, pair_split=pending)
                if pending:
                    pending, g = False, Geom(2 * g.B, g.F, g.H, g.W)
                x = self._transformer_temporal(x, f"{prefix}.temp_attentions.{j}", c // dh, g, tape=tape)
            return x, g

        def paired(x, gx):  # a tensor produced before the split, used behind it
            return (self._pair_rows(x, gx), Geom(2 * gx.B, gx.F, gx.H, gx.W)) if cfg_pairs and not pending and gx.B * 2 == B else (x, gx)

        try:
            skips = [(x, g)]
            for i, btype in enumerate(cfg.down_block_types):
                c = boc[i]
                for j in range(cfg.layers_per_block):
                    x, g = layer(f"down_blocks.{i}", j, ("down", i, j, 0), x, g, btype == "CrossAttnDownBlock3D", c)
                    skips.append((x, g))
                if i != len(boc) - 1:
                    x, g = self._conv3x3(x, f"down_blocks.{i}.downsamplers.0.conv", g, tape=tape, stride=2)
                    skips.append((x, g))
            c = boc[-1]
            x = self._resnet(x, "mid_block.resnets.0", g, temb_rows, tape=tape)
            x = self._temporal_conv(x, "mid_block.temp_convs.0", g, tape=tape)
            x = self._transformer2d(x, "mid_block.attentions.0", c // dh, g, text, tape=tape, objs=objs, key=("mid", 0, 0, 0), collect=collect, pair_split=pending)
            if pending:  # a topology without attention in the down path
                pending, g = False, Geom(2 * g.B, g.F, g.H, g.W)
            x = self._transformer_temporal(x, "mid_block.temp_attentions.0", c // dh, g, tape=tape)
            x = self._resnet(x, "mid_block.resnets.1", g, temb_rows, tape=tape)
            x = self._temporal_conv(x, "mid_block.temp_convs.1", g, tape=tape)
            rev = list(reversed(boc))
            for i, btype in enumerate(cfg.up_block_types):
                c = rev[i]
                for j in range(cfg.layers_per_block + 1):
                    skip, gs = paired(*skips.pop())
                    assert (gs.H, gs.W, gs.B) == (g.H, g.W, g.B)
                    x, g = layer(f"up_blocks.{i}", j, ("up", i, j, 0), x, g, btype == "CrossAttnUpBlock3D", c, skip=skip)
                if i != len(boc) - 1:
                    gt = skips[-1][1]
                    x, g = self._conv3x3(x, f"up_blocks.{i}.upsamplers.0.conv", g, tape=tape, upsample=1, up_to=(gt.H, gt.W))
        except StopForward:
            self._tokens_in = tokens
            return None
        self._tokens_in = tokens
        h = self._groupnorm(x, "conv_norm_out", g.HW, tape=tape, eps=cfg.norm_eps, silu=True)
        out32, _ = self._conv3x3(h, "conv_out", g, tape=None, out_fp32=True)
        return ops.tokens_to_latents(out32, B, cfg.out_channels, Fr, H, W)

    def forward_cfg(self, latents, timestep, *, text: TextCache, gligen=None, fuser_enabled=True):
        """Noise predictions of V samples for classifier-free guidance: latents (V,4,F,h,w), text / gligen of 2V items ordered
        [uncond_0, cond_0, ...] -> (2V,4,F,h,w) in that order (controllable_pipeline…py:908-923).  The shared prefix runs once per sample."""
        if self.cfg_shared_prefix:
            return self.forward(latents, timestep, text=text, gligen=gligen, fuser_enabled=fuser_enabled, cfg_pairs=True)
        x2 = latents.repeat_interleave(2, 0).contiguous()
        return self.forward(x2, timestep, text=text, gligen=gligen, fuser_enabled=fuser_enabled)

    def input_gradient(self, tape: Tape, g: Geom, scale=1.0):
        """After tape.backward(): gradient w.r.t. the (B,4,F,h,w) latents."""
        dtok = tape.pop(self._tokens_in)
        assert dtok is not None, "no gradient reached the latents"
        return ops.tokens_grad_to_latents(dtok, g.B, self.cfg.in_channels, g.F, g.H, g.W, scale=scale)
