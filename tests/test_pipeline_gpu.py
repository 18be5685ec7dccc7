"""GPU tests of the reference-shaped host surface: UNet3DConditionModel / HipAttnProcessor / TextToVideoSDPipeline."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

import lvd_amd  # noqa: E402
from lvd_amd.guidance import hip_latent_backward_guidance  # noqa: E402
from lvd_amd.models.attention_processor import HipAttnProcessor  # noqa: E402
from lvd_amd.models.controllable_pipeline_text_to_video_synth import TextToVideoSDPipeline  # noqa: E402
from lvd_amd.models.unet_3d_condition import UNet3DConditionModel  # noqa: E402
from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict  # noqa: E402
from oracle import guidance_ref, scheduler_ref, unet_ref  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_model_forward_and_saved_maps_vs_reference_golden():
    g = np.load(os.path.join(G, "unet_tiny.npz"))
    sd = synthetic_state_dict(UNetConfig(**TINY), seed=0)
    unet = UNet3DConditionModel.from_state_dict(sd, **TINY)
    keys = [("down", 1, 0, 0), ("down", 2, 0, 0), ("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 2, 1, 0)]
    saved = {}
    out = unet(torch.from_numpy(g["sample"]).cuda(), int(g["timestep"]), torch.from_numpy(g["ehs"]).cuda(),
               cross_attention_kwargs={"save_attn_to_dict": saved, "save_keys": keys}, return_dict=False)[0]
    assert rel(out, g["out"]) < 3e-2
    for k in keys:
        assert rel(saved[k], g["attn_" + "_".join(map(str, k))]) < 3e-2, k
    assert unet(torch.from_numpy(g["sample"]).cuda(), 500, torch.from_numpy(g["ehs"]).cuda()).sample.shape == out.shape
    with pytest.raises(RuntimeError):  # no autograd graph: loud, not silent
        with torch.enable_grad():
            unet(torch.from_numpy(g["sample"]).cuda().requires_grad_(True), 500, torch.from_numpy(g["ehs"]).cuda())


class _Attn(nn.Module):
    def __init__(self, dim, ctx, heads):
        super().__init__()
        self.heads, self.scale = heads, 0.125
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, dim, bias=False), nn.Linear(ctx, dim, bias=False), nn.Linear(ctx, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])


def test_attn_processor_plugin():
    torch.manual_seed(0)
    attn = _Attn(128, 96, 2).cuda()
    x, ctx = torch.randn(3, 50, 128, device="cuda"), torch.randn(3, 77, 96, device="cuda")
    proc = HipAttnProcessor()
    saved = {}
    out = proc(attn, x, encoder_hidden_states=ctx, attn_key=["down", 1, 0, 0], save_attn_to_dict=saved, save_keys=[("down", 1, 0, 0)])
    q, k, v = attn.to_q(x), attn.to_k(ctx), attn.to_v(ctx)
    sp = lambda t: t.reshape(3, -1, 2, 64).permute(0, 2, 1, 3)
    probs = (sp(q) @ sp(k).transpose(-1, -2) * 0.125).softmax(-1)
    ref = attn.to_out[0]((probs @ sp(v)).permute(0, 2, 1, 3).reshape(3, 50, 128))
    assert rel(out, ref) < 2e-2 and rel(saved[("down", 1, 0, 0)], probs) < 2e-2
    out_self = proc(_Attn(128, 128, 2).cuda(), x, attn_key=["in", 0, 0])
    assert out_self.shape == x.shape and torch.isfinite(out_self).all()
    # attention_mask on cross-attention: the additive per-text-position bias of models/transformer_2d.py:303-307 ((1 - mask) * -10000, (batch, 1, text))
    keep = torch.ones(3, 77, device="cuda")
    keep[0, 40:] = 0
    keep[2, 9:] = 0
    bias = ((1 - keep) * -10000.0).unsqueeze(1)
    sm = {}
    om = proc(attn, x, encoder_hidden_states=ctx, attention_mask=bias, attn_key=["down", 1, 0, 0], save_attn_to_dict=sm, save_keys=[("down", 1, 0, 0)])
    pm = (sp(q) @ sp(k).transpose(-1, -2) * 0.125 + bias[:, None]).softmax(-1)
    refm = attn.to_out[0]((pm @ sp(v)).permute(0, 2, 1, 3).reshape(3, 50, 128))
    assert rel(om, refm) < 2e-2 and rel(sm[("down", 1, 0, 0)], pm) < 2e-2 and float(sm[("down", 1, 0, 0)][0, :, :, 40:].abs().max()) == 0.0
    assert rel(proc(attn, x, encoder_hidden_states=ctx, attention_mask=bias[:, 0]), refm) < 2e-2  # (batch, text) spelling
    with pytest.raises(NotImplementedError):  # a mask over the queries, a boolean mask, or a self-attention mask: no kernel
        proc(attn, x, encoder_hidden_states=ctx, attention_mask=torch.zeros(3, 50, 77, device="cuda"))
    with pytest.raises(NotImplementedError):
        proc(attn, x, encoder_hidden_states=ctx, attention_mask=keep.bool())
    with pytest.raises(NotImplementedError):
        proc(_Attn(128, 128, 2).cuda(), x, attention_mask=torch.zeros(3, 1, 50, device="cuda"))
    # slicing / pairing options of the saved map (models/attention_processor.py:566-583)
    o2, p2 = proc(attn, x[:2], encoder_hidden_states=ctx[:2], attn_key=["up", 1, 0, 0], return_attntion_probs=True, return_token_ca_only=5,
                  return_cond_ca_only=True, offload_cross_attn_to_cpu=True)
    assert p2.device.type == "cpu" and p2.shape == (1, 2, 50, 1) and rel(p2, probs[1:2, :, :, 5:6]) < 2e-2 and rel(o2, ref[:2]) < 2e-2
    _, p3 = proc(attn, x, encoder_hidden_states=ctx, attn_key=["up", 1, 0, 0], return_attntion_probs=True, return_token_ca_only=torch.tensor([2, 7], device="cuda"))
    assert p3.shape == (3, 2, 50, 2) and rel(p3, probs[..., [2, 7]]) < 2e-2


def test_attn_processor_attn_process_fn():
    """`attn_process_fn` (models/attention_processor.py:537-549): the callback receives the (batch*heads, L, T) cross-attention probabilities
    with head-batched q / k / v and its return value is what multiplies V; the map saved / returned is the one BEFORE the rewrite (:553-556).
    Self-attention never calls it (:459-474)."""
    torch.manual_seed(1)
    attn = _Attn(128, 96, 2).cuda()
    x, ctx = torch.randn(2, 70, 128, device="cuda"), torch.randn(2, 77, 96, device="cuda")
    proc = HipAttnProcessor()
    seen = {}

    def boost_token_3(p, query, key, value, attn_key=None, cross_attn=None, batch_size=None, heads=None):
        seen.update(shape=tuple(p.shape), q=tuple(query.shape), k=tuple(key.shape), v=tuple(value.shape), key=attn_key, cross=cross_attn, b=batch_size, h=heads)
        p = p.clone()
        p[:, :, 3] += 0.5
        return p / p.sum(-1, keepdim=True)

    saved = {}
    out = proc(attn, x, encoder_hidden_states=ctx, attn_key=["mid", 0, 0, 0], attn_process_fn=boost_token_3, save_attn_to_dict=saved, save_keys=[("mid", 0, 0, 0)])
    assert seen == dict(shape=(4, 70, 77), q=(4, 70, 64), k=(4, 77, 64), v=(4, 77, 64), key=["mid", 0, 0, 0], cross=True, b=2, h=2)
    q, k, v = attn.to_q(x), attn.to_k(ctx), attn.to_v(ctx)
    sp = lambda t: t.reshape(2, -1, 2, 64).permute(0, 2, 1, 3)
    probs = (sp(q) @ sp(k).transpose(-1, -2) * 0.125).softmax(-1)
    rewritten = boost_token_3(probs.reshape(4, 70, 77), sp(q).reshape(4, 70, 64), sp(k).reshape(4, 77, 64), sp(v).reshape(4, 77, 64)).reshape(2, 2, 70, 77)
    ref = attn.to_out[0]((rewritten @ sp(v)).permute(0, 2, 1, 3).reshape(2, 70, 128))
    plain = attn.to_out[0]((probs @ sp(v)).permute(0, 2, 1, 3).reshape(2, 70, 128))
    assert rel(out, ref) < 2e-2 and rel(out, plain) > 5 * rel(out, ref)
    assert rel(saved[("mid", 0, 0, 0)], probs) < 2e-2  # the map before the rewrite
    called = []
    proc(_Attn(128, 128, 2).cuda(), x, attn_key=["in", 0, 0], attn_process_fn=lambda *a, **kw: called.append(1))
    assert not called


def test_pipeline_guided_sampling_vs_oracle_loop():
    """4 DPM steps with backward guidance on the first 2, tiny UNet: final latents vs the all-oracle loop (fp32 CPU)."""
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    unet = UNet3DConditionModel.from_state_dict(sd, **TINY)
    pipe = TextToVideoSDPipeline(unet=unet).to("cuda")
    gen = torch.Generator().manual_seed(3)
    lat0 = torch.randn(1, 4, 4, 16, 16, generator=gen)
    pe, ne = torch.randn(1, 77, 64, generator=gen), torch.randn(1, 77, 64, generator=gen)
    keys = [("down", 1, 0, 0), ("up", 1, 1, 0)]
    boxes, pos = [[[0.1, 0.2, 0.6, 0.8], [0.2, 0.2, 0.7, 0.8], [0.3, 0.2, 0.8, 0.8], [0.4, 0.2, 0.9, 0.8]]], [[2]]
    bg = dict(bboxes=boxes, object_positions=pos, loss_scale=5.0, loss_threshold=0.01, max_iter=1, max_index_step=2, fg_top_p=0.5,
              bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03, guidance_attn_keys=keys, verbose=False)
    out = pipe(prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), height=128, width=128, num_frames=4, num_inference_steps=4,
               guidance_scale=9.0, latents=lat0.clone(), output_type="latent", backward_guidance_kwargs=bg,
               custom_latent_backward_guidance=hip_latent_backward_guidance).frames
    # oracle loop
    sch = scheduler_ref.DPMSolverPP2M(timestep_spacing="leading", steps_offset=1)
    sch.set_timesteps(4)
    lat, loss = lat0.clone(), 10000.0
    both = torch.cat([ne, pe])

    def unet_fn(x, t, cond, save, save_keys):
        unet_ref.unet_forward(sd, cfg, x, int(t), cond, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=keys[-1])

    hp = {k: v for k, v in bg.items() if k not in ("bboxes", "object_positions", "verbose")}
    for i, t in enumerate(sch.timesteps):
        lat, loss = guidance_ref.latent_backward_guidance(unet_fn, sch.alphas_cumprod, pe, i, boxes, pos, int(t), lat, loss, base_attn_dim=(16, 16), **hp)
        with torch.no_grad():
            eps = unet_ref.unet_forward(sd, cfg, lat.expand(2, -1, -1, -1, -1), int(t), both)
        e = eps[0:1] + 9.0 * (eps[1:2] - eps[0:1])
        lat = sch.step(e, lat)
    err = rel(out, lat)
    print("pipeline 4-step latents rel-L2 vs oracle loop:", err)
    assert err < 8e-2


def test_pipeline_lvd_plus_loop_vs_oracle_loop():
    """lvd_plus = backward guidance + GLIGEN adapters (BASELINE configs[4] control flow, generation/lvd_plus.py:75-210): 4 DPM
    steps on the gated TINY topology, guidance on the first 2, gligen_scheduled_sampling_beta = 0.5 so the fusers are switched off
    from step 2 on (controllable_pipeline_text_to_video_synth.py:816-839); the guidance forward runs without the GLIGEN inputs
    (models/pipelines.py:66-82).  Final latents vs the all-oracle loop."""
    cfg = UNetConfig(attention_type="gated", **TINY)
    sd = synthetic_state_dict(cfg, seed=1)
    unet = UNet3DConditionModel.from_state_dict(sd, attention_type="gated", **TINY)
    pipe = TextToVideoSDPipeline(unet=unet).to("cuda")
    gen = torch.Generator().manual_seed(5)
    Fr = 4
    lat0 = torch.randn(1, 4, Fr, 16, 16, generator=gen)
    pe, ne = torch.randn(1, 77, 64, generator=gen), torch.randn(1, 77, 64, generator=gen)
    keys = [("down", 1, 0, 0), ("up", 1, 1, 0)]
    boxes = [[[0.1 + 0.1 * f, 0.2, 0.6 + 0.1 * f, 0.8] for f in range(Fr)]]
    pos = [[2]]
    bg = dict(bboxes=boxes, object_positions=pos, loss_scale=5.0, loss_threshold=0.01, max_iter=1, max_index_step=2, fg_top_p=0.5,
              bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03, guidance_attn_keys=keys, verbose=False)
    gl_boxes = [[boxes[0][f]] for f in range(Fr)]
    gl_phr = [["a bear"]] * Fr
    phr_emb = torch.randn(Fr, 1, 64, generator=gen)
    out = pipe(prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), height=128, width=128, num_frames=Fr, num_inference_steps=4,
               guidance_scale=9.0, latents=lat0.clone(), output_type="latent", backward_guidance_kwargs=bg,
               custom_latent_backward_guidance=hip_latent_backward_guidance, gligen_boxes=gl_boxes, gligen_phrases=gl_phr,
               gligen_phrase_embeds=phr_emb, gligen_scheduled_sampling_beta=0.5).frames
    # oracle loop
    gb, ge, gm = torch.zeros(2 * Fr, 30, 4), torch.zeros(2 * Fr, 30, 64), torch.zeros(2 * Fr, 30)
    for f in range(Fr):
        for half in range(2):
            gb[half * Fr + f, 0] = torch.tensor(boxes[0][f])
            ge[half * Fr + f, 0] = phr_emb[f, 0]
        gm[Fr + f, 0] = 1.0
    gl = {"boxes": gb, "positive_embeddings": ge, "masks": gm}
    sch = scheduler_ref.DPMSolverPP2M(timestep_spacing="leading", steps_offset=1)
    sch.set_timesteps(4)
    lat, loss = lat0.clone(), 10000.0
    both = torch.cat([ne, pe])

    def unet_fn(x, t, cond, save, save_keys):
        unet_ref.unet_forward(sd, cfg, x, int(t), cond, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=keys[-1])

    hp = {k: v for k, v in bg.items() if k not in ("bboxes", "object_positions", "verbose")}
    for i, t in enumerate(sch.timesteps):
        lat, loss = guidance_ref.latent_backward_guidance(unet_fn, sch.alphas_cumprod, pe, i, boxes, pos, int(t), lat, loss, base_attn_dim=(16, 16), **hp)
        with torch.no_grad():
            eps = unet_ref.unet_forward(sd, cfg, lat.expand(2, -1, -1, -1, -1), int(t), both, gligen=gl, fuser_enabled=i < 2)
        lat = sch.step(eps[0:1] + 9.0 * (eps[1:2] - eps[0:1]), lat)
    with torch.no_grad():  # the same loop with the fusers never on must differ: the adapters were really exercised
        lat_nf = lat0.clone()
        sch2 = scheduler_ref.DPMSolverPP2M(timestep_spacing="leading", steps_offset=1)
        sch2.set_timesteps(4)
        e0 = unet_ref.unet_forward(sd, cfg, lat_nf.expand(2, -1, -1, -1, -1), int(sch2.timesteps[0]), both, gligen=gl, fuser_enabled=False)
        e1 = unet_ref.unet_forward(sd, cfg, lat_nf.expand(2, -1, -1, -1, -1), int(sch2.timesteps[0]), both, gligen=gl, fuser_enabled=True)
    assert rel(e0, e1) > 1e-3
    err = rel(out, lat)
    print("lvd_plus 4-step latents rel-L2 vs oracle loop:", err)
    assert err < 8e-2


def test_sample_many_pairs_every_sample_with_its_own_text_and_noise():
    """pipeline.sample_many (V samples through one denoising loop: V guidance passes + ONE CFG forward of batch 2V per step): the batch must
    be a set of independent samples.  (a) Two copies of one (prompt embeddings, latents, layout) give BIT-IDENTICAL videos (a mis-paired text
    or layout would separate them); (b) swapping the order of two different samples swaps the outputs bit for bit (same batch, other slots);
    (c) each sample of a V = 2 run equals its own V = 1 run up to the batch-consistency distance (tile geometries are chosen per M), checked
    on the latents after ONE step — before classifier-free guidance at scale 9 has amplified it — below 4e-2, and the two samples differ from
    each other by far more than that."""
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    unet = UNet3DConditionModel.from_state_dict(sd, **TINY)
    pipe = TextToVideoSDPipeline(unet=unet).to("cuda")
    gen = torch.Generator().manual_seed(5)
    keys = [("down", 1, 0, 0), ("up", 1, 1, 0)]

    def sample(seed, x0):
        g = torch.Generator().manual_seed(seed)
        boxes = [[[x0 + 0.05 * f, 0.2, x0 + 0.5 + 0.05 * f, 0.8] for f in range(4)]]
        return dict(prompt_embeds=torch.randn(1, 77, 64, generator=g).cuda(), negative_prompt_embeds=torch.randn(1, 77, 64, generator=g).cuda(),
                    latents=torch.randn(1, 4, 4, 16, 16, generator=g),
                    backward_guidance_kwargs=dict(bboxes=boxes, object_positions=[[2]], loss_scale=5.0, loss_threshold=0.01, max_iter=1, max_index_step=2,
                                                  fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03, guidance_attn_keys=keys,
                                                  verbose=False))

    def run(samples, steps):
        return pipe.sample_many([dict(s, latents=s["latents"].clone()) for s in samples], height=128, width=128, num_frames=4, num_inference_steps=steps,
                                guidance_scale=9.0, output_type="latent", custom_latent_backward_guidance=hip_latent_backward_guidance)

    a, b = sample(11, 0.1), sample(12, 0.3)
    # (a) duplicates
    twice = run([a, a], 3)
    assert torch.equal(twice[0], twice[1]), "two copies of one sample came out different: samples of a batch are not independent"
    # (b) permutation
    ab, ba = run([a, b], 3), run([b, a], 3)
    assert torch.equal(ab[0], ba[1]) and torch.equal(ab[1], ba[0]), "swapping two samples did not swap their outputs"
    assert rel(ab[0], ab[1]) > 0.3
    # (c) V = 2 against V = 1 on the step-0 noise prediction (what sample_many feeds the DPM update: engine.forward_cfg of the 2V batch)
    engine = unet._ensure_engine()
    pe = lambda s: torch.cat([s["negative_prompt_embeds"], s["prompt_embeds"]])
    xa, xb = a["latents"].cuda(), b["latents"].cuda()
    t0 = int(pipe.scheduler.timesteps[0]) if len(pipe.scheduler.timesteps) else 961
    e2 = engine.forward_cfg(torch.cat([xa, xb]), t0, text=engine.encode_text(torch.cat([pe(a), pe(b)])))
    ea1 = engine.forward_cfg(xa, t0, text=engine.encode_text(pe(a)))
    eb1 = engine.forward_cfg(xb, t0, text=engine.encode_text(pe(b)))
    ea, eb = rel(e2[0:2], ea1), rel(e2[2:4], eb1)
    print(f"sample_many's CFG forward, V = 2 vs V = 1 noise prediction: {ea:.2e} / {eb:.2e}; sample a vs sample b {rel(ea1, eb1):.2f}")
    assert ea < 4e-2 and eb < 4e-2
    assert rel(e2[0:2], eb1) > 0.3 and rel(e2[2:4], ea1) > 0.3
