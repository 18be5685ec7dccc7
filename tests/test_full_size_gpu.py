"""The headline configuration itself against the fp32 oracle: BASELINE configs[1] / [2] / [4] at 576x320x24 (latent 40x72, 24 frames) on the
1411 M / 1624 M topologies — every product shape of the timed step (M = 138 240 / 69 120 rows at level 0, the half-tile tails, the K-split
plans, the shipped autotune table) checked by value, not only by properties.

The oracle side is `oracle/unet_ref.unet_forward` in fp32 on the host (16 threads: the count bench.py's thread probe picks on the GPU
hosts; 42.8 TFLOP at 0.8-1.0 TFLOP/s ~ 50 s per CFG forward).  The guidance iteration needs autograd through the oracle at full size
(every activation of the cond branch retained in fp32, ~70 GB): it runs only when the host reports enough free memory and says so otherwise.

Reference: /root/reference/models/unet_3d_condition.py:642-859 (forward), /root/reference/models/controllable_pipeline_text_to_video_synth.py:908-923
(the CFG batch), /root/reference/models/pipelines.py:21-150 (guidance iteration)."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

import lvd_amd  # noqa: E402,F401
from lvd_amd import guidance  # noqa: E402
from lvd_amd.engine import HipUNet3D  # noqa: E402
from lvd_amd.weights import UNetConfig, synthetic_state_dict  # noqa: E402
from oracle import guidance_ref, scheduler_ref, unet_ref  # noqa: E402

FRAMES, LAT_H, LAT_W = 24, 40, 72
KEYS = [("down", 1, 0, 0), ("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 2, 0)]  # generation/lvd.py:66-73
ORACLE_THREADS = 16


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def mem_available_gb():
    with open("/proc/meminfo") as fh:
        for line in fh:
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2**20
    return 0.0


@pytest.fixture(scope="module")
def oracle_threads():
    before = torch.get_num_threads()
    torch.set_num_threads(min(ORACLE_THREADS, before) if before >= 8 else before)
    yield torch.get_num_threads()
    torch.set_num_threads(before)


@pytest.fixture(scope="module")
def full():
    cfg = UNetConfig()
    sd = synthetic_state_dict(cfg, seed=0, device="cuda")
    net = HipUNet3D(cfg, sd, device="cuda")
    return cfg, net, {k: v.float().cpu() for k, v in sd.items()}


def test_headline_cfg_forward_vs_oracle(full, oracle_threads):
    """configs[1]: what bench.py and the pipeline time — engine.forward_cfg (shared classifier-free-guidance prefix on) at (1,4,24,40,72) —
    against the oracle of the reference's form unet(cat([latents] * 2)) with the (negative, positive) text pair."""
    cfg, net, sd = full
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, FRAMES, LAT_H, LAT_W, generator=gen)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=gen)
    t0 = time.time()
    with torch.no_grad():
        ref = unet_ref.unet_forward(sd, cfg, torch.cat([x] * 2), 500, ehs)
    dt = time.time() - t0
    text = net.encode_text(ehs.cuda())
    out = net.forward_cfg(x.cuda(), 500, text=text)
    dup = net.forward(torch.cat([x] * 2).cuda(), 500, text=text)  # the duplicated batch through the engine (no shared prefix)
    e, e_dup = rel(out, ref), rel(dup, ref)
    worst_frame = max(rel(out[:, :, f], ref[:, :, f]) for f in range(FRAMES))
    print(f"576x320x24 CFG forward vs fp32 oracle ({dt:.0f} s on {oracle_threads} threads): forward_cfg rel-L2 {e:.4f} (worst frame {worst_frame:.4f}), "
          f"duplicated batch {e_dup:.4f}")
    assert out.shape == ref.shape == (2, 4, FRAMES, LAT_H, LAT_W)
    assert e < 3e-2 and e_dup < 3e-2
    assert worst_frame < 4e-2  # no frame (row block of 2880 tokens) stands out: a tile-geometry bug would be local


def test_headline_guidance_iteration_vs_oracle_autograd(full, oracle_threads):
    """configs[1], the guided half of the step: ONE guidance iteration (recorded forward to the last guidance key, fused loss, hand-written
    backward, latent update) at (1,4,24,40,72) with the README's weak-guidance hyper-parameters and a two-object layout with a
    disappearing box, against the oracle's forward + compute_ca_loss + torch.autograd.grad."""
    need = 160.0
    have = mem_available_gb()
    if have < need:
        pytest.skip(f"full-size autograd through the fp32 oracle retains ~70 GB of activations; MemAvailable is {have:.0f} GB < {need:.0f} GB")
    cfg, net, sd = full
    gen = torch.Generator().manual_seed(1)
    lat0 = torch.randn(1, 4, FRAMES, LAT_H, LAT_W, generator=gen)
    cond = torch.randn(1, 77, cfg.cross_attention_dim, generator=gen)
    bear = [[0.0 + 0.8301 * f / 23, 0.5, 0.1953 + 0.8301 * f / 23, 0.6953] for f in range(FRAMES)]
    ball = [([0.45, 0.7, 0.6, 0.9] if not 9 <= f < 15 else [0.0] * 4) for f in range(FRAMES)]
    boxes, pos = [bear, ball], [[2], [7, 8]]
    hp = dict(loss_scale=2.5, loss_threshold=0.0, max_index_step=10, fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0,
              com_loss_scale=0.03, guidance_attn_keys=KEYS)
    sched = scheduler_ref.DPMSolverPP2M()
    t = 801

    def unet_fn(x, tt, c, save, save_keys):
        unet_ref.unet_forward(sd, cfg, x, int(tt), c, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=KEYS[-1])

    t0 = time.time()
    ref_lat, ref_loss = guidance_ref.latent_backward_guidance(unet_fn, sched.alphas_cumprod, cond, 0, boxes, pos, t, lat0.clone(), 10000.0,
                                                              max_iter=1, base_attn_dim=(LAT_H, LAT_W), **hp)
    dt = time.time() - t0
    lat, loss = guidance.hip_latent_backward_guidance(sched, net, cond.cuda(), 0, boxes, pos, t, lat0.clone().cuda(), torch.tensor(10000.0),
                                                      max_iter=1, **hp)
    d, d_ref = (lat.cpu() - lat0).double(), (ref_lat - lat0).double()
    cos = float((d * d_ref).sum() / (d.norm() * d_ref.norm()))
    print(f"576x320x24 guidance iteration vs oracle autograd ({dt:.0f} s, MemAvailable was {have:.0f} GB): loss {float(loss):.5f} vs {ref_loss:.5f}; "
          f"update rel-L2 {rel(d, d_ref):.4f}, cosine {cos:.5f}")
    assert abs(float(loss) - ref_loss) < 2e-2 * abs(ref_loss)
    # Yardstick: the fp32 oracle with activations / activation gradients rounded to bf16 where the product stores them (oracle/bf16_storage.py)
    # is 0.0729 away from the fp32 oracle on exactly this problem (tools/full_size_oracle_probe.py --floor, gpurun_out -> profiles/r06_full_size_parity.txt);
    # the HIP path measured 0.0727, cosine 0.99736.  Bound = 1.4 x that floor.
    assert rel(d, d_ref) < 0.10 and cos > 0.995


def test_headline_gated_cfg_forward_vs_oracle(oracle_threads):
    """configs[2] / [4]: the GLIGEN topology (1624 M parameters) with the fusers ON at (2,4,24,40,72): 2880 queries + 30 grounding tokens per
    frame, PositionNet over 48 x 30 slots."""
    from test_full_topology_gpu import _gligen_inputs
    cfg = UNetConfig(attention_type="gated")
    sd = synthetic_state_dict(cfg, seed=1, device="cuda")
    net = HipUNet3D(cfg, sd, device="cuda")
    sd_cpu = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(1, 4, FRAMES, LAT_H, LAT_W, generator=gen)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=gen)
    gl = _gligen_inputs(cfg, 2, FRAMES, 3, gen)
    x2 = torch.cat([x] * 2)
    t0 = time.time()
    with torch.no_grad():
        ref = unet_ref.unet_forward(sd_cpu, cfg, x2, 500, ehs, gligen=gl)
    dt = time.time() - t0
    out = net.forward(x2.cuda(), 500, ehs.cuda(), gligen=gl)
    off = net.forward(x2.cuda(), 500, ehs.cuda(), gligen=gl, fuser_enabled=False)
    e = rel(out, ref)
    print(f"576x320x24 gated CFG forward (fusers on) vs fp32 oracle ({dt:.0f} s): rel-L2 {e:.4f}; fusers-off run is {rel(off, ref):.3f} away from it")
    assert e < 3e-2
    assert rel(off, ref) > 3 * e  # the adapters' contribution is resolved, not lost in the error
