"""CPU tests of the host logic: C-ABI symbol export, reference-shaped error behaviour, schedule/guidance host maths."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import lvd_amd  # noqa: F401
from lvd_amd import guidance, hip
from lvd_amd.models.controllable_pipeline_text_to_video_synth import TextToVideoSDPipeline
from lvd_amd.models.unet_3d_condition import UNet3DConditionModel
from lvd_amd.sampler import DPMSolverPP2MSchedule
from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict
from oracle import guidance_ref, scheduler_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/lvdhip.h <-> liblvdhip.so <-> hip.SYMBOLS agree (no compute call is made: no GPU here)."""
    header = open(os.path.join(ROOT, "include", "lvdhip.h")).read()
    declared = set(re.findall(r"\b(lvdhip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.lib().lvdhip_version() == hip.ABI_VERSION  # lib() itself refuses a library of another ABI version


def test_struct_sizes_match_header():
    """Field order/size drift between include/lvdhip.h and the ctypes mirrors would corrupt launches silently."""
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "lvdhip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(lvd_gemm_params), sizeof(lvd_gn_stats_params), sizeof(lvd_gn_apply_params), sizeof(lvd_gn_bwd_stats_params), sizeof(lvd_gn_bwd_apply_params), sizeof(lvd_ln_params), sizeof(lvd_ln_bwd_params), sizeof(lvd_attn_params), sizeof(lvd_attn_bwd_params), sizeof(lvd_ca_probs_params), sizeof(lvd_ca_select_params), sizeof(lvd_ca_dq_params), sizeof(lvd_ca_probs_full_params), sizeof(lvd_ca_apply_probs_params));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        sizes = list(map(int, subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split()))
    mirrors = [hip.GemmParams, hip.GnStatsParams, hip.GnApplyParams, hip.GnBwdStatsParams, hip.GnBwdApplyParams, hip.LnParams, hip.LnBwdParams,
               hip.AttnParams, hip.AttnBwdParams, hip.CaProbsParams, hip.CaSelectParams, hip.CaDqParams, hip.CaProbsFullParams, hip.CaApplyProbsParams]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]


def test_shipped_gemm_autotune_table_is_loadable():
    """profiles/gemm_autotune_576x320x24.json (loaded by default by bench.py / generate.py) has the key layout `ops.gemm` looks up and
    only names tile geometries the library knows."""
    from lvd_amd import ops
    path = os.path.join(ROOT, "profiles", "gemm_autotune_576x320x24.json")
    assert os.path.exists(path)
    saved = dict(ops._gemm_choice)
    try:
        ops._gemm_choice.clear()
        ops.load_gemm_autotune_table(path)
        tab = ops.gemm_autotune_table()
        assert len(tab) >= 150
        known = set(ops.GEMM_CANDIDATES) | {ops.SPLITK_VARIANT, ops.SPLITK_WIDE_VARIANT} | set(ops.TAIL_VARIANTS) | set(ops.HALO_VARIANTS)
        folded = 0
        for k, v in tab.items():
            assert len(k) in (14, 15) and v in known, (k, v)
            if len(k) == 15:  # a LayerNorm-folded product (round 4): plain loader, asm-DMA ring or K-split variants only
                folded += 1
                assert k[14] is True and k[0] == ops.A_PLAIN and (v >= 100 or v in (ops.SPLITK_VARIANT, ops.SPLITK_WIDE_VARIANT)), (k, v)
        assert folded >= 20
        for k, v in tab.items():
            assert k[0] in (ops.A_PLAIN, ops.A_CONV3X3, ops.A_TCONV3, ops.A_CONV3X3_T2) and (k[7] is None or len(k[7]) == 4)
        # the headline shapes are in it: level-0 QKV projection of the CFG batch and the level-0 resnet conv
        assert any(k[:4] == (ops.A_PLAIN, 138240, 960, 320) for k in tab) and any(k[:4] == (ops.A_CONV3X3, 138240, 320, 2880) for k in tab)
    finally:
        ops._gemm_choice.clear()
        ops._gemm_choice.update(saved)


def test_unet_constructor_and_loading_errors_match_reference_behaviour():
    with pytest.raises(NotImplementedError):
        UNet3DConditionModel(num_attention_heads=8)
    with pytest.raises(ValueError):
        UNet3DConditionModel(down_block_types=("DownBlock3D",) * 3)
    with pytest.raises(ValueError):
        UNet3DConditionModel(block_out_channels=(320, 640))
    m = UNet3DConditionModel(**TINY)
    assert m.config.cross_attention_dim == 64 and m.config.in_channels == 4 and m.dtype == torch.bfloat16
    sd = synthetic_state_dict(UNetConfig(**TINY), seed=0)
    bad = dict(sd)
    bad.pop("conv_in.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    bad = dict(sd)
    bad["conv_in.weight"] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    m.load_state_dict(sd)
    assert list(m.state_dict().keys()) == list(sd.keys())
    with pytest.raises(RuntimeError):  # no CPU fallback
        m.to("cpu")._ensure_engine()
    with pytest.raises(RuntimeError):
        UNet3DConditionModel.from_pretrained("cerspense/zeroscope_v2_576w", subfolder="unet")
    g = UNet3DConditionModel(attention_type="gated", **TINY)
    fusers = [x for x in g.modules() if type(x).__name__ == "GatedSelfAttentionDense"]
    assert len(fusers) == 10 and all(f.enabled for f in fusers)


def test_pipeline_input_checks():
    pipe = TextToVideoSDPipeline(unet=UNet3DConditionModel(sample_size=16, **TINY))
    pe = torch.zeros(1, 77, 64)
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, negative_prompt_embeds=pe, height=100, width=128, num_frames=4)
    with pytest.raises(ValueError):
        pipe(prompt="a", prompt_embeds=pe, negative_prompt_embeds=pe, height=128, width=128)
    with pytest.raises(ValueError):
        pipe(height=128, width=128)
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, negative_prompt_embeds=pe, height=128, width=128, num_frames=4, gligen_boxes=[[[0, 0, 1, 1]]] * 3, gligen_phrases=[["a"]] * 3)
    pipe.enable_fuser(False)


def test_generation_schedule_is_the_ddim_config_one():
    """generation/lvd.py:46 converts the checkpoint's DDIM scheduler: DDIM's registered timestep_spacing="leading" and the SD-family
    steps_offset=1 travel with the config (diffusers 0.27.2 set_timesteps, 'leading' branch): 40 steps -> 961, 937, ..., 25."""
    s = DPMSolverPP2MSchedule.from_ddim_config()
    s.set_timesteps(40)
    assert list(s.timesteps) == [961 - 24 * i for i in range(40)]
    s.set_timesteps(50)
    assert list(s.timesteps) == [951 - 19 * i for i in range(50)]
    from lvd_amd.models.controllable_pipeline_text_to_video_synth import TextToVideoSDPipeline
    sch = TextToVideoSDPipeline(unet=None).scheduler
    assert (sch.timestep_spacing, sch.steps_offset) == ("leading", 1)
    d = DPMSolverPP2MSchedule()  # the bare class keeps the diffusers class defaults
    d.set_timesteps(40)
    assert int(d.timesteps[0]) == 999 and int(d.timesteps[-1]) == 25


@pytest.mark.parametrize("spacing", [dict(), dict(timestep_spacing="leading", steps_offset=1)])
def test_schedule_coefficients_match_oracle_scheduler(spacing):
    a, b = DPMSolverPP2MSchedule(**spacing), scheduler_ref.DPMSolverPP2M(**spacing)
    for spacing_steps in (10, 40):
        a.set_timesteps(spacing_steps)
        b.set_timesteps(spacing_steps)
        assert list(a.timesteps) == list(b.timesteps)
        x = torch.randn(5, generator=torch.Generator().manual_seed(0))
        xa, prev = x.clone(), torch.zeros(5)
        for i in range(spacing_steps):
            eps = torch.randn(5, generator=torch.Generator().manual_seed(i + 1))
            ref = b.step(eps, xa.clone() if i == 0 else xb)
            al, sg, cx, c0, c1 = a.coefficients(i)
            x0 = (xa - sg * eps) / al
            xa = cx * xa + c0 * x0 + c1 * prev
            prev = x0
            a.advance()
            xb = ref
            assert torch.allclose(xa, ref, atol=2e-5, rtol=1e-4), i


def test_guidance_layout_matches_oracle_box_logic():
    boxes = [[[0.1, 0.2, 0.55, 0.8], [0.0, 0.0, 0.0, 0.0], [0.45, 0.05, 1.2, 0.5]]]
    lay = guidance.GuidanceLayout(boxes, [[2, 5]], 3, 20, 36, 0.75, 0.25, "cpu")
    arr = lay.boxes.numpy()[0]
    for f, box in enumerate(boxes[0]):
        x0, y0, x1, y1 = guidance_ref.scale_proportion(box, 20, 36)
        n = max(0, x1 - x0) * max(0, y1 - y0)
        kfg = int((torch.tensor(float(n)) * 0.75).long().clamp_(min=1))
        kbg = int((torch.tensor(float(20 * 36 - n)) * 0.25).long().clamp_(min=1))
        assert list(arr[f]) == [x0, y0, x1, y1, kfg, kbg]
    assert lay.tok_ids.tolist() == [2, 5] and lay.tok_weight.tolist() == [0.5, 0.5]
    with pytest.raises(NotImplementedError):
        guidance.hip_latent_backward_guidance(None, None, None, 0, boxes, [[2]], 1, None, 1.0, upsample_scale=2)  # the reference's own max-based / CE forms raise on it
    assert [guidance._energy_form(*a) for a in ((False, True, False), (True, True, True), (False, False, True))] == [0, 1, 2]  # the chain of utils/guidance.py:312,346,363
    with pytest.raises(ValueError, match="no loss selected"):
        guidance._energy_form(False, False, False)
    with pytest.raises(KeyError):
        guidance._last_key_in_order(type("E", (), {"cfg": UNetConfig(**TINY)})(), [("down", 3, 0, 0)])


def test_from_pretrained_reads_a_local_snapshot(tmp_path):
    """generation/lvd.py:39-44 loads `UNet3DConditionModel.from_pretrained(key, subfolder="unet")`: a local snapshot directory
    (config.json + diffusion_pytorch_model.safetensors) loads by key name; a hub id without files raises loudly."""
    import json
    from safetensors.torch import save_file
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    d = tmp_path / "snap" / "unet"
    d.mkdir(parents=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    conf = dict(_class_name="UNet3DConditionModel", _diffusers_version="0.27.2", sample_size=32, in_channels=4, out_channels=4,
                block_out_channels=list(TINY["block_out_channels"]), layers_per_block=TINY["layers_per_block"],
                cross_attention_dim=TINY["cross_attention_dim"], attention_head_dim=64, norm_num_groups=32, norm_eps=1e-5, act_fn="silu",
                down_block_types=["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"], up_block_types=["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3,
                downsample_padding=1, mid_block_scale_factor=1)
    (d / "config.json").write_text(json.dumps(conf))
    m = UNet3DConditionModel.from_pretrained(str(tmp_path / "snap"), subfolder="unet")
    assert m.config.block_out_channels == tuple(TINY["block_out_channels"]) and m.config.sample_size == 32
    got = m.state_dict()
    assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(RuntimeError):
        UNet3DConditionModel.from_pretrained("cerspense/zeroscope_v2_576w", subfolder="unet")


def test_two_rank_sharding_is_seed_invariant():
    """world_size-2 gloo: ranks take prompt indices i % 2 == rank; seeds are a function of the global index
    (generate.py:325-335), so the union over ranks equals the single-process assignment."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(sum((q.get(timeout=120) for _ in procs), []))
    for p in procs:
        p.join(timeout=60)
    assert got == [(i, rep, i + rep * 6789 + 11) for i in range(5) for rep in range(2)]


def _shard_worker(rank, world, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29581"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lvd_amd.sharding import shard_jobs, gather_frames
    jobs = shard_jobs(num_prompts=5, repeats=2, seed_offset=11, rank=rank, world=world)
    frames = torch.full((2, 3), float(rank))
    allf = gather_frames(frames)
    assert [float(f[0, 0]) for f in allf] == [0.0, 1.0]
    q.put(jobs)
    dist.destroy_process_group()


def test_two_rank_generate_agrees_on_one_run_directory(tmp_path):
    """generate.py under torch.distributed.run (two ranks, gloo, --dry-run: no GPU): without --force_run_ind the run directory is
    chosen by rank 0 and broadcast, so a rank that arrives late (here: rank 1 sleeps before main) does not see the run0/ its peer has
    created and move on to run1/ (reference generate.py:225-234 probes per process; it is single-process)."""
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    driver = tmp_path / "driver.py"
    driver.write_text("import os, sys, time\nsys.path.insert(0, %r)\nimport generate\n"
                      "time.sleep(3.0 if os.environ.get('RANK') == '1' else 0.0)\ngenerate.main(sys.argv[1:])\n" % repo)
    (tmp_path / "prompts.txt").write_text("a cat walking\na dog running\na bird flying\na fish swimming\n")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for attempt in range(2):  # the second launch must take run1/ on both ranks
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), str(driver), "--model", "gpt-4", "--run-model", "zeroscope", "--prompt-type", "plain",
                            "--prompts-file", str(tmp_path / "prompts.txt"), "--template_version", "v0.1", "--dry-run", "--img-root", str(tmp_path / "out")],
                           env=env, capture_output=True, text=True, timeout=300, cwd=repo)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        base = tmp_path / "out" / "imgs_plain_templatev0.1_zeroscope"
        assert sorted(os.listdir(base)) == [f"run{i}" for i in range(attempt + 1)], (os.listdir(base), r.stdout[-1500:])
        assert sorted(os.listdir(base / f"run{attempt}")) == ["0", "1", "2", "3"]  # both ranks wrote their prompt directories into the same run


def test_gemm_workspace_query_needs_no_gpu():
    """Pure host arithmetic of the C ABI: the split-K workspace the caller should offer (lvdhip_gemm_workspace_bytes)."""
    p, n = hip.GemmParams(), ctypes.c_int64(-1)
    for (m, nn, k, act), want in [((1080, 1280, 11520, 0), 16 * 1080 * 1280 * 4), ((138240, 1280, 11520, 0), 64 << 20), ((138240, 1280, 320, 0), 0),
                                  ((1080, 2560, 1280, 1), 0)]:
        p.M, p.N, p.K, p.act = m, nn, k, act
        assert hip.lib().lvdhip_gemm_workspace_bytes(ctypes.byref(p), ctypes.byref(n)) == 0 and n.value == want
    p.M = 0
    assert hip.lib().lvdhip_gemm_workspace_bytes(ctypes.byref(p), ctypes.byref(n)) != 0
    assert b"gemm_workspace_bytes" in hip.lib().lvdhip_last_error()


def test_gemm_autotune_table_round_trip(tmp_path):
    """The per-shape variant choices survive save -> load (tuple keys incl. the conv (stride, upsample) pair): what
    generate.py --gemm_autotune_table relies on to make every rank run the same kernels."""
    from lvd_amd import ops
    saved = dict(ops._gemm_choice)
    try:
        ops._gemm_choice.clear()
        ops._gemm_choice[(0, 138240, 960, 320, 0, 320, 320, None, 0, False, False, 0, 0, False)] = 117
        ops._gemm_choice[(1, 138240, 320, 2880, 0, 320, 320, (1, 0, 40, 72), 0, True, False, 0, 0, True)] = 47
        ops._gemm_choice[(2, 4320, 1280, 3840, 0, 1280, 1280, None, 0, True, True, 24, 180, False)] = 45
        want = dict(ops._gemm_choice)
        path = tmp_path / "table.json"
        ops.save_gemm_autotune_table(str(path))
        ops._gemm_choice.clear()
        ops.load_gemm_autotune_table(str(path))
        assert ops.gemm_autotune_table() == want
    finally:
        ops._gemm_choice.clear()
        ops._gemm_choice.update(saved)


def test_groupnorm_chunking_rules():
    """Statistics chunks: enough workgroups for 256 CUs, never fewer than the minimum rows per chunk, never zero."""
    from lvd_amd import ops
    assert ops._gn_chunks(48, 2880) == 10            # level 0, 2-D norm: 512 // 48
    assert ops._gn_chunks(2, 69120) == 256           # level 0, 5-D norm: capped (the apply workgroups fold the chunk partials themselves)
    assert ops._gn_chunks(2, 1080) == 16             # 5x9 level, forward: 64-row chunks
    assert ops._gn_chunks(2, 1080, 32) == 33         # ... backward statistics: 32-row chunks
    assert ops._gn_chunks(4096, 10) == 1 and ops._gn_chunks(1, 1) == 1
    assert [ops._gn_min_rows(c) for c in (320, 640, 1280, 1920, 2560)] == [64, 64, 64, 48, 24]
    assert ops._gn_chunks(24, 180, ops._gn_min_rows(2560)) == 7   # level 2, norm over [x, skip]: one row lane per workgroup, 24-row chunks (was 2 chunks)


def test_groupnorm_single_launch_plan_for_the_step_shapes():
    """Which GroupNorm of the 576x320x24 step runs as ONE launch with its (sample, group) slab in registers (lvdhip_groupnorm_slab_loads /
    lvdhip_groupnorm_bwd_slab_loads are host-side queries: no GPU needed) — and which do not, and why."""
    from lvd_amd import hip
    fwd, bwd = hip.lib().lvdhip_groupnorm_slab_loads, hip.lib().lvdhip_groupnorm_bwd_slab_loads
    # (c, c1, groups, rows_per_sample) -> loads per thread
    assert fwd(1280, 1280, 32, 1080) == 6            # 5x9 level, 5-D norm of a TemporalConvLayer: 40 channels per group, 16-byte loads
    assert fwd(2560, 1280, 32, 180) == 2             # 10x18 level, norm over [x, skip]
    assert fwd(640, 640, 32, 720) == 4               # 20x36 level, 2-D norm: 20 channels per group, 8-byte loads
    assert fwd(320, 320, 32, 720) == 4               # first resnet of the 20x36 level: 10 channels per group, 4-byte loads
    assert fwd(1280, 1280, 32, 720) == 4             # 20x36 level, norm over [x, skip] (640 + 640)
    assert fwd(1280, 1280, 32, 4320) == 0            # 10x18 level, 5-D norm: a 345 KB slab (22 loads) loses to the two launches
    assert fwd(320, 320, 32, 2880) == 0              # 40x72 level, 2-D norm: 15 narrow loads per thread
    assert fwd(640, 640, 32, 17280) == 0             # 20x36 level, 5-D norm
    assert fwd(1920, 1280, 32, 180) == 0             # 1280 + 640 channels, 60 per group: a group would straddle the two sources
    assert fwd(960, 640, 32, 2880) == 0 and fwd(96, 96, 32, 100) == 0
    assert bwd(640, 640, 32, 720) == 4 and bwd(2560, 1280, 32, 180) == 2 and bwd(320, 320, 32, 720) == 4
    assert bwd(1280, 1280, 32, 1080) == 0            # x and dy: 6 + 6 loads of 16 bytes do not fit the registers of a 1024-thread workgroup


def test_test_tokenizer_ids_do_not_depend_on_call_order():
    """The stand-in tokenizer of the test suite assigns ids from the token text alone, so two processes that see different subsets
    of a prompt list (sharded generate.py) embed the same prompt identically."""
    from oracle.fake_tokenizer import FakeClipTokenizer
    a, b = FakeClipTokenizer(), FakeClipTokenizer()
    a(["a dog runs on the beach"], return_tensors="np")
    ids_a = a(["a red car turns left"], return_tensors="np")["input_ids"]
    ids_b = b(["a red car turns left"], return_tensors="np")["input_ids"]
    assert (ids_a == ids_b).all()
    assert a._convert_id_to_token(ids_a[0][2]) == "red</w>"
