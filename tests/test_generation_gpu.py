"""GPU tests of the run-model protocol and the batch driver (generate.py) with injected stand-ins for the 'next' rows
(CLIP tokenizer/text encoder, VAE): files, shapes, resume/skip behaviour, seed rule, GLIGEN variants."""
import json
import os

import joblib
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import lvd_amd  # noqa: E402
from lvd_amd import dsl  # noqa: E402
from lvd_amd.generation import _common, lvd, lvd_gligen, lvd_plus, zeroscope_dpm  # noqa: E402
from lvd_amd.weights import UNetConfig, synthetic_state_dict  # noqa: E402
from oracle.fake_tokenizer import FakeClipTokenizer, FakeTextEncoder, fake_vae_decode  # noqa: E402

CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dsl.json")))
SMALL = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=2, cross_attention_dim=64, attention_head_dim=64)


def _configure(tmp_path, gated):
    cfg = UNetConfig(attention_type="gated" if gated else "default", **SMALL)
    _common.configure(state_dict=synthetic_state_dict(cfg, seed=0), unet_config=dict(SMALL), tokenizer=FakeClipTokenizer(),
                      text_encoder=FakeTextEncoder(64), vae=fake_vae_decode, device="cuda", img_dir=str(tmp_path))


def test_lvd_run_writes_reference_file_contract(tmp_path):
    _configure(tmp_path, gated=False)
    assert lvd.version == "lvd" and lvd.init("zeroscope") == (320, 576)
    demo = [c for c in CASES if c["cache"].startswith("cache_demo")][0]
    layout = dsl.parse_layout_response(demo["prompt"], demo["response"])
    frames = lvd.run(layout, seed=7, num_inference_steps=4, num_frames=24, repeat_ind=0, loss_scale=2.5, loss_threshold=0.5, max_iter=1,
                     max_index_step=2, fg_top_p=0.25, bg_top_p=0.25, bg_weight=2.0)
    assert frames.dtype == np.uint8 and frames.shape == (24, 320, 576, 3)
    back = joblib.load(tmp_path / "video_0.joblib")
    assert np.array_equal(back, frames) and (tmp_path / "video_0.gif").exists()
    assert lvd.run(layout, seed=7, num_inference_steps=4, num_frames=24, repeat_ind=0) is None  # existing gif -> skipped
    again = lvd.run(layout, seed=7, num_inference_steps=4, num_frames=24, repeat_ind=1, loss_scale=2.5, loss_threshold=0.5, max_iter=1,
                    max_index_step=2, fg_top_p=0.25, bg_top_p=0.25, bg_weight=2.0)
    other = lvd.run(layout, seed=8, num_inference_steps=4, num_frames=24, repeat_ind=2, max_index_step=0, save_annotated_videos=True)
    assert np.array_equal(again, frames) and not np.array_equal(other, frames)  # output is a function of the seed
    assert (tmp_path / "video_2_with_box.gif").exists() and np.array_equal(joblib.load(tmp_path / "video_2.joblib"), other)  # annotation is a copy


def test_lvd_run_with_hip_vae_decoder(tmp_path):
    """Same run with the frames produced by the HIP VAE decoder (random-init decoder of a small AutoencoderKL topology)
    instead of the injected stand-in: latents -> decode -> tensor2vid -> uint8 file contract, end to end on the GPU."""
    cfg = UNetConfig(**SMALL)
    _common.configure(state_dict=synthetic_state_dict(cfg, seed=0), unet_config=dict(SMALL), tokenizer=FakeClipTokenizer(),
                      text_encoder=FakeTextEncoder(64), vae="synthetic", vae_config=dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1),
                      device="cuda", img_dir=str(tmp_path))
    try:
        assert lvd.init("modelscope256") == (256, 256)
        demo = [c for c in CASES if c["cache"].startswith("cache_demo")][0]
        layout = dsl.parse_layout_response(demo["prompt"], demo["response"])
        frames = lvd.run(layout, seed=11, num_inference_steps=3, num_frames=16, repeat_ind=0, max_index_step=1, max_iter=1)
        assert frames.dtype == np.uint8 and frames.shape == (16, 256, 256, 3)
        assert 5 < frames.mean() < 250 and frames.std() > 1
    finally:
        _common.configure(vae=None, vae_config=None)


def test_decoded_frames_end_to_end_vs_oracle_loop(tmp_path):
    """BASELINE.json north_star: "decoded frames match the reference PyTorch path on the same cached DSLs within a stated fp tolerance"
    (/root/reference/models/controllable_pipeline_text_to_video_synth.py:960-979, generation/lvd.py:161-196).  The whole product path —
    cached demo DSL (the reference's own cache entry) -> layout -> boxes / object token positions -> 6 DPM-Solver++ steps, backward
    guidance on the first 3 -> HIP VAE decoder -> tensor2vid -> uint8 frames — against the all-oracle loop (fp32 UNet with autograd
    guidance, fp32 scheduler, oracle/vae_ref.py) on the same embeddings, latents and layout.
    Stated tolerance, in frame units (8-bit): PSNR >= 33 dB, mean |difference| <= 4 LSB, 99.8 % of the samples within 16 LSB
    (measured on MI355X: 36.8 dB, 2.7 LSB, 99.95 %, worst sample 25 LSB).  (The UNet
    oracle is pinned by the reference's goldens; the VAE restatement is NOT — diffusers is not vendored — so this bounds HIP vs the
    restated path, frame for frame.)"""
    from lvd_amd.weights import VAEConfig, synthetic_vae_state_dict
    from oracle import guidance_ref, scheduler_ref, unet_ref, vae_ref
    cfg = UNetConfig(**SMALL)
    sd = synthetic_state_dict(cfg, seed=0)
    vkw = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1)
    vcfg = VAEConfig(**vkw)
    vsd = synthetic_vae_state_dict(vcfg, seed=0)
    tok = FakeClipTokenizer()
    _common.configure(state_dict=sd, unet_config=dict(SMALL), tokenizer=tok, text_encoder=FakeTextEncoder(64), vae=vsd, vae_config=vkw,
                      device="cuda", img_dir=str(tmp_path))
    try:
        assert lvd.init("modelscope256") == (256, 256)
        demo = [c for c in CASES if c["cache"].startswith("cache_demo")][0]
        layout = dsl.parse_layout_response(demo["prompt"], demo["response"])
        Fr, steps, guided = 8, 6, 3
        gen = torch.Generator().manual_seed(21)
        lat0 = torch.randn(1, 4, Fr, 32, 32, generator=gen)
        pe, ne = torch.randn(1, 77, 64, generator=gen), torch.randn(1, 77, 64, generator=gen)
        hp = dict(loss_scale=5.0, loss_threshold=0.01, max_iter=1, max_index_step=guided, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0,
                  com_loss_scale=0.03)
        frames = lvd.run(layout, seed=0, num_inference_steps=steps, num_frames=Fr, repeat_ind=0, prompt_embeds=pe.cuda(),
                         negative_prompt_embeds=ne.cuda(), latents=lat0.clone(), **hp)
        assert frames.dtype == np.uint8 and frames.shape == (Fr, 256, 256, 3)
        # ---- the all-oracle loop on the same inputs
        cond = dsl.layout_to_condition(layout, height=dsl.LAYOUT_SIZE[0], width=dsl.LAYOUT_SIZE[1], num_condition_frames=Fr, tokenizer=tok)
        keys = _common.GUIDANCE_ATTN_KEYS
        sch = scheduler_ref.DPMSolverPP2M(timestep_spacing="leading", steps_offset=1)
        sch.set_timesteps(steps)
        both = torch.cat([ne, pe])

        def unet_fn(x, t, c, save, save_keys):
            unet_ref.unet_forward(sd, cfg, x, int(t), c, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=("up", 2, 2, 0))

        lat, loss = lat0.clone(), 10000.0
        for i, t in enumerate(sch.timesteps):
            lat, loss = guidance_ref.latent_backward_guidance(unet_fn, sch.alphas_cumprod, pe, i, cond.boxes, cond.object_positions, int(t), lat, loss,
                                                              base_attn_dim=(32, 32), guidance_attn_keys=keys, **hp)
            with torch.no_grad():
                eps = unet_ref.unet_forward(sd, cfg, lat.expand(2, -1, -1, -1, -1), int(t), both)
            lat = sch.step(eps[0:1] + 9.0 * (eps[1:2] - eps[0:1]), lat)
        with torch.no_grad():
            ref = (vae_ref.decode_latents_to_video(vsd, vcfg, lat)[0].numpy() * 255.0).astype(np.uint8)
        d = np.abs(frames.astype(np.int32) - ref.astype(np.int32))
        psnr = 10 * np.log10(255.0 ** 2 / max(float((d.astype(np.float64) ** 2).mean()), 1e-12))
        within16 = float((d <= 16).mean())
        print(f"decoded frames vs oracle loop ({steps} steps, {guided} guided, {Fr}x256x256): PSNR {psnr:.1f} dB, mean |d| {d.mean():.2f} LSB, "
              f"max {d.max()} LSB, within 16 LSB {within16:.5f}; frame mean {ref.mean():.1f} std {ref.std():.1f}")
        assert ref.std() > 5, "degenerate reference frames"
        assert psnr >= 33.0 and d.mean() <= 4.0 and within16 >= 0.998
    finally:
        _common.configure(vae=None, vae_config=None)


@pytest.mark.parametrize("mod,name", [(lvd_gligen, "lvd-gligen"), (lvd_plus, "lvd-plus"), (zeroscope_dpm, "zeroscope")])
def test_other_run_models(tmp_path, mod, name):
    _configure(tmp_path, gated=name != "zeroscope")
    assert mod.version == name
    hw = mod.init("modelscope256") if name != "zeroscope" else mod.init("")
    case = [c for c in CASES if any(all(v == 0 for v in fb) for b in c["boxes"] for fb in b)][0]  # an object disappears
    layout = dsl.parse_layout_response(case["prompt"], case["response"])
    kw = dict(gligen_scheduled_sampling_beta=0.5) if name != "zeroscope" else {}
    frames = mod.run(layout, seed=3, num_inference_steps=4, num_frames=16, repeat_ind=0, max_index_step=1, max_iter=1, **kw)
    assert frames.shape == (16, hw[0], hw[1], 3) and frames.dtype == np.uint8


def test_generate_cli_resume_and_seed_rule(tmp_path, monkeypatch):
    import generate
    demo = [c for c in CASES if c["cache"].startswith("cache_demo")][0]
    cache_dir = tmp_path / "cache"
    cache_dir.mkdir()
    (cache_dir / "cache_demo_v0.1_gpt-4-1106-preview.json").write_text(json.dumps({demo["prompt"]: [demo["response"]]}))
    _configure(tmp_path, gated=False)
    seeds = []
    orig = lvd._m.run
    monkeypatch.setattr(lvd._m, "run", lambda layout, seed, **kw: (seeds.append(seed), orig(layout, seed, **kw))[1])
    argv = ["--model", "gpt-4", "--run-model", "lvd_zeroscope", "--prompt-type", "demo", "--template_version", "v0.1", "--num_frames", "24",
            "--num_inference_steps", "3", "--max_index_step", "1", "--max_iter", "1", "--repeats", "2", "--seed_offset", "5", "--force_run_ind", "0",
            "--cache-dir", str(cache_dir), "--img-root", str(tmp_path / "out")]
    assert generate.main(argv) == 2
    assert seeds == [0 + 5, 6789 + 5]
    run_dir = tmp_path / "out" / "imgs_demo_templatev0.1_lvd_zeroscope" / "run0" / "0"
    assert sorted(f for f in os.listdir(run_dir) if f.endswith(".joblib")) == ["video_0.joblib", "video_1.joblib"]
    assert generate.main(argv) == 0  # resumed: both repeats exist
    with pytest.raises(ValueError):
        generate.main([a if a != "24" else "16" for a in argv])  # zeroscope with < 24 frames is refused like in the reference


def test_generate_videos_per_gpu_batches_jobs_and_keeps_seeds_and_files(tmp_path):
    """generate.py --videos-per-gpu 2 (throughput mode, reference loop generate.py:325-338 batched): the two repeats of a prompt go through
    ONE denoising loop (their CFG forwards as one batch-4 pass, guidance per sample).  Same seed rule and file names as the one-video-at-a-
    time run.  The videos are NOT bit-equal to it: a batch-4 forward picks other tile geometries / K-split plans / GroupNorm chunkings than
    a batch-2 forward (M is part of every choice), i.e. other fp32 summation orders and other bf16 roundings — the same ~1e-2 distance on
    the noise prediction that test_gated_full_size_properties allows between a batch-2 and a batch-1 run (4e-2) — and classifier-free
    guidance at scale 9 on RANDOM weights amplifies it (measured: 6-8 % on the latents after one step, 27 dB on these frames).  Stated
    bound: PSNR >= 24 dB on the 8-bit frames of the stand-in decoder; seeds, file names and the resume rule are exact."""
    import generate
    demo = [c for c in CASES if c["cache"].startswith("cache_demo")][0]
    cache_dir = tmp_path / "cache"
    cache_dir.mkdir()
    (cache_dir / "cache_demo_v0.1_gpt-4-1106-preview.json").write_text(json.dumps({demo["prompt"]: [demo["response"]]}))
    _configure(tmp_path, gated=False)

    def argv(out, v):
        return ["--model", "gpt-4", "--run-model", "lvd_zeroscope", "--prompt-type", "demo", "--template_version", "v0.1", "--num_frames", "24",
                "--num_inference_steps", "4", "--max_index_step", "2", "--max_iter", "1", "--repeats", "3", "--seed_offset", "7", "--force_run_ind", "0",
                "--cache-dir", str(cache_dir), "--img-root", str(tmp_path / out), "--videos-per-gpu", str(v)]
    assert generate.main(argv("one", 1)) == 3
    assert generate.main(argv("two", 2)) == 3  # a batch of two and a last batch of one
    sub = os.path.join("imgs_demo_templatev0.1_lvd_zeroscope", "run0", "0")
    worst = 1e9
    for r in range(3):
        a = joblib.load(tmp_path / "one" / sub / f"video_{r}.joblib").astype(np.float64)
        b = joblib.load(tmp_path / "two" / sub / f"video_{r}.joblib").astype(np.float64)
        assert a.shape == b.shape == (24, 320, 576, 3)
        mse = float(((a - b) ** 2).mean())
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        worst = min(worst, psnr)
    print("videos-per-gpu 2 vs 1: worst PSNR over 3 seeds", worst)
    assert worst >= 24.0, worst
    assert generate.main(argv("two", 2)) == 0  # resumed: everything exists


_DRIVER = '''
import sys
sys.path.insert(0, {repo!r})
import lvd_amd  # noqa: F401
import generate
from lvd_amd.generation import _common
from lvd_amd.weights import UNetConfig, synthetic_state_dict
from oracle.fake_tokenizer import FakeClipTokenizer, FakeTextEncoder, fake_vae_decode
SMALL = {small!r}
_common.configure(state_dict=synthetic_state_dict(UNetConfig(**SMALL), seed=0), unet_config=dict(SMALL), tokenizer=FakeClipTokenizer(),
                  text_encoder=FakeTextEncoder(64), vae=fake_vae_decode)
n = generate.main(sys.argv[1:])
# one file per rank: ranks of one torch.distributed.run share a stdout pipe and their writes may interleave mid-line
import json, os
rank = int(os.environ.get("RANK", "0"))
with open(os.path.join(os.environ["LVD_TEST_TALLY_DIR"], f"generated_rank{{rank}}.json"), "w") as fh:
    json.dump({{"rank": rank, "generated": n}}, fh)
'''


def _tally(d):
    """{rank: videos generated} from the per-rank files the driver script leaves in `d` (then removed, so that the next run starts clean)."""
    out = {}
    for name in sorted(os.listdir(d)):
        if name.startswith("generated_rank"):
            with open(os.path.join(d, name)) as fh:
                rec = json.load(fh)
            out[rec["rank"]] = rec["generated"]
            os.remove(os.path.join(d, name))
    return out


def test_generate_two_ranks_reproduce_the_single_process_run(tmp_path):
    """generate.py under torch.distributed.run, two ranks (both on this one GPU, gloo for the end-of-run tally): every rank binds a
    device, takes the prompts `sharding.owns` gives it, and — with the GEMM autotune table of the single-process run loaded — writes
    bit-identical videos (global prompt index -> seed, same tile geometry per shape in every process).  Three runs: the single-process run
    that tunes and writes the table, a single-process run that loads it ("pinned"), and the two-rank run that loads it."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases, seen = [], set()
    for c in CASES:  # four distinct prompts (a repeated prompt would consume a second cache entry)
        if c["prompt"] not in seen and len(cases) < 4:
            seen.add(c["prompt"])
            cases.append(c)
    cache_dir = tmp_path / "cache"
    cache_dir.mkdir()
    (cache_dir / "cache_shardtest_v0.1_gpt-4-1106-preview.json").write_text(json.dumps({c["prompt"].strip().rstrip("."): [c["response"]] for c in cases}))
    (tmp_path / "prompts.txt").write_text("\n".join(c["prompt"] for c in cases) + "\n")
    driver = tmp_path / "driver.py"
    driver.write_text(_DRIVER.format(repo=repo, small=SMALL))
    table = tmp_path / "gemm_table.json"

    def argv(out, pin_run=True):
        return ["--model", "gpt-4", "--run-model", "lvd_zeroscope", "--prompt-type", "shardtest", "--prompts-file", str(tmp_path / "prompts.txt"),
                "--template_version", "v0.1", "--num_frames", "24", "--num_inference_steps", "3", "--max_index_step", "1", "--max_iter", "1",
                "--repeats", "1", "--cache-dir", str(cache_dir), "--img-root", str(tmp_path / out),
                "--gemm_autotune_table", str(table)] + (["--force_run_ind", "0"] if pin_run else [])

    tally_dir = tmp_path / "tally"
    tally_dir.mkdir()
    env = dict(os.environ, LVD_DIST_BACKEND="gloo", PYTHONPATH=repo + os.pathsep + os.environ.get("PYTHONPATH", ""), LVD_TEST_TALLY_DIR=str(tally_dir))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, str(driver)] + argv("one"), env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert one.returncode == 0 and _tally(tally_dir) == {0: 4}, one.stdout[-2000:] + one.stderr[-2000:]
    assert table.exists()
    pinned = subprocess.run([sys.executable, str(driver)] + argv("pinned"), env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert pinned.returncode == 0 and _tally(tally_dir) == {0: 4}, pinned.stdout[-2000:] + pinned.stderr[-2000:]
    import socket
    with socket.socket() as sk:  # a port nothing else on this box holds
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(driver)] + argv("two", pin_run=False), env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    assert _tally(tally_dir) == {0: 2, 1: 2}, two.stdout[-2000:] + two.stderr[-2000:]  # two prompts per rank
    root = "imgs_shardtest_templatev0.1_lvd_zeroscope/run0"
    # no --force_run_ind in the sharded run: rank 0 picks the run directory and broadcasts it, so there is exactly one
    assert sorted(os.listdir(tmp_path / "two" / "imgs_shardtest_templatev0.1_lvd_zeroscope")) == ["run0"]
    vids = {k: [joblib.load(tmp_path / k / root / str(i) / "video_0.joblib") for i in range(4)] for k in ("one", "pinned", "two")}
    report = [(i, np.array_equal(vids["one"][i], vids["pinned"][i]), np.array_equal(vids["two"][i], vids["pinned"][i])) for i in range(4)]
    print("prompt: (tuning run == pinned run, two ranks == pinned run):", report)
    for i in range(4):
        assert vids["pinned"][i].shape == (24, 320, 576, 3)
        assert report[i][2], f"prompt {i} differs between the sharded run and the single-process run on the same table: {report}"
        assert report[i][1], f"prompt {i} differs between the run that tuned the table and a run that loaded it: {report}"
