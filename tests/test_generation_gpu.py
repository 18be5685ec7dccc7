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
print("GENERATED", n, flush=True)
'''


def test_generate_two_ranks_reproduce_the_single_process_run(tmp_path):
    """generate.py under torch.distributed.run, two ranks (both on this one GPU, gloo for the end-of-run tally): every rank binds a
    device, takes the prompts `sharding.owns` gives it, and — with the GEMM autotune table of the single-process run loaded — writes
    bit-identical videos (global prompt index -> seed, same tile geometry per shape in every process)."""
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

    def argv(out):
        return ["--model", "gpt-4", "--run-model", "lvd_zeroscope", "--prompt-type", "shardtest", "--prompts-file", str(tmp_path / "prompts.txt"),
                "--template_version", "v0.1", "--num_frames", "24", "--num_inference_steps", "3", "--max_index_step", "1", "--max_iter", "1",
                "--repeats", "1", "--force_run_ind", "0", "--cache-dir", str(cache_dir), "--img-root", str(tmp_path / out),
                "--gemm_autotune_table", str(table)]

    env = dict(os.environ, LVD_DIST_BACKEND="gloo", PYTHONPATH=repo + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, str(driver)] + argv("one"), env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert one.returncode == 0 and "GENERATED 4" in one.stdout, one.stdout[-2000:] + one.stderr[-2000:]
    assert table.exists()
    port = 29600 + os.getpid() % 300
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(driver)] + argv("two"), env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    assert two.stdout.count("GENERATED 2") == 2, two.stdout[-2000:]  # two prompts per rank
    root = "imgs_shardtest_templatev0.1_lvd_zeroscope/run0"
    for i in range(4):
        a = joblib.load(tmp_path / "one" / root / str(i) / "video_0.joblib")
        b = joblib.load(tmp_path / "two" / root / str(i) / "video_0.joblib")
        assert a.shape == (24, 320, 576, 3) and np.array_equal(a, b), f"prompt {i} differs between the sharded and the single-process run"
