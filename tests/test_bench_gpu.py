"""bench.py as the driver runs it: a plain `python bench.py --gpus N` (no torch.distributed.run around it) must launch its own ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_plain_bench_command_self_launches_its_ranks(world):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run (one rank per GPU;
    here all ranks share the one GPU of the test box and LVD_BENCH_BACKEND=gloo carries the barriers, the MAX-reduce and the frame
    gather).  One JSON line, from rank 0 and nothing else on stdout, with every rank seen by the collective backend, each rank on its own
    (prompt, seed) sample, the frame gather exercised, and the whole-job value = N x 24 frames / step.  N = 4 is the rehearsal of the
    driver's 1/2/4/8-GPU sweep (the manual sharding of /root/reference/README.md:140-148 it replaces): plumbing must not be what a first
    real --gpus 8 run dies on."""
    env = dict(os.environ, LVD_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
                        "--unguided-steps", "1"], env=env, capture_output=True, text=True, timeout=1200, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]  # stdout IS the one JSON line (library banners of the ranks go to stderr)
    j = json.loads(lines[0])
    seen = j["rccl_ranks_seen"]
    assert j["n_gpus"] == world and seen["world_size"] == world and seen["ranks"] == list(range(world))
    assert seen["sample_seeds"] == [1234 + r for r in range(world)] and seen["distinct_samples"] == world
    assert j["scaling"] == "weak" and abs(j["value"] - world * 24 / (j["ms_per_step"] * 1e-3)) < 0.02 * j["value"]
    assert j["frame_gather_ms_untimed"] is not None and j["loss_finite"]
    assert j["roofline"]["hbm_kernels"]["guidance_loss"]["GB_per_s"] > 0 and j["c_abi_calls_per_guided_step"] > 500
