"""bench.py as the driver runs it: a plain `python bench.py --gpus N` (no torch.distributed.run around it) must launch its own ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_bench_command_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run (one rank per GPU;
    here both ranks share the one GPU of the test box and LVD_BENCH_BACKEND=gloo carries the barriers, the MAX-reduce and the frame
    gather).  One JSON line, from rank 0, with both ranks seen by the collective backend and the whole-job value = 2 x 24 frames / step."""
    env = dict(os.environ, LVD_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
                        "--unguided-steps", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]  # stdout IS the one JSON line (library banners of the ranks go to stderr)
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"]["world_size"] == 2 and j["rccl_ranks_seen"]["ranks"] == [0, 1]
    assert j["scaling"] == "weak" and abs(j["value"] - 2 * 24 / (j["ms_per_step"] * 1e-3)) < 0.02 * j["value"]
    assert j["frame_gather_ms_untimed"] is not None and j["loss_finite"]
