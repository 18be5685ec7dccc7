"""Developer probe: the two-rank determinism test by hand, with the autotune log of every process."""
import json, os, subprocess, sys, tempfile
import numpy as np, joblib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import types
src = open(os.path.join(ROOT, "tests", "test_generation_gpu.py")).read()
drv = src[src.index("_DRIVER = '''") + len("_DRIVER = '''"):]
drv = drv[:drv.index("'''")]
SMALL = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=2, cross_attention_dim=64, attention_head_dim=64)
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "dsl.json")))
tmp = tempfile.mkdtemp()
cases, seen = [], set()
for c in CASES:
    if c["prompt"] not in seen and len(cases) < 4:
        seen.add(c["prompt"]); cases.append(c)
os.makedirs(f"{tmp}/cache")
open(f"{tmp}/cache/cache_shardtest_v0.1_gpt-4-1106-preview.json", "w").write(json.dumps({c["prompt"].strip().rstrip("."): [c["response"]] for c in cases}))
open(f"{tmp}/prompts.txt", "w").write("\n".join(c["prompt"] for c in cases) + "\n")
open(f"{tmp}/driver.py", "w").write(drv.format(repo=ROOT, small=SMALL))
table = f"{tmp}/gemm_table.json"
def argv(out):
    return ["--model", "gpt-4", "--run-model", "lvd_zeroscope", "--prompt-type", "shardtest", "--prompts-file", f"{tmp}/prompts.txt",
            "--template_version", "v0.1", "--num_frames", "24", "--num_inference_steps", "3", "--max_index_step", "1", "--max_iter", "1",
            "--repeats", "1", "--force_run_ind", "0", "--cache-dir", f"{tmp}/cache", "--img-root", f"{tmp}/{out}", "--gemm_autotune_table", table]
env = dict(os.environ, LVD_DIST_BACKEND="gloo", LVD_GEMM_LOG="1", PYTHONPATH=ROOT)
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
one = subprocess.run([sys.executable, f"{tmp}/driver.py"] + argv("one"), env=env, capture_output=True, text=True, cwd=ROOT)
print("one rc", one.returncode, "tuned", one.stderr.count("[lvd gemm autotune]"))
two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29711",
                      f"{tmp}/driver.py"] + argv("two"), env=env, capture_output=True, text=True, cwd=ROOT)
print("two rc", two.returncode, "tuned", two.stderr.count("[lvd gemm autotune]"))
for l in two.stderr.split("\n"):
    if "[lvd gemm autotune]" in l:
        print("  TWO:", l)
three = subprocess.run([sys.executable, f"{tmp}/driver.py"] + argv("three"), env=env, capture_output=True, text=True, cwd=ROOT)
print("three (single process, table loaded) rc", three.returncode, "tuned", three.stderr.count("[lvd gemm autotune]"))
root = "imgs_shardtest_templatev0.1_lvd_zeroscope/run0"
for i in range(4):
    a = joblib.load(f"{tmp}/one/{root}/{i}/video_0.joblib"); b = joblib.load(f"{tmp}/two/{root}/{i}/video_0.joblib"); c = joblib.load(f"{tmp}/three/{root}/{i}/video_0.joblib")
    print(f"prompt {i}: one==two {np.array_equal(a, b)} ({(a != b).sum()} bytes differ), one==three {np.array_equal(a, c)}, two==three {np.array_equal(b, c)}")
