import os, sys, subprocess, threading, time
sys.path.insert(0, "/root/repo")
import torch
import lvd_amd
from lvd_amd import ops
dev = "cuda"
log = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            log.append((time.time(), out.strip()[:600]))
        except Exception as e:
            log.append((time.time(), "err " + str(e)))
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
def bench(a, w, v, n=200):
    for _ in range(5): ops.gemm(a, w, variant=v)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): ops.gemm(a, w, variant=v)
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192), (34560, 640, 2560)]:
    for fill in ("randn", "zeros", "signconst"):
        if fill == "randn":
            a, w = torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) * 0.03).bfloat16()
        elif fill == "zeros":
            a, w = torch.zeros(M, K, device=dev).bfloat16(), torch.zeros(N, K, device=dev).bfloat16()
        else:
            a, w = torch.rand(M, K, device=dev).bfloat16(), (torch.rand(N, K, device=dev) * 0.03).bfloat16()
        for v in (111, 211):
            t0 = time.time()
            us = bench(a, w, v, 300 if M < 8000 else 60)
            print(f"M={M} N={N} K={K} {fill:9s} v{v}: {us:8.1f} us {2.0*M*N*K/us/1e6:6.0f} TF  (t={t0 - log[0][0] if log else 0:.2f}..{time.time() - (log[0][0] if log else 0):.2f})", flush=True)
stop = True; th.join()
t00 = log[0][0]
for t, o in log[::4]:
    print(f"{t - t00:6.2f} {o}")
