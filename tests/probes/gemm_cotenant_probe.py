"""Developer probe: bit-reproducibility of single GEMM launches while ANOTHER process runs a mix of small kernels on the same GPU
(co-resident workgroups on the same CUs).    python tests/probes/gemm_cotenant_probe.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import lvd_amd
from lvd_amd import ops

dev = "cuda"
if len(sys.argv) > 1 and sys.argv[1] == "load":
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(138240, 64, device=dev, generator=g).bfloat16()
    x5 = torch.randn(138240, 512, device=dev, generator=g).bfloat16()
    gam, bet = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    g5, b5 = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    w = (torch.randn(512, 64, device=dev, generator=g) * 0.05).bfloat16()
    ops.groupnorm(x, gam, bet, 2880, groups=32, silu=True)
    torch.cuda.synchronize()
    print("READY", flush=True)  # the first launch has run: a parent that waits for this line measures next to a live co-tenant
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(20):
            ops.groupnorm(x, gam, bet, 2880, groups=32, silu=True)
            ops.layernorm(x5, g5, b5)
            ops.gemm(x, w, variant=1)
            ops.silu(x)
        torch.cuda.synchronize()
    sys.exit(0)

load = subprocess.Popen([sys.executable, os.path.abspath(__file__), "load", "150"])
time.sleep(10)
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K, geglu) in [(138240, 1536, 512, 0), (138240, 4096, 512, 1), (69120, 4096, 512, 0), (138240, 512, 512, 0)]:
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    for v in (1, 5, 10, 9, 11, 17, 105, 109, 111, 117, 131, 137, 211, 231):
        run = lambda: ops.gemm(a, w, bias=bias, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE, variant=v)
        ref = run()
        bad = 0
        worst = 0
        for _ in range(25):
            out = run()
            if not torch.equal(out, ref):
                bad += 1
                worst = max(worst, int((out != ref).sum().item()))
        print(f"M={M} N={N} K={K} g{geglu} v{v}: {bad}/25 differ" + (f" (up to {worst} elements)" if bad else ""), flush=True)
load.terminate()
