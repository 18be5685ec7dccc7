"""Developer probe: GroupNorm per shape of the step by KERNEL duration (the timing loop of tools/small_ops_bench.py is host-bound below
~20 us per launch): the slab-in-registers single launch where the shape qualifies, and statistics + apply over the statistics chunk
floor (rows per chunk; 0 = the shipped rule ops._gn_min_rows).
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/gn_probe.py      launches
  python tools/gn_probe.py parse <kernel_trace.csv>                                    table
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (label, rows, C, rows_per_sample)
SHAPES = [("L3 b1 5d", 1080, 1280, 1080), ("L3 b2 5d", 2160, 1280, 1080), ("L2 b1 5d", 4320, 1280, 4320), ("L2 b2 5d", 8640, 1280, 4320),
          ("L1 b1 5d", 17280, 640, 17280), ("L1 b1 2d", 17280, 640, 720), ("L1 b2 5d", 34560, 640, 17280), ("L1 b2 2d", 34560, 640, 720),
          ("L0 b1 5d", 69120, 320, 69120), ("L0 b1 2d", 69120, 320, 2880), ("L0 b2 5d", 138240, 320, 69120), ("L0 b2 2d", 138240, 320, 2880),
          ("L2 b1 2d cat", 4320, 2560, 180), ("L2 b2 2d cat", 8640, 2560, 180), ("L1 b2 2d cat", 34560, 1280, 720), ("L0 b2 2d cat", 138240, 640, 2880),
          ("L0 b1 2d cat", 69120, 640, 2880), ("L1 b1 2d 320", 17280, 320, 720), ("L2 b2 2d 640", 8640, 640, 180)]
FLOORS = [0] if os.environ.get("GN_PROBE_FLOORS") is None else [int(v) for v in os.environ["GN_PROBE_FLOORS"].split(",")]
N = 12

if len(sys.argv) > 2 and sys.argv[1] == "parse":
    import csv
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
          if "gn_partial_kernel" in r["Kernel_Name"] or "gn_apply_kernel" in r["Kernel_Name"] or "gn_slab_kernel" in r["Kernel_Name"]]
    i = 0
    print("# us per launch: slab-in-registers single launch | statistics + apply = total, per statistics chunk floor (rows per chunk; 0 = shipped rule)")
    for label, rws, c, rps in SHAPES:
        line = f"{label:13s} rows={rws:6d} C={c:5d} rps={rps:6d} {rws * c * 2 / 1e6:6.1f} MB |"
        if i < len(ks) and "slab" in ks[i][0]:
            seg = [d for n, d in ks[i:i + N]][2:]
            i += N
            line += f" slab {sum(seg) / len(seg) / 1e3:5.1f} |"
        else:
            line += "  slab   -   |"
        for fl in FLOORS:
            seg = ks[i:i + 2 * N]
            i += 2 * N
            part = [d for n, d in seg if "partial" in n][2:]
            appl = [d for n, d in seg if "apply" in n][2:]
            a, b = sum(part) / len(part) / 1e3, sum(appl) / len(appl) / 1e3
            line += f" {fl:2d}: {a:5.1f}+{b:5.1f}={a + b:5.1f} |"
        print(line)
    sys.exit(0)

import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

dev = "cuda"
ops._gn_fused.update(max_bytes=0, max_bytes_slab=1 << 40)  # never the 256-thread kernel; the slab kernel wherever the shape qualifies
shipped_rule = ops._gn_min_rows
for label, rows, c, rps in SHAPES:
    x = torch.randn(rows, c, device=dev).bfloat16()
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    if ops.groupnorm_slab_ok(rows, c, rps, 32):
        for _ in range(N):
            ops.groupnorm_fused(x, g, b, rps, silu=True, slab=True)
        torch.cuda.synchronize()
    for fl in FLOORS:
        ops._gn_min_rows = (lambda c, fl=fl: fl) if fl else shipped_rule
        for _ in range(N):
            st = ops.groupnorm_stats(x, g, b, rps)
            ops.groupnorm_apply(x, st, rps, silu=True)
        torch.cuda.synchronize()
