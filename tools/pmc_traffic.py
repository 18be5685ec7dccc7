"""Roll the per-shape rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gemm_pmc.py into profiles/r02_gemm_traffic.json
(read by bench.py for roofline.traffic) and a text table.
    python tools/pmc_traffic.py <dir> <out.json> [reps]
<dir> holds fetch_<i>/r_counter_collection.csv and write_<i>/r_counter_collection.csv for shape index i.  Only the dispatches
after the LAST silu marker kernel are counted (the autotuner's trial launches come before it) and divided by `reps`.
FETCH_SIZE is in KB and tallies 128-B requests at 64 B for wide streaming reads on gfx950: doubled (MI355X_MICROARCH.md, HBM)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_pmc import SHAPES, algorithmic_bytes

root, out = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3


def total(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    last = max((i for i, r in enumerate(rows) if "silu_kernel" in r["Kernel_Name"]), default=-1)
    rows = rows[last + 1:]
    kernels = sorted({r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] for r in rows})
    t = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows) / 1e3 / reps
    return sum(float(r["Counter_Value"]) for r in rows) * 1024.0 / reps, kernels, t


res, lines = {}, []
for i, (M, N, K, kind) in enumerate(SHAPES):
    fp, wp = (os.path.join(root, f"{c}_{i}", "r_counter_collection.csv") for c in ("fetch", "write"))
    if not (os.path.exists(fp) and os.path.exists(wp)):
        continue
    fetch, kern, t = total(fp, "FETCH_SIZE")
    write, _, _ = total(wp, "WRITE_SIZE")
    fetch *= 2.0
    alg = algorithmic_bytes(M, N, K, kind)
    lines.append(f"{kind:8s} M={M:6d} N={N:4d} K={K:5d}  {t:7.1f} us/product (profiled)  FETCH x2 {fetch / 1e6:7.1f} MB  WRITE {write / 1e6:6.1f} MB  "
                 f"algorithmic {alg / 1e6:6.1f} MB  ratio {(fetch + write) / alg:4.2f}   kernels: {', '.join(k[:60] for k in kern)}")
    if kind not in res:  # first (dominant) shape of each class
        res[kind] = {"shape": f"M={M} N={N} K={K}", "fetch_bytes": round(fetch), "write_bytes": round(write), "algorithmic_bytes": alg,
                     "note": "rocprofv3 --pmc FETCH_SIZE (x2: gfx950 tallies 128-B requests at 64 B) and WRITE_SIZE, separate passes, per product "
                             "(all kernels of the product: head, split-K tail, slab reduce); Infinity-Cache hits are counted"}
json.dump(res, open(out, "w"), indent=1)
print("\n".join(lines))
