"""Summarise a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run: one line per (kernel, grid) with the derived
ratios DESIGN.md quotes.    python tools/pmc_rollup.py <dir>/r_counter_collection.csv [name-substring]
SQ set:  SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
         SQ_BUSY_CYCLES (one pass);  TCC set: FETCH_SIZE or WRITE_SIZE (one pass each).
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); FETCH_SIZE is KB and counts 128-B requests
as 64 B on gfx950 for wide streaming reads (x2 column = corrected, MI355X_MICROARCH.md HBM section)."""
import collections
import csv
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.OrderedDict()
allrows = list(csv.DictReader(open(path)))
# tools/gemm_pmc.py launches a silu marker kernel between a shape's first (autotuning) call and its measured calls: dispatches between a
# shape's first launch and its marker are tuning trials and are dropped
marks = sorted({int(r["Dispatch_Id"]) for r in allrows if "silu_kernel" in r["Kernel_Name"]})
rnd_ids = sorted({int(r["Dispatch_Id"]) for r in allrows if "distribution_" in r["Kernel_Name"] or "elementwise" in r["Kernel_Name"]})


def measured(d):
    if not marks:
        return True
    prev_marks = [m for m in marks if m < d]
    if not prev_marks:
        return False
    later_inputs = [x for x in rnd_ids if prev_marks[-1] < x < d]  # a new shape's input generation after the marker: tuning again
    return not later_inputs


for r in allrows:
    name = r["Kernel_Name"]
    if want and want not in name:
        continue
    if not measured(int(r["Dispatch_Id"])):
        continue
    key = r["Dispatch_Id"]
    d = rows.setdefault(key, {"name": name, "grid": int(r["Grid_Size"]), "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:70]


agg = collections.OrderedDict()
for d in rows.values():
    k = (short(d["name"]), d["grid"])
    a = agg.setdefault(k, collections.defaultdict(float))
    a["n"] += 1
    for c, v in d.items():
        if c not in ("name", "grid"):
            a[c] += v
for (name, grid), a in agg.items():
    n = a["n"]
    line = f"{name:70s} grid={grid:8d} x{int(n):3d} {a['t'] / n:9.1f} us"
    if a.get("GRBM_GUI_ACTIVE"):
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        line += f"  MFMA busy {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 256 * 4):5.1f}%"
        wc = a.get("SQ_WAVE_CYCLES", 0) or 1
        line += f"  wait_any {100 * a['SQ_WAIT_ANY'] / wc:5.1f}%  wait_inst {100 * a['SQ_WAIT_INST_ANY'] / wc:5.1f}%  active {100 * a['SQ_ACTIVE_INST_ANY'] / wc:5.1f}%"
        line += f"  LDS-conflict cyc/CU {a.get('SQ_LDS_BANK_CONFLICT', 0) / n / 256:9.0f}"
        # GRBM_GUI_ACTIVE / duration as a clock estimate is only meaningful for launches long against the counter's start / stop skew
        # (it read 4.6 - 5.4 "GHz" on 5 us kernels): printed from 20 us up
        line += f"  clock {cyc / n / (a['t'] / n) / 1e3:4.2f} GHz" if a["t"] / n >= 20.0 else "  clock    -    "
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if c in a:
            mb = a[c] / n / 1024.0
            line += f"  {c} {mb:8.1f} MB" + (f" (x2 = {2 * mb:8.1f} MB)" if c == "FETCH_SIZE" else "")
    print(line)
