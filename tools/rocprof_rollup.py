"""Developer probe: per-kernel roll-up of a rocprofv3 --kernel-trace run (rocpd sqlite output).
usage: python tools/rocprof_rollup.py <dir with *_results.db> [top_n] [skip_first_fraction]"""
import glob, sqlite3, sys, collections

path = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[0]
top_n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
db = sqlite3.connect(path)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # drop the leading part (weight init, autotuning, warm-up)
rows = rows[int(len(rows) * skip):]
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    a = agg[n]; a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values())
print(f"# {path}: {len(rows)} dispatches, {tot / 1e6:.1f} ms of kernel time")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top_n]:
    print(f"{n[:120]:120s} calls={c:6d} total_ms={t / 1e6:8.2f} avg_us={t / c / 1e3:8.1f} {100 * t / tot:5.1f}%")
