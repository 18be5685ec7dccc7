"""Per-kernel / per-class roll-up of a rocprofv3 --kernel-trace run (rocpd sqlite output).
usage: python tools/rocprof_rollup.py <dir with *_results.db> [top_n] [skip_first_fraction]
The leading fraction of dispatches (weight init, GEMM autotuning, warm-up) can be skipped."""
import collections
import glob
import sqlite3
import sys

CLASSES = [
    ("tap GEMM, LDS-resident im2col (conv_halo.hip: 3x3 conv, temporal conv, their split-K reduce)", ("tap_gemm_kernel", "tap_reduce_kernel")),
    ("GEMM family (gemm.hip, gemm_ring.hip, gemm_stream.hip, split-K reduce)", ("gemm_kernel", "gemm_ring_kernel", "gemm_stream_kernel", "splitk_reduce")),
    ("GroupNorm", ("gn_",)),
    ("attention fwd/bwd", ("attn_",)),
    ("LayerNorm", ("ln_",)),
    ("guidance loss (3 launches for all keys of an iteration)", ("ca_",)),
    ("elementwise / layout (incl. the temporal-conv combine pass)", ("geglu_", "add_kernel", "silu_kernel", "tokens", "upsample2x", "timestep_embedding", "cfg_dpm", "axpy", "reduce_sum", "tconv_combine")),
]


def main():
    top_n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    dbs = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))
    if dbs:
        rows = sqlite3.connect(dbs[0]).execute("select name, start, end from kernels order by start").fetchall()
    else:  # --output-format csv: *_kernel_trace.csv
        import csv
        path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
        rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path))), key=lambda r: r[1])
    # dispatches per denoising step: a step ends with its fused CFG / DPM-Solver++ update (one cfg_dpm_step launch); it is a guided step when
    # it holds the guidance-loss launches (ca_probs*).  Counted over the whole trace, before the leading fraction is dropped.
    guided, unguided, cur, has_loss = [], [], 0, False
    for n, s, e in rows:
        cur += 1
        has_loss = has_loss or "ca_probs" in n
        if "cfg_dpm_step" in n:
            (guided if has_loss else unguided).append(cur)
            cur, has_loss = 0, False
    med = lambda v: sorted(v)[len(v) // 2] if v else 0
    per_step = f"# dispatches per step (median over {len(guided)} guided / {len(unguided)} unguided steps of the trace): guided {med(guided)}, unguided {med(unguided)}"
    rows = rows[int(len(rows) * skip):]
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s, e in rows:
        a = agg[n]
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    wall = rows[-1][2] - rows[0][1]
    print(f"# {len(rows)} dispatches, {tot / 1e6:.1f} ms of kernel time in {wall / 1e6:.1f} ms of wall time")
    print(per_step)
    cls = collections.OrderedDict((c[0], [0, 0]) for c in CLASSES)
    cls["torch housekeeping (fills, copies)"] = [0, 0]
    for n, (c, t) in agg.items():
        for title, keys in CLASSES:
            if any(k in n for k in keys):
                cls[title][0] += c
                cls[title][1] += t
                break
        else:
            cls["torch housekeeping (fills, copies)"][0] += c
            cls["torch housekeeping (fills, copies)"][1] += t
    for title, (c, t) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
        print(f"{title:58s} calls={c:7d} total_ms={t / 1e6:9.1f} share={100 * t / tot:5.1f}%")
    print("\ntop kernels:")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top_n]:
        print(f"{n[:118]:118s} calls={c:6d} total_ms={t / 1e6:8.2f} avg_us={t / c / 1e3:8.1f} {100 * t / tot:5.1f}%")


if __name__ == "__main__":
    main()
