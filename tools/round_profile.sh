export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.log 2>&1; tail -1 $O/r04_smoke.log
timeout 600 python bench.py > $O/r04_bench.json 2> $O/r04_bench.err
rm -rf $O/prof_r4; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_r4 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r04_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof_r4.err)
python tools/rocprof_rollup.py $O/prof_r4 50 0.5 > $O/r04_bench_summary.txt 2>&1
cp $(find $O/prof_r4 -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv 2>/dev/null
find $O/prof_r4 -name "*kernel_trace.csv" -delete; find $O/prof_r4 -name "*.db" -delete
rm -rf $O/pmc_r4; (cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_r4 -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $GRAFT_REPO_ROOT/$O/pmc_r4.log 2>&1)
python tools/pmc_rollup.py $(find $O/pmc_r4 -name "*counter_collection.csv" | head -1) > $O/r04_pmc_mfma_util.txt 2>&1
find $O/pmc_r4 -name "*.csv" -size +20M -delete
timeout 400 python bench.py --videos-per-gpu 2 --no-cpu-baseline --steps 10 --warmup 1 > $O/r04_bench_2videos.json 2> $O/r04_bench_2videos.err
timeout 400 python bench.py --gligen --no-cpu-baseline --steps 10 --warmup 1 > $O/r04_bench_gligen.json 2> $O/r04_bench_gligen.err
python - <<'PY'
import json
for f in ("r04_bench", "r04_bench_under_rocprof", "r04_bench_2videos", "r04_bench_gligen"):
    try:
        j = json.load(open(f"gpurun_out/{f}.json"))
        print(f, j["value"], j["ms_per_step"], j.get("unguided_ms_per_step"), j.get("step_mfma_frac"), {k: (v["ms_per_step"], v["frac"]) for k, v in j["roofline"]["all_gemm"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
head -12 $O/r04_bench_summary.txt; cat $O/r04_pmc_mfma_util.txt | head -12
