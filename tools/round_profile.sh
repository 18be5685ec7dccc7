# Round-end profile set (run on the GPU box: gpurun -- 'bash tools/round_profile.sh'); everything lands in gpurun_out/, the summaries that
# are judged are copied to profiles/ by hand afterwards.  ROUND tag: r06.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=r06
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; tail -1 $O/${T}_smoke.log
timeout 600 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
rm -rf $O/prof_r6; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_r6 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${T}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof_r6.err)
python tools/rocprof_rollup.py $O/prof_r6 70 0.5 > $O/${T}_bench_summary.txt 2>&1
cp $(find $O/prof_r6 -name "*kernel_stats.csv" | head -1) $O/${T}_bench_kernel_stats.csv 2>/dev/null
find $O/prof_r6 -name "*kernel_trace.csv" -delete; find $O/prof_r6 -name "*.db" -delete
# MFMA utilisation of the step's own GEMM variants (the shipped autotune table is loaded by tools/gemm_pmc.py)
rm -rf $O/pmc_r6; (cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_r6 -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $GRAFT_REPO_ROOT/$O/pmc_r6.log 2>&1)
python tools/pmc_rollup.py $(find $O/pmc_r6 -name "*counter_collection.csv" | head -1) > $O/${T}_pmc_mfma_util.txt 2>&1
find $O/pmc_r6 -name "*.csv" -size +20M -delete
# HBM traffic of the GEMM classes (separate FETCH_SIZE / WRITE_SIZE passes)
ROUND=r06 bash tools/traffic_passes.sh > $O/${T}_traffic_passes.log 2>&1
# the memory-bound kernels: microseconds and TB/s per level, the guidance loss, the short-K quantisation probe
timeout 300 python tools/small_ops_bench.py > $O/${T}_hbm_kernels.txt 2>&1
timeout 200 python tools/loss_bench.py > $O/${T}_guidance_loss.txt 2>&1
timeout 300 python tools/quant_probe.py > $O/${T}_short_k_quantisation.txt 2>&1
timeout 300 python tools/host_profile.py > $O/${T}_host_profile.txt 2>&1
# GroupNorm per shape by kernel duration: the single-launch slab-in-registers kernel against statistics + apply
bash tools/gn_probe.sh > /dev/null 2>&1; cp $O/gn_probe.txt $O/${T}_groupnorm_kernels.txt
# round 5: the per-shape roofline model of the plain GEMMs, the persistent walker against the one-shot kernels, its phase trace (needs
# tools/build_ablations.sh on the build host), spatial attention v2 / v3
timeout 300 python tools/gemm_roofline_model.py > $O/${T}_gemm_roofline_model.txt 2> $O/${T}_gemm_roofline_model.err
timeout 300 python tools/stream_bench.py --no-check > $O/${T}_stream_walker.txt 2>&1
[ -f build/abl/liblvdhip_strace.so ] && LVD_LIB=build/abl/liblvdhip_strace.so timeout 200 python tools/stream_trace.py > $O/${T}_stream_trace.txt 2>&1
for v in 2 3; do echo "== LVD_ATTN_VARIANT=$v"; LVD_ATTN_VARIANT=$v timeout 200 python tools/attn_bench.py 2>&1 | grep -E "spatial" | cut -c1-52; done > $O/${T}_attention_fwd.txt
timeout 400 python bench.py --videos-per-gpu 2 --no-cpu-baseline --steps 10 --warmup 1 > $O/${T}_bench_2videos.json 2> $O/${T}_bench_2videos.err
timeout 400 python bench.py --videos-per-gpu 2 --guidance-one-by-one --no-cpu-baseline --steps 10 --warmup 1 > $O/${T}_bench_2videos_one_by_one.json 2> $O/${T}_bench_2videos_one_by_one.err
timeout 400 python bench.py --gligen --no-cpu-baseline --steps 10 --warmup 1 > $O/${T}_bench_gligen.json 2> $O/${T}_bench_gligen.err
LVD_CFG_SHARED_PREFIX=0 timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 1 > $O/${T}_bench_no_shared_prefix.json 2> $O/${T}_bench_no_shared_prefix.err
python - <<'PY'
import json
for f in ("r06_bench", "r06_bench_under_rocprof", "r06_bench_2videos", "r06_bench_2videos_one_by_one", "r06_bench_gligen", "r06_bench_no_shared_prefix"):
    try:
        j = json.load(open(f"gpurun_out/{f}.json"))
        print(f, j["value"], j["ms_per_step"], j.get("unguided_ms_per_step"), j.get("step_mfma_frac"), {k: (v["ms_per_step"], v["frac"]) for k, v in j["roofline"]["all_gemm"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
head -12 $O/${T}_bench_summary.txt; cat $O/${T}_pmc_mfma_util.txt | head -12
