export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gn; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gn -- python $GRAFT_REPO_ROOT/tools/gn_probe.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/gn_probe.err)
python tools/gn_probe.py parse $(find gpurun_out/prof_gn -name "*kernel_trace.csv" | head -1) | tee gpurun_out/gn_probe.txt
rm -rf gpurun_out/prof_gn
