#!/bin/bash
# FETCH_SIZE and WRITE_SIZE of the GEMM classes' largest products, one rocprofv3 --pmc pass per counter and shape (TCC slots: the two do not
# fit one pass; counters only next to --kernel-trace, as the pool requires) -> gpurun_out/traffic_${ROUND:-r05}/{fetch,write}_<i>/, rolled up by
# tools/pmc_traffic.py into profiles/${ROUND:-r05}_gemm_traffic.json (read by bench.py for roofline.traffic).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traffic_${ROUND:-r05}
rm -rf $O; mkdir -p $O
for i in 0 1 2 3 4 5; do
  for c in fetch write; do
    ctr=FETCH_SIZE; [ $c = write ] && ctr=WRITE_SIZE
    (cd /tmp && timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${c}_$i -o r -- python $R/tools/gemm_pmc.py --shape $i > $O/${c}_$i.log 2>&1)
    f=$(find $O/${c}_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mv $f $O/${c}_$i/r_counter_collection.csv
    find $O/${c}_$i -name "*kernel_trace.csv" -delete
  done
done
cd $R && python tools/pmc_traffic.py $O gpurun_out/${ROUND:-r05}_gemm_traffic.json > gpurun_out/${ROUND:-r05}_gemm_pmc_traffic.txt 2>&1
cat gpurun_out/${ROUND:-r05}_gemm_pmc_traffic.txt
