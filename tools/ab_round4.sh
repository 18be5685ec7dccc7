#!/bin/bash
# Same-box A/B of the whole step: the round-4 library + its autotune table (build/r4src, made by `git archive 2a030e6 ... | tar -x` + make on the build
# host) against the library and table at HEAD, interleaved, two runs each.  Box-to-box spread on this pool is +-3 %, as large as a round's gain.
cd $GRAFT_REPO_ROOT
O=gpurun_out
for i in 1 2; do
  LVD_LIB=build/r4src/llm-groundedvideodiffusion_amd/liblvdhip.so timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 2 --gemm_autotune_table build/r4src/profiles/gemm_autotune_576x320x24.json > $O/ab_r4_$i.json 2> $O/ab_r4_$i.err
  timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 2 > $O/ab_head_$i.json 2> $O/ab_head_$i.err
done
python - <<'PY'
import json
for f in ("ab_r4_1", "ab_head_1", "ab_r4_2", "ab_head_2"):
    try:
        j = json.load(open(f"gpurun_out/{f}.json"))
        print(f"{f:10s} guided {j['ms_per_step']:7.2f} ms  unguided {j['unguided_ms_per_step']:6.2f} ms  linear {j['roofline']['all_gemm']['linear']['frac']:.4f}")
    except Exception as e:
        print(f, "ERR", e)
PY
