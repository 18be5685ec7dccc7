# Developer probe: same-box A/B of the engine's knobs on the bench step (gpurun -- 'bash tools/knob_sweep.sh'); prints guided / unguided ms per setting.
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', j['ms_per_step'], j['unguided_ms_per_step'], j['loss_finite'])"; }
for k in ${KNOBS:-A=default}; do run $k; done
