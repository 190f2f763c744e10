# A/B of the native stepper's scheduling options on ONE box (FLID_STEP_OPT bits, csrc/tg_step.hip): ms per step of the headline line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for o in ${OPTS:-0 1 3 7 15 31}; do
  FLID_STEP_OPT=$o python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('opt', $o, d['ms_per_step'], d['value'])"
done
done
