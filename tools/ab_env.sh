# A/B of one environment variable on ONE box: VAR=name VALS="a b" [BENCH_ARGS=..] bash tools/ab_env.sh  -> ms per step of the bench line, twice each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in $VALS; do
  env FLID_GEMM_TUNE=1 $VAR=$v python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR', '$v', d['ms_per_step'], d['value'])"
done
done
