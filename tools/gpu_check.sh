# The round's standing GPU check: the -m gpu suite (or TESTS=...) and the three models' bench lines; everything also lands in
# gpurun_out/check_summary.txt so that a cut-off tool output loses nothing.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/check_summary.txt; mkdir -p gpurun_out; : > $O
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], {k: v.get('value') for k, v in d.items() if isinstance(v, dict) and 'ms_per_step' in v})"; }
(timeout 1500 python -m pytest ${TESTS:-tests} -m gpu -q 2>&1 | tail -${TAILN:-4}) | tee -a $O
for m in ${MODELS:-tgat tgn dygformer}; do
  extra=""; [ "$m" = "tgat" ] || extra="--model $m"
  python bench.py $extra --no-cpu-baseline 2>/dev/null | p $m | tee -a $O
done
