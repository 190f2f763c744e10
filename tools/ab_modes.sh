# A/B of two library builds on the secondary bench lines (forward only, TGAT / TGN link-prediction step), two rounds each, on one box:
#   OLD=flid_amd/csrc/variants/libflid_tg_old.so bash tools/ab_modes.sh      (run on the GPU box from the repo root)
cd $GRAFT_REPO_ROOT
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for lib in new old; do
  if [ $lib = old ]; then export FLID_TG_LIB=$GRAFT_REPO_ROOT/${OLD:-flid_amd/csrc/variants/libflid_tg_old.so}; else unset FLID_TG_LIB; fi
  python bench.py --mode fwd --no-cpu-baseline 2>/dev/null | p "$lib fwd"
  python bench.py --model tgn --mode lp --no-cpu-baseline 2>/dev/null | p "$lib tgn_lp"
  python bench.py --mode lp --no-cpu-baseline 2>/dev/null | p "$lib lp"
done; done
