# per-kernel durations of tools/rows_bench.py --quick under rocprofv3, for the production library and the ablation builds
# (flid_amd/csrc/variants/libflid_tg_exp*.so, built with -DFLID_ROWS_EXP=n).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rowsprof; mkdir -p $O
for v in prod exp1 exp2 exp3 exp4; do
  if [ $v = prod ]; then unset FLID_TG_LIB; else export FLID_TG_LIB=$GRAFT_REPO_ROOT/flid_amd/csrc/variants/libflid_tg_$v.so; [ -f $FLID_TG_LIB ] || continue; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o t -- python3 tools/rows_bench.py --quick --rows ${ROWS:-13622} > $O/$v.log 2>&1
  echo "== $v"; python3 - "$O/$v" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows = [r for r in rows if "gemm_rows" in r["Kernel_Name"]]
# group consecutive launches of 30
out = []
for i in range(0, len(rows), 30):
    grp = rows[i:i + 30]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp)
    out.append(f"{d[len(d)//2]:.1f}")
print(" ".join(out))
PY
done
