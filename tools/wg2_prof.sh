# grouped weight-gradient launch of a 13.6 k-row layer: form 2 (tg_wgrad.hip) and its ablation builds under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/wg2prof; mkdir -p $O
for v in prod wg1 wg2 wg3; do
  if [ $v = prod ]; then unset FLID_TG_LIB; else export FLID_TG_LIB=$GRAFT_REPO_ROOT/flid_amd/csrc/variants/libflid_tg_$v.so; [ -f $FLID_TG_LIB ] || continue; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o t -- python3 tools/gemm_bench.py --wgrad ${ROWS:-13622} > $O/$v.log 2>&1
  echo "== $v"; python3 - "$O/$v" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
by = collections.defaultdict(list)
for r in rows:
    if "wgrad" in r["Kernel_Name"]:
        by[(r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in by.items():
    d.sort()
    print(k, len(d), f"median {d[len(d)//2]:.1f} us  min {d[0]:.1f}")
PY
done
