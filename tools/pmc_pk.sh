# PMC counters of the pre-split-weight products on one deep shape (separate passes, kernel-trace only): bash tools/pmc_pk.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCP_PENDING_STALL_CYCLES_sum SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  PK_SHAPES=${PK_SHAPES:-200x800} rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcp$i -- python3 tools/pk_bench.py > /dev/null 2> gpurun_out/pmcp$i.err
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmcp$i/**/*counter_collection.csv", recursive=True)
if not f: print("no csv for set $i"); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"]
    for k in ("gemm_pk_l", "gemm_pk_s", "gemm_bf16x3_nt"):
        if k in n:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
rm -rf gpurun_out/pmcp*
