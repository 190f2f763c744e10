# PMC counters of selected kernels of the headline step (separate passes, kernel-trace only): KERNELS="chain wgrad2" bash tools/pmc_step.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcs$i; rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcs$i -- python3 bench.py $BENCH_ARGS --steps 3 --warmup 2 --no-cpu-baseline --no-breakdown > /dev/null 2> gpurun_out/pmcs$i.err
  KERNELS="${KERNELS:-chain wgrad2_kernel}" python3 - <<PY
import csv,glob,collections,os
f=glob.glob("gpurun_out/pmcs$i/**/*counter_collection.csv", recursive=True)
if not f: print("no csv for set $i"); raise SystemExit
keys=os.environ["KERNELS"].split()
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"]
    for k in keys:
        if k in n:
            acc[k+" g"+r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
