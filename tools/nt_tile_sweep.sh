# tile-width sweep of the split-bf16 NT product (FLID_NT_TNW: 32-column tiles per wave) on the DyGFormer / TGAT shapes; rocprof gives the
# kernel time (the Python loop around it is host-bound)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 2 3 4; do
  rm -rf gpurun_out/nts
  FLID_GEMM_TUNE=1 FLID_NT_TNW=$t rocprofv3 --kernel-trace -d gpurun_out/nts -o x -- python3 tools/gemm_bench.py --modes 1 --shape 0,1,38400,600,200 --shape 0,1,38400,800,200 --shape 0,1,38400,200,800 --shape 0,1,38400,200,200 --shape 0,1,38400,200,600 --shape 0,1,13000,888,172 --shape 0,1,38400,200,496 > /dev/null 2>&1
  echo "TNW=$t"; python3 tools/rocpd_by_grid.py gpurun_out/nts/x_results.db 2>/dev/null | grep gemm_bf16x3_nt | cut -c1-120
done
rm -rf gpurun_out/nts
