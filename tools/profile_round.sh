# Everything under profiles/${R}_*: run on the GPU box from the repo root (bash tools/profile_round.sh), then copy gpurun_out/${R}p/* into profiles/.
R=${ROUND:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${R}p; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o headline -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-breakdown > $O/kt_bench.json 2> $O/kt.err
python3 tools/rocpd_stats.py $O/kt/headline_results.db --csv $O/${R}_headline_kernel_stats.csv --timeline > $O/${R}_headline_timeline.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown > $O/pmc_$c.json 2> $O/pmc_$c.err; done
python3 tools/traffic_from_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic_${R}.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown" > $O/traffic.log 2>&1
# (the headline line quotes the PMC traffic only from a file taken on the same kernel source: put it in place first)
cp $O/traffic_${R}.json profiles/traffic_${R}.json
python3 bench.py --host-profile > $O/${R}_headline_bench.json 2> $O/headline.err
KERNELS="chain_fwd chain_bwd wgrad2_kernel attn_bwd_fast attn_fwd_fast" bash tools/pmc_step.sh > $O/${R}_pmc_sq.txt 2>&1
python3 bench.py --gemm-mode 0 --no-cpu-baseline > $O/${R}_headline_bench_exact_f32.json 2> $O/exact.err
python3 bench.py --python-step --no-cpu-baseline --no-breakdown --host-profile > $O/${R}_headline_bench_python_step.json 2> $O/pystep.err
python3 bench.py --model tgn > $O/${R}_tgn_bench.json 2> $O/tgn.err
python3 bench.py --model tgn --mode lp > $O/${R}_tgn_lp_bench.json 2> $O/tgn_lp.err
python3 bench.py --model tgn --gemm-mode 0 --no-cpu-baseline > $O/${R}_tgn_bench_exact_f32.json 2> $O/tgn_exact.err
python3 bench.py --model tgn --simulate-world 8 --no-cpu-baseline > $O/${R}_tgn_simulate_world8_bench.json 2> $O/tgn_sim.err
rocprofv3 --kernel-trace -d $O/kt_tgn -o tgn -- python3 bench.py --model tgn --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/kt_tgn.err
python3 tools/rocpd_stats.py $O/kt_tgn/tgn_results.db --csv $O/${R}_tgn_kernel_stats.csv --timeline > $O/${R}_tgn_timeline.txt 2>&1
python3 bench.py --model dygformer > $O/${R}_dygformer_bench.json 2> $O/dyg.err
python3 bench.py --model dygformer --python-step --no-cpu-baseline > $O/${R}_dygformer_bench_autograd.json 2> $O/dyg_py.err
python3 bench.py --model dygformer --gemm-mode 0 --no-cpu-baseline > $O/${R}_dygformer_bench_exact_f32.json 2> $O/dyg_exact.err
rocprofv3 --kernel-trace -d $O/kt_dyg -o dyg -- python3 bench.py --model dygformer --steps 20 --warmup 6 --no-cpu-baseline --no-breakdown > /dev/null 2> $O/kt_dyg.err
python3 tools/rocpd_stats.py $O/kt_dyg/dyg_results.db --csv $O/${R}_dygformer_kernel_stats.csv --timeline > $O/${R}_dygformer_timeline.txt 2>&1
python3 tools/dyg_host_prof.py > $O/${R}_dygformer_host_issue.txt 2> $O/dyg_host.err
python3 tools/pk_bench.py 2> /dev/null | grep "^R=" > $O/${R}_pk_bench.txt
python3 bench.py --mode sweep > $O/${R}_sweep_bench.json 2> $O/sweep.err
python3 bench.py --mode fwd > $O/${R}_fwd_bench.json 2> $O/fwd.err
python3 bench.py --mode lp > $O/${R}_lp_bench.json 2> $O/lp.err
python3 bench.py --workload scale --no-cpu-baseline > $O/${R}_scale_config5_bench.json 2> $O/scale.err
rocprofv3 --kernel-trace -d $O/kt_scale -o scale -- python3 bench.py --workload scale --steps 20 --warmup 6 --no-cpu-baseline --no-breakdown > /dev/null 2> $O/kt_scale.err
python3 tools/rocpd_stats.py $O/kt_scale/scale_results.db --csv $O/${R}_scale_config5_kernel_stats.csv --timeline > $O/${R}_scale_config5_timeline.txt 2>&1
BENCH_ARGS="--workload scale" KERNELS="chain_fwd chain_bwd wgrad2_kernel attn_bwd_fast attn_fwd_fast" bash tools/pmc_step.sh > $O/${R}_scale_config5_pmc_sq.txt 2>&1
BENCH_ARGS="--model dygformer" KERNELS="seq_attn_bwd seq_attn_fwd gemm_pk_s gemm_pk_l wgrad2_kernel" bash tools/pmc_step.sh > $O/${R}_dygformer_pmc_sq.txt 2>&1
rm -rf $O/kt $O/kt_tgn $O/kt_dyg $O/kt_scale gpurun_out/pmcs* $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -c 300 $O/${R}_headline_bench.json; tail -3 $O/traffic.log; grep -h "host issue" $O/*.err
