# Everything under profiles/r03_*: run on the GPU box from the repo root (bash tools/profile_round.sh), then copy gpurun_out/r03/* into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
python3 bench.py > $O/r03_headline_bench.json 2> $O/headline.err
rocprofv3 --kernel-trace -d $O/kt -o headline -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-breakdown > $O/kt_bench.json 2> $O/kt.err
python3 tools/rocpd_stats.py $O/kt/headline_results.db --csv $O/r03_headline_kernel_stats.csv --timeline > $O/r03_headline_timeline.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown > $O/pmc_$c.json 2> $O/pmc_$c.err; done
python3 tools/traffic_from_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic_r03.json "commit $(cat .git_head 2>/dev/null); rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown" > $O/traffic.log 2>&1
KERNELS="chain_fwd chain_bwd wgrad2_kernel attn_bwd_fast attn_fwd_fast" bash tools/pmc_step.sh > $O/r03_pmc_sq.txt 2>&1
python3 bench.py --gemm-mode 0 --no-cpu-baseline > $O/r03_headline_bench_exact_f32.json 2> $O/exact.err
python3 bench.py --roofline-kernel gemm --no-cpu-baseline > $O/r03_headline_bench_mfma.json 2> $O/mfma.err
python3 bench.py --model tgn > $O/r03_tgn_bench.json 2> $O/tgn.err
rocprofv3 --kernel-trace -d $O/kt_tgn -o tgn -- python3 bench.py --model tgn --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/kt_tgn.err
python3 tools/rocpd_stats.py $O/kt_tgn/tgn_results.db --csv $O/r03_tgn_kernel_stats.csv --timeline > $O/r03_tgn_timeline.txt 2>&1
python3 bench.py --model dygformer > $O/r03_dygformer_bench.json 2> $O/dyg.err
python3 bench.py --mode sweep > $O/r03_sweep_bench.json 2> $O/sweep.err
python3 bench.py --mode fwd > $O/r03_fwd_bench.json 2> $O/fwd.err
python3 bench.py --mode lp > $O/r03_lp_bench.json 2> $O/lp.err
python3 bench.py --workload scale --no-cpu-baseline > $O/r03_scale_config5_bench.json 2> $O/scale.err
rm -rf $O/kt/*.db $O/kt_tgn/*.db gpurun_out/pmcs*
tail -c 400 $O/r03_headline_bench.json; cat $O/traffic.log | tail -5
