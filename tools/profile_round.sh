cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
python3 bench.py > $O/r02_headline_bench.json 2> $O/headline.err
rocprofv3 --kernel-trace -d $O/kt -o headline -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-breakdown > $O/kt_bench.json 2> $O/kt.err
python3 tools/rocpd_stats.py $O/kt/headline_results.db --csv $O/r02_headline_kernel_stats.csv --timeline > $O/r02_headline_timeline.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown > $O/pmc_$c.json 2> $O/pmc_$c.err; done
python3 tools/traffic_from_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic_r02.json "commit 3c4aa30; rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown" > $O/traffic.log 2>&1
python3 bench.py --model tgn > $O/r02_tgn_bench.json 2> $O/tgn.err
rocprofv3 --kernel-trace -d $O/kt_tgn -o tgn -- python3 bench.py --model tgn --steps 40 --warmup 10 --no-cpu-baseline > $O/kt_tgn.json 2> $O/kt_tgn.err
python3 tools/rocpd_stats.py $O/kt_tgn/tgn_results.db --csv $O/r02_tgn_kernel_stats.csv > $O/tgn_stats.txt 2>&1
python3 bench.py --model dygformer > $O/r02_dygformer_bench.json 2> $O/dyg.err
python3 bench.py --model tcl --steps 30 --warmup 5 > $O/r02_tcl_bench.json 2> $O/tcl.err
python3 bench.py --model graphmixer --steps 30 --warmup 5 > $O/r02_graphmixer_bench.json 2> $O/mixer.err
python3 bench.py --mode sweep > $O/r02_sweep_bench.json 2> $O/sweep.err
python3 bench.py --mode fwd > $O/r02_fwd_bench.json 2> $O/fwd.err
python3 bench.py --mode lp > $O/r02_lp_bench.json 2> $O/lp.err
python3 bench.py --roofline-kernel gemm --no-cpu-baseline > $O/r02_headline_bench_mfma.json 2> $O/mfma.err
rm -rf $O/kt/*.db $O/kt_tgn/*.db
tail -c 600 $O/r02_headline_bench.json; cat $O/traffic.log | tail -30
