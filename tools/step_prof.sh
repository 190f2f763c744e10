# kernel trace of the headline step: per-kernel totals and the timeline of one step (tools/rocpd_stats.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-step}; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o headline -- python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-breakdown $BENCH_ARGS > $O/kt_bench.json 2> $O/kt.err
python3 tools/rocpd_stats.py $O/kt/headline_results.db --csv $O/kernel_stats.csv --timeline > $O/timeline.txt 2>&1
rm -rf $O/kt
head -${LINES_OUT:-45} $O/timeline.txt | cut -c1-150
