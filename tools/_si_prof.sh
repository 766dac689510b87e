cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/si; mkdir -p $R
rocprofv3 --kernel-trace --stats -d $R/kt -o kt --output-format csv -- python bench.py --images 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-kernel-timing --no-other-configs > $R/kt.log 2>&1
cp $(find $R/kt -name "*kernel_stats.csv" | head -1) $R/kernel_stats.csv
cp $(find $R/kt -name "*kernel_trace.csv" | head -1) $R/kernel_trace.csv
tail -1 $R/kt.log | cut -c1-200
head -30 $R/kernel_stats.csv
rm -rf $R/kt
