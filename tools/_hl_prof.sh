cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/hl; mkdir -p $R
rocprofv3 --kernel-trace --stats -d $R/kt -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-kernel-timing --no-other-configs > $R/kt.log 2>&1
cp $(find $R/kt -name "*kernel_stats.csv" | head -1) $R/kernel_stats.csv
tail -1 $R/kt.log | cut -c1-200
rm -rf $R/kt
