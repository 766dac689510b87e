cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r5c; mkdir -p $R
rocprofv3 --kernel-trace --stats -d $R/kt -o kt --output-format csv -- python tools/quick_bench.py 10 > $R/kt.log 2>&1
cp $(find $R/kt -name "*kernel_stats.csv" | head -1) $R/kernel_stats.csv
rm -rf $R/kt
head -25 $R/kernel_stats.csv | cut -c1-200
