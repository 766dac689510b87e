#!/bin/bash
# usage (GPU box, via gpurun): tools/trace_step.sh <images per step> <out name>
# rocprofv3 kernel trace of a few steps -> gpurun_out/<out>/kernel_trace.csv (+ stats), for tools/trace_gaps.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IM=$1; OUT=gpurun_out/$2; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --images $IM --steps 6 --warmup 3 --cpu-seconds 0 --no-kernel-timing --no-other-configs > $OUT/log.txt 2>&1
cp $(find $OUT/kt -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/kt
tail -1 $OUT/log.txt | cut -c1-200
python tools/trace_gaps.py $OUT/kernel_trace.csv | tee $OUT/gaps.txt
