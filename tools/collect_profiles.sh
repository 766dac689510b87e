#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: refreshes the round's measurement artefacts under
# gpurun_out/prof/ (copy the summaries into profiles/ afterwards):
#   pmc_fetch.txt / pmc_write.txt / pmc_sq.txt   separate --pmc passes, per-kernel per-launch averages
#   traffic.json              HBM bytes + MFMA-busy fraction per launch and kernel (bench.py reads profiles/r06_traffic.json:
#                             copied there ON THE BOX before the bench line is taken, so that the line carries the counters of
#                             the same kernel sources)
#   bench.json                default bench.py line (with other_configs and cpu_baseline)
#   kernel_stats.csv          rocprofv3 --kernel-trace --stats of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/prof; mkdir -p $R
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --pmc "$@" -d $R/pmc_$name -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-kernel-timing --no-other-configs > $R/pmc_$name.log 2>&1
  cp $(find $R/pmc_$name -name "*counter_collection.csv" | head -1) $R/pmc_$name.csv
  python tools/pmc_sum.py $R/pmc_$name.csv > $R/pmc_$name.txt 2>&1
done
python tools/make_traffic.py $R/pmc_fetch.csv $R/pmc_write.csv $R/pmc_sq.csv $R/traffic.json > $R/traffic.log 2>&1
cp $R/traffic.json profiles/r06_traffic.json
python bench.py > $R/bench.log 2>&1; tail -1 $R/bench.log > $R/bench.json
rocprofv3 --kernel-trace --stats -d $R/kt -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-kernel-timing --no-other-configs > $R/kt.log 2>&1
cp $(find $R/kt -name "*kernel_stats.csv" | head -1) $R/kernel_stats.csv
rm -rf $R/kt $R/pmc_fetch $R/pmc_write $R/pmc_sq $R/pmc_fetch.csv $R/pmc_write.csv $R/pmc_sq.csv
ls -la $R; cat $R/bench.json | cut -c1-300; cat $R/pmc_fetch.txt | head -8; cat $R/pmc_write.txt | head -8; cat $R/pmc_sq.txt | head -12
