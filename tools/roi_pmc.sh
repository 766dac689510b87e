#!/bin/bash
# Run on the GPU box from the repo root: PMC passes over tools/roi_bench.py (the RoI-pool kernels only); summaries in gpurun_out/roi/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/roi; mkdir -p $R
for pass in "fetch FETCH_SIZE" "sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; name=$1; shift
  rocprofv3 --pmc "$@" -d $R/pmc_$name -o pmc --output-format csv -- python tools/roi_bench.py > $R/pmc_$name.log 2>&1
  cp $(find $R/pmc_$name -name "*counter_collection.csv" | head -1) $R/pmc_$name.csv
  python tools/pmc_sum.py $R/pmc_$name.csv > $R/pmc_$name.txt 2>&1
  rm -rf $R/pmc_$name
  grep -i "roi_pool" $R/pmc_$name.txt | head -20
done
