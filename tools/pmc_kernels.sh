#!/bin/bash
# usage (GPU box, via gpurun): bash tools/pmc_kernels.sh <outdir under gpurun_out> [env assignments...] -- one --pmc pass of SQ wait / busy counters
# over a short run of the 8-image step; prints per-kernel per-launch averages (tools/pmc_sum.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift
mkdir -p $out
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES"; do
  set -- $pass; name=$1; shift
  env "${EXTRA_ENV[@]}" rocprofv3 --pmc "$@" -d $out/pmc_$name -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-kernel-timing --no-other-configs > $out/pmc_$name.log 2>&1
  cp $(find $out/pmc_$name -name "*counter_collection.csv" | head -1) $out/pmc_$name.csv
  python tools/pmc_sum.py $out/pmc_$name.csv > $out/pmc_$name.txt 2>&1
  rm -rf $out/pmc_$name $out/pmc_$name.csv
  head -8 $out/pmc_$name.txt
done
