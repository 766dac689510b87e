#!/bin/bash
# usage: pmc_pass.sh <outdir> <counters...>   (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
mkdir -p gpurun_out/$out
rocprofv3 --pmc "$@" -d gpurun_out/$out -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-kernel-timing > gpurun_out/$out/log.txt 2>&1
tail -2 gpurun_out/$out/log.txt
