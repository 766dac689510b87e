#!/bin/bash
# roi_pool_bwd_block under measurement builds (tools/ab/roi_bx<mask>.so: ROI_BX in csrc/roi_pool.hip), durations only.  GPU box.
cd "$(dirname "$0")/.."
for n in main "$@"; do
  if [ $n = main ]; then L=""; else L="GNET_LIB_AB=$PWD/tools/ab/$n.so"; fi
  echo "$n: $(env $L python tools/roi_bench.py 2>/dev/null | grep -A3 'reference order' | grep '"us"')"
done
