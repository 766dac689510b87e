#!/bin/bash
# roi_pool_fwd_rows under measurement builds (tools/ab/roi_x<mask>.so: ROI_X in csrc/roi_pool.hip), durations only.  GPU box.
cd "$(dirname "$0")/.."
for n in main "$@"; do
  if [ $n = main ]; then L=""; else L="GNET_LIB_AB=$PWD/tools/ab/$n.so"; fi
  echo "$n: $(env $L python tools/roi_bench.py 2>/dev/null | grep -A3 '"roi_pool_fwd"' | grep '"us"')"
done
