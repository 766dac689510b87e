#!/bin/bash
# Cost map of pw_bwd_bf's pieces: measurement builds with one piece compiled out each (results are WRONG in those builds; only the
# kernel's duration is read), timed with tools/kprobe.py through GNET_LIB_AB.  Build here (no GPU needed):  tools/pw_ablate.sh build
# Run on the GPU box: tools/pw_ablate.sh run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/ab
  for x in 1 2 4 8 16 32 64 127; do
    GNET_EXTRA_FLAGS="-DPBB_X=$x" python -m gossipnet_amd.build > /dev/null 2>&1 && cp gossipnet_amd/libgossipnet_hip_probe.so tools/ab/pbb_x$x.so && echo built $x
  done
else
  for x in 0 1 2 4 8 16 32 64 127; do
    if [ $x = 0 ]; then L=""; else L="GNET_LIB_AB=$PWD/tools/ab/pbb_x$x.so"; fi
    echo "PBB_X=$x $(env $L python tools/kprobe.py 8 2>/dev/null | grep ^train | tr ' ' '\n' | grep -E 'pw_bwd_main|sum')"
  done
fi
