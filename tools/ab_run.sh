#!/bin/bash
# A/B timing of library builds on one box (GPU box): tools/ab_run.sh <name> [<name> ...]; "main" = the shipped library, another name =
# tools/ab/<name>.so (a measurement build made here with GNET_EXTRA_FLAGS).  Two rounds, alternating; prints the per-class milliseconds.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for n in "$@"; do
    if [ $n = main ]; then L=""; else L="GNET_LIB_AB=$PWD/tools/ab/$n.so"; fi
    echo "$n: $(env $L python tools/kprobe.py 8 2>/dev/null | grep -E '^train|^infer' | sed -E 's/(gather_winners|node_bwd|pw_w1_nodesums|node_fwd|winner_lists|graph|reduce_partials|loss|head_bwd|edge_geometry|pw_w1_classrows|pack)=[0-9.]+ //g' | tr '\n' ' ')"
  done
done
