#!/bin/bash
# One kernel class under several library builds on one box (GPU box): tools/ab_class.sh <class> <images> <name> [<name> ...]; "main" = the
# shipped library, another name = tools/ab/<name>.so (a measurement build made with GNET_EXTRA_FLAGS).  Prints the class's milliseconds per
# training step and the sum over all classes, two rounds.
cd "$(dirname "$0")/.."
cls=$1; images=$2; shift 2
for round in 1 2; do
  for n in "$@"; do
    if [ $n = main ]; then L=""; else L="GNET_LIB_AB=$PWD/tools/ab/$n.so"; fi
    echo "$n: $(env $L python tools/kprobe.py $images 2>/dev/null | grep ^train | tr ' ' '\n' | grep -E "^($cls)=|^sum" | tr '\n' ' ')"
  done
done
