#!/bin/bash
# A/B of two builds of liblsrecon.so on ONE box (box-to-box spread exceeds most kernel changes):
#   tools/probes/lib_ab.sh <other.so> <rounds> -- <command printing a result line>
# The library file is swapped in place between runs; the in-tree build is restored at the end.
set -e
other=$1; rounds=$2; shift 3
cur=shrimpy_amd/csrc/liblsrecon.so
cp $cur /tmp/lsr_ab_new.so
for round in $(seq 1 $rounds); do
  for which in new other; do
    if [ $which = new ]; then cp /tmp/lsr_ab_new.so $cur; else cp "$other" $cur; fi
    echo "== round $round: $which"
    "$@"
  done
done
cp /tmp/lsr_ab_new.so $cur
