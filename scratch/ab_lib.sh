#!/bin/bash
# A/B on ONE box: alternates the in-tree libanyref_hip.so (new) with another build (old) under any lab script.
# usage (on the GPU box): bash scratch/ab_lib.sh scratch/bin/lib_other.so rounds python scratch/some_bench.py
OLD=$1; R=$2; shift; shift
cp anyref_amd/libanyref_hip.so /tmp/lib_new.so
cp "$OLD" /tmp/lib_old.so
for r in $(seq 1 $R); do
  for w in old new; do
    cp /tmp/lib_$w.so anyref_amd/libanyref_hip.so
    echo "== $w"; "$@" 2>/dev/null
  done
done
cp /tmp/lib_new.so anyref_amd/libanyref_hip.so
