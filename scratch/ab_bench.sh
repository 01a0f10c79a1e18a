#!/bin/bash
# A/B on ONE box: alternates the in-tree libanyref_hip.so (new) with another build (old) under bench.py.
# usage (on the GPU box): bash scratch/ab_bench.sh scratch/bin/lib_prev.so [rounds] [bench args...]
set -e
OLD=$1; R=${2:-3}; shift; shift || true
cp anyref_amd/libanyref_hip.so /tmp/lib_new.so
cp "$OLD" /tmp/lib_old.so
for r in $(seq 1 $R); do
  for w in old new; do
    cp /tmp/lib_$w.so anyref_amd/libanyref_hip.so
    python bench.py --no-cpu-baseline --no-parity --steps 10 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('isolated', {}).get('frac'))"
  done
done
cp /tmp/lib_new.so anyref_amd/libanyref_hip.so
