#!/bin/bash
# usage: bash scratch/ab_stage.sh scratch/bin/lib_prev.so   -- per-stage times, old vs new .so on one box
cp anyref_amd/libanyref_hip.so /tmp/lib_new.so; cp "$1" /tmp/lib_old.so
for w in old new; do cp /tmp/lib_$w.so anyref_amd/libanyref_hip.so; echo -n "$w "; python scratch/stage_times.py 2>/dev/null | tail -1; done
cp /tmp/lib_new.so anyref_amd/libanyref_hip.so
