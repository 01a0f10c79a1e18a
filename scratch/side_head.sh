#!/bin/bash
# A/B of ANYREF_SIDE_HEAD (encoder blocks beside CLIP) on one box: separate processes, alternated
for r in 1 2; do
for h in 0 2 3 5; do
  ANYREF_SIDE_HEAD=$h timeout -k 10 200 python scratch/side_share.py 128:6 2>&1 | grep "round 1" | sed "s/^/head $h: /" | cut -c1-120
done; done
