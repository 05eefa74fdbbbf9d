#!/bin/bash
for s in "10 2 3 4" "30 7 9 4" "50 25 50 4" "60 0 20 4" "60 20 0 4" "200 30 56 2" "15 3 4 4"; do
  timeout 40 python scripts/gpu_dbg2.py $s 2>&1 | grep -v amdgpu | tail -2 | cut -c1-400
  echo "  rc=$? for $s"
done
