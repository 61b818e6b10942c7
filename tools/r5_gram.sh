#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2 3; do
for z in 128 256 512; do echo -n "C3 gram_wgs=$z: "; MVF_GRAM_STATS_WGS=$z run; done
for z in 128 512; do echo -n "C3 q_wgs=$z: "; MVF_DZFREE_Q_WGS=$z run; done
done
