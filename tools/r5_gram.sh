#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2 3; do
for z in 0 128; do echo -n "C3 q_mink=$z: "; MVF_DZFREE_Q_MINK=$z run; done
for z in 0 128; do echo -n "C4 q_mink=$z: "; MVF_DZFREE_Q_MINK=$z run --depth 101 --frames 16 --clips 16; done
done
timeout 300 python bench.py --steps 10 --warmup 3 --per-layer --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep "M802816" | cut -c1-150
