#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2 3; do
for w in 192 256 384; do echo -n "C3 wgs=$w: "; MVF_DZFREE_Q_WGS=$w run; done
echo -n "C3 q=2: "; MVF_DZFREE_Q=2 run
for w in 192 256 384; do echo -n "C4 wgs=$w: "; MVF_DZFREE_Q_WGS=$w run --depth 101 --frames 16 --clips 16; done
echo -n "C4 q=2: "; MVF_DZFREE_Q=2 run --depth 101 --frames 16 --clips 16
echo -n "12 clips q=1: "; run --clips 12
echo -n "12 clips q=2: "; MVF_DZFREE_Q=2 run --clips 12
echo -n "12 clips q=2 maxk: "; MVF_DZFREE_Q=2 MVF_DZFREE_Q_WGS=128 run --clips 12
done
