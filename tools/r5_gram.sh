#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2 3; do
for z in 1 3 2; do echo -n "C3 bn3_apply=$z: "; MVF_FUSE_BN3_APPLY=$z run; done
for z in 1 3 2; do echo -n "C4 bn3_apply=$z: "; MVF_FUSE_BN3_APPLY=$z run --depth 101 --frames 16 --clips 16; done
done
