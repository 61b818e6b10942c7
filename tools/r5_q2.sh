#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2 3; do
for q in 0 1 2; do echo -n "C3 q=$q: "; MVF_DZFREE_Q=$q run; done
for q in 0 1 2; do echo -n "C4 q=$q: "; MVF_DZFREE_Q=$q run --depth 101 --frames 16 --clips 16; done
for q in 0 1; do echo -n "12 clips q=$q: "; MVF_DZFREE_Q=$q run --clips 12; done
done
