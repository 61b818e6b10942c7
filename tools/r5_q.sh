#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 900 python -m pytest tests/test_dzfree_gpu.py -q -m gpu -p no:cacheprovider -x -s 2>&1 | grep -v "^$" | tail -25)
for q in 0 1 0 1; do echo -n "C3 q=$q: "; MVF_DZFREE_Q=$q timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
for q in 0 1 0 1; do echo -n "C4 q=$q: "; MVF_DZFREE_Q=$q timeout 300 python bench.py --depth 101 --frames 16 --clips 16 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
