#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 1200 python -m pytest tests/test_dzfree_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4)
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2 3; do
for z in 0 1; do echo -n "C3 q_slabs=$z: "; MVF_DZFREE_Q_SLABS=$z run; done
for z in 0 1; do echo -n "C4 q_slabs=$z: "; MVF_DZFREE_Q_SLABS=$z run --depth 101 --frames 16 --clips 16; done
done
