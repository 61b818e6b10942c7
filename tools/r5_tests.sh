#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8)
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --per-layer --no-cpu-baseline --no-other-configs 2>gpurun_out/per_layer_q.txt | tail -1 | cut -c1-400
