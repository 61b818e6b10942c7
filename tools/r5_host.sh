#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for c in 12 32; do timeout 300 python tools/host_overhead.py bf16 $c noprofile 2>&1 | grep clips; done
for c in 12 32 12 32; do echo -n "clips $c: "; timeout 300 python bench.py --clips $c --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | cut -c1-200 | grep -o '"value": [0-9.]*, "unit"\|"ms_per_step": [0-9.]*, "higher' | paste - -; done
(timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_dzfree_gpu.py tests/test_dist_gpu.py tests/test_bf16_parity_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep "passed\|failed")
