#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 2400 python -m pytest tests/test_train_gpu.py tests/test_bf16_parity_gpu.py tests/test_dzfree_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/gram_tests.log 2>&1; grep "passed\|failed\|FAILED" gpurun_out/gram_tests.log | tail -5
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2 3; do
for z in 0 1; do echo -n "C3 gram_ds=$z: "; MVF_GRAM_STATS_DS=$z run; done
for z in 0 1; do echo -n "C4 gram_ds=$z: "; MVF_GRAM_STATS_DS=$z run --depth 101 --frames 16 --clips 16; done
done
