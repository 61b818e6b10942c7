#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 900 python -m pytest tests/test_dzfree_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -14) > gpurun_out/r5_gsum_tests.txt
(timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -p no:cacheprovider -k "engine_switch or bit_identical or tight_over" 2>&1 | tail -8) >> gpurun_out/r5_gsum_tests.txt
run() { echo "## $*" >> gpurun_out/r5_gsum_ab.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_gsum_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_gsum_ab.txt; }
: > gpurun_out/r5_gsum_ab.txt
X=""
for i in 1 2; do run MVF_GATE_SUMS=0; run MVF_GATE_SUMS=1; done
X="--depth 101 --frames 16 --clips 16"
for i in 1 2; do run MVF_GATE_SUMS=0; run MVF_GATE_SUMS=1; done
cat gpurun_out/r5_gsum_ab.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
tail -5 gpurun_out/r5_gsum_tests.txt
