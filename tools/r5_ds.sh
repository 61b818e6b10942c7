#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 900 python -m pytest tests/test_train_gpu.py tests/test_bf16_parity_gpu.py -q -m gpu -p no:cacheprovider -k "engine_switch or bit_identical or every_block or bottleneck" 2>&1 | tail -5)
run() { echo "## $*" >> gpurun_out/r5_ds_ab.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_late_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_ds_ab.txt; }
: > gpurun_out/r5_ds_ab.txt
for X in "" "--depth 101 --frames 16 --clips 16"; do
for i in 1 2 3; do run MVF_DZFREE_DS=0; run MVF_DZFREE_DS=1; done
done
cat gpurun_out/r5_ds_ab.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
