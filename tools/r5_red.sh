#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*" >> gpurun_out/r5_red_ab.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_late_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_red_ab.txt; }
: > gpurun_out/r5_red_ab.txt
X=""
for i in 1 2; do run MVF_WGRAD_REDUCE_WGS=0; run MVF_WGRAD_REDUCE_WGS=512; run MVF_WGRAD_REDUCE_WGS=128; run MVF_WGRAD_REDUCE_WGS=2048; done
X="--depth 101 --frames 16 --clips 16"
for i in 1 2; do run MVF_WGRAD_REDUCE_WGS=0; run MVF_WGRAD_REDUCE_WGS=512; run MVF_WGRAD_REDUCE_WGS=128; run MVF_WGRAD_REDUCE_WGS=2048; done
cat gpurun_out/r5_red_ab.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
(timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -p no:cacheprovider -k "wgrad" 2>&1 | tail -2)
