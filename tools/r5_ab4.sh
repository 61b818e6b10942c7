#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(MVF_GSUM_GLDS=1 timeout 900 python -m pytest tests/test_dzfree_gpu.py -q -m gpu -p no:cacheprovider -k fused_sums 2>&1 | tail -4) > gpurun_out/r5_gsum_tests2.txt
run() { echo "## $*" >> gpurun_out/r5_gsum_ab2.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_gsum_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_gsum_ab2.txt; }
: > gpurun_out/r5_gsum_ab2.txt
X=""
for i in 1 2; do run MVF_GATE_SUMS=0; run MVF_GATE_SUMS=1 MVF_GSUM_GLDS=1; done
X="--depth 101 --frames 16 --clips 16"
for i in 1 2; do run MVF_GATE_SUMS=0; run MVF_GATE_SUMS=1 MVF_GSUM_GLDS=1; done
cat gpurun_out/r5_gsum_ab2.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
tail -3 gpurun_out/r5_gsum_tests2.txt
