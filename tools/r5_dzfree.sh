#!/bin/bash
# Round 5: the dz3-free bn3 backward -- unit tests, then the A/B in the step (C3 and C4), then the judge's item 1(b) proxy.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_dzfree_gpu.py -q -m gpu -x -s -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r5_dzfree_tests.txt
(timeout 900 python -m pytest tests/test_train_gpu.py tests/test_bf16_parity_gpu.py tests/test_mvf_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -12) >> gpurun_out/r5_dzfree_tests.txt
run() { echo "## $*" >> gpurun_out/r5_dzfree_ab.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_dzfree_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_dzfree_ab.txt; }
: > gpurun_out/r5_dzfree_ab.txt
X=""
for i in 1 2; do run MVF_DZFREE=0; run MVF_DZFREE=1; run MVF_DZFREE=1 MVF_GATE_PRODUCER=0; done
run MVF_DZFREE=2
run MVF_FUSE_MVF_STATS=0
run MVF_FUSE_MVF_STATS=1
X="--depth 101 --frames 16 --clips 16"
for i in 1 2; do run MVF_DZFREE=0; run MVF_DZFREE=1; done
run MVF_DZFREE=1 MVF_GATE_PRODUCER=0
(python tools/kbench.py conv "l3.c1 dgrad"; python tools/kbench.py conv "l3.c3 dgrad"; python tools/kbench.py bn 2>&1 | grep -i "1024\|C1024" | head -8) > gpurun_out/r5_item1b_proxy.txt 2>&1
cat gpurun_out/r5_dzfree_ab.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - - 
