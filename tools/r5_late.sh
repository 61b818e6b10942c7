#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*" >> gpurun_out/r5_late_ab.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_late_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_late_ab.txt; }
: > gpurun_out/r5_late_ab.txt
X=""
for i in 1 2 3; do run MVF_SIDE_LATE=0; run MVF_SIDE_LATE=1; done
X="--depth 101 --frames 16 --clips 16"
for i in 1 2; do run MVF_SIDE_LATE=0; run MVF_SIDE_LATE=1; done
X="--clips 12"
for i in 1 2; do run MVF_SIDE_LATE=0; run MVF_SIDE_LATE=1; done
cat gpurun_out/r5_late_ab.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
(timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -p no:cacheprovider -k "bit_identical or engine_switch or reuses" 2>&1 | tail -4)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace -d /tmp/tl -- python $R/tools/trace_steps.py bf16 5 > /dev/null 2>&1
python $R/tools/step_timeline.py $(find /tmp/tl -name "*.db" | head -1) 4 | head -12
