#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 900 python -m pytest tests/test_dzfree_gpu.py -q -m gpu -x -s -p no:cacheprovider 2>&1 | tail -14) > gpurun_out/r5_dzfree_tests2.txt
run() { echo "## $*" >> gpurun_out/r5_dzfree_ab2.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_dzfree_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_dzfree_ab2.txt; }
: > gpurun_out/r5_dzfree_ab2.txt
X=""
for i in 1 2; do run MVF_DZFREE=0; run MVF_DZFREE=1; run MVF_DZFREE=2; done
X="--depth 101 --frames 16 --clips 16"
for i in 1 2; do run MVF_DZFREE=0; run MVF_DZFREE=1; run MVF_DZFREE=2; done
cd /tmp && export TMPDIR=/tmp
MVF_DZFREE=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kz2 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-overlap > /dev/null 2>&1
cp $(find /tmp/kz2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r5_ks_dzfree2.csv
cat $R/gpurun_out/r5_dzfree_ab2.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
