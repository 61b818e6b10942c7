#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
MVF_DZFREE=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kz$v -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-overlap > /dev/null 2>&1
cp $(find /tmp/kz$v -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r5_ks_dzfree$v.csv
done
cd $R
MVF_DZFREE=1 timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2> gpurun_out/r5_per_layer_dzfree1.txt > /dev/null
