#!/bin/bash
# A/B sweeps of environment switches on bench.py (C3, then C4); edit the `run` list.  Output: gpurun_out/r5_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*" >> gpurun_out/r5_sweep.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_late_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_sweep.txt; }
: > gpurun_out/r5_sweep.txt
for X in "" "--depth 101 --frames 16 --clips 16"; do
for i in 1 2; do run A=0; run MVF_WGRAD_REDUCE4=1; run MVF_SIDE_HOLD=1; run MVF_AUX_DOWNSAMPLE_BWD=1; run MVF_SIDE_DOWNSAMPLE=0; done
done
cat gpurun_out/r5_sweep.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
