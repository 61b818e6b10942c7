#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*" >> gpurun_out/r5_sweep.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $X 2>>gpurun_out/r5_late_err.txt | tail -1 | cut -c1-330 >> gpurun_out/r5_sweep.txt; }
: > gpurun_out/r5_sweep.txt
for X in "" "--depth 101 --frames 16 --clips 16"; do
for i in 1 2; do run A=0; run MVF_FUSE_BNWG=3; run MVF_FUSE_BNWG=0; run MVF_STEM_WGRAD_MAIN=0; run MVF_WGRAD_BIG=2; run MVF_FUSE_BN3_APPLY=2; run MVF_FUSE_BN3_APPLY=0 MVF_Z3_FREE=0; done
done
cat gpurun_out/r5_sweep.txt | grep -o '## .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
