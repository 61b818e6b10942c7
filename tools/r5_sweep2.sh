#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
both() { echo -n "$1 | C3: "; run | tr '\n' ' '; echo -n " C4: "; run --depth 101 --frames 16 --clips 16; }
for rep in 1 2; do
both "default"
MVF_FUSE_BN3_APPLY=0 both "bn3_apply=0"
MVF_FUSE_BN3_APPLY=2 both "bn3_apply=2"
MVF_Z3_FREE=0 both "z3_free=0"
MVF_FUSE_BNWG=0 both "bnwg=0"
MVF_FUSE_BNWG=3 both "bnwg=3"
MVF_FUSE_BNWG=5 both "bnwg=5"
MVF_SIDE_BATCH=2 both "side_batch=2"
MVF_SIDE_HOLD=1 both "side_hold=1"
MVF_STEM_WGRAD_MAIN=0 both "stem_wgrad_main=0"
MVF_AUX_DOWNSAMPLE_BWD=1 both "aux_ds_bwd=1"
done
