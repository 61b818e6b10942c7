#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2; do
echo -n "C3 default: "; run
echo -n "C3 conv_prio=1: "; MVF_CONV_PRIO=1 run
echo -n "C3 mask_lds=0: "; MVF_MASK_LDS=0 run
echo -n "C3 glds1=4: "; MVF_CONV_GLDS1=4 run
echo -n "C3 glds1=12: "; MVF_CONV_GLDS1=12 run
echo -n "C3 wgrad_map=0: "; MVF_WGRAD_MAP=0 run
done
grep -n "MVF_CONV_GLDS1\|g_glds1_max" mvfnet_amd/csrc/conv_nhwc.hip | head -5
