#!/bin/bash
# Round-4 evidence refresh after the fp32 weight gradients moved to the bf16 matrix cores (wgrad_x3_kernel): the bf16 kernels are unchanged since tools/r4_final.sh ran,
# so only the fp32 profiles, the table of every configuration, the fp32 per-layer table, the weight-gradient micro-benchmark and the default bench line are re-collected.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/final_numbers.sh
bash tools/collect_profiles.sh f32 > gpurun_out/collect_f32.log 2>&1
cd $R
timeout 600 python bench.py --dtype f32 --steps 5 --warmup 2 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2> gpurun_out/per_layer_f32_train.txt > /dev/null
(echo "== default (wgrad_x3_kernel)"; python tools/kbench.py wgrad; echo "== MVF_WGRAD_X3=0 (fp32 MFMA kernel)"; MVF_WGRAD_X3=0 python tools/kbench.py wgrad; echo "== bf16"; python tools/kbench.py wgrad16) > gpurun_out/r4_kbench_wgrad.txt 2>&1
timeout 1500 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err
tail -c 400 gpurun_out/default_bench.json
