#!/bin/bash
# Round-3 evidence refresh: final numbers, profiles (bf16), per-layer table, one default bench line.  Output under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/final_numbers.sh
bash tools/collect_profiles.sh bf16 > gpurun_out/collect_bf16.log 2>&1
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2> gpurun_out/per_layer_bf16.txt > /dev/null
timeout 1200 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err
tail -c 600 gpurun_out/default_bench.json
