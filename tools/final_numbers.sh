#!/bin/bash
# The table of DESIGN.md section 5: every bench configuration once (gpurun box). Output: gpurun_out/final_numbers.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final_numbers.txt; : > $O
run() { echo "## bench.py $*" >> $O; timeout 900 python $R/bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | tail -1 >> $O; }
run --steps 20 --warmup 5
run --steps 20 --warmup 5
run --dtype f32
run --mode infer --dtype f32 --steps 20 --warmup 5
run --mode infer --dtype bf16 --steps 30 --warmup 5
run --mode video --dtype bf16 --steps 30 --warmup 5
run --mode video --dtype f32 --steps 20 --warmup 5
run --clips 12
run --depth 101 --frames 16 --clips 16
run --depth 101 --frames 16 --clips 32 --steps 5
