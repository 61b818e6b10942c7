#!/bin/bash
# Round-4 evidence refresh on the round's FINAL kernels: every bench configuration, profiles (bf16 + fp32), per-layer tables, the kernel micro-benchmarks behind
# DESIGN.md's round-4 decisions, one default bench line.  Output under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/final_numbers.sh
bash tools/collect_profiles.sh bf16 > gpurun_out/collect_bf16.log 2>&1
cd $R
bash tools/collect_profiles.sh f32 > gpurun_out/collect_f32.log 2>&1
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2> gpurun_out/per_layer_bf16.txt > /dev/null
timeout 600 python bench.py --mode infer --dtype f32 --steps 10 --warmup 3 --per-layer --no-cpu-baseline 2> gpurun_out/per_layer_f32_infer.txt > /dev/null
timeout 600 python bench.py --dtype f32 --steps 5 --warmup 2 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2> gpurun_out/per_layer_f32_train.txt > /dev/null
(echo "== fp32, default (wgrad_x3_kernel)"; python tools/kbench.py wgrad; echo "== fp32, MVF_WGRAD_X3=0 (fp32 MFMA kernel)"; MVF_WGRAD_X3=0 python tools/kbench.py wgrad; echo "== bf16, default (layer1 3x3 on the direct kernel)"; python tools/kbench.py wgrad16; echo "== bf16, MVF_WGRAD3X3_DIRECT=0"; MVF_WGRAD3X3_DIRECT=0 python tools/kbench.py wgrad16 l1.c2) > gpurun_out/r4_kbench_wgrad.txt 2>&1
(python tools/kbench.py c3bwd; PYTHONPATH=. python tools/stem_bench.py; MVF_WGRAD_STEM_DIRECT=0 PYTHONPATH=. python tools/stem_bench.py | tail -1; python tools/kbench.py bnwg; python tools/kbench.py mvf; for f in 256 334; do echo "frames $f"; KBENCH_FRAMES=$f python tools/kbench.py conv "l3.c2"; KBENCH_FRAMES=$f python tools/kbench.py conv "l3.c1 fwd"; done) > gpurun_out/r4_kbench.txt 2>&1
timeout 1500 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err
tail -c 400 gpurun_out/default_bench.json
