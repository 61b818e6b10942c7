#!/bin/bash
# Round-5 evidence on the round's kernels: every bench configuration, profiles (C3 bf16 + fp32 train, C4, C2, C5), per-layer tables, step timeline, one default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/final_numbers.sh
bash tools/collect_profiles.sh bf16 > gpurun_out/collect_bf16.log 2>&1
cd $R; bash tools/collect_profiles.sh f32 > gpurun_out/collect_f32.log 2>&1
cd $R; SKIP_SQ=1 bash tools/collect_profiles.sh bf16 _c4 --depth 101 --frames 16 --clips 16 > gpurun_out/collect_c4.log 2>&1
cd $R; SKIP_SQ=1 bash tools/collect_profiles.sh f32 _c2 --mode infer > gpurun_out/collect_c2.log 2>&1
cd $R; SKIP_SQ=1 bash tools/collect_profiles.sh f32 _c5 --mode video > gpurun_out/collect_c5f.log 2>&1
cd $R; SKIP_SQ=1 bash tools/collect_profiles.sh bf16 _c5 --mode video > gpurun_out/collect_c5b.log 2>&1
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2> gpurun_out/per_layer_bf16.txt > /dev/null
timeout 600 python bench.py --depth 101 --frames 16 --clips 16 --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2> gpurun_out/per_layer_c4.txt > /dev/null
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/tl -- python $R/tools/trace_steps.py bf16 5 > /dev/null 2>&1; python $R/tools/step_timeline.py $(find /tmp/tl -name "*.db" | head -1) 4 > $R/gpurun_out/step_timeline.txt 2>&1)
(for c in 12 32; do timeout 300 python tools/host_overhead.py bf16 $c noprofile; done) > gpurun_out/host_overhead.txt 2>&1
timeout 1500 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err
tail -c 600 gpurun_out/default_bench.json
