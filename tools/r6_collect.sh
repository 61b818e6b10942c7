#!/bin/bash
# round 6 evidence: rocprofv3 kernel stats (overlapped + no-overlap), HBM counters (separate --pmc passes), SQ counters for C3; kernel stats + HBM counters for C4
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/collect_profiles.sh bf16 "" 2>&1 | tail -2
SKIP_SQ=1 bash tools/collect_profiles.sh bf16 _c4 --depth 101 --frames 16 --clips 16 2>&1 | tail -2
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python bench.py --per-layer --steps 10 --warmup 5 --no-cpu-baseline --no-other-configs --no-eager-compare > gpurun_out/r6_per_layer.json 2> gpurun_out/r6_per_layer_bf16.txt
python bench.py --per-layer --depth 101 --frames 16 --clips 16 --steps 10 --warmup 5 --no-cpu-baseline --no-other-configs --no-eager-compare > gpurun_out/r6_per_layer_c4.json 2> gpurun_out/r6_per_layer_c4.txt
ls -la gpurun_out/prof_bf16 gpurun_out/prof_bf16_c4 | head -30
