#!/bin/bash
# Round 5, first GPU pass: the round's new gates, the C4 evidence the judge asked for (item 1a), the host-side measurement at the reference's recipe (item 7b).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_train_gpu.py -q -m gpu -x -s -k "tight_over_seeds or trailing_partial or test_norm_eval_training_vs_reference_golden" -p no:cacheprovider 2>&1 | tail -25) > gpurun_out/r5_tests_a.txt
(timeout 900 python -m pytest tests/test_dist_gpu.py -q -m gpu -x -k "two_ranks_on_one_gpu or rccl_path" -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r5_tests_b.txt
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > gpurun_out/r5_smoke.txt
(for c in 12 32; do timeout 300 python tools/host_overhead.py bf16 $c noprofile; done; timeout 300 python bench.py --clips 12 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | cut -c1-400) > gpurun_out/r5_clips12.txt 2>&1
SKIP_SQ=1 bash tools/collect_profiles.sh bf16 _c4 --depth 101 --frames 16 --clips 16 > gpurun_out/collect_c4.log 2>&1
cd $R
timeout 600 python bench.py --depth 101 --frames 16 --clips 16 --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2> gpurun_out/per_layer_c4.txt > gpurun_out/bench_c4.json
tail -c 300 gpurun_out/bench_c4.json
