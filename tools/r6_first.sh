#!/bin/bash
# round 6, first GPU call: the new full-size oracle rows, the touched tests, smoke, weight-gradient launches at half / full chip, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_fullsize_default_kernels_gpu.py -x -q -s -m gpu > $O/r6_fullsize_tests.txt 2>&1
tail -3 $O/r6_fullsize_tests.txt
python -m pytest tests/test_dzfree_gpu.py -x -q -s -m gpu -k "without_dz_match" > $O/r6_dzfree_mean.txt 2>&1
tail -3 $O/r6_dzfree_mean.txt
python -m pytest tests/test_bf16_parity_gpu.py -x -q -s -m gpu -k "full_size_bf16_train_step" > $O/r6_perm.txt 2>&1
tail -3 $O/r6_perm.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.txt 2>&1
tail -2 $O/r6_smoke.txt
for w in 128 256; do echo "== MVF_WGRAD_BIG_WGS=$w"; MVF_WGRAD_BIG_WGS=$w python tools/wgbench.py; done > $O/r6_wgbench.txt 2>&1
python bench.py > $O/r6_bench0.json 2> $O/r6_bench0.err
tail -c 600 $O/r6_bench0.json
