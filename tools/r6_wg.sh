#!/bin/bash
# round 6: the NS-stage LDS-DMA ring of the four-wave weight-gradient tile -- parity (oracle test at ring depth 4 and 3) and single-launch timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for st in 4 3; do
  MVF_POLICY=wgrad_stages=$st python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "conv_dgrad_wgrad_vs_oracle and bf16" 2>&1 | tail -2
done > $O/r6_wg_tests.txt 2>&1
cat $O/r6_wg_tests.txt
for cfg in "wgrad_stages=2" "wgrad_stages=3" "wgrad_stages=4" "wgrad_stages=2,wgrad_big=0" "wgrad_stages=4,wgrad_big=0" "wgrad_stages=4,wgrad_big=0,wgrad_wgs=128" "wgrad_stages=4,wgrad_big=0,wgrad_wgs=512"; do
  echo "== MVF_POLICY=$cfg"; MVF_POLICY=$cfg python tools/wgbench.py 2>&1 | grep -v amdgpu.ids
done > $O/r6_wgbench2.txt 2>&1
cat $O/r6_wgbench2.txt
python -m pytest tests/test_fullsize_default_kernels_gpu.py -x -q -s -m gpu > $O/r6_fullsize_tests.txt 2>&1
grep -v "^$" $O/r6_fullsize_tests.txt | tail -25
python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.txt 2>&1
tail -2 $O/r6_smoke.txt
