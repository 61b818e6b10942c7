#!/bin/bash
# round 6: the NS-stage LDS-DMA ring of the four-wave weight-gradient tile -- parity (oracle test at ring depth 4 and 3) and single-launch timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for st in 4 3; do
  MVF_WGRAD_STAGES=$st python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "conv_dgrad_wgrad_vs_oracle and bf16" 2>&1 | tail -2
done > $O/r6_wg_tests.txt 2>&1
cat $O/r6_wg_tests.txt
for cfg in "MVF_WGRAD_STAGES=2" "MVF_WGRAD_STAGES=3" "MVF_WGRAD_STAGES=4" "MVF_WGRAD_STAGES=2 MVF_WGRAD_BIG=0" "MVF_WGRAD_STAGES=4 MVF_WGRAD_BIG=0" "MVF_WGRAD_STAGES=4 MVF_WGRAD_BIG=0 MVF_WGRAD_WGS=128" "MVF_WGRAD_STAGES=4 MVF_WGRAD_BIG=0 MVF_WGRAD_WGS=512"; do
  echo "== $cfg"; env $cfg python tools/wgbench.py 2>&1 | grep -v amdgpu.ids
done > $O/r6_wgbench2.txt 2>&1
cat $O/r6_wgbench2.txt
python -m pytest tests/test_fullsize_default_kernels_gpu.py -x -q -s -m gpu > $O/r6_fullsize_tests.txt 2>&1
grep -v "^$" $O/r6_fullsize_tests.txt | tail -25
python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.txt 2>&1
tail -2 $O/r6_smoke.txt
