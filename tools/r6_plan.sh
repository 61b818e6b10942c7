#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_launch_plan_gpu.py -x -q -s -m gpu > $O/r6_plan_tests.txt 2>&1
grep -v "^$" $O/r6_plan_tests.txt | tail -30
for p in 1 0; do for c in 12 32; do echo "== MVF_POLICY=plan=$p"; MVF_POLICY=plan=$p python tools/host_overhead.py bf16 $c noprofile 2>&1 | grep -v amdgpu.ids; done; done > $O/r6_host_overhead.txt 2>&1
cat $O/r6_host_overhead.txt
