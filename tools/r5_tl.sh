#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "MVF_DZFREE=0" "MVF_DZFREE=2" "MVF_DZFREE=2 MVF_GATE_PRODUCER=0" "MVF_DZFREE=2 MVF_FUSE_MVF_STATS=0"; do
  rm -rf /tmp/tl; env $v rocprofv3 --kernel-trace -d /tmp/tl -- python $R/tools/trace_steps.py bf16 5 > /dev/null 2>&1
  echo "=== $v"; python $R/tools/step_timeline.py $(find /tmp/tl -name "*.db" | head -1) 4 | head -9
done > $R/gpurun_out/r5_tl_variants.txt 2>&1
(timeout 300 python $R/tools/probes/hipgraph_probe.py 12; timeout 300 python $R/tools/probes/hipgraph_probe.py 32) > $R/gpurun_out/r5_hipgraph.txt 2>&1
cat $R/gpurun_out/r5_tl_variants.txt; grep -v amdgpu $R/gpurun_out/r5_hipgraph.txt
