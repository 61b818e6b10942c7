#!/bin/bash
# round 6: per-queue timeline of the plan-replayed step at 32 and 12 clips per GPU (kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for c in 32 12; do
  rm -rf /tmp/tl_$c
  rocprofv3 --kernel-trace --output-format rocpd -d /tmp/tl_$c -- python $R/tools/trace_steps.py bf16 6 $c > /tmp/tl_$c.log 2>&1
  db=$(find /tmp/tl_$c -name "*.db" | head -1)
  { echo "## $c clips per GPU: tools/step_timeline.py (last 4 steps)"; python $R/tools/step_timeline.py $db 4 2>&1 | head -80; } > $O/r6_timeline_$c.txt
done
head -45 $O/r6_timeline_12.txt
