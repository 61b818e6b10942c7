#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace -d /tmp/tl -- python $R/tools/trace_steps.py bf16 3 > /dev/null 2>&1
python $R/tools/gap_dump.py $(find /tmp/tl -name "*.db" | head -1) mvf_nhwc_apply_chunked 15 4 > $R/gpurun_out/r5_gap_dump.txt 2>&1
cat $R/gpurun_out/r5_gap_dump.txt
