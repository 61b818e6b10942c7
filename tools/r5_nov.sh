#!/bin/bash
# no-overlap kernel stats of the bf16 train step (every kernel's alone-time) for C3 and C4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "c3:" "c4:--depth 101 --frames 16 --clips 16"; do
  tag=${cfg%%:*}; X=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kn_$tag -- python $R/bench.py $X --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-overlap > /tmp/kn_$tag.log 2>&1
  cp $(find /tmp/kn_$tag -name "*kernel_stats.csv" | head -1) $R/gpurun_out/nov_$tag.csv
done
