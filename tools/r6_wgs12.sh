#!/bin/bash
# round 6: weight-gradient grid targets at the reference's own batch size (12 clips per GPU): fewer, deeper splits = fewer fp32 slabs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
{
echo "## wgrad_big_wgs (256 x 256 tile), 12 clips"; python tools/ab_env.py wgrad_big_wgs 128 64 32 --reps 2 -- --clips 12
echo "## wgrad_wgs (128 x 128 plans), 12 clips"; python tools/ab_env.py wgrad_wgs 256 128 64 --reps 2 -- --clips 12
} > $O/r6_wgs12.txt 2>&1
cat $O/r6_wgs12.txt
