#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
{
echo "## launch plan A/B (tools/ab_env.py plan 0 1): C3"; python tools/ab_env.py plan 0 1 --reps 3
echo "## 12 clips"; python tools/ab_env.py plan 0 1 --reps 3 -- --clips 12
echo "## C4"; python tools/ab_env.py plan 0 1 --reps 2 -- --depth 101 --frames 16 --clips 16
echo "## wgrad ring depth: C3"; python tools/ab_env.py wgrad_stages 2 3 --reps 3
echo "## C4"; python tools/ab_env.py wgrad_stages 2 3 --reps 2 -- --depth 101 --frames 16 --clips 16
} > $O/r6_plan_ab.txt 2>&1
grep -v amdgpu.ids $O/r6_plan_ab.txt
