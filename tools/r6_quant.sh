#!/bin/bash
# round 6: do the latency-chain-bound expand launches (128 x 128 tiles, four workgroups per CU = 1024 slots) pay for partial last rounds of tiles?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
for n in 236 250 256 262 334 342; do echo "## frames $n: l3.c3 tiles $(( (n*196+127)/128*8 )), l4.c3 tiles $(( (n*49+127)/128*16 ))"; KBENCH_FRAMES=$n python tools/kbench.py conv 3.c3 2>&1 | grep -v amdgpu.ids; KBENCH_FRAMES=$n python tools/kbench.py conv 4.c3 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r6_quant.txt 2>&1
cat gpurun_out/r6_quant.txt
