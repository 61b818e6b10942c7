#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2 3; do
for z in 400 150; do echo -n "infer bf16 minwg=$z: "; MVF_STENCIL_LDS_MINWG=$z run --mode infer --dtype bf16; done
for z in 400 150; do echo -n "video bf16 minwg=$z: "; MVF_STENCIL_LDS_MINWG=$z run --mode video --dtype bf16; done
for z in 400 150; do echo -n "C4 minwg=$z: "; MVF_STENCIL_LDS_MINWG=$z run --depth 101 --frames 16 --clips 16; done
done
