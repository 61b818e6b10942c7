#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2; do
for z in 16 8 4; do echo -n "C3 big2=$z: "; MVF_CONV_BIG2=$z run; done
for z in 16 8 4; do echo -n "C4 big2=$z: "; MVF_CONV_BIG2=$z run --depth 101 --frames 16 --clips 16; done
done
MVF_CONV_BIG2=4 timeout 300 python bench.py --steps 10 --warmup 3 --per-layer --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep "N1024 K256\|N256 K1024\|N512 K128\|N2048 K512" | cut -c1-150
