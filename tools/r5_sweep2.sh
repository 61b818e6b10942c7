#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
both() { echo -n "$1 | C3: "; run | tr '\n' ' '; echo -n " C4: "; run --depth 101 --frames 16 --clips 16; }
for rep in 1 2; do
both "default"
MVF_WGRAD_BIG_WGS=96 both "big_wgs=96"
MVF_WGRAD_BIG_WGS=160 both "big_wgs=160"
MVF_WGRAD_WGS=192 both "wgs=192"
MVF_WGRAD_WGS=384 both "wgs=384"
MVF_GRAM_WGS=16 both "gram=16"
MVF_GRAM_WGS=64 both "gram=64"
MVF_SIDE_LATE=0 both "side_late=0"
MVF_DZFREE=1 both "dzfree=1"
done
