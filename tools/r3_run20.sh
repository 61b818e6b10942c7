export PYTHONPATH=$PWD
python tools/c3_bench.py 20 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -p no:cacheprovider -k "conv3x3_c64" 2>&1 | grep -v amdgpu.ids | tail -5
for v in 1 0; do
MVF_CONV3X3_DIRECT=$v timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -E "N64 K576" | sed "s/^/DIRECT=$v /"
done
