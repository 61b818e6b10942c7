timeout 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -p no:cacheprovider -k "conv3x3_c64 or stem_direct" 2>&1 | grep -v amdgpu.ids | tail -25
export PYTHONPATH=$PWD
python tools/c3_bench.py 20
MVF_CONV3X3_DIRECT=0 python tools/c3_bench.py 20
