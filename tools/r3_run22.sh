export PYTHONPATH=$PWD
python tools/c3_bench.py 20 2>&1 | grep -v amdgpu
python tools/c3_bench.py 20 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -p no:cacheprovider -k "conv3x3_c64" 2>&1 | grep -v amdgpu.ids | tail -5
