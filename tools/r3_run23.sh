export PYTHONPATH=$PWD
export MVF_LIB_PATH=$PWD/mvfnet_amd/libmvfnet_hip_ablate.so
MVF_CONV3X3_TRACE=2 python tools/c3_bench.py 20 2>&1 | grep -v amdgpu
