export PYTHONPATH=$PWD
export MVF_LIB_PATH=$PWD/mvfnet_amd/libmvfnet_hip_ablate.so
for abl in 0 1 2 3 4 5 6; do echo ABL=$abl; MVF_CONV3X3_ABL=$abl python tools/c3_bench.py 20 2>&1 | grep -v amdgpu; done
