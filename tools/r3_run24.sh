export PYTHONPATH=$PWD
for s in 0 -1 -2 -3 0 -1; do echo STAGGER=$s; MVF_CONV3X3_STAGGER=$s python tools/c3_bench.py 20 2>&1 | grep -v amdgpu; done
