export MVF_LIB_PATH=$PWD/mvfnet_amd/libmvfnet_hip_ablate.so
echo "== conv3 forward + statistics as today (EPI 1)"; MVF_CONV_PRIO=0 python tools/kbench.py conv "c3 fwd" 2>&1 | grep -v amdgpu
echo "== statistics-only pass (no output stores: MVF_CONV_PRIO=32)"; MVF_CONV_PRIO=32 python tools/kbench.py conv "c3 fwd" 2>&1 | grep -v amdgpu
echo "== conv3 + bias + residual + ReLU (proxy of the fused bn3-apply pass)"; MVF_CONV_PRIO=0 python tools/kbench.py conv "c3 apply" 2>&1 | grep -v amdgpu
unset MVF_LIB_PATH
python tools/kbench.py bn 2>&1 | grep -v amdgpu | head -30
