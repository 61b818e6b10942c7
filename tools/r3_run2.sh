export MVF_LIB_PATH=$PWD/mvfnet_amd/libmvfnet_hip_ablate.so
for sel in "l3.c2 fwd" "l3.c1 fwd" "l3.c2 dgrad"; do
for prio in 0 16 8; do
  echo "== $sel  MVF_CONV_PRIO=$prio (0 full, 16 no epilogue, 8 launch only)"
  MVF_CONV_PRIO=$prio python tools/kbench.py conv "$sel"
done
done
echo "== two-barrier loop"
for sel in "l3.c2 fwd" "l3.c1 fwd"; do
for prio in 0 16; do
  MVF_CONV_P4=0 MVF_CONV_PRIO=$prio python tools/kbench.py conv "$sel"
done
done
