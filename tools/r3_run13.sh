export MVF_LIB_PATH=$PWD/mvfnet_amd/libmvfnet_hip_ablate.so
for abl in 0 1 2 3 4 7 16 20 23; do
MVF_STEM_ABL=$abl timeout 600 python bench.py --steps 6 --warmup 2 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -E "K147" | head -1 | sed "s/^/ABL=$abl /"
done
