timeout 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -p no:cacheprovider -k "conv3x3_c64 or stem_direct" 2>&1 | grep -v amdgpu.ids | tail -25
for v in 1 0; do
MVF_CONV3X3_DIRECT=$v timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -E "N64 K576" | sed "s/^/DIRECT=$v /"
done
for v in 1 0 1 0; do
MVF_CONV3X3_DIRECT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-eager-compare --no-cpu-baseline --no-other-configs > gpurun_out/r3_c3_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_c3_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("MVF_CONV3X3_DIRECT=$v", d["value"], d["ms_per_step"], "conv", r["ms_per_step"], r["launches_per_step"], r["frac"], r["mfma_frac"])
PY
done
for v in 1 0; do
MVF_CONV3X3_DIRECT=$v timeout 600 python bench.py --mode infer --dtype bf16 --steps 30 --warmup 5 --no-eager-compare --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('infer bf16 CONV3X3_DIRECT=$v', d['value'], d['ms_per_step'])"
done
