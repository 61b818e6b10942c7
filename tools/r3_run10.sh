for v in 1 0 1 0; do
MVF_FUSE_BN3_APPLY=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-eager-compare --no-cpu-baseline --no-other-configs > gpurun_out/r3_bn3_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bn3_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("MVF_FUSE_BN3_APPLY=$v", d["value"], d["ms_per_step"], "conv", r["ms_per_step"], r["launches_per_step"], r["frac"], r["mfma_frac"], "bn", r["bn"]["ms_per_step"], r["bn"]["launches_per_step"])
PY
done
