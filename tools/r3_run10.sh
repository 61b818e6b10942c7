timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_bf16_parity_gpu.py -q -m gpu -p no:cacheprovider -k "recomputed_conv or bnapply_pass or engine_switch_variants or c1_train_two or bottleneck_train or every_block or norm_eval_training or frozen_stages or side_stream_overlap" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|^E |Error" | head -20
for v in 1 0 2 1 0 2; do
MVF_Z3_FREE=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-eager-compare --no-cpu-baseline --no-other-configs > gpurun_out/r3_z3_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_z3_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("MVF_Z3_FREE=$v", d["value"], d["ms_per_step"], "conv", r["ms_per_step"], r["launches_per_step"], r["frac"], r["mfma_frac"], "bn", r["bn"]["ms_per_step"], r["bn"]["launches_per_step"])
PY
done
MVF_Z3_FREE=1 timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -E "bwd-|fwd\(s\)|fwd\+bn"
