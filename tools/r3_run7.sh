timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs > gpurun_out/r3_wg6_bench.json 2> gpurun_out/r3_wg6_perlayer.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-eager-compare --no-cpu-baseline --no-other-configs > gpurun_out/r3_wg6b_bench.json 2>/dev/null
python - <<PY
import json
for f in ["r3_wg6_bench.json","r3_wg6b_bench.json"]:
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], "conv", r["ms_per_step"], r["frac"], r["mfma_frac"], "wgrad", r["wgrad"]["ms_per_step"], r["wgrad"]["tflops"], r["wgrad"]["hbm_frac"], "bn", r["bn"]["ms_per_step"])
PY
