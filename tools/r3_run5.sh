timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv_dgrad_wgrad or bottleneck_train or c1_train_two" 2>&1 | grep -v amdgpu.ids | tail -12
timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs > gpurun_out/r3_wp4_bench.json 2> gpurun_out/r3_wp4_perlayer.txt
MVF_WGRAD_P4=0 timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs > gpurun_out/r3_wp4off_bench.json 2> gpurun_out/r3_wp4off_perlayer.txt
python - <<PY
import json
for f in ["r3_wp4_bench.json","r3_wp4off_bench.json"]:
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], "conv", r["ms_per_step"], "wgrad", r["wgrad"]["ms_per_step"], r["wgrad"]["tflops"])
PY
grep wgrad gpurun_out/r3_wp4_perlayer.txt | head -12; echo; grep wgrad gpurun_out/r3_wp4off_perlayer.txt | head -12
