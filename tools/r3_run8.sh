timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -p no:cacheprovider -k "device_prefetcher or train_network_shim or dist_eval" 2>&1 | grep -E "passed|failed|^E " | head
python bench.py --host-input --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r3_hostinput_bench.json 2> gpurun_out/r3_hostinput_bench.err
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r3_resident_bench.json 2> /dev/null
python - <<PY
import json
for f in ["r3_hostinput_bench.json","r3_resident_bench.json"]:
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("host_input"))
PY
