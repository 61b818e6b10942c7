for cfg in "8" "16" "4"; do
MVF_CONV_BIG2=$cfg MVF_CONV_BIG2_FORCE=1 timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline > gpurun_out/r3_p4force${cfg}_bench.json 2> gpurun_out/r3_p4force${cfg}_perlayer.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_p4force${cfg}_bench.json").read().strip().splitlines()[-1])
print("force from ${cfg} chunks:", d["value"], d["ms_per_step"], "conv", d["roofline"]["ms_per_step"])
PY
done
