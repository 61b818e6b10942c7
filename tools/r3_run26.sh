timeout 1200 python -m pytest tests/test_mvf_gpu.py tests/test_net_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -4
timeout 1200 python -m pytest tests/test_train_gpu.py -q -m gpu -p no:cacheprovider -x -k "c1_train or bottleneck_train or mvf or norm_eval_training" 2>&1 | grep -v amdgpu.ids | tail -4
for v in 1 0; do
MVF_STENCIL_CHUNKED=$v timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -E "mvf " | sed "s/^/CHUNKED=$v /"
done
for v in 1 0 1 0; do
MVF_STENCIL_CHUNKED=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-eager-compare --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train CHUNKED=$v', d['value'], d['ms_per_step'], 'mvf', d['roofline']['mvf']['ms_per_step'])"
done
for v in 1 0; do
MVF_STENCIL_CHUNKED=$v timeout 600 python bench.py --mode infer --dtype bf16 --steps 30 --warmup 5 --no-eager-compare --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('infer bf16 CHUNKED=$v', d['value'], d['ms_per_step'])"
done
