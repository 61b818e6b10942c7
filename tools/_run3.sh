cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_gpu.py tests/test_bf16_parity_gpu.py -m gpu -q -x 2>&1 | tail -2
for w in 1 0 1 0; do echo "== MVF_SIDE_DOWNSAMPLE=$w"; MVF_SIDE_DOWNSAMPLE=$w python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'])"; done
