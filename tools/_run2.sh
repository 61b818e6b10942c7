cd $GRAFT_REPO_ROOT
for fk in 0 1 2; do echo "== fused kernel variant $fk"; MVF_FUSE_KERNEL=$fk python -m pytest tests/test_conv_gpu.py -m gpu -q -k mvf_fused 2>&1 | tail -1; done
python -m pytest tests/test_net_gpu.py -m gpu -q 2>&1 | tail -1
for cfg in "1 -1" "1 0" "1 1" "1 2" "0 -1"; do set -- $cfg; for dt in bf16 f32; do
  echo "== FUSE=$1 KERNEL=$2 $dt infer"; MVF_FUSE_LOADER=$1 MVF_FUSE_KERNEL=$2 python bench.py --mode infer --dtype $dt --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
