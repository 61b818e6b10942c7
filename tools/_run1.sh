cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_gpu.py -m gpu -q -k "batched_weight_pack or side_stream_overlap or c1_train" > gpurun_out/r2_t.log 2>&1; tail -3 gpurun_out/r2_t.log
python bench.py --no-cpu-baseline > gpurun_out/r2_b.json 2> gpurun_out/r2_b.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
DB=$(find /tmp/tr -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/probes/trace_adjacency.py $DB > $GRAFT_REPO_ROOT/gpurun_out/r2_adj.txt 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_breakdown.py $DB > $GRAFT_REPO_ROOT/gpurun_out/r2_kb.txt 2>&1
