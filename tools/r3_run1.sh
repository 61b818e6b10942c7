set -x
export MVF_CONV_BIG2=1 MVF_CONV_BIG2_FORCE=1
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "not forced_by_env" -p no:cacheprovider 2>&1 | tail -15
unset MVF_CONV_BIG2 MVF_CONV_BIG2_FORCE
timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline > gpurun_out/r3_p4_bench.json 2> gpurun_out/r3_p4_perlayer.txt; tail -c 1500 gpurun_out/r3_p4_bench.json
MVF_CONV_P4=0 timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline > gpurun_out/r3_p4off_bench.json 2> gpurun_out/r3_p4off_perlayer.txt; tail -c 600 gpurun_out/r3_p4off_bench.json
