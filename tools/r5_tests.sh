#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12) > gpurun_out/r5_full_gpu_tests.txt
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) >> gpurun_out/r5_full_gpu_tests.txt
tail -6 gpurun_out/r5_full_gpu_tests.txt
