#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 1200 python -m pytest tests/test_dzfree_gpu.py -q -m gpu -p no:cacheprovider -s 2>&1 | grep "Gram\|shape\|passed\|failed\|Error\|error" | tail -14)
timeout 2400 python -m pytest tests/test_train_gpu.py tests/test_bf16_parity_gpu.py tests/test_net_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/gram_tests.log 2>&1; tail -5 gpurun_out/gram_tests.log
