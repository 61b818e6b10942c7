python tools/wgbench.py l3 2>&1 | grep wgrad
python tools/wgbench.py l4 2>&1 | grep wgrad
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv_dgrad_wgrad" 2>&1 | tail -2
