#!/bin/bash
# the whole GPU suite + smoke (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/r6_full_gpu_tests.txt 2>&1
tail -15 $O/r6_full_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.txt 2>&1
tail -2 $O/r6_smoke.txt
