#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gputests_r5.log 2>&1
grep -n "passed\|failed\|FAILED" gpurun_out/gputests_r5.log | tail -12
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep "smoke ok"
