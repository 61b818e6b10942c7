#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
(timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5)
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
