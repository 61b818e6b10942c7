timeout 3000 python -m pytest tests/ -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|^FAILED|^ERROR" | tail -15
