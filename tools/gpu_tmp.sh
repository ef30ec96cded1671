cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|AssertionError|launch" | head -20
timeout 2400 python -m pytest tests/ -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|AssertionError|launch" | head -20
