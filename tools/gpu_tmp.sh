cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -m gpu -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -4
