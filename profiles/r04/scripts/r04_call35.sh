#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_errdiff.py -m gpu -q 2>&1 | tail -2
python __graft_entry__.py --smoke 2>&1 | tail -1 | cut -c1-120
