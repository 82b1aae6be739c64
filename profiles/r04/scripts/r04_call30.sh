#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python -m pytest tests/test_errdiff.py -m gpu -q -k "dolby_vision_batch" 2>&1 | tail -12 | cut -c1-300
