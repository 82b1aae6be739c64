#!/bin/bash
# smoke() with the error-diffusion case and the pass's tests with the three new cases (Dolby Vision, RGB48 without a convert draw, v210)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python __graft_entry__.py --smoke 2>&1 | tail -6 | cut -c1-200
python -m pytest tests/test_errdiff.py -m gpu -q 2>&1 | tail -3
