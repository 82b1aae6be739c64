#!/bin/bash
# the round's last library: the GPU suite and smoke() as the driver runs them
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python __graft_entry__.py --smoke 2>&1 | tail -3 | cut -c1-160
