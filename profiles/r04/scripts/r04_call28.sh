#!/bin/bash
# the general fuzz net on the round's final library (tests/tools/fuzz_strip.py: default planner against the plain kernels / the oracle)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python tests/tools/fuzz_strip.py 3000 20260927 2>&1 | tail -12 > $O/fuzz_final_3000.txt
cut -c1-500 $O/fuzz_final_3000.txt
