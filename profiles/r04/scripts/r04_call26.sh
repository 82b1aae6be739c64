#!/bin/bash
# error-diffusion pass behind every kind of plan: fuzz with v210 / interleaved RGB / Dolby Vision / rotation / flip thrown in
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 500 python tests/tools/fuzz_errdiff.py 250 11 2>&1 | tail -12 > $O/fuzz_errdiff2.txt
timeout 500 python tests/tools/fuzz_errdiff.py 250 12 2>&1 | tail -12 >> $O/fuzz_errdiff2.txt
cat $O/fuzz_errdiff2.txt | cut -c1-600
