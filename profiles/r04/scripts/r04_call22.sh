#!/bin/bash
# error-diffusion pass: 400 random shapes against the serial model (tests/tools/fuzz_errdiff.py), two seeds
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 500 python tests/tools/fuzz_errdiff.py 250 1 2>&1 | tail -8 > $O/fuzz_errdiff.txt
timeout 500 python tests/tools/fuzz_errdiff.py 250 2 2>&1 | tail -8 >> $O/fuzz_errdiff.txt
cat $O/fuzz_errdiff.txt | cut -c1-400
