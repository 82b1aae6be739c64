#!/bin/bash
# round 5, call 22: the fuzz tool with the tier flags on the product side (every tier against the plain kernels and the oracle on random geometries)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for f in 64 128 256 8 4 72 1; do    # NO_STRIP, NO_PERIOD, FORCE_PERIOD, NO_FAST_CONVERT, NO_LUT, NO_STRIP|NO_FAST_CONVERT, LANCZOS3_FIXED
  ( time MPCVR_FUZZ_FLAGS=$f timeout 600 python tests/tools/fuzz_strip.py 2500 $((80 + f)) ) > $O/fuzz_2500_flags$f.txt 2>&1; echo "rc=$?" >> $O/fuzz_2500_flags$f.txt
  echo "== flags $f"; grep -E "^rc=|Error|^cases" $O/fuzz_2500_flags$f.txt | cut -c1-260
done
