#!/bin/bash
# round 5: the fuzz evidence — 8000 + 3000 random fused-path cases (the tool now FAILS on a channel beyond the bar where no transcendental decides
# the last code), 500 random error-diffusion shapes against the serial model
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c14; mkdir -p $O
( time timeout 1700 python tests/tools/fuzz_strip.py 8000 ) > $O/fuzz_8000.txt 2>&1; echo "rc=$?" >> $O/fuzz_8000.txt
( time timeout 900 python tests/tools/fuzz_strip.py 3000 777 ) > $O/fuzz_3000_seed777.txt 2>&1; echo "rc=$?" >> $O/fuzz_3000_seed777.txt
timeout 500 python tests/tools/fuzz_errdiff.py 250 1 2>&1 | tail -8 > $O/fuzz_errdiff.txt
timeout 500 python tests/tools/fuzz_errdiff.py 250 11 2>&1 | tail -8 >> $O/fuzz_errdiff.txt
tail -4 $O/fuzz_8000.txt | cut -c1-600; tail -4 $O/fuzz_3000_seed777.txt | cut -c1-600; tail -6 $O/fuzz_errdiff.txt | cut -c1-300
