#!/bin/bash
# round 5, call 21: more fuzz seeds on the final code (spare GPU minutes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
( time timeout 900 python tests/tools/fuzz_strip.py 8000 123 ) > $O/fuzz_8000_seed123.txt 2>&1; echo "rc=$?" >> $O/fuzz_8000_seed123.txt
( time timeout 900 python tests/tools/fuzz_strip.py 8000 999 ) > $O/fuzz_8000_seed999.txt 2>&1; echo "rc=$?" >> $O/fuzz_8000_seed999.txt
( time MPCVR_FUZZ_JINC=1 timeout 900 python tests/tools/fuzz_strip.py 4000 9 ) > $O/fuzz_4000_jinc_seed9.txt 2>&1; echo "rc=$?" >> $O/fuzz_4000_jinc_seed9.txt
( time MPCVR_FUZZ_PERIODIC=1 timeout 900 python tests/tools/fuzz_strip.py 3000 31 ) > $O/fuzz_3000_periodic_seed31.txt 2>&1; echo "rc=$?" >> $O/fuzz_3000_periodic_seed31.txt
( timeout 400 python tests/tools/fuzz_errdiff.py 250 21 2>&1 | tail -3; timeout 400 python tests/tools/fuzz_errdiff.py 250 33 2>&1 | tail -3 ) > $O/fuzz_errdiff_seeds21_33.txt
for f in fuzz_8000_seed123 fuzz_8000_seed999 fuzz_4000_jinc_seed9 fuzz_3000_periodic_seed31; do echo "== $f"; grep -E "^rc=|Error|^cases" $O/$f.txt | cut -c1-200; done; cut -c1-120 $O/fuzz_errdiff_seeds21_33.txt
