#!/bin/bash
# round 5, call 23: the exact convert form in front of an HDR10 tone-mapping operator — the suite, the fuzz runs that found it and a default-mode seed
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
( time MPCVR_FUZZ_FLAGS=128 timeout 600 python tests/tools/fuzz_strip.py 2500 208 ) > $O/fuzz_2500_flags128.txt 2>&1; echo "rc=$?" >> $O/fuzz_2500_flags128.txt
( time timeout 600 python tests/tools/fuzz_strip.py 6000 303 ) > $O/fuzz_6000_seed303.txt 2>&1; echo "rc=$?" >> $O/fuzz_6000_seed303.txt
for f in fuzz_2500_flags128 fuzz_6000_seed303; do echo "== $f"; grep -E "^rc=|Error|^cases" $O/$f.txt | cut -c1-260; done
for wl in hdrpass_1440 c3hdr; do python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['value']))"; done
