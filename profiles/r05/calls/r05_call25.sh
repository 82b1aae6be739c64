#!/bin/bash
# round 5, call 25: fuzz modes combined (spare GPU minutes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
run() { name=$1; shift; ( time env "$@" timeout 500 python tests/tools/fuzz_strip.py 2500 $SEED ) > $O/$name.txt 2>&1; echo "rc=$?" >> $O/$name.txt; echo "== $name"; grep -E "^rc=|Error|^cases" $O/$name.txt | cut -c1-220; }
SEED=401 run fuzz_2500_jinc_flags8 MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8
SEED=402 run fuzz_2500_jinc_flags64 MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=64
SEED=403 run fuzz_2500_scalers_unaligned_flags72 MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72
SEED=404 run fuzz_2500_host_unaligned_flags4 MPCVR_FUZZ_HOST=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=4
