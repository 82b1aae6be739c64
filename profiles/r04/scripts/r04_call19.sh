#!/bin/bash
# error-diffusion pass: where the time goes (SQ counters of k_error_diffusion on the c4ed workload) and what larger batches return
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for b in 64 128; do
  timeout 300 python bench.py --workload c4ed --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null | tail -n 1 | sed "s/^{/{\"batch\": $b, /" | cut -c1-260
done
KFILTER=k_error_diffusion bash tools/prof_headline.sh errdiff_c4ed --workload c4ed --steps 8 --warmup 2 2>&1 | tail -45
rm -rf $O/errdiff_c4ed
