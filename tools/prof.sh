#!/bin/bash
# tools/prof.sh <tag> [bench args...] — run on the GPU box (via gpurun): kernel-trace stats + PMC passes of bench.py.
# Writes CSVs under gpurun_out/prof_<tag>/ ; summarise with tools/prof_summary.py, copy what matters to profiles/.
# PMC passes are run WITHOUT sys/hip/hsa tracing (only --kernel-trace), as the GPU pool requires.
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
BARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-host-path $*"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path "$@" > $OUT/bench_kt.log 2>&1
pmc() { n=$1; shift; timeout -k 5 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o $n -- python bench.py $BARGS > $OUT/bench_$n.log 2>&1; }
pmc pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pmc pmc2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
pmc pmc3 FETCH_SIZE
pmc pmc4 WRITE_SIZE
pmc pmc5 GRBM_GUI_ACTIVE GRBM_COUNT
pmc pmc6 TCC_HIT_sum TCC_MISS_sum
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path "$@" > $OUT/bench_plain.log 2>&1
find $OUT -name "*.csv" | head -40
tail -1 $OUT/bench_plain.log | cut -c1-400
