# second soak on the final code: new seeds in every mode, bounded runs
O=gpurun_out/r06/soak2; mkdir -p $O
run() { n=$1; cases=$2; seed=$3; shift 3; ( time env "$@" timeout 1200 python tests/tools/fuzz_strip.py $cases $seed ) > $O/$n.txt 2>&1; echo "rc=$?" >> $O/$n.txt; }
for seed in 3001 3002 3003 3004 3005 3006 3007 3008; do run default_8000_seed$seed 8000 $seed X=1; done
for seed in 3101 3102 3103; do run jinc_4000_seed$seed 4000 $seed MPCVR_FUZZ_JINC=1; done
for seed in 3201 3202 3203; do run jinc_flags8_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8; done
for seed in 3301 3302; do run jinc_flags64_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=64; done
for seed in 3401 3402; do run scalers_unaligned_flags72_4000_seed$seed 4000 $seed MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72; done
for seed in 3501 3502; do run host_unaligned_flags4_4000_seed$seed 4000 $seed MPCVR_FUZZ_HOST=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=4; done
for seed in 3601 3602; do run periodic_4000_seed$seed 4000 $seed MPCVR_FUZZ_PERIODIC=1; done
for seed in 3701 3702; do run scalers_4000_seed$seed 4000 $seed MPCVR_FUZZ_SCALERS=1; done
( timeout 600 python tests/tools/fuzz_errdiff.py 400 21 2>&1 | tail -6 ) > $O/errdiff_400_seed21.txt; echo "rc=$?" >> $O/errdiff_400_seed21.txt
grep -H "^rc=" $O/*.txt > $O/SUMMARY.txt; grep -v "rc=0" $O/SUMMARY.txt; echo "runs: $(wc -l < $O/SUMMARY.txt)"
