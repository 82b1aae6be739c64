# second soak on the final code: new seeds in every mode, bounded runs
O=gpurun_out/r06/soak3; mkdir -p $O
run() { n=$1; cases=$2; seed=$3; shift 3; ( time env "$@" timeout 1200 python tests/tools/fuzz_strip.py $cases $seed ) > $O/$n.txt 2>&1; echo "rc=$?" >> $O/$n.txt; }
for seed in 4001 4002 4003 4004 4005 4006; do run default_8000_seed$seed 8000 $seed X=1; done
for seed in 4101 4102; do run jinc_4000_seed$seed 4000 $seed MPCVR_FUZZ_JINC=1; done
for seed in 4201 4202; do run jinc_flags8_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8; done
for seed in 4301 4302; do run jinc_flags64_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=64; done
for seed in 4401 4402; do run scalers_unaligned_flags72_4000_seed$seed 4000 $seed MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72; done
for seed in 4501 4502; do run host_unaligned_flags4_4000_seed$seed 4000 $seed MPCVR_FUZZ_HOST=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=4; done
for seed in 4601 4602 4603; do run periodic_4000_seed$seed 4000 $seed MPCVR_FUZZ_PERIODIC=1; done
for seed in 4701 4702; do run scalers_4000_seed$seed 4000 $seed MPCVR_FUZZ_SCALERS=1; done
( timeout 600 python tests/tools/fuzz_errdiff.py 400 31 2>&1 | tail -6 ) > $O/errdiff_400_seed31.txt; echo "rc=$?" >> $O/errdiff_400_seed31.txt
grep -H "^rc=" $O/*.txt > $O/SUMMARY.txt; grep -v "rc=0" $O/SUMMARY.txt; echo "runs: $(wc -l < $O/SUMMARY.txt)"
O=gpurun_out/r06/soak3
for seed in 4801 4802; do run flags16_4000_seed$seed 4000 $seed MPCVR_FUZZ_FLAGS=16; done
for seed in 4901 4902; do run periodic_flags256_4000_seed$seed 4000 $seed MPCVR_FUZZ_PERIODIC=1 MPCVR_FUZZ_FLAGS=256; done
grep -H "^rc=" $O/*.txt > $O/SUMMARY.txt; grep -v "rc=0" $O/SUMMARY.txt; echo "runs: $(wc -l < $O/SUMMARY.txt)"
