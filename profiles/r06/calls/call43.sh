# second soak on the final code: new seeds in every mode, bounded runs
O=gpurun_out/r06/soak4; mkdir -p $O
run() { n=$1; cases=$2; seed=$3; shift 3; ( time env "$@" timeout 1200 python tests/tools/fuzz_strip.py $cases $seed ) > $O/$n.txt 2>&1; echo "rc=$?" >> $O/$n.txt; }
for seed in 5001 5002 5003 5004 5005 5006; do run default_8000_seed$seed 8000 $seed X=1; done
for seed in 5101 5102; do run jinc_5000_seed$seed 5000 $seed MPCVR_FUZZ_JINC=1; done
for seed in 5201 5202; do run jinc_flags8_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8; done
for seed in 5301 5302; do run jinc_flags64_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=64; done
for seed in 5401 5402; do run scalers_unaligned_flags72_5000_seed$seed 5000 $seed MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72; done
for seed in 5501 5502; do run host_unaligned_flags4_5000_seed$seed 5000 $seed MPCVR_FUZZ_HOST=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=4; done
for seed in 5601 5602 5603; do run periodic_5000_seed$seed 5000 $seed MPCVR_FUZZ_PERIODIC=1; done
for seed in 5701 5702; do run scalers_5000_seed$seed 5000 $seed MPCVR_FUZZ_SCALERS=1; done
( timeout 600 python tests/tools/fuzz_errdiff.py 400 41 2>&1 | tail -6 ) > $O/errdiff_400_seed41.txt; echo "rc=$?" >> $O/errdiff_400_seed41.txt
grep -H "^rc=" $O/*.txt > $O/SUMMARY.txt; grep -v "rc=0" $O/SUMMARY.txt; echo "runs: $(wc -l < $O/SUMMARY.txt)"
O=gpurun_out/r06/soak4
for seed in 5801 5802; do run flags16_5000_seed$seed 5000 $seed MPCVR_FUZZ_FLAGS=16; done
for seed in 5901 5902; do run periodic_flags256_5000_seed$seed 5000 $seed MPCVR_FUZZ_PERIODIC=1 MPCVR_FUZZ_FLAGS=256; done
grep -H "^rc=" $O/*.txt > $O/SUMMARY.txt; grep -v "rc=0" $O/SUMMARY.txt; echo "runs: $(wc -l < $O/SUMMARY.txt)"
