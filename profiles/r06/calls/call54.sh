# fifth soak, with every witness of the round in the tool: new seeds
O=gpurun_out/r06/soak5; mkdir -p $O
run() { n=$1; cases=$2; seed=$3; shift 3; ( time env "$@" timeout 1200 python tests/tools/fuzz_strip.py $cases $seed ) > $O/$n.txt 2>&1; echo "rc=$?" >> $O/$n.txt; }
for seed in 6001 6002 6003 6004; do run default_8000_seed$seed 8000 $seed X=1; done
for seed in 6101 6102; do run jinc_5000_seed$seed 5000 $seed MPCVR_FUZZ_JINC=1; done
run jinc_flags8_3000_seed6201 3000 6201 MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8
run jinc_flags64_3000_seed6301 3000 6301 MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=64
run scalers_unaligned_flags72_5000_seed6401 5000 6401 MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72
run host_unaligned_flags4_5000_seed6501 5000 6501 MPCVR_FUZZ_HOST=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=4
run periodic_5000_seed6601 5000 6601 MPCVR_FUZZ_PERIODIC=1
run scalers_5000_seed6701 5000 6701 MPCVR_FUZZ_SCALERS=1
grep -H "^rc=" $O/*.txt > $O/SUMMARY.txt; grep -v "rc=0" $O/SUMMARY.txt; echo "runs: $(wc -l < $O/SUMMARY.txt)"
