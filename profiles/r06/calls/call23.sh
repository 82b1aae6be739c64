O=gpurun_out/r06/soak; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "soak_case_2367 or give_up_path or fuzz_case or tonemap" > gpurun_out/r06/gpu_new_tests_call23.txt 2>&1; tail -3 gpurun_out/r06/gpu_new_tests_call23.txt
run() { n=$1; cases=$2; seed=$3; shift 3; ( time env "$@" timeout 1200 python tests/tools/fuzz_strip.py $cases $seed ) > $O/$n.txt 2>&1; echo "rc=$?" >> $O/$n.txt; }
for seed in 2102 2103 2104 2105 2106; do run jinc_flags8_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8; done
for seed in 2004 2005; do run jinc_4000_seed$seed 4000 $seed MPCVR_FUZZ_JINC=1; done
for seed in 1013 1014 1015 1016 1017 1018; do run default_8000_seed$seed 8000 $seed X=1; done
for seed in 2304 2305; do run scalers_unaligned_flags72_4000_seed$seed 4000 $seed MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72; done
for seed in 2701 2702; do run flags16_4000_seed$seed 4000 $seed MPCVR_FUZZ_FLAGS=16; done
for seed in 2801 2802; do run flags128_periodic_4000_seed$seed 4000 $seed MPCVR_FUZZ_PERIODIC=1 MPCVR_FUZZ_FLAGS=128; done
grep -H "^rc=" $O/*.txt > $O/SUMMARY.txt; grep -v "rc=0" $O/SUMMARY.txt; echo "runs: $(wc -l < $O/SUMMARY.txt)"
