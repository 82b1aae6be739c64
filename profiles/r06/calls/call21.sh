# soak: new seeds in every mode of the fuzz tool, each run bounded; a failing run stops nothing
O=gpurun_out/r06/soak; mkdir -p $O
run() { n=$1; cases=$2; seed=$3; shift 3; ( time env "$@" timeout 1200 python tests/tools/fuzz_strip.py $cases $seed ) > $O/$n.txt 2>&1; echo "rc=$?" >> $O/$n.txt; }
for seed in 1001 1002 1003 1004 1005 1006 1007 1008 1009 1010 1011 1012; do run default_8000_seed$seed 8000 $seed X=1; done
for seed in 2001 2002 2003; do run jinc_4000_seed$seed 4000 $seed MPCVR_FUZZ_JINC=1; done
for seed in 2101 2102; do run jinc_flags8_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8; done
for seed in 2201 2202; do run jinc_flags64_3000_seed$seed 3000 $seed MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=64; done
for seed in 2301 2302 2303; do run scalers_unaligned_flags72_4000_seed$seed 4000 $seed MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72; done
for seed in 2401 2402 2403; do run host_unaligned_flags4_4000_seed$seed 4000 $seed MPCVR_FUZZ_HOST=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=4; done
for seed in 2501 2502 2503; do run periodic_4000_seed$seed 4000 $seed MPCVR_FUZZ_PERIODIC=1; done
for seed in 2601 2602; do run scalers_4000_seed$seed 4000 $seed MPCVR_FUZZ_SCALERS=1; done
grep -H "^rc=" $O/*.txt > $O/SUMMARY.txt; cat $O/SUMMARY.txt
