O=gpurun_out/r06; mkdir -p $O/soak4
timeout 900 python -m pytest tests -q -m gpu -x -k "soak_case" 2>&1 | tail -2
( time MPCVR_FUZZ_JINC=1 timeout 1200 python tests/tools/fuzz_strip.py 5000 5102 ) > $O/soak4/jinc_5000_seed5102_rerun3.txt 2>&1; echo "rc=$?" >> $O/soak4/jinc_5000_seed5102_rerun3.txt; tail -2 $O/soak4/jinc_5000_seed5102_rerun3.txt
for seed in 5101 2001; do ( time MPCVR_FUZZ_JINC=1 timeout 1200 python tests/tools/fuzz_strip.py 4000 $seed ) > $O/soak4/jinc_4000_seed${seed}_pow10.txt 2>&1; echo "rc=$?" >> $O/soak4/jinc_4000_seed${seed}_pow10.txt; tail -1 $O/soak4/jinc_4000_seed${seed}_pow10.txt; done
