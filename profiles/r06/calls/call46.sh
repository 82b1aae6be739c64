O=gpurun_out/r06; mkdir -p $O/soak4
timeout 900 python -m pytest tests -q -m gpu -x -k "soak_case" 2>&1 | tail -3
( time timeout 1200 python tests/tools/fuzz_strip.py 8000 5001 ) > $O/soak4/default_8000_seed5001_rerun.txt 2>&1; echo "rc=$?" >> $O/soak4/default_8000_seed5001_rerun.txt; tail -2 $O/soak4/default_8000_seed5001_rerun.txt
( time MPCVR_FUZZ_JINC=1 timeout 1200 python tests/tools/fuzz_strip.py 5000 5102 ) > $O/soak4/jinc_5000_seed5102_rerun.txt 2>&1; echo "rc=$?" >> $O/soak4/jinc_5000_seed5102_rerun.txt; tail -4 $O/soak4/jinc_5000_seed5102_rerun.txt | cut -c1-300
