O=gpurun_out/r06; mkdir -p $O
for seed in 4242 31337 2718 55 77; do ( time timeout 1200 python tests/tools/fuzz_strip.py 8000 $seed ) > $O/fuzz_8000_seed$seed.txt 2>&1; echo "rc=$?" >> $O/fuzz_8000_seed$seed.txt; done
grep -H "^rc=" $O/fuzz_8000_seed*.txt
