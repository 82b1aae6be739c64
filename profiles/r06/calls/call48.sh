O=gpurun_out/r06; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite_last2.txt 2>&1; tail -5 $O/gpu_suite_last2.txt | head -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
