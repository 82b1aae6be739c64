O=gpurun_out/r06; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite_final_plain.txt 2>&1; tail -4 $O/gpu_suite_final_plain.txt
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke_final.txt 2>&1; tail -5 $O/smoke_final.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape_final.json 2> $O/bench_driver_shape_final.err; tail -c 600 $O/bench_driver_shape_final.json
