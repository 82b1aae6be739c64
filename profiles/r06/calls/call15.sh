mkdir -p gpurun_out/r06
python tools/debug/dovi_tail_stages.py > gpurun_out/r06/dovi_tail_stages.txt 2>&1
