mkdir -p gpurun_out/r06
python tools/debug/case6375.py > gpurun_out/r06/case6375.txt 2>&1
