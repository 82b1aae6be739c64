mkdir -p gpurun_out/r06
python tools/debug/devdiv.py > gpurun_out/r06/devdiv.txt 2>&1
