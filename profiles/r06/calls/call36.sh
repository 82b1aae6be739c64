mkdir -p gpurun_out/r06
for wlk in c3hdr up1440; do echo "== $wlk"; timeout 300 python tools/debug/half_batches.py $wlk 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06/half_batches_call36.txt 2>&1; cat gpurun_out/r06/half_batches_call36.txt
