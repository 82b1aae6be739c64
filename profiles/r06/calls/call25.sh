mkdir -p gpurun_out/r06
for wlk in c3hdr c4ed up1440; do echo "== $wlk"; timeout 300 python tools/debug/batch_overlap.py $wlk 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06/batch_overlap.txt 2>&1; cat gpurun_out/r06/batch_overlap.txt
