mkdir -p gpurun_out/r06
python tools/debug/case1820.py > gpurun_out/r06/case1820_after2.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06/gpu_suite_crmath2.txt
( MPCVR_FUZZ_PLAIN_STATS=1 timeout 600 python tests/tools/fuzz_strip.py 1500 61 ) > gpurun_out/r06/fuzz_stats_default.txt 2>&1
( MPCVR_FUZZ_PLAIN_STATS=1 MPCVR_FUZZ_JINC=1 timeout 600 python tests/tools/fuzz_strip.py 1500 62 ) > gpurun_out/r06/fuzz_stats_jinc.txt 2>&1
( MPCVR_FUZZ_PLAIN_STATS=1 MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 timeout 600 python tests/tools/fuzz_strip.py 1500 63 ) > gpurun_out/r06/fuzz_stats_scalers.txt 2>&1
