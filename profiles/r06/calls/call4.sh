mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/parity_log.jsonl
MPCVR_PARITY_LOG=$PWD/gpurun_out/r06/parity_log.jsonl timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r06/gpu_suite_strict_plain.txt
( time MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=64 timeout 900 python tests/tools/fuzz_strip.py 2500 402 ) > gpurun_out/r06/fuzz_2500_jinc_flags64.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/fuzz_2500_jinc_flags64.txt
( time MPCVR_FUZZ_JINC=1 MPCVR_FUZZ_FLAGS=8 timeout 900 python tests/tools/fuzz_strip.py 2500 401 ) > gpurun_out/r06/fuzz_2500_jinc_flags8.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/fuzz_2500_jinc_flags8.txt
( time MPCVR_FUZZ_SCALERS=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=72 timeout 900 python tests/tools/fuzz_strip.py 2500 403 ) > gpurun_out/r06/fuzz_2500_scalers_unaligned_flags72.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/fuzz_2500_scalers_unaligned_flags72.txt
( time MPCVR_FUZZ_HOST=1 MPCVR_FUZZ_UNALIGNED=1 MPCVR_FUZZ_FLAGS=4 timeout 900 python tests/tools/fuzz_strip.py 2500 404 ) > gpurun_out/r06/fuzz_2500_host_unaligned_flags4.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/fuzz_2500_host_unaligned_flags4.txt
