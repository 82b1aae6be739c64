#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/prof.sh jinc1080 --workload jinc1080 > gpurun_out/prof_jinc1080.log 2>&1
python tools/prof_summary.py gpurun_out/prof_jinc1080 jinc > gpurun_out/prof_jinc1080_quad_summary.txt 2>&1
python tools/prof_summary.py gpurun_out/prof_jinc1080 convert > gpurun_out/prof_jinc1080_convert_summary.txt 2>&1
bash tools/prof.sh dovi4k --workload dovi4k > gpurun_out/prof_dovi4k.log 2>&1
python tools/prof_summary.py gpurun_out/prof_dovi4k convert > gpurun_out/prof_dovi4k_summary.txt 2>&1
cat gpurun_out/prof_jinc1080_quad_summary.txt gpurun_out/prof_jinc1080_convert_summary.txt gpurun_out/prof_dovi4k_summary.txt | cut -c1-200
tail -2 gpurun_out/prof_jinc1080.log | cut -c1-300; tail -2 gpurun_out/prof_dovi4k.log | cut -c1-300
