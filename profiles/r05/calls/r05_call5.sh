#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c5; mkdir -p $O
MPCVR_NO_FRAME_LANES=1 timeout 600 python tools/ed_probe.py 3840 > $O/probe.jsonl 2>&1; cat $O/probe.jsonl
