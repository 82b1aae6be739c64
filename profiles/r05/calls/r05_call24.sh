#!/bin/bash
# round 5, call 24: the new test, then the traffic / issue-slot passes on the final sources (vp_fused.hip changed: the hashes of call 20 are stale)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -k "tone_mapping_operator_carries" 2>&1 | tail -6 | cut -c1-400
for w in c3hdr c1 hdr4k up1440 down1440 up2160 c5 c4ed jinc1080 dovi4k; do bash tools/pmc_traffic.sh $w > /dev/null 2>&1; done
K=/tmp/keep24; rm -rf $K; mkdir -p $K; cp $O/traffic_*.json $K/; rm -rf $O/*; cp $K/* $O/; cat $O/traffic_c3hdr.json | cut -c1-200
