#!/bin/bash
# round 5, call 20: the traffic / issue-slot passes again, without bench.py's bandwidth probe in the sums, and the lines that carry them
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for w in c3hdr c1 hdr4k up1440 down1440 up2160 c5 c4ed jinc1080 dovi4k; do bash tools/pmc_traffic.sh $w > /dev/null 2>&1; done
K=/tmp/keep20; rm -rf $K; mkdir -p $K; cp $O/traffic_*.json $K/; rm -rf $O/*; cp $K/* $O/; cat $O/traffic_c3hdr.json | cut -c1-300
