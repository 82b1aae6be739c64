cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for w in c3hdr c5 c4ed; do bash tools/pmc_traffic.sh $w > /dev/null 2>&1; done
K=/tmp/keep38; rm -rf $K; mkdir -p $K; cp gpurun_out/traffic_*.json $K/; rm -rf gpurun_out/*; cp $K/* gpurun_out/; ls -la gpurun_out
