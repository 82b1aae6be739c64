#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q > $O/suite.txt 2>&1; grep -E "passed|failed|Error" $O/suite.txt | tail -8
python tools/bench_general.py "Dolby" 2>/dev/null | grep "^{" > $O/bench_general_dovi.jsonl
cat $O/bench_general_dovi.jsonl | cut -c1-200
python tests/tools/diag_dovi_tiers.py 2>/dev/null | grep "^{" > $O/dovi_tiers3.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/dovi_tiers3.jsonl"):
    r = json.loads(l); print(r["lib"], r["case"], r["flags"], r["beyond_1lsb"], r["max"], round(r["identical"], 5), r["path"][:40])
PY
