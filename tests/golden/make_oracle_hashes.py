"""Freeze sha256 of the oracle's output for every named case (tests/golden/cases.py) into
oracle_hashes.json.  Re-run ONLY after a reviewed, intentional change of the oracle.

    python tests/golden/make_oracle_hashes.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.golden.cases import GOLDEN_CASES, run_case  # noqa: E402


def main():
    O.build(ref=False)
    out = {}
    for name in GOLDEN_CASES:
        out[name] = hashlib.sha256(run_case(O, name).tobytes()).hexdigest()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"froze {len(out)} cases")


if __name__ == "__main__":
    main()
