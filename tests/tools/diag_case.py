#!/usr/bin/env python3
"""tests/tools/diag_case.py '<case dict>' — one case through the default planner, the plain kernels and the CPU oracle: where do they differ?"""
import sys, os, ast
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
from tests.golden.cases import case_frame, oracle_params
from tests.test_parity_gpu import run_product, BG, _codes10
from oracle import oracle

c = ast.literal_eval(sys.argv[1])
flags = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, api.FLAG_NO_FUSED]
outs = {}
for f in flags:
    out, info = run_product(api, torch, c, extra_flags=f)
    outs[f] = out
    print(f"flags {f}: {info}")
want = oracle.process(oracle_params(oracle, c), *case_frame(c), dst=np.full((outs[flags[0]].shape[0], outs[flags[0]].shape[1], 4), BG, dtype=np.uint8))
codes = _codes10 if c.get("output_format", 0) == 1 else (lambda a: a[..., :3].astype(np.int16))
def report(name, a, b):
    d = np.abs(codes(a) - codes(b))
    ys, xs, cs = np.nonzero(d)
    print(f"{name}: identical {float((d == 0).mean()):.5f} max {int(d.max())}  rows mod 3 {np.bincount(ys % 3, minlength=3).tolist()} cols mod 3 {np.bincount(xs % 3, minlength=3).tolist()} channels {np.bincount(cs, minlength=3).tolist()}")
    if len(ys):
        print("   first:", [(int(y), int(x), int(ch), int(codes(a)[y, x, ch]), int(codes(b)[y, x, ch])) for y, x, ch in list(zip(ys, xs, cs))[:6]])
for f in flags:
    report(f"flags {f} vs oracle", outs[f], want)
if len(flags) > 1:
    report(f"flags {flags[0]} vs flags {flags[1]}", outs[flags[0]], outs[flags[1]])
