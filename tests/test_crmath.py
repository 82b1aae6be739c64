"""The shader transcendentals as DEFINED functions (round 6): HLSL's pow(x, y) = exp2(y * log2(x)) with every step the correctly rounded
fp32 function.  Three statements of the one definition exist — the oracle's (oracle/crmath.h, C), the reference-shader executor's
(oracle/ref_hlsl/hlsl_shim.h includes the same header) and the product's plain tier (videorenderer_amd/csrc/vp_crmath.h, fp64 on the GPU,
also compiled for the host) — and they have to agree on every bit; against extended precision the definition has to BE the correctly
rounded function.  CPU part here; the device part is tests/test_parity_gpu.py::test_defined_transcendentals_device_equals_host."""
import ctypes as C

import numpy as np
import pytest


def _args(seed=7, n=400_000):
    """Arguments that matter on this path: every binade of fp32, the neighbourhood of 1, [0, 1] (PQ / HLG signals), subnormals, powers of 2."""
    rng = np.random.default_rng(seed)
    pos = rng.integers(1, 0x7f800000, n, dtype=np.uint32).view(np.float32)
    x = np.concatenate([pos, (1 + rng.uniform(-0.3, 0.42, n)).astype(np.float32), rng.uniform(0, 1, n).astype(np.float32),
                        np.arange(1, 4096, dtype=np.uint32).view(np.float32), np.float32(2.0) ** np.arange(-149, 128).astype(np.float32)])
    t = np.concatenate([rng.uniform(-152, 129, n), rng.uniform(-1, 1, n), rng.uniform(-40, 4, n), np.arange(-152, 130, 0.5)]).astype(np.float32)
    return x.astype(np.float32), t


def test_definition_is_the_correctly_rounded_function(oracle):
    """log2f / exp2f / expf / sinf / cosf of oracle/crmath.h against numpy's 80-bit long double rounded once to fp32: identical on every argument
    (2 M points; the double evaluation is accurate to ~4e-16, so a misrounding is expected once per ~1e8 arguments)."""
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("no 80-bit long double on this host")
    x, t = _args()
    with np.errstate(over="ignore"):
        assert np.array_equal(oracle.eval_transcendental("log2", x), np.log2(x.astype(np.longdouble)).astype(np.float32))
        assert np.array_equal(oracle.eval_transcendental("exp2", t), np.exp2(t.astype(np.longdouble)).astype(np.float32))
        e = t[(t > -104) & (t < 89)]
        assert np.array_equal(oracle.eval_transcendental("exp", e), np.exp(e.astype(np.longdouble)).astype(np.float32))
        # sin / cos: the arguments of the windowed sinc / jinc weights are below 10; the three-piece pi/2 holds to |x| ~ 1e6
        rng = np.random.default_rng(17)
        a = np.concatenate([rng.uniform(-12, 12, 600_000), rng.uniform(-0.01, 0.01, 100_000), rng.uniform(-1e5, 1e5, 100_000),
                            np.arange(-40, 41) * (np.pi / 2), np.arange(-4000, 4001) * np.pi]).astype(np.float32)
        assert np.array_equal(oracle.eval_transcendental("sin", a), np.sin(a.astype(np.longdouble)).astype(np.float32))
        assert np.array_equal(oracle.eval_transcendental("cos", a), np.cos(a.astype(np.longdouble)).astype(np.float32))


def test_special_values(oracle):
    f = np.float32
    with np.errstate(all="ignore"):
        l = oracle.eval_transcendental("log2", np.array([0.0, -0.0, -1.0, np.inf, np.nan, 1.0, 8.0, 1e-45], f))
    assert l[0] == -np.inf and l[1] == -np.inf and np.isnan(l[2]) and l[3] == np.inf and np.isnan(l[4]) and l[5] == 0 and l[6] == 3 and l[7] == -149
    e = oracle.eval_transcendental("exp2", np.array([-np.inf, np.inf, np.nan, 0.0, -149.0, -150.0, -151.0, 128.0, 127.0, -126.5, 1e9, -1e9], f))
    assert e[0] == 0 and e[1] == np.inf and np.isnan(e[2]) and e[3] == 1 and e[4] == f(1e-45) and e[5] == 0 and e[6] == 0 and e[7] == np.inf
    assert e[8] == f(2.0) ** f(127) and e[10] == np.inf and e[11] == 0
    assert e[9] == np.exp2(np.longdouble(-126.5)).astype(f)            # a subnormal result: gradual underflow, rounded once
    # pow as the shader has it: pow(0, y > 0) = exp2(y * -inf) = 0; pow(x < 0, y) = NaN (the callers saturate first)
    p = oracle.eval_transcendental("pow", np.array([0.0, 1.0, 4.0, -1.0, 0.25], f), np.array([1 / 2.2, 5.0, 0.5, 2.0, 0.5], f))
    assert p[0] == 0 and p[1] == 1 and p[2] == 2 and np.isnan(p[3]) and p[4] == 0.5


def test_product_statement_equals_the_oracles_on_the_host(oracle, mpcvr):
    """videorenderer_amd/csrc/vp_crmath.h compiled for the host (mpcvr_eval_transcendental_host) == oracle/crmath.h, bit for bit, NaNs included."""
    from videorenderer_amd import api
    L = api.load_library()
    x, t = _args(seed=11)
    rng = np.random.default_rng(3)
    xs = np.concatenate([x, np.array([0.0, -0.0, -1.0, np.inf, -np.inf, np.nan], np.float32)])
    ts = np.concatenate([t, np.array([np.inf, -np.inf, np.nan, 1e9, -1e9], np.float32)])
    ys = rng.choice(np.array([1 / 2.2, 2.2, 0.2, 2610 / 16384, 2523 / 32, 32 / 2523, 16384 / 2610, 1.961, 0.1], np.float32), xs.size)
    sc = np.concatenate([rng.uniform(-12, 12, 300_000), rng.uniform(-1e5, 1e5, 50_000), np.arange(-400, 401) * (np.pi / 2), [0.0, -0.0, np.inf, -np.inf, np.nan]]).astype(np.float32)
    for fn, name, a, b in ((0, "log2", xs, None), (1, "exp2", ts, None), (2, "exp", ts, None), (3, "pow", xs, ys), (4, "sin", sc, None), (5, "cos", sc, None)):
        out = np.empty_like(a)
        yy = b if b is not None else a
        assert L.mpcvr_eval_transcendental_host(fn, a.ctypes.data_as(C.c_void_p), yy.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.size) == 0
        with np.errstate(all="ignore"):
            want = oracle.eval_transcendental(name, a, b)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), f"{name}: {int((out.view(np.uint32) != want.view(np.uint32)).sum())} of {a.size} differ"


def test_pow_is_the_three_rounded_steps(oracle):
    """pow(x, y) = exp2f(fl(y * log2f(x))) — d3dcompiler's lowering with each step rounded to fp32 — not a correctly rounded x^y."""
    rng = np.random.default_rng(5)
    x = rng.uniform(1e-6, 1.0, 200_000).astype(np.float32)
    y = rng.choice(np.array([1 / 2.2, 16384 / 2610, 32 / 2523], np.float32), x.size)
    l = oracle.eval_transcendental("log2", x)
    assert np.array_equal(oracle.eval_transcendental("pow", x, y), oracle.eval_transcendental("exp2", (y * l).astype(np.float32)))
