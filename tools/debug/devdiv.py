"""fp32 division and sqrt on the device against numpy (IEEE, correctly rounded): the plain tier's bit-exactness rests on them too."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videorenderer_amd import api
L = api.load_library()
rng = np.random.default_rng(3)
n = 8_000_000
def rnd(n):
    return np.concatenate([rng.uniform(0, 1, n // 2), np.exp(rng.uniform(-40, 10, n // 4)), rng.integers(1, 0x7f000000, n // 4, dtype=np.uint32).view(np.float32)]).astype(np.float32)
x, y = rnd(n), rnd(n)
for fn, name, want in ((6, "div", None), (7, "sqrt", None)):
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = torch.empty_like(dx)
    assert L.mpcvr_eval_transcendental(fn, C.c_void_p(dx.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(out.data_ptr()), x.size, None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    with np.errstate(all="ignore"):
        ref = (x / y) if fn == 6 else np.sqrt(x)
    bad = got.view(np.uint32) != ref.view(np.uint32)
    print(name, "device != IEEE:", int(bad.sum()), "of", x.size)
    for i in np.nonzero(bad)[0][:8]:
        print("   x", repr(x[i]), "y", repr(y[i]), "device", repr(got[i]), "ieee", repr(ref[i]))
