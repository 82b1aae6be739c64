"""The C-ABI library loads and exports every symbol include/mpcvr.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mpcvr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcvr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_library_agree(mpcvr):
    from videorenderer_amd import api
    L = api.load_library()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mpcvr.h but not exported by libmpcvr.so"
    # and the python binding table covers the header
    assert sorted(api.EXPORTS) == names
    assert api.load_library().mpcvr_version().decode().startswith("mpcvr-mi355x")


def test_header_compiles_as_c(tmp_path):
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "mpcvr.h"\nint main(void){ mpcvr_settings s; return sizeof(s) == 44 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_create_fails_loudly_without_gpu(mpcvr):
    import torch
    from videorenderer_amd import api
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    ctx = C.c_void_p()
    s = api.default_settings()
    hr = api.load_library().mpcvr_create(C.byref(s), 0, C.byref(ctx))
    assert hr == api.E_FAIL and not ctx.value          # no silent CPU fallback
    with pytest.raises(api.MpcvrError):
        api.VideoProcessor(use_torch_stream=False)


def test_null_context_is_rejected(mpcvr):
    from videorenderer_amd import api
    L = api.load_library()
    assert L.mpcvr_render(None, 0) == api.E_POINTER
    assert L.mpcvr_process(None, None, 0, None, None, 0) == api.E_POINTER
    assert L.mpcvr_destroy(None) == api.E_POINTER
    assert L.mpcvr_settings_default(None) == api.E_POINTER
