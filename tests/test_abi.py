"""C-ABI boundary checks that need no GPU: the library loads, exports every symbol the header
declares, and the product path fails loudly (no CPU fallback) when it cannot run."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt_api):
    declared = _declared_symbols("rt_hip.h")
    assert set(declared) >= set(rt_api.ABI_SYMBOLS)
    lib = rt_api.load()
    for name in declared:
        assert hasattr(lib, name), f"librt_hip.so does not export {name}"
    assert "gfx950" in rt_api.version()


def test_library_is_in_tree_and_has_gfx950_code_object(rt_api):
    assert os.path.dirname(rt_api.LIB_PATH) == os.path.join(ROOT, "gpu_raytracer_amd")
    blob = open(rt_api.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_render_reference" in blob


def test_product_does_not_import_or_link_the_oracle(rt_api):
    pkg = os.path.join(ROOT, "gpu_raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
                assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', src), f
    assert b"liboracle" not in open(rt_api.LIB_PATH, "rb").read()


def test_missing_extension_fails_loudly(monkeypatch):
    from gpu_raytracer_amd import api
    monkeypatch.setattr(api, "_lib", None)
    monkeypatch.setattr(api, "LIB_PATH", "/nonexistent/librt_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        api.load()


def test_no_device_is_an_error_not_a_fallback(rt_api):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt_api.RtError) as e:
        rt_api.Context()
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_null_context_calls_return_bad_arg(rt_api):
    lib = rt_api.load()
    null = C.c_void_p(0)
    assert lib.rt_render(null, null) == -1
    assert lib.rt_dispatch_tile(null, null) == -1
    assert lib.rt_get_stats(null, null) == -1
    assert lib.rt_read_rgb32f(null, null, C.c_size_t(0)) == -1
    h = C.c_void_p(0)
    assert lib.rt_create(C.byref(h), null, C.c_int(0)) == -1
    lib.rt_destroy(null)  # no-op


def test_python_flag_values_are_the_headers(rt_api):
    """gpu_raytracer_amd/api.py restates the RT_FLAG_* / RT_MODE_* constants of include/rt_hip.h: they must agree (a flag that drifts
    silently selects another code path), and the debug entry points the tests and bench.py call must be exported."""
    import re
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    defines = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(RT_(?:FLAG|MODE)_[A-Z0-9_]+)\s+(\d+)u", header)}
    assert defines["RT_FLAG_NO_SHADOW_GRID"] == 16
    for name, value in defines.items():
        py = name[3:]  # RT_FLAG_X -> FLAG_X
        if hasattr(rt_api, py):
            assert getattr(rt_api, py) == value, name
    for flag in ("FLAG_COUNTERS", "FLAG_NO_SHADOWS", "FLAG_KERNEL_V1", "FLAG_KERNEL_SM", "FLAG_NO_SHADOW_GRID", "FLAG_KERNEL_PIPELINE", "FLAG_NO_BEAMS", "FLAG_STAGE_TIMES"):
        assert "RT_" + flag in defines and hasattr(rt_api, flag), flag
    lib = rt_api.load()
    for sym in ("rt_debug_shadow_grid", "rt_debug_counters", "rt_debug_check_bvh", "rt_debug_stage_times", "rt_debug_beams"):
        assert hasattr(lib, sym), sym
