"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports every symbol that
include/llamole_hip.h declares; layout queries (pure host code) agree with the reference state-dict
key tables.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from llamole_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from llamole_amd.build import build
    build(verbose=False)
    return _lib.load()          # tests/conftest.py selects the LL_TUNING=1 build


def _declared_in(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ll_[a-z0-9_]+)\s*\(", src)))


def _declared_functions():
    return sorted(set(_declared_in("llamole_hip.h")) | set(_declared_in("llamole_hip_tuning.h")))


def test_tuning_hooks_are_fenced_off_the_product_header():
    """process-global switches, micro-benchmarks and single-kernel test entry points live in their own header; what a
    maintainer binds (llamole_hip.h) carries none of them, and the product wrappers call none of them."""
    product, tuning = _declared_in("llamole_hip.h"), _declared_in("llamole_hip_tuning.h")
    assert not set(product) & set(tuning)
    assert not [n for n in product if n.startswith(("ll_set_", "ll_debug_")) or n.endswith(("_bench", "_probe")) or n in ("ll_linear_cfg",)]
    assert all(n.startswith(("ll_set_", "ll_debug_")) or "bench" in n or "probe" in n or n in ("ll_linear_cfg",) for n in tuning), tuning
    pkg = os.path.join(ROOT, "llamole_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f not in ("_lib.py", "benchlib.py"):      # benchlib.py = bench.py's measurement helpers (round-5 split), not a product path
            text = open(os.path.join(pkg, f)).read()
            used = [n for n in tuning if re.search(r"\b" + n + r"\b", text)]
            assert not used, f"{f} (product path) calls tuning hooks {used}"


def test_every_declared_symbol_is_exported(lib):
    """The tuning build exports both headers; the PRODUCT library exports exactly include/llamole_hip.h and none of the switches, benchmarks
    or probes -- checked on the file itself, whatever build this process runs."""
    import ctypes
    product, tuning = _declared_in("llamole_hip.h"), _declared_in("llamole_hip_tuning.h")
    assert len(product) >= 25
    for n in product + tuning:
        assert hasattr(lib, n), f"{n} declared but not exported by the tuning build"
    assert sorted(_lib.SIGNATURES) == product, "ctypes signature table out of sync with include/llamole_hip.h"
    assert sorted(_lib.TUNING_SIGNATURES) == tuning, "ctypes signature table out of sync with include/llamole_hip_tuning.h"
    prod = ctypes.CDLL(_lib.LIB_PATH)
    for n in product:
        assert hasattr(prod, n), f"{n} declared in llamole_hip.h but not exported by the product library"
    leaked = [n for n in tuning if hasattr(prod, n)]
    assert not leaked, f"the product library exports tuning hooks: {leaked}"
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\b(ll_[a-z0-9_]+)$", syms, flags=re.M)))
    assert exported == product, (set(exported) ^ set(product))


def test_product_library_is_what_a_plain_import_loads(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("LLAMOLE_TUNING", raising=False)
    plain = _lib.load()
    assert plain._ll_path == _lib.LIB_PATH and not hasattr(plain, "ll_set_rows64_ksplit") and hasattr(plain, "ll_linear_rows64_bf16")
    monkeypatch.setattr(_lib, "_lib", None)


def test_version_and_error_string(lib):
    assert lib.ll_version() >= 100
    odd = _lib.LLDitConfig(100, 2, 4, 250, 32, 50, 2.0, 0)      # any width the reference constructs is a valid config (padded inside the engine)
    assert lib.ll_dit_param_count(C.byref(odd)) > 0
    bad = _lib.LLDitConfig(4096, 2, 32, 400, 32, 50, 2.0, 0)    # the documented bounds: hidden <= 2048, head_dim <= 128, max_nodes <= 128
    assert lib.ll_dit_param_count(C.byref(bad)) < 0
    assert b"hidden" in lib.ll_last_error()
    with pytest.raises(RuntimeError, match="hidden"):
        _lib.check(lib.ll_dit_param_info(C.byref(bad), 0, None, 0, None, None))
    assert lib.ll_dit_param_count(C.byref(_lib.LLDitConfig(512, 2, 2, 2048, 32, 50, 2.0, 0))) < 0 and b"head_dim" in lib.ll_last_error()
    assert lib.ll_dit_param_count(C.byref(_lib.LLDitConfig(128, 2, 4, 512, 128, 50, 2.0, 0))) > 0
    assert lib.ll_dit_param_count(C.byref(_lib.LLDitConfig(128, 2, 4, 512, 129, 50, 2.0, 0))) < 0 and b"max_nodes" in lib.ll_last_error()


@pytest.mark.parametrize("H,L,heads,N,ratio", [(128, 2, 4, 32, 4.0), (256, 2, 4, 50, 4.0), (1024, 28, 16, 32, 4.0),
                                               (1152, 3, 16, 38, 4.0), (600, 3, 8, 9, 2.0), (300, 2, 4, 17, 2.5), (48, 1, 3, 6, 4.0)])
def test_dit_layout_matches_reference_state_dict(lib, H, L, heads, N, ratio):
    """The public arena layout is the checkpoint's own at every width (the engine's zero padding is internal)."""
    cfg = _lib.LLDitConfig(H, L, heads, int(H * ratio), N, 50, 2.0, 1)
    table = _lib.param_table("dit", cfg)
    shapes = synth.dit_weight_shapes(synth.make_dit_config(H, L, heads, mlp_ratio=ratio), N)
    assert [t[0] for t in table] == list(shapes.keys())
    end = 0
    for (name, numel, off), shp in zip(table, shapes.values()):
        n = 1
        for d in shp:
            n *= d
        assert numel == n, name
        assert off >= end and off % 64 == 0
        end = off + numel
    assert lib.ll_dit_arena_elems(C.byref(cfg)) >= end
    if (H, L) == (1024, 28):   # parameter count of the reference-default denoiser
        assert sum(t[1] for t in table) == 573_662_736


@pytest.mark.parametrize("kind", ["encoder", "predictor"])
def test_gin_layout_matches_reference_state_dict(lib, kind):
    L, H, out_dim = 3, 64, 1000
    cfg = _lib.LLGinConfig(L, H, 0 if kind == "encoder" else 1, out_dim, 768, 0)
    table = {t[0]: t[1] for t in _lib.param_table("gin", cfg)}
    shapes = dict(synth.gin_weight_shapes(L, H, kind, out_dim))
    if kind == "encoder":
        shapes.update({"proj." + k: v for k, v in synth.proj_weight_shapes(H).items()})
    assert sorted(table) == sorted(shapes)
    for k, shp in shapes.items():
        n = 1
        for d in shp:
            n *= d
        assert table[k] == n, k


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "TUNING_LIB_PATH", str(tmp_path / "nope_tuning.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "llamole_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
