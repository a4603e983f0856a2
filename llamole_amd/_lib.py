"""ctypes binding of libllamole_hip.so (the C ABI declared in include/llamole_hip.h).

The product path has no CPU fallback: importing this module on a machine where the shared
library has not been built raises, and every wrapper raises ``RuntimeError`` with the
library's own message when a call returns a negative code.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libllamole_hip.so")
TUNING_LIB_PATH = os.path.join(_HERE, "libllamole_hip_tuning.so")


def tuning_build() -> bool:
    """LLAMOLE_TUNING=1 (tests/conftest.py, tools/, bench.py): this process runs the LL_TUNING=1 build, which adds the A/B switches,
    micro-benchmarks and probes of include/llamole_hip_tuning.h to the same kernels and entry points.  Everything else -- `import
    llamole_amd`, main.py eval / train -- loads the product library, which exports none of them."""
    return os.environ.get("LLAMOLE_TUNING") == "1"

LL_F32, LL_BF16 = 0, 1


class LLDitConfig(C.Structure):
    _fields_ = [("hidden", C.c_int), ("depth", C.c_int), ("heads", C.c_int), ("mlp_hidden", C.c_int),
                ("max_nodes", C.c_int), ("T", C.c_int), ("guide_scale", C.c_float), ("dtype", C.c_int)]


class LLDitTables(C.Structure):
    _fields_ = [("h_x_marg", C.c_void_p), ("h_e_marg", C.c_void_p), ("h_u_xe", C.c_void_p),
                ("h_u_ex", C.c_void_p), ("h_betas", C.c_void_p), ("h_alphas_bar", C.c_void_p)]


class LLGinConfig(C.Structure):
    _fields_ = [("num_layer", C.c_int), ("hidden", C.c_int), ("kind", C.c_int), ("out_dim", C.c_int),
                ("text_dim", C.c_int), ("dtype", C.c_int)]


_P, _I, _I64, _U64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float

# name -> (restype, argtypes); mirrors include/llamole_hip.h one to one (the product library exports exactly these)
SIGNATURES = {
    "ll_version": (_I, []),
    "ll_last_error": (C.c_char_p, []),
    "ll_linear": (_I, [_I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ll_linear_splitk_bf16": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "ll_dit_param_count": (_I, [C.POINTER(LLDitConfig)]),
    "ll_dit_param_info": (_I, [C.POINTER(LLDitConfig), _I, C.c_char_p, _I, C.POINTER(_I64), C.POINTER(_I64)]),
    "ll_dit_arena_elems": (_I64, [C.POINTER(LLDitConfig)]),
    "ll_dit_create": (_I, [C.POINTER(LLDitConfig), C.POINTER(LLDitTables), _P, C.POINTER(_P)]),
    "ll_dit_destroy": (_I, [_P]),
    "ll_dit_begin": (_I, [_P, _I, _P, _P, _P, _P]),
    "ll_dit_init_state": (_I, [_P, _P, _P, _U64, _P]),
    "ll_dit_step": (_I, [_P, _I, _P, _P, _U64, _P]),
    "ll_dit_run": (_I, [_P, _U64, _I, _P]),
    "ll_dit_set_state": (_I, [_P, _P, _P, _P]),
    "ll_dit_get_state": (_I, [_P, _P, _P, _P]),
    "ll_dit_denoise": (_I, [_P, _I, _P, _P, _P, _I, _P]),
    "ll_dit_step_probs": (_I, [_P, _I, _P, _P, _P]),
    "ll_dit_denoise_rows": (_I, [_P, _P, _P, _P, _P]),
    "ll_dit_cvec": (_I, [_P, _I, _P, _P]),
    "ll_dit_last_run_ms": (_I, [_P, C.POINTER(_F), C.POINTER(_I)]),
    "ll_gin_param_count": (_I, [C.POINTER(LLGinConfig)]),
    "ll_gin_param_info": (_I, [C.POINTER(LLGinConfig), _I, C.c_char_p, _I, C.POINTER(_I64), C.POINTER(_I64)]),
    "ll_gin_arena_elems": (_I64, [C.POINTER(LLGinConfig)]),
    "ll_gin_create": (_I, [C.POINTER(LLGinConfig), _P, C.POINTER(_P)]),
    "ll_gin_destroy": (_I, [_P]),
    "ll_graph_csr": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ll_gin_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "ll_softmax_topk": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "ll_cost_mlp": (_I, [_P, _P, _I, _P, _P]),
    "ll_rmsnorm_bf16": (_I, [_P, _P, _P, _I, _I, _F, _P]),
    "ll_rope_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), _P]),
    "ll_silu_mul_bf16": (_I, [_P, _P, _P, _I, _I, _I64, _P]),
    "ll_kv_append_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(_I64), C.POINTER(_I64), _P]),
    "ll_decode_attn_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, C.POINTER(_I64), C.POINTER(_I64), _P]),
    "ll_gemv_fused_bf16": (_I, [_P, _I, _P, _I, _P, _P, _F, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "ll_decode_attn_rope_bf16": (_I, [_P, _I64, _P, _P, _I64, _P, _P, _P, _P, _I64, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ll_decode_prologue": (_I, [_P, _P, _F, _P, _I64, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ll_suffix_prologue": (_I, [_P, _P, _F, _P, _I64, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ll_suffix_attn_rope_bf16": (_I, [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "ll_dit_set_overlap": (_I, [_P, _I]),
    "ll_dit_set_option": (_I, [_P, _I, _I]),
    "ll_linear_rows16_bf16": (_I, [_P, _I, _P, _I, _P, _P, _F, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "ll_rows64_packed_elems": (_I64, [_I, _I]),
    "ll_rows64_pack_bf16": (_I, [_P, _I, _I, _I, _P, _P]),
    "ll_linear_rows64_bf16": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _F, _P, _P, _I, _P, _P, _I64, _P]),
    "ll_rows64_ssq_chunks": (_I, [_I]),
    "ll_rows64_prenorm_bf16": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _P]),
    "ll_linear_rows64_workspace_bytes": (_I64, [_I, _I]),
    "ll_gin_forward_train": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "ll_gin_backward_c": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "ll_sample_token_bf16": (_I, [_P, _I64, _I, _I, _F, _F, _I, _P, _P, _I, _I64, _P, _P, _P, _I64, _I, _P, _P, _P, _I, _P, _P]),
    "ll_sample_token_topk_bf16": (_I, [_P, _I64, _I, _I, _F, _F, _I, _I, _P, _P, _I, _I64, _P, _P, _P, _I64, _I, _P, _P, _P, _I, _P, _P]),
    "ll_sample_workspace_bytes": (_I64, [_I]),
    "ll_sample_token_topk_ws_bf16": (_I, [_P, _I64, _I, _I, _F, _F, _I, _I, _P, _P, _I, _I64, _P, _P, _P, _I64, _I, _P, _P, _P, _I, _P, _P, _I64, _P]),
}

# include/llamole_hip_tuning.h: exported by libllamole_hip_tuning.so only (the LL_TUNING=1 build of the same sources)
TUNING_SIGNATURES = {
    "ll_host_launch_probe": (_I, [_I, _I, C.POINTER(_F)]),
    "ll_gemm_bench": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F)]),
    "ll_linear_cfg": (_I, [_I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ll_launch_bench": (_I, [_I, _I, _I, C.POINTER(_F)]),
    "ll_launch_bench_set_buffers": (_I, [_P, _P]),
    "ll_set_topk_single": (_I, [_I]),
    "ll_gemv_fused_bench": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F)]),
    "ll_set_gemv_nt": (_I, [_I]),
    "ll_set_gemv_stage": (_I, [_I]),
    "ll_set_m64_waves": (_I, [_I]),
    "ll_set_m128_panel": (_I, [_I]),
    "ll_set_m64_packed": (_I, [_I]),
    "ll_set_gemm_krot": (_I, [_I]),
    "ll_set_lnmod_multiwave": (_I, [_I]),
    "ll_set_stage_mod": (_I, [_I]),
    "ll_set_attn_waves": (_I, [_I]),
    "ll_debug_check_guards": (_I, []),
    "ll_debug_guard_selftest": (_I, []),
    "ll_philox_probe": (_I, [_P, _P, _I, _P]),
    "ll_dit_noise_probe": (_I, [_U64, _I, _I, _I, _P, _P, _P]),
    "ll_dit_class_probe": (_I, [_P, _I]),
    "ll_dit_class_probe_read": (_I, [_P, C.POINTER(_F), C.POINTER(_I)]),
    "ll_set_rows16_geometry": (_I, [_I, _I, _I]),
    "ll_rows16_bench": (_I, [_I, _I, _I, _I, _I, _I, _I, C.POINTER(_F)]),
    "ll_set_rows64_ksplit": (_I, [_I]),
    "ll_rows64_bench": (_I, [_I, _I, _I, _I, _I, _I, _I, C.POINTER(_F)]),
}

_lib = None


def load():
    """Load the shared library (once): the product build, or the tuning build under LLAMOLE_TUNING=1.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    tuning = tuning_build()
    path = TUNING_LIB_PATH if tuning else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m llamole_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the graph hot path.")
    # PyTorch-ROCm ships its own HIP runtime; the one that is loaded FIRST in a process is the one that can open the device
    # (build() followed by smoke() in one process loaded this library before torch had touched HIP, and every hipMalloc of the
    # library then failed with "no ROCm-capable device is detected").  Torch provides the device memory and streams this library
    # is handed, so its runtime goes first; libamdhip64 of this library then resolves to the copy already in the process.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    lib = C.CDLL(path)
    table = dict(SIGNATURES, **TUNING_SIGNATURES) if tuning else SIGNATURES
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._ll_path = path
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ll_last_error()
        raise RuntimeError(f"{what or 'llamole_hip'} failed (code {rc}): {msg.decode() if msg else '?'}")


def param_table(kind: str, cfg) -> list:
    """[(name, numel, offset)] of the weight arena for a 'dit' or 'gin' config."""
    lib = load()
    n = getattr(lib, f"ll_{kind}_param_count")(C.byref(cfg))
    if n < 0:
        check(n, f"ll_{kind}_param_count")
    out = []
    buf = C.create_string_buffer(256)
    ne, off = _I64(), _I64()
    for i in range(n):
        check(getattr(lib, f"ll_{kind}_param_info")(C.byref(cfg), i, buf, 256, C.byref(ne), C.byref(off)))
        out.append((buf.value.decode(), ne.value, off.value))
    return out


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    """Device pointer of a (contiguous) torch tensor, or NULL."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    return C.c_void_p(t.data_ptr())
