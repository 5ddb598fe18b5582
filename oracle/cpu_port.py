"""ctypes wrapper of oracle/cpu_port.c (TEST INFRASTRUCTURE ONLY — see the C file's header).

`build()` compiles it with gcc (oracle/Makefile); the functions take torch CPU tensors.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "libcpu_port.so"
_lib = None


def build() -> Path:
    src = HERE / "cpu_port.c"
    if not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
    return LIB


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(LIB))
        _lib.cpu_port_num_threads.restype = ctypes.c_int
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def num_threads() -> int:
    return lib().cpu_port_num_threads()


def set_num_threads(n: int) -> None:
    lib().cpu_port_set_num_threads(int(n))


def rms_norm(x, w, eps):
    out = torch.empty_like(x)
    lib().cpu_rms_norm_bf16(_p(out), _p(x), _p(w), ctypes.c_float(eps), x.shape[0], x.shape[1])
    return out


def fused_add_rms_norm(x, res, w, eps):
    lib().cpu_fused_add_rms_norm_bf16(_p(x), _p(res), _p(w), ctypes.c_float(eps), x.shape[0], x.shape[1])


def rotary_neox(pos, q, k, cache, heads, kv_heads, head_size):
    lib().cpu_rotary_neox_bf16(_p(pos), _p(q), _p(k), _p(cache), pos.numel(), cache.shape[1],
                               ctypes.c_int64(q.stride(0)), ctypes.c_int64(k.stride(0)), heads,
                               kv_heads, head_size)


def silu_and_mul(x):
    d = x.shape[1] // 2
    out = torch.empty(x.shape[0], d, dtype=x.dtype)
    lib().cpu_silu_and_mul_bf16(_p(out), _p(x), x.shape[0], d)
    return out


def reshape_and_cache(key, value, kc, vc, slots):
    lib().cpu_reshape_and_cache_bf16(_p(key), _p(value), _p(kc), _p(vc), _p(slots), slots.numel(),
                                     ctypes.c_int64(key.stride(0)), ctypes.c_int64(value.stride(0)),
                                     key.shape[1], key.shape[2], kc.shape[3])


def paged_attention_v1(q, kc, vc, kv_heads, scale, bt, sl):
    out = torch.empty_like(q)
    lib().cpu_paged_attention_v1_bf16(_p(out), _p(q), _p(kc), _p(vc), q.shape[0], q.shape[1], kv_heads,
                                      q.shape[2], kc.shape[3], ctypes.c_float(scale), _p(bt), _p(sl),
                                      bt.shape[1], ctypes.c_int64(q.stride(0)),
                                      ctypes.c_int64(kc.stride(0)), ctypes.c_int64(kc.stride(1)))
    return out


def w4a16_gemm(x, qw_mem, scales, qz, zero_mode, group):
    """qw_mem: shuffled words viewed as [K/8, N] (int32)."""
    m, k = x.shape
    n = qw_mem.shape[1]
    out = torch.empty(m, n, dtype=x.dtype)
    lib().cpu_w4a16_gemm_bf16(_p(out), _p(x), _p(qw_mem), _p(scales), _p(qz), zero_mode, m, n, k, group,
                              ctypes.c_int64(x.stride(0)))
    return out


def gemm_bf16(x, w):
    m, k = x.shape
    n = w.shape[1]
    out = torch.empty(m, n, dtype=torch.float32)
    lib().cpu_gemm_bf16(_p(out), _p(x), _p(w), m, n, k)
    return out
