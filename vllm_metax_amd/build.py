"""In-tree build of the MI355X hot-path libraries (gfx950 only).

Two artefacts, both written next to this file so they travel with the repo
snapshot to the GPU box (they are git-ignored, not gpurun-ignored):

  libmi355x_hotpath.so  pure C-ABI (include/mi355x_hotpath.h); hipcc, no torch.
  _C.so                 thin torch.library bindings (csrc/torch_bindings.cpp) that
                        register torch.ops._C / _C_cache_ops / _C_cuda_utils and
                        forward raw pointers to the C-ABI.

`python -m vllm_metax_amd.build` builds both; `build_hotpath()` / `build_torch_ops()`
are used by __graft_entry__.build().
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
OBJ_DIR = PKG_DIR / "csrc" / "_obj"
HOTPATH_LIB = PKG_DIR / "libmi355x_hotpath.so"
TORCH_OPS_LIB = PKG_DIR / "_C.so"

ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", shutil.which("hipcc") or "/opt/rocm/bin/hipcc")

HIP_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-fPIC",
    "-std=c++17",
    "-fno-gpu-rdc",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-variable",
    f"-I{INCLUDE}",
] + os.environ.get("MI355X_EXTRA_HIPFLAGS", "").split()   # (kernel ablation experiments)


def _newer(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _run(cmd: list[str]) -> None:
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    if res.stderr.strip():
        sys.stderr.write(res.stderr)


def hip_sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def build_hotpath(force: bool = False, jobs: int | None = None, verbose: bool = True) -> Path:
    """Compile every csrc/*.hip for gfx950 and link libmi355x_hotpath.so."""
    OBJ_DIR.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.cuh")) + sorted(INCLUDE.glob("*.h"))
    todo = []
    objs = []
    for src in hip_sources():
        obj = OBJ_DIR / (src.stem + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            todo.append((src, obj))
    jobs = jobs or min(6, os.cpu_count() or 1)

    def compile_one(pair):
        src, obj = pair
        if verbose:
            print(f"[build] hipcc {src.name}", flush=True)
        _run([HIPCC, *HIP_FLAGS, "-c", str(src), "-o", str(obj)])

    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))
    if force or todo or _newer(HOTPATH_LIB, objs):
        if verbose:
            print(f"[build] link {HOTPATH_LIB.name}", flush=True)
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(HOTPATH_LIB),
              *map(str, objs)])
    return HOTPATH_LIB


def build_torch_ops(force: bool = False, verbose: bool = True) -> Path:
    """Compile csrc/torch_bindings.cpp against the installed torch and link it to
    the C-ABI library (rpath $ORIGIN)."""
    import torch
    from torch.utils import cpp_extension

    src = CSRC / "torch_bindings.cpp"
    deps = [src, INCLUDE / "mi355x_hotpath.h"]
    if not force and not _newer(TORCH_OPS_LIB, deps + [HOTPATH_LIB]):
        return TORCH_OPS_LIB
    if verbose:
        print("[build] torch_bindings.cpp", flush=True)
    inc = [f"-I{p}" for p in cpp_extension.include_paths()]
    torch_lib = Path(torch.__file__).parent / "lib"
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [
        "g++", "-O2", "-fPIC", "-shared", "-std=c++17",
        f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
        "-DTORCH_EXTENSION_NAME=_C", f"-I{INCLUDE}", f"-I{rocm}/include", "-I/usr/include/python3.10", *inc,
        str(src), "-o", str(TORCH_OPS_LIB),
        f"-L{PKG_DIR}", "-lmi355x_hotpath", f"-L{torch_lib}", "-ltorch", "-ltorch_cpu",
        "-lc10", "-ltorch_hip", "-lc10_hip", f"-L{rocm}/lib", "-lamdhip64",
        "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{torch_lib}", "-Wl,--no-as-needed",
    ]
    _run(cmd)
    return TORCH_OPS_LIB


def build_all(force: bool = False) -> None:
    build_hotpath(force=force)
    build_torch_ops(force=force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("ok")
