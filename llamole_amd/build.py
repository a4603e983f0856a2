"""Build the gfx950 shared library (libllamole_hip.so) in-tree with hipcc.

``python -m llamole_amd.build`` or ``llamole_amd.build.build()``.  hipcc cross-compiles for gfx950
without a GPU; the built .so is git-ignored but travels to the GPU box with the source snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libllamole_hip.so")
SOURCES = ["gemm.hip", "graphdit.hip", "gin.hip", "llm_ops.hip", "llm_layer.hip", "llm_rows16.hip", "llm_rows64.hip", "llm_sample.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _newer(a, deps):
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "llamole_hip.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "llamole_hip_tuning.h"))
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not _newer(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or not _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
