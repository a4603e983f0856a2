"""Build the gfx950 shared libraries in-tree with hipcc.

``python -m llamole_amd.build`` or ``llamole_amd.build.build()``.  hipcc cross-compiles for gfx950 without a GPU; the built .so files
are git-ignored but travel to the GPU box with the source snapshot.

Two libraries from the same sources:
  * ``libllamole_hip.so``         -- the product: exactly the entry points of include/llamole_hip.h (what a maintainer of the reference binds);
  * ``libllamole_hip_tuning.so``  -- the same translation units compiled with -DLL_TUNING=1, which adds the process-global A/B switches,
                                     micro-benchmarks and probes of include/llamole_hip_tuning.h.  tests/, tools/ and bench.py load this one
                                     (LLAMOLE_TUNING=1; _lib.load()).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libllamole_hip.so")
TUNING_LIB = os.path.join(HERE, "libllamole_hip_tuning.so")
SOURCES = ["gemm.hip", "graphdit.hip", "gin.hip", "llm_ops.hip", "llm_layer.hip", "llm_rows16.hip", "llm_rows64.hip", "llm_sample.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]
JOBS = max(1, min(8, (os.cpu_count() or 2)))


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
    todo = []           # (cmd) compile jobs of both variants
    objs = {0: [], 1: []}
    for tuning in (0, 1):
        for s in SOURCES:
            src = os.path.join(CSRC, s)
            if not os.path.exists(src):
                continue
            obj = os.path.join(CSRC, s.replace(".hip", ".t.o" if tuning else ".o"))
            objs[tuning].append(obj)
            if force or not _newer(obj, [src] + headers):
                todo.append([hipcc] + FLAGS + [f"-DLL_TUNING={tuning}", "-c", src, "-o", obj])
    running = []
    while todo or running:
        while todo and len(running) < JOBS:
            cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        cmd, p = running.pop(0)
        if p.wait() != 0:
            for _, q in running:
                q.kill()
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    for tuning, lib in ((0, LIB), (1, TUNING_LIB)):
        if force or not _newer(lib, objs[tuning]):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs[tuning]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB, "and", TUNING_LIB)
