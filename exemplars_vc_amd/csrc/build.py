"""Build libevc_hip.so for gfx950 with hipcc (in-tree, next to the package).

    python -m exemplars_vc_amd.csrc.build [--force]

Each .hip translation unit is compiled to an object file under csrc/_build/ (in parallel)
and linked into exemplars_vc_amd/libevc_hip.so.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(PKG, "libevc_hip.so")
SOURCES = ["evc_gemm.hip", "evc_gemm2.hip", "evc_aux.hip", "evc_fused.hip", "evc_fused_res.hip", "evc_fused_all.hip", "evc_fused_xy.hip", "evc_wide.hip", "evc_wide64.hip", "evc_gl.hip", "evc_dtw.hip", "evc_api.hip"]
HEADERS = ["evc_internal.h", "evc_fused_common.h", os.path.join("..", "..", "include", "evc.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
EXTRA_FLAGS = {}

def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        jobs.append((s, o, force or _newer(o, [s] + hdrs)))

    def compile_one(job):
        s, o, needed = job
        if not needed:
            return 0, ""
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, capture_output=True, text=True)
        return p.returncode, p.stdout + p.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(compile_one, jobs))
    for (s, _, _), (rc, out) in zip(jobs, results):
        if rc != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if out.strip() and verbose:
            print(out)
    objs = [o for _, o, _ in jobs]
    if force or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("link failed:\n" + p.stdout + p.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
