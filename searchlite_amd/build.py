"""Build the native pieces in-tree.

* lib/libsearchlite_gpu.so — the product: HIP kernels + C ABI for gfx950 (hipcc).
* lib/libslg_corpus.so     — harness tool: synthetic Zipf corpus generator (g++, host only).
* lib/libslg_segfile.so    — host-only decoder of searchlite's segment files (g++).
* lib/libslg_plan.so       — the host planner behind a test C ABI (g++; CPU unit tests).
* lib/libslg_harness.so    — bench harness: native caller threads over the C ABI (bench.py).

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the built
.so files travel to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
GPU_LIB = os.path.join(LIBDIR, "libsearchlite_gpu.so")
CORPUS_LIB = os.path.join(LIBDIR, "libslg_corpus.so")
SEGFILE_LIB = os.path.join(LIBDIR, "libslg_segfile.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off",  # f32 ops rounded one by one, as the Rust reference does
               "-Wall", "-Wno-unused-function"]


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libsearchlite_gpu.so")


SCORE_KREGS = (1, 2, 4, 8, 16)


def build_gpu(force: bool = False, verbose: bool = False, stamps: bool = False, defines=(), tag: str = "") -> str:
    """hipcc every translation unit for gfx950 (score-kernel variants in parallel), then link.
    stamps=True builds the diagnostic library with in-kernel s_memtime stamps (tools/stamps.py);
    defines/tag build an experiment library libsearchlite_gpu_<tag>.so with extra -D flags."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    tag = "stamps" if stamps else tag
    objdir = os.path.join(LIBDIR, f"obj_{tag}" if tag else "obj")
    out_lib = os.path.join(LIBDIR, f"libsearchlite_gpu_{tag}.so") if tag else GPU_LIB
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in ("slg_desc.hpp", "slg_kernels.hpp", "slg_rerank.hpp", "slg_score.hpp", "slg_score_uni.hpp", "slg_score_uni3.hpp", "slg_score_uni4.hpp",
                                           "slg_score_multi.hpp", "slg_plan.hpp")]
    hdrs.append(os.path.join(_HERE, "..", "include", "searchlite_gpu.h"))
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + (["-DSLG_STAMPS"] if stamps else []) \
        + [f"-D{d}" for d in defines]
    kregs = (1,) if stamps else SCORE_KREGS
    jobs = [(os.path.join(CSRC, "slg_api.hip"), os.path.join(objdir, "slg_api.o"), []),
            # the request coalescer: host code over the public ABI
            (os.path.join(CSRC, "slg_coalesce.hip"), os.path.join(objdir, "slg_coalesce.o"), []),
            # the host planner: plain C++ (the same source builds with g++ for the CPU unit tests)
            (os.path.join(CSRC, "slg_plan.cpp"), os.path.join(objdir, "slg_plan.o"), ["-x", "c++"])]
    for kr in kregs:
        jobs.append((os.path.join(CSRC, "slg_score_inst.hip"),
                     os.path.join(objdir, f"slg_score_k{kr}.o"), [f"-DSLG_INST_KREGS={kr}"]))

    def compile_one(job):
        src, obj, extra = job
        if force or _newer(obj, [src] + hdrs):
            cmd = [_hipcc(), *compile_flags, *extra, "-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return True
        return False

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
        rebuilt = list(ex.map(compile_one, jobs))
    objs = [j[1] for j in jobs]
    if force or any(rebuilt) or _newer(out_lib, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_lib, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out_lib


def build_corpus_tool(force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, "tools", "corpus_gen.cpp")
    if force or _newer(CORPUS_LIB, [src]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                               "-o", CORPUS_LIB, src])
    return CORPUS_LIB


def build_segfile(force: bool = False) -> str:
    """Host-only decoder of searchlite's segment files (include/searchlite_segfile.h)."""
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, "slg_segfile.cpp")
    hdr = os.path.join(_HERE, "..", "include", "searchlite_segfile.h")
    if force or _newer(SEGFILE_LIB, [src, hdr]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
                               "-o", SEGFILE_LIB, src])
    return SEGFILE_LIB


PLAN_LIB = os.path.join(LIBDIR, "libslg_plan.so")


def build_plan_lib(force: bool = False, extra_flags=(), out: str | None = None) -> str:
    """The host planner (csrc/slg_plan.cpp: the same source libsearchlite_gpu.so links) behind a
    small C ABI for the CPU unit tests (tests/test_plan.py).  g++, host only."""
    os.makedirs(LIBDIR, exist_ok=True)
    out = out or PLAN_LIB
    srcs = [os.path.join(CSRC, "slg_plan.cpp"), os.path.join(CSRC, "slg_plan_capi.cpp")]
    hdrs = [os.path.join(CSRC, "slg_plan.hpp"), os.path.join(CSRC, "slg_desc.hpp"),
            os.path.join(_HERE, "..", "include", "searchlite_gpu.h")]
    if force or _newer(out, srcs + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", *extra_flags,
                               "-o", out, *srcs])
    return out


HARNESS_LIB = os.path.join(LIBDIR, "libslg_harness.so")


def build_harness(force: bool = False) -> str:
    """bench harness: native caller threads over the C ABI (csrc/tools/host_harness.cpp); links
    libsearchlite_gpu.so.  Not part of the product library."""
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, "tools", "host_harness.cpp")
    hdr = os.path.join(_HERE, "..", "include", "searchlite_gpu.h")
    if force or _newer(HARNESS_LIB, [src, hdr, GPU_LIB]):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                               "-x", "hip", src, "-L" + LIBDIR, "-lsearchlite_gpu", "-Wl,-rpath,$ORIGIN",
                               "-o", HARNESS_LIB])
    return HARNESS_LIB


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_gpu(force, verbose)
    build_harness(force)
    build_corpus_tool(force)
    build_segfile(force)
    build_plan_lib(force)


if __name__ == "__main__":
    build_all(verbose=True)
