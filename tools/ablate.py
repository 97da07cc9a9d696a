"""Diagnostic: time the uniform scoring kernel (config 2) with phases compiled out.
Build first (CPU container): python tools/ablate.py build; then on the GPU box: python tools/ablate.py
Bits: 1 P3b, 2 P4, 4 atomics->plain stores, 8 P3a, 16 P2, 32 whole accumulate,
64 collision-free bitmap words (lane-based).  ABL=1,2,... selects the variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searchlite_amd import build
VARIANTS = [int(x) for x in os.environ.get('ABL', '1,2,4,8,16,3,11,27,31,32').split(',')]
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for v in VARIANTS:
        print(v, build.build_gpu(ablate=v))
    sys.exit(0)
for v in [0] + VARIANTS:
    lib = build.GPU_LIB if v == 0 else os.path.join(build.LIBDIR, f"libsearchlite_gpu_abl{v}.so")
    code = f"""
import sys, os, numpy as np
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from searchlite_amd import build, corpus
build.GPU_LIB = {lib!r}
from searchlite_amd import searcher
seg = corpus.zipf_segment(1_000_000, 1 << 18, seed=42)
offs, terms, w = corpus.zipf_queries(1024, 3, seed=7, vocab=1 << 18)
ix = searcher.GpuIndex([seg]); b = ix.prepare(offs, terms, w, 11)
for _ in range(5): b.run()
b.sync(); ix.profile(True); ix.profile_read()
for _ in range(20): b.run()
b.sync(); n, ms = ix.profile_read()
print("abl", {v}, "kernel_ms", round(ms / n, 4), flush=True)
"""
    import subprocess
    subprocess.run([sys.executable, "-c", code], check=False)
