for lib in libsearchlite_gpu.so libsearchlite_gpu_candall.so; do for inf in 1 2; do
python - <<PY
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
from searchlite_amd import build
build.GPU_LIB = os.path.join(build.LIBDIR, "$lib")
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--inflight", "$inf"]
import runpy, io, contextlib, json
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit as e:
        pass
d = json.loads(buf.getvalue().strip().split("\n")[-1])
print("$lib inflight $inf", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("parity"))
PY
done; done
