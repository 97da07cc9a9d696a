"""bench.py's launcher behaviour (CPU): `--gpus N` must either start N ranks itself or fail loudly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1"})
    assert r.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """No launcher in the environment: bench.py starts a child torch.distributed.run with N ranks
    before touching any GPU and exits with its code.  Here (no GPU) both ranks stop at the
    'no HIP device' check, which is the product's no-CPU-fallback rule."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the ranks would run the whole benchmark")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "needs an MI355X" in out
    assert "local_rank: 1" in out or "rank      : 1" in out or out.count("needs an MI355X") >= 2
