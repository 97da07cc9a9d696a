"""The C-ABI library loads without a GPU and exports exactly what include/searchlite_gpu.h
declares; host-side argument checking works before any device call."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "searchlite_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slg_[a-z_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from searchlite_amd import _native
    if not os.path.exists(_native.lib_path()):
        from searchlite_amd import build
        build.build_gpu()
    return _native.load()


def test_header_declares_expected_surface():
    names = declared_functions()
    for must in ["slg_index_create", "slg_index_destroy", "slg_search_batch", "slg_batch_prepare",
                 "slg_batch_run", "slg_batch_fetch", "slg_rerank_batch", "slg_last_error",
                 "slg_merge_shards_device"]:
        assert must in names


def test_rust_ffi_binds_every_declared_function():
    """integration/searchlite-core/src/gpu/ffi.rs (the reference-side `extern "C"` block a
    maintainer adds behind the `gpu` feature) lists the whole ABI."""
    rs = open(os.path.join(ROOT, "integration", "searchlite-core", "src", "gpu", "ffi.rs")).read()
    missing = [n for n in declared_functions() if f"pub fn {n}(" not in rs]
    assert not missing, f"not bound in gpu/ffi.rs: {missing}"


def test_rust_shim_uses_only_bound_items():
    """Every ffi:: item the shim sources use exists in ffi.rs."""
    base = os.path.join(ROOT, "integration", "searchlite-core", "src")
    ffi = open(os.path.join(base, "gpu", "ffi.rs")).read()
    for rel in ("gpu/mod.rs", "gpu/rerank.rs"):
        src = open(os.path.join(base, rel)).read()
        used = set(re.findall(r"ffi::(slg_[a-z_]+|SLG_[A-Z0-9_]+)", src))
        assert used, rel
        for name in used:
            assert re.search(r"\b" + re.escape(name) + r"\b", ffi), \
                f"{rel} uses ffi::{name}, which ffi.rs does not declare"


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"not exported: {missing}"


def test_abi_version_and_error_string(lib):
    assert lib.slg_abi_version() == 3
    assert isinstance(lib.slg_last_error(), bytes)


def test_null_arguments_fail_cleanly(lib):
    """searchlite-ffi conventions (searchlite-ffi/src/lib.rs:24-43): NULL / negative, no crash."""
    from searchlite_amd import _native as N
    assert lib.slg_index_create(None, 0, 0) is None
    assert b"segs" in lib.slg_last_error()
    assert lib.slg_batch_run(None) == N.ERR_INVALID
    assert lib.slg_batch_fetch(None, None, None, None, None, None) == N.ERR_INVALID
    assert lib.slg_search_batch(None, None, 0, 11, 1, None, None, None, None, None) == N.ERR_INVALID
    assert lib.slg_index_info(None, None, None, None) == N.ERR_INVALID
    assert lib.slg_index_add_filter(None, None) == N.ERR_INVALID
    assert lib.slg_index_remove_filter(None, 0) == N.ERR_INVALID
    assert lib.slg_batch_prepare_filtered(None, 0, None, None, None, None, 11, 1) is None
    assert lib.slg_batch_set_stream(None, None) == N.ERR_INVALID
    lib.slg_index_destroy(None)
    lib.slg_batch_destroy(None)


def test_malformed_segment_is_rejected_before_touching_a_device(lib):
    import numpy as np
    from searchlite_amd import _native as N
    offs = np.array([0, 2, 1], dtype=np.uint64)  # not monotone
    docs = np.array([1, 2], dtype=np.uint32)
    tfs = np.array([1, 1], dtype=np.uint32)
    dl = np.ones(4, dtype=np.float32)
    avg = np.ones(1, dtype=np.float32)
    ptrs = (C.c_void_p * 1)(dl.ctypes.data)
    d = N.SegmentDesc(4, 2, offs.ctypes.data, docs.ctypes.data, tfs.ctypes.data, None, 1,
                      C.addressof(ptrs), avg.ctypes.data, 4.0, 0.9, 0.4, None, 0, 0, None, None, 0)
    arr = (N.SegmentDesc * 1)(d)
    assert lib.slg_index_create(arr, 1, 0) is None
    assert b"monotone" in lib.slg_last_error() or b"too long" in lib.slg_last_error()


def test_no_cpu_fallback_without_device():
    """Without a GPU the product must fail loudly, not compute on the CPU."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import searchlite_amd as sa
    from searchlite_amd import _native as N
    from tests.util import random_segment
    seg = random_segment(np.random.default_rng(0), 50, 5, 5)
    with pytest.raises(N.SlgError) as ei:
        sa.GpuIndex([seg])
    assert ei.value.code in (N.ERR_DEVICE, N.ERR_INVALID)


def _seg_desc(n_docs, offs, docs, tfs, keep):
    import numpy as np
    from searchlite_amd import _native as N
    offs = np.asarray(offs, dtype=np.uint64)
    docs = np.asarray(docs, dtype=np.uint32)
    tfs = np.asarray(tfs, dtype=np.uint32)
    dl = np.ones(max(n_docs, 1), dtype=np.float32)
    avg = np.ones(1, dtype=np.float32)
    ptrs = (C.c_void_p * 1)(dl.ctypes.data)
    keep += [offs, docs, tfs, dl, avg, ptrs]
    return N.SegmentDesc(n_docs, len(offs) - 1, offs.ctypes.data, docs.ctypes.data, tfs.ctypes.data,
                         None, 1, C.addressof(ptrs), avg.ctypes.data, float(n_docs), 0.9, 0.4, None,
                         0, 0, None, None, 0)


@pytest.mark.parametrize("validate", [1, 0])
def test_doc_id_beyond_n_docs_is_rejected(lib, validate):
    """The kernels index per-doc bitmaps with the raw doc id and the round planner interpolates on
    doc / n_docs: a posting with doc >= n_docs must fail with SLG_ERR_INVALID at the boundary,
    also when per-posting validation is switched off (then the last posting of each list is checked)."""
    from searchlite_amd import _native as N
    keep = []
    d = _seg_desc(4, [0, 2, 3], [1, 9, 3], [1, 1, 1], keep)  # doc 9 in a 4-doc segment
    arr = (N.SegmentDesc * 1)(d)
    t = N.default_tuning()
    t.validate = validate
    assert lib.slg_index_create_tuned(arr, 1, 0, C.addressof(t)) is None
    assert b">= n_docs" in lib.slg_last_error()
    assert lib.slg_last_error_code() == N.ERR_INVALID


def test_last_error_code_follows_handle_returning_functions(lib):
    from searchlite_amd import _native as N
    assert lib.slg_index_create(None, 0, 0) is None
    assert lib.slg_last_error_code() == N.ERR_INVALID
    assert lib.slg_index_info(None, None, None, None) == N.ERR_INVALID
    assert lib.slg_last_error_code() == N.ERR_INVALID


def test_tuning_defaults_come_from_the_environment_once(lib, monkeypatch):
    """slg_tuning_default() is the only reader of SLG_* variables."""
    from searchlite_amd import _native as N
    monkeypatch.delenv("SLG_MAXSCORE", raising=False)
    t = N.default_tuning()
    assert t.struct_size == C.sizeof(N.Tuning) and t.pruning == -1 and t.uniform_max_terms == 8
    assert t.validate == 1 and t.champions == 1 and t.cand_mode == 1 and t.block_max == 1
    monkeypatch.setenv("SLG_MAXSCORE", "1")
    monkeypatch.setenv("SLG_UNIFORM_MAX_TERMS", "0")
    t = N.default_tuning()
    assert t.pruning == 1 and t.uniform_max_terms == 0
    bad = N.Tuning()
    bad.struct_size = 3
    keep = []
    arr = (N.SegmentDesc * 1)(_seg_desc(4, [0, 1], [1], [1], keep))
    assert lib.slg_index_create_tuned(arr, 1, 0, C.addressof(bad)) is None
    assert b"struct_size" in lib.slg_last_error()


def test_segfile_library_exports_its_header():
    """include/searchlite_segfile.h (host-only decoder of searchlite's segment files)."""
    from searchlite_amd import index_files as IF
    text = open(os.path.join(ROOT, "include", "searchlite_segfile.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(slf_[a-z0-9_]+)\s*\(", text)))
    assert "slf_postings_decode" in names and "slf_postings_scan" in names
    L = IF._load()
    assert not [n for n in names if not hasattr(L, n)]


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors (searchlite_amd/_native.py) and the Rust mirrors (integration/.../ffi.rs) of the
    header's public structs: sizes from a C program compiled against include/searchlite_gpu.h, field counts
    from the sources (a struct that grows in the header must grow in both mirrors)."""
    import ctypes as C
    import re
    import subprocess
    from searchlite_amd import _native as N
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = {"slg_segment_desc": N.SegmentDesc, "slg_vector_field_desc": N.VectorFieldDesc, "slg_stats": N.Stats,
             "slg_tuning": N.Tuning, "slg_score_plans": N.ScorePlans, "slg_ticket": N.Ticket, "slg_query": N.Query}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "searchlite_gpu.h"\nint main(void) {\n' +
                   "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    sizes = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, cls in names.items():
        assert int(sizes[n]) == C.sizeof(cls), (n, sizes[n], C.sizeof(cls))
    ffi = open(os.path.join(root, "integration", "searchlite-core", "src", "gpu", "ffi.rs")).read()
    for n, cls in names.items():
        m = re.search(r"pub struct %s \{(.*?)\}" % n, ffi, re.S)
        if m is None:
            continue  # (not every struct is bound by the shim)
        body = re.sub(r"//[^\n]*", "", m.group(1))  # (comments may hold commas and colons)
        n_rust = len(re.findall(r"pub\s+\w+\s*:", body))
        assert n_rust == len(cls._fields_), (n, n_rust, len(cls._fields_))
