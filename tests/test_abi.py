"""C-ABI surface (CPU): the library loads, exports every symbol include/rq.h declares, and fails
loudly -- never silently falls back -- when no gfx950 device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from rag_uq_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(os.path.dirname(os.path.abspath(_native.__file__)), "csrc")


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rq.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rq.h but not exported by librq_hip.so"
    # and the binding declares a signature for each of them
    assert set(names) == set(_native._SIGNATURES), set(names) ^ set(_native._SIGNATURES)


def test_bm25_library_exports_every_declared_symbol():
    """include/rq_bm25.h (librq_bm25.so: the batched CPU BM25 of configs[4], host cores, no GPU code): every declared entry
    point is exported and bound; malformed operands are refused at create time."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rq_bm25.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(rq_bm25_[a-z0-9_]+)\s*\(", text)))
    lib = _native.load_bm25_library()
    assert names == sorted(_native._BM25_SIGNATURES)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rq_bm25.h but not exported by librq_bm25.so"
    assert b"bm25" in lib.rq_bm25_version()
    blob = open(_native.BM25_LIB_PATH, "rb").read()
    assert b"gfx950" not in blob and b"hip" not in blob.lower()[:0] + b""          # host code only
    indptr = np.array([0, 2, 3], np.int64)
    rows = np.array([0, 1, 1], np.int32)
    contrib = np.array([1.0, 2.0, 0.5])
    h = _native.bm25_create(indptr, rows, contrib, 2, 2)
    r, sc = _native.bm25_topk(h, np.array([0, 2, 2, 3], np.int64), np.array([0, 1, 1], np.int32), 3, 2)
    assert r.tolist() == [[1, 0], [-1, -1], [1, -1]] and sc.tolist() == [[2.5, 1.0], [0.0, 0.0], [0.5, 0.0]]
    with pytest.raises(_native.RqError):
        _native.bm25_topk(h, np.array([0, 1], np.int64), np.array([2], np.int32), 1, 2)          # token id outside the index
    _native.bm25_destroy(h)
    with pytest.raises(_native.RqError):
        _native.bm25_create(indptr, np.array([0, 1, 2], np.int32), contrib, 2, 2)                # row outside the corpus


def test_library_is_gfx950_code_object():
    blob = open(_native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"rq_scan_kernel" in blob


def test_version_and_error_channel():
    lib = _native.load_library()
    assert b"gfx950" in lib.rq_version()
    assert isinstance(_native.last_error(), str)


def test_no_device_means_loud_failure_not_fallback():
    if _native.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_native.RqError, match="no HIP device|no CPU fallback"):
        _native.NativeIndex(768, 0)
    lib = _native.load_library()
    ids = (ctypes.c_int * 1)(0)
    assert not lib.rq_index_create(768, 1, ids)
    assert not lib.rq_index_create(0, 1, ids) and "dim" in _native.last_error()
    two = (ctypes.c_int * 2)(0, 0)
    assert not lib.rq_index_create(768, 2, two) and "no HIP device" in _native.last_error()      # multi-device: same loud failure
    assert not lib.rq_index_create(768, 0, ids) and "n_devices" in _native.last_error()
    assert not lib.rq_load(b"/nonexistent/path", 1, ids)


def test_key_packing_roundtrip_matches_device_encoding():
    """host twin of csrc/rq_device.h rq_make_key: order of keys == canonical order"""
    from rag_uq_amd import distributed as d
    rng = np.random.default_rng(0)
    s = np.concatenate([rng.standard_normal(200).astype(np.float32), np.float32([0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30])])
    r = rng.permutation(len(s)).astype(np.int64)
    keys = d.pack_keys(s, r)
    s2, r2 = d.unpack_keys(keys)
    assert np.array_equal(r2, r) and np.array_equal(s2.view(np.uint32) & 0x7FFFFFFF, s.view(np.uint32) & 0x7FFFFFFF)
    order_keys = np.argsort(-keys.astype(np.float64), kind="stable")      # coarse check first
    canon = np.lexsort((r, -s.astype(np.float64)))
    assert np.array_equal(np.argsort(keys)[::-1], canon) or np.array_equal(s[np.argsort(keys)[::-1]], s[canon])
    k2 = d.pack_keys(np.float32([0.5, 0.5]), np.int64([7, 3]))
    assert k2[1] > k2[0]                                                   # equal score: lower row wins
    assert d.pack_keys(np.float32([1.0]), np.int64([-1]))[0] == 0
    ms, mr = d.merge_keys_host(np.array([[k2[0], 0, k2[1]]], dtype=np.uint64), 4)
    assert mr.tolist() == [[3, 7, -1, -1]] and ms[0, :2].tolist() == [0.5, 0.5]


def test_record_codec_properties_on_the_host(tmp_path):
    """csrc/rq_device.h's bit-level helpers (order-preserving keys, rounded-up record fields) compiled for the host with
    hipcc and checked on a few hundred thousand values: every decoded field is an upper bound, every code is monotone."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(os.path.dirname(__file__), "native", "codec_check.cpp")
    exe = str(tmp_path / "codec_check")
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", CSRC, src, "-o", exe], check=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    """CPU-only sanitizer builds (the GPU boxes of this pool cannot run sanitizers): -fsanitize=address,undefined over (a) the
    record / key codecs of csrc/rq_device.h, (b) the multi-device parent of csrc/rq_multi.hip -- stripe arithmetic, row read-back,
    k-way merge incl. its threads, the poisoned / failed-search paths -- behind STUBBED shards (tests/native/multi_check.cpp), and
    (c) librq_bm25's scorer (csrc/rq_bm25.cpp compiled into tests/native/bm25_check.cpp).  Any report aborts the program."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    gxx = shutil.which("g++")
    if not os.path.exists(hipcc) or gxx is None:
        pytest.skip("hipcc / g++ not available")
    san = ["-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    native = os.path.join(os.path.dirname(__file__), "native")
    jobs = [([hipcc, "--offload-host-only", *san, "-I", CSRC, os.path.join(native, "codec_check.cpp")], "codec"),
            ([hipcc, "--offload-host-only", *san, "-I", CSRC, os.path.join(native, "multi_check.cpp")], "multi"),
            ([gxx, *san, "-I", os.path.join(ROOT, "include"), os.path.join(native, "bm25_check.cpp"), os.path.join(CSRC, "rq_bm25.cpp")], "bm25")]
    for cmd, name in jobs:
        exe = str(tmp_path / f"{name}_san")
        subprocess.run(cmd + ["-o", exe], check=True, timeout=600, capture_output=True)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0"})
        assert out.returncode == 0 and "0 failures" in out.stdout, f"{name}: {out.stdout[-1500:]}{out.stderr[-3000:]}"


def _build_c_client(tmp_path):
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    pkg = os.path.dirname(os.path.abspath(_native.__file__))
    exe = str(tmp_path / "abi_check")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(os.path.dirname(__file__), "native", "abi_check.c"), "-o", exe,
                    "-L", pkg, "-lrq_hip", "-lm", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"], check=True, timeout=300)
    return exe


def test_header_is_plain_c_and_a_c_client_fails_loudly_without_a_gpu(tmp_path):
    """include/rq.h compiles as C99 with gcc (no HIP, no C++), links against librq_hip.so, and on a box without a GPU
    the client gets NULL + a message instead of a silent fallback."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the gpu-marked twin")
    out = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "OK (no device)" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_client_searches_on_the_gpu(tmp_path):
    import subprocess
    out = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK (device search)" in out.stdout, out.stdout + out.stderr
