"""The C-ABI library loads on a CPU-only box and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mola_icp_amd.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(mola_(?:icp|lo)_[a-z0-9_]+)\s*\(", txt))
    names -= {"mola_icp_allreduce_fn", "mola_lo_align_fn"}
    return sorted(names)


def test_header_symbols_exported(pkg):
    lib = pkg._lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mola_icp_amd.h but not exported"
    # and the binding binds exactly the header's functions
    assert sorted(pkg._lib.SIGNATURES) == declared


def test_struct_layout_matches_header(pkg):
    # sizes the C side reports through a default-params call + known field offsets
    p = pkg._lib.CParams()
    assert pkg._lib.lib().mola_icp_params_default(ctypes.byref(p)) == 0
    assert p.max_iterations == 40 and p.matcher_threshold == pytest.approx(0.5)
    assert p.quality_threshold == pytest.approx(0.10) and p.knn == 6
    assert p.solver_class == pkg._lib.SOLVER_HORN and p.nn_kernel == pkg._lib.NN_AUTO


def test_status_strings_and_errors(pkg):
    lib = pkg._lib.lib()
    assert lib.mola_icp_status_string(0) == b"ok"
    assert b"device" in lib.mola_icp_status_string(pkg._lib.E_NODEVICE)
    assert lib.mola_icp_params_default(None) == pkg._lib.E_BADARG
    assert b"null" in lib.mola_icp_last_error()


def test_no_cpu_fallback_without_gpu(pkg):
    n = ctypes.c_int(-1)
    assert pkg._lib.lib().mola_icp_device_count(ctypes.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.IcpError) as e:
        pkg.ICP()
    assert e.value.status == pkg._lib.E_NODEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_never_references_oracle():
    """the product path must not import, link, include or call anything under oracle/"""
    pdir = os.path.join(ROOT, "mola-fe-lidar_amd")
    banned = [r"\bimport\s+oracle", r"\bfrom\s+oracle", r"libicp_oracle", r"\borc_[a-z]", r"#include\s*[<\"][^>\"]*oracle",
              r"oracle/_build", r"oracle/_ref", r"icp_oracle\.c"]
    for dp, _, files in os.walk(pdir):
        if os.path.basename(dp) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                for pat in banned:
                    assert not re.search(pat, txt), f"{f} uses the oracle ({pat})"


def test_comm_api_without_gpu(pkg):
    # argument checking of the RCCL entry points needs neither a GPU nor RCCL
    lib = pkg._lib.lib()
    assert lib.mola_icp_comm_unique_id(None) == pkg._lib.E_BADARG
    assert lib.mola_icp_comm_init(None, None, 1, 0) == pkg._lib.E_BADARG
    assert lib.mola_icp_comm_destroy(None) == pkg._lib.E_BADARG
    assert lib.mola_icp_comm_set_library(None) == 0


def test_device_pool_assignment_is_round_robin(pkg):
    """config[3]: 64 independent pairs over the 8 GPUs of a node -- pair i goes to device slot i mod 8;
    the dealing rule is a pure function of the C-ABI and needs no GPU."""
    a = pkg.pool_assignment(64, 8)
    assert a == [i % 8 for i in range(64)]
    assert all(a.count(d) == 8 for d in range(8))                     # 8 pairs per GPU
    assert pkg.pool_assignment(10, 4) == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1]
    assert pkg.pool_assignment(0, 3) == []
    assert pkg.pool_assignment(3, 1) == [0, 0, 0]
    with pytest.raises(pkg.IcpError):
        pkg.pool_assignment(4, 0)


def test_device_pool_needs_a_gpu_or_fails_loudly(pkg):
    import ctypes
    n = ctypes.c_int(0)
    pkg._lib.lib().mola_icp_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.IcpError) as e:
        pkg.DevicePool()
    assert e.value.status == pkg._lib.E_NODEVICE


def test_roctx_ranges_are_optional(pkg, monkeypatch):
    """MOLA_ICP_ROCTX asks for named ranges around the stages (rocprofv3 --marker-trace); with or without the marker library
    present the host loop behaves the same (run over caller-supplied stages: no GPU needed)"""
    import subprocess
    import sys
    code = ("import importlib, numpy as np; pkg = importlib.import_module('mola-fe-lidar_amd');"
            "p = pkg.Parameters(); p.max_iterations = 3;"
            "acc = np.zeros(24); acc[0] = acc[16] = 3.0; acc[1:4] = 1.0; acc[4:7] = 1.5; acc[7] = acc[11] = acc[15] = 2.0;"
            "r = pkg.run_loop(lambda T, thr: 3, lambda p_, T, s, cl, cg, rs: acc, np.eye(4), p, 3, 3);"
            "print('ITS', r.nIterations)")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env_extra in ({}, {"MOLA_ICP_ROCTX": "1"}):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "ITS" in out.stdout, out.stdout + out.stderr
