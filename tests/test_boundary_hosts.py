"""the drop-in boundary from the hosts' side.
 * a plain C11 program includes include/mola_icp_amd.h under -Wall -Wextra -Werror -pedantic, links the library and walks
   create -> params_from_yaml -> align along their error paths (no GPU needed);
 * shim/mola_icp_amd_shim.h -- the class a maintainer adds to mola-fe-lidar (initialize_solvers / _matchers /
   _quality_evaluators, align: src/LidarOdometry.cpp:81-87, 869-871) -- is compiled with stand-in value types under
   -Wall -Wextra -Werror and exercised; on a GPU one align through it equals the direct call."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "mola-fe-lidar_amd", "lib")
OUT = os.path.join(ROOT, "tests", "hosts", "_build")


def _env():
    env = dict(os.environ)
    # the library's libamdhip64 dependency: the system ROCm for a stand-alone host program
    env["LD_LIBRARY_PATH"] = LIBDIR + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return env


def _build(compiler, std, src, exe, extra=()):
    os.makedirs(OUT, exist_ok=True)
    cmd = [compiler, std, "-O1", "-Wall", "-Wextra", "-Werror", *extra, "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "shim"), os.path.join(ROOT, "tests", "hosts", src), "-o", os.path.join(OUT, exe),
           "-L", LIBDIR, "-lmola_icp_amd", "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return os.path.join(OUT, exe)


@pytest.fixture(scope="module")
def c_host(pkg):
    return _build("gcc", "-std=c11", "c_abi_walk.c", "c_abi_walk", ("-pedantic",))


@pytest.fixture(scope="module")
def shim_host(pkg):
    return _build("g++", "-std=c++17", "shim_test.cpp", "shim_test")


def test_c11_host_walks_the_error_paths(c_host):
    r = subprocess.run([c_host], capture_output=True, text=True, env=_env(), timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and "PASSED" in r.stdout and "FAIL " not in r.stdout, r.stdout + r.stderr
    assert "no GPU: create -> E_NODEVICE" in r.stdout or "GPU: a 4-point align runs" in r.stdout


def test_shim_compiles_and_loads_the_reference_yaml_sections(shim_host):
    r = subprocess.run([shim_host, "config"], capture_output=True, text=True, env=_env(), timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout + r.stderr
    for what in ("solvers", "matchers", "unknown solver class named", "align before initialize_*"):
        assert "ok   " + what in r.stdout


@pytest.mark.gpu
def test_c11_host_on_gpu(c_host):
    r = subprocess.run([c_host], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0 and "GPU: a 4-point align runs" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_shim_align_equals_direct_call(shim_host, pkg, synth, tmp_path):
    g, l, _ = synth.make_pair(20000, 24000, seed=17)
    f = tmp_path / "clouds.bin"
    with open(f, "wb") as fh:
        np.array([g.shape[1], l.shape[1]], dtype=np.uint64).tofile(fh)
        g.tofile(fh)
        l.tofile(fh)
    r = subprocess.run([shim_host, "align", str(f)], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    vals = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0].split()[1:]
    T = np.array([float(v) for v in vals[:16]]).reshape(4, 4)
    # the same call through the Python binding: shipped pipeline (icp-settings-regular.yaml), same per-call Parameters
    p = pkg.Parameters.load_from_file(os.path.join(ROOT, "params", "icp-settings-regular.yaml"))
    icp = pkg.ICP(device=0)
    d = icp.align(g, l, pkg.pose_from_xyzypr([0.05, 0, 0, 0.004, 0, 0]), p)
    assert np.array_equal(T, d.optimal_tf) and float(vals[16]) == d.quality
    assert int(vals[17]) == d.nIterations and int(vals[18]) == d.terminationReason
    icp.close()
