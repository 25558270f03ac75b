"""the threaded host scaffolding that only runs WITH a device -- workspace leases, mola_icp_align_batch's lockstep
lanes / prepare-ahead tasks / worker pool, mola_icp_align_multi_init, the cloud cache under concurrent put / align_cached /
align_cached_put / drop -- driven from 8 threads on one handle on the CPU: the library's REAL host translation units linked against a
test-only host-memory backend (tests/hosts/fake_backend.cpp; the product itself has no CPU path), every threaded result compared bit
for bit with the serial call (tests/hosts/race_host.cpp).  Plain build here and under ThreadSanitizer; ASan + UBSan and the committed
reports: tools/sanitize.sh -> profiles/r05/sanitizers/.  The reference's contract: src/LidarOdometry.cpp:94-96, 869."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(san=None):
    cmd = ["make", "-f", os.path.join(ROOT, "tests", "hosts", "Makefile.race")] + ([f"SAN={san}"] if san else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return os.path.join(ROOT, "tests", "hosts", "_build", "race_host" + (("_" + san.replace(",", "_")) if san else ""))


def test_eight_threads_on_one_handle_equal_the_serial_calls():
    exe = _build()
    r = subprocess.run([exe, "8", "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "mismatches: 0" in r.stdout
    for name in ("align_batch", "align_multi_init", "cloud_put", "align_cached", "align_cached_put", "cloud_drop"):
        assert f"{name} 0" not in r.stdout, r.stdout        # every entry point was actually driven


def test_the_same_under_thread_sanitizer():
    """(round 5 found a use-after-scope with it: a pool worker notified the batch frame's condition variable after releasing the
    lock the frame's owner waits under -- csrc/c_api.cpp, job_done -- profiles/r05/sanitizers/tsan_race_host_BEFORE_the_fix.txt)"""
    exe = _build("thread")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    r = subprocess.run([exe, "8", "6"], capture_output=True, text=True, timeout=600, env=env)
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel configuration")
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[:6000]
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_the_fake_backend_is_not_in_the_product_library():
    """the host-memory stages are test infrastructure: the shipped library holds none of it (and still refuses to run without a device)"""
    lib = os.path.join(ROOT, "mola-fe-lidar_amd", "lib", "libmola_icp_amd.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    assert "g_fake_" not in syms and "hipGetDeviceCount" not in syms
    mk = open(os.path.join(ROOT, "mola-fe-lidar_amd", "csrc", "Makefile")).read()
    assert "fake_backend" not in mk and "tests/" not in mk
