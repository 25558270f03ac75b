#!/bin/bash
# Sanitizer runs of the HOST code, on the CPU (never on the GPU box: GPU AddressSanitizer / XNACK are not available on the pool):
#   bash tools/sanitize.sh            -> profiles/r06/sanitizers/{asan_ubsan,tsan}_*.txt
# Builds ../lib/san/libmola_icp_amd_<kind>.so (host .cpp with -fsanitize, device objects as they are), then runs the CPU test suite
# and the plain-C / C++ hosts of tests/hosts against it with the sanitizer runtime preloaded (Python itself is not instrumented).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
OUT=profiles/r06/sanitizers; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run_kind() {
  kind=$1; san=$2; rt=$3; opts=$4; skip=${5:-}
  make -C mola-fe-lidar_amd/csrc san SAN=$san > $OUT/${kind}_build.log 2>&1 || { tail -5 $OUT/${kind}_build.log; return 1; }
  lib=$ROOT/mola-fe-lidar_amd/lib/san/libmola_icp_amd_$(echo $san | tr ',' '_').so
  # a directory in which the sanitizer build answers to the product library's name (the hosts link -lmola_icp_amd)
  d=/tmp/mola_san_$kind; rm -rf $d; mkdir -p $d; ln -s $lib $d/libmola_icp_amd.so
  # (libstdc++ is preloaded with the runtime: Python does not link it, and a sanitizer runtime that initialises without it has no
  #  __cxa_throw to forward to -- the first C++ exception inside the library then aborts (ASan) or hangs (TSan) the run)
  pre="$(g++ -print-file-name=$rt) $(g++ -print-file-name=libstdc++.so.6)"
  env LD_PRELOAD="$pre" $opts MOLA_ICP_LIB_PATH=$lib \
      timeout -k 10 600 python -m pytest tests -q -m "not gpu" -p no:cacheprovider $skip > $OUT/${kind}_pytest.txt 2>&1
  echo "pytest exit code: $?" >> $OUT/${kind}_pytest.txt
  for h in c_abi_walk "shim_test config" sort_net_test; do
    exe=tests/hosts/_build/$(echo $h | cut -d' ' -f1)
    [ -x $exe ] || continue
    env LD_PRELOAD="$pre" $opts LD_LIBRARY_PATH=$d:/opt/rocm/lib timeout -k 10 300 $exe $(echo $h | cut -s -d' ' -f2) > $OUT/${kind}_host_$(echo $h | cut -d' ' -f1).txt 2>&1
    echo "exit code: $?" >> $OUT/${kind}_host_$(echo $h | cut -d' ' -f1).txt
  done
  # the device-only threaded scaffolding (leases, batch lanes, deferred builds, the cloud cache under concurrent put / align / drop):
  # the real host translation units + the test-only host-memory backend, 8 threads on one handle (tests/hosts/race_host.cpp)
  make -f tests/hosts/Makefile.race SAN=$san >> $OUT/${kind}_build.log 2>&1 && {
    env $opts timeout -k 10 900 tests/hosts/_build/race_host_$(echo $san | tr ',' '_') 8 16 > $OUT/${kind}_race_host.txt 2>&1
    echo "exit code: $?" >> $OUT/${kind}_race_host.txt
  }
  n=$(grep -l -E "ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer|ERROR: LeakSanitizer" $OUT/${kind}_*.txt 2>/dev/null | grep -v BEFORE_the_fix | wc -l)
  echo "$kind: $(grep -h -E 'passed|failed' $OUT/${kind}_pytest.txt | tail -1) ; files with sanitizer reports: $n" | tee $OUT/${kind}_summary.txt
}
# (tests/test_race_host.py builds and runs sanitizer executables of its own: not under a preloaded runtime -- the race host runs below)
run_kind asan_ubsan address,undefined libasan.so "ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1" "--ignore=tests/test_race_host.py"
# (ThreadSanitizer: without the tests that start child interpreters or compilers -- torch / g++ under a preloaded TSan runtime do not
#  come back -- i.e. the bench launcher, the gloo ranks, the compiled hosts -- and without the roctx probe, which dlopens the
#  profiler's library; those run under ASan + UBSan above)
run_kind tsan thread libtsan.so "TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1:report_signal_unsafe=0" \
  "--timeout 120 --ignore=tests/test_bench_launch.py --ignore=tests/test_sharded_gloo.py --ignore=tests/test_boundary_hosts.py --ignore=tests/test_gpu_prepare.py --ignore=tests/test_local_comm.py --ignore=tests/test_race_host.py --deselect tests/test_c_abi.py::test_roctx_ranges_are_optional"
