#!/bin/bash
# counters of k_nn_tiled on configs[4] (1M queries, 10M-point map) for the libraries given (mola-fe-lidar_amd/lib/variants/*.so):
#   bash tools/pmc_c5_ab.sh a.so b.so      (two --pmc passes each; no trace domains beside them)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/mola-fe-lidar_amd/lib; cp $L/libmola_icp_amd.so /tmp/orig_lib.so
OUT=$ROOT/gpurun_out/pmc_c5; mkdir -p $OUT
cat > /tmp/c5_run.py <<PY
import importlib, os, sys, numpy as np
sys.path.insert(0, "$ROOT")
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair(1_000_000, int(os.environ.get("C5_MAP", "10000000")), seed=42)
icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
p = pkg.Parameters(); p.matcher_threshold, p.fixed_iterations, p.skip_quality, p.max_iterations = 1.0, 1, 1, 6
icp.align_resident(np.eye(4), p); icp.align_resident(np.eye(4), p)
PY
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  n=$(basename $v .so); cp $L/variants/$n.so $L/libmola_icp_amd.so
  rm -rf $OUT/$n; mkdir -p $OUT/$n
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/$n/h -- python3 /tmp/c5_run.py > $OUT/$n/h.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/$n/g -- python3 /tmp/c5_run.py > $OUT/$n/g.log 2>&1
  echo "== $n"; python3 $ROOT/tools/pmc_summary.py $OUT/$n --kernel k_nn_tiled | grep -v "^$"
  rm -rf $OUT/$n/h $OUT/$n/g
done
cp /tmp/orig_lib.so $L/libmola_icp_amd.so
