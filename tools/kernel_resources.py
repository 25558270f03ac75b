"""Registers / LDS / scratch of every kernel in a built object (the code object's metadata notes).
   python tools/kernel_resources.py [object] [name filter ...]      (default: build/hip_backend.hip.o)"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "mola-fe-lidar_amd/build/hip_backend.hip.o")
filters = [a for a in sys.argv[1:] if not os.path.exists(a)]
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                           "--output=" + co, "--unbundle"])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    k = ".agpr_count" + k
    def g(key):
        m = re.search(r"\.%s:\s+(\d+)" % key, k)
        return int(m.group(1)) if m else -1
    sym = re.search(r"\.name:\s+(\S+)", k).group(1)
    name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().split("(")[0]
    name = name.replace("void mola_icp_amd::", "")
    if filters and not any(f in name for f in filters):
        continue
    print("%-52s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spills v %3d s %3d" % (name[-52:], g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
          g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
