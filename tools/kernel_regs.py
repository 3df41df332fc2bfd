"""Register / LDS / scratch usage of the kernels in a HIP object (from the code object's metadata notes).
    python tools/kernel_regs.py lightning_gan_zoo_amd/csrc/gz_conv.o [substring ...]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin/"
obj = sys.argv[1]
pats = sys.argv[2:]
d = tempfile.mkdtemp()
subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, d + "/fat.bin"], check=True)
subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + d + "/fat.bin",
                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + d + "/k.co"], check=True)
notes = subprocess.run([LLVM + "llvm-readelf", "--notes", d + "/k.co"], capture_output=True, text=True).stdout
for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name:
        continue
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*$", "", dem).replace("gz::", "").replace("void ", "")
    if pats and not all(p in dem for p in pats):
        continue
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))      # noqa: E731
    print("agpr %3d vgpr %3d sgpr %3d scratch %4d lds %6d  %s" % (int(re.match(r"\s*(\d+)", blk).group(1)), g("vgpr_count"),
          g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), dem[:150]))
