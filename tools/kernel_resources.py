#!/usr/bin/env python3
"""Register / spill / LDS table of every kernel of libnerfca_hip.so, from the metadata notes of the code objects in nerf-ca_amd/csrc/*.o
(what the hardware is told: .vgpr_count, .vgpr_spill_count, .sgpr_count, .sgpr_spill_count, scratch bytes, LDS):
    python tools/kernel_resources.py [filter] > profiles/r02_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1] if len(sys.argv) > 1 else ""
notes = ""
with tempfile.TemporaryDirectory() as td:
    # one offload bundle per translation unit: take them from the object files the library was linked from
    import glob
    for i, obj in enumerate(sorted(glob.glob(os.path.join(ROOT, "nerf-ca_amd", "csrc", "*.o")))):
        fat, co = f"{td}/fat{i}.bin", f"{td}/dev{i}.co"
        if subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True).returncode:
            continue            # a translation unit without device code
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
rows = []
for blk in notes.split("- .agpr_count:")[1:]:
    def field(name):
        m = re.search(rf"\.{name}:\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = field("name")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*\)$", "", dem).replace("void ", "")
    if pat and pat not in dem:
        continue
    rows.append((dem, field("vgpr_count"), field("vgpr_spill_count"), field("sgpr_count"), field("sgpr_spill_count"), field("private_segment_fixed_size"),
                 field("group_segment_fixed_size"), blk.split()[0]))
print(f"{'kernel':58s} {'vgpr':>5s} {'vspill':>6s} {'sgpr':>5s} {'sspill':>6s} {'scratchB':>8s} {'ldsB':>6s} {'agpr':>5s}")
for r in sorted(rows):
    print(f"{r[0]:58s} {r[1]:>5s} {r[2]:>6s} {r[3]:>5s} {r[4]:>6s} {r[5]:>8s} {r[6]:>6s} {r[7]:>5s}")
