#!/usr/bin/env python3
"""Register / scratch usage of every kernel of one HIP translation unit (from the code object metadata):
    python tools/kernel_regs.py sqlite-vector_amd/csrc/vg_multi.hip [--all]
prints name, VGPRs, spilled VGPRs, scratch bytes; without --all only kernels that spill or use scratch."""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
show_all = "--all" in sys.argv
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value",
                           "-Wno-unused-result", "--cuda-device-only", "-S", "-o", f.name, src], stderr=subprocess.DEVNULL)
    text = open(f.name).read()
meta = text[text.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk).group(1)
    name, vg, sp, scr = g("name"), int(g("vgpr_count")), int(g("vgpr_spill_count")), int(g("private_segment_fixed_size"))
    if show_all or sp or scr:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print("%4d vgpr %4d spilled %5d scratch  %s" % (vg, sp, scr, dem.replace("(ScanArgs)", "")))
