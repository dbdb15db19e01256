#!/usr/bin/env python3
"""Kernel name -> (vgpr, agpr, sgpr, spills, LDS, scratch) from the amdhsa metadata of a `hipcc -save-temps` .s file.
    tools/isa_regs.py <file.s> [name-substring]"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if sub in name:
        print("%-90s agpr %s vgpr %s sgpr %s vspill %s sspill %s lds %s scratch %s" % (name[:90], blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
