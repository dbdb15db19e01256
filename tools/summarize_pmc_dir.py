#!/usr/bin/env python3
"""Per-dispatch counter values of the scan kernels found under a directory of rocprofv3 --pmc runs (one sub-directory per run):
    python tools/summarize_pmc_dir.py gpurun_out/<tag>/tlb [kernel-name-substring]
prints, per run, one line per matching dispatch: duration (us) and every counter collected for it."""
import csv
import glob
import os
import sys
from collections import OrderedDict

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "vg_scan_kernel"
csv.field_size_limit(1 << 30)
for run in sorted(glob.glob(os.path.join(root, "*"))):
    files = glob.glob(os.path.join(run, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("%s: no counter_collection.csv" % os.path.basename(run))
        continue
    disp = OrderedDict()
    for row in csv.DictReader(open(files[0])):
        if needle not in row["Kernel_Name"]:
            continue
        d = disp.setdefault(row["Dispatch_Id"], {"grid": row["Grid_Size"], "us": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3,
                                                 "name": row["Kernel_Name"][:60], "c": OrderedDict()})
        d["c"][row["Counter_Name"]] = d["c"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    print("== %s" % os.path.basename(run))
    for i, (k, d) in enumerate(disp.items()):
        if int(d["grid"]) < 100_000:          # (the pre-pass / short launches)
            continue
        print("  #%d %s grid %s %.1f us  " % (i, d["name"], d["grid"], d["us"]) + "  ".join("%s=%.4g" % kv for kv in d["c"].items()))
