"""Summarise the rocprofv3 passes of tools/measure.sh.

    python tools/summarize_profiles.py gpurun_out/<tag> [--update-traffic <tag>]

  stats/   rocprofv3 --kernel-trace --stats: the per-kernel table, and - from the per-dispatch trace - the same kernels split
           into their FULL-CORPUS dispatches and the short ones (a filter scan's pre-pass / a probe run the same kernel over a
           prefix: their durations must not pollute the average the roofline is checked against)
  pmc/     rocprofv3 --pmc FETCH_SIZE (its own run): FETCH_SIZE per dispatch, x 1024 B/KB x 2 (gfx950 counts a 128-byte request
           as 64 bytes, MI355X_MICROARCH.md, HBM section), full-corpus dispatches only
  --update-traffic: rewrite profiles/pmc_traffic.json for the kernels bench.py reports, stamped with the hash of the kernel
           sources of THIS build (bench.py refuses an entry whose hash is stale)
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]

# rocprofv3 kernel name (prefix) -> bench.py's kernel name @ rows, for the kernels bench.py attaches `traffic` to
KNOWN = {
    "vg_scan_kernel<1, 0, 3, true, false>": "scan_f32_l2_u3_lpr32_nt",
    "vg_scan_kernel<4, 1, 3, true, false>": "scan_u8_cos_u3_lpr16_nt",
    "vg_scan_filter_kernel<1, 0, 3, true, true>": "scan_filter_f32_l2_q8_u3_lpr8_nt",
    "vg_scan_filter_n4_kernel<4, 2, 3, true>": "scan_filter_u8_cos_n4_u3_lpr8_nt",
}


def find(pat):
    r = glob.glob(os.path.join(out, pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name else name


for name in ("bench_default.json", "bench_under_rocprof.json"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        print("# %s" % name)
        for l in open(p):
            if l.startswith("{"):
                print(l.strip())

st = find("stats/**/*kernel_stats.csv")
print("\n# rocprofv3 --kernel-trace --stats: per-kernel summary (%s)" % (os.path.relpath(st, out) if st else "MISSING"))
if st:
    for r in list(csv.DictReader(open(st)))[:24]:
        print("%-90s calls %6s  total %12s ns  avg %12s ns  min %10s  max %10s  %6s%%" % (
            short(r.get("Name", ""))[:90], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("MinNs"),
            r.get("MaxNs"), r.get("Percentage")))
tr = find("stats/**/*kernel_trace.csv")
if tr:
    print("\n# the same trace per dispatch: FULL-CORPUS dispatches (duration >= half the kernel's longest) apart from the short ones")
    per = {}
    for r in csv.DictReader(open(tr)):
        try:
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        except Exception:
            continue
        per.setdefault(short(r.get("Kernel_Name", "")), []).append(dur)
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:16]:
        mx = max(v)
        full = [d for d in v if d >= mx / 2]
        rest = [d for d in v if d < mx / 2]
        print("%-90s full: %4d x avg %10.0f ns (min %d max %d)%s" % (
            k[:90], len(full), sum(full) / len(full), min(full), max(full),
            ("   short: %4d x avg %8.0f ns" % (len(rest), sum(rest) / len(rest))) if rest else ""))

pm = find("pmc/**/*counter_collection.csv")
print("\n# rocprofv3 --pmc FETCH_SIZE (%s)" % (os.path.relpath(pm, out) if pm else "MISSING"))
traffic = {}
if pm:
    per = {}
    for r in csv.DictReader(open(pm)):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        per.setdefault(short(r.get("Kernel_Name", "")), []).append(float(r.get("Counter_Value", 0)))
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:16]:
        mx = max(v)
        full = [x for x in v if x >= mx / 2]
        kb = sum(full) / len(full)
        print("%-90s dispatches %4d (full-corpus %4d)  FETCH_SIZE/full dispatch %.1f KB  -> x1024 x2 = %.6g bytes" % (
            k[:90], len(v), len(full), kb, kb * 2048))
        for pref, bname in KNOWN.items():
            if k.startswith(pref):
                traffic[bname] = (kb, len(full))

if "--update-traffic" in sys.argv and traffic:
    tag = sys.argv[sys.argv.index("--update-traffic") + 1]
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_hash()
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    old_tab = {}
    try:
        old_tab = json.load(open(path))
    except Exception:
        pass
    tab = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE passes (their own runs, tools/measure.sh pmc): "
                       "FETCH_SIZE[KB] x 1024 x 2 (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM), "
                       "full-corpus dispatches only.  bench.py copies the entry matching its kernel and row count into "
                       "roofline.traffic - only while kernel_source_hash equals the hash of the kernel sources it runs."}
    rows = 10_000_000
    for bname, (kb, n) in traffic.items():
        tab["%s@%d" % (bname, rows)] = {
            "bytes_per_launch": int(round(kb * 2048)), "fetch_size_kb": round(kb, 1), "dispatches": n, "kernel_source_hash": h,
            "source": "profiles/%s_default_bench_pmc_fetch_size.csv (rocprofv3 --pmc FETCH_SIZE of `python bench.py`, its own run; "
                      "KB x 1024 x 2 = gfx950 128-B request correction; full-corpus dispatches only)" % tag}
    for kk, vv in old_tab.items():                       # (the batched workloads' entries - tools/pmc_batch.sh - are not this pass' to drop)
        if kk.startswith("batch_") and kk not in tab:
            tab[kk] = vv
    json.dump(tab, open(path, "w"), indent=1)
    print("\n# profiles/pmc_traffic.json rewritten for kernel sources %s: %s" % (h, ", ".join(sorted(traffic))))
