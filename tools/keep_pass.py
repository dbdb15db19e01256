#!/usr/bin/env python3
"""Copy what a tools/r6_final_pass.sh / tools/measure.sh pass left in gpurun_out/<tag>/ into profiles/ under the names the docs and
profiles/pmc_traffic.json cite (the counter CSV trimmed to the columns the traffic figures use; lines cut to 400 characters in summaries).
    tools/keep_pass.py <tag>"""
import csv, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))
        print("profiles/%s_%s" % (tag, b))
cp("bench_default_with_traffic.json", "default_bench_line.json")
if not os.path.exists(os.path.join(src, "bench_default_with_traffic.json")):
    cp("bench_default.json", "default_bench_line.json")
cp("bench_under_rocprof.json", "bench_line_under_rocprof.json")
cp("stats/run_kernel_stats.csv", "default_bench_kernel_stats.csv")
for a, b in (("c5_default_path_int8_filter_pmc.txt",) * 2, ("long_rows_int8_filter_pmc_10m_1536.txt",) * 2, ("c3b_uint8_batch_pmc.txt",) * 2,
             ("bench_lines_other_workloads.jsonl",) * 2, ("bench_stage.json",) * 2, ("dist_rccl_1rank.json", "n_rank_path_1_rank_over_rccl.json"),
             ("dist_selflaunch_2ranks_shared_device.json", "n_rank_path_2_ranks_sharing_the_device_gloo.json")):
    cp(a, b)
p = os.path.join(src, "summary.txt")
if os.path.exists(p):
    with open(os.path.join(dst, "%s_default_bench_line_rocprof_summary.txt" % tag), "w") as f:
        for line in open(p):
            f.write(line[:400].rstrip("\n") + "\n")
    print("profiles/%s_default_bench_line_rocprof_summary.txt" % tag)
p = os.path.join(src, "pmc", "run_counter_collection.csv")
if os.path.exists(p):
    keep = ("Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp")
    with open(os.path.join(dst, "%s_default_bench_pmc_fetch_size.csv" % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(keep)
        for r in csv.DictReader(open(p)):
            if r["Kernel_Name"].startswith(("void vg_", "vg_")):
                w.writerow([r[k][:96] if k == "Kernel_Name" else r[k] for k in keep])
    print("profiles/%s_default_bench_pmc_fetch_size.csv" % tag)
p = os.path.join(src, "pytest_gpu.txt")
if os.path.exists(p):
    last = open(p).read().strip().splitlines()[-1]
    n = last.split()[0] if last.split() and last.split()[0].isdigit() else "x"
    shutil.copy(p, os.path.join(dst, "%s_pytest_gpu_%s_tests.txt" % (tag, n)))
    print("profiles/%s_pytest_gpu_%s_tests.txt" % (tag, n))
p = os.path.join(src, "pmc_traffic.json")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, "pmc_traffic.json"))
    print("profiles/pmc_traffic.json")
