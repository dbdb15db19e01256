#!/usr/bin/env python3
"""The shader clock and the package power the chip holds under a batch kernel (a DIRECT clock figure instead of an
inference from s_memtime ticks): a thread polls `rocm-smi --showclocks --showpower --json` every 0.2 s while the main thread runs
batches back to back for `--seconds`; prints the samples' median / min / max sclk and power next to the kernel's mean time.
    VG_F32_FILTER=1 python tools/clock_probe.py --type f32 [--single]     (--single: single-query plain scans instead of batches)"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d.get("card0", {})
            s = {"t": time.time()}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "speed" in kl:
                    s["sclk"] = v
                if "mclk" in kl and "speed" in kl:
                    s["mclk"] = v
                if "power" in kl and ("average" in kl or "current" in kl or "socket" in kl):
                    s["power"] = v
            out.append(s)
        except Exception as e:
            out.append({"err": repr(e)})
        time.sleep(0.2)


def num(x):
    import re
    m = re.search(r"[-+]?\d*\.?\d+", str(x).replace("(", " "))
    return float(m.group(0)) if m else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--type", default="f32")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--metric", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--single", action="store_true")
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    import bench
    pkg = g.load_package()
    vt = {"f32": pkg.F32, "f16": pkg.F16, "bf16": pkg.BF16, "u8": pkg.U8}[a.type]
    c = bench.make_shard(pkg, torch, vt, a.dim, a.rows, 42, 0)
    rng = np.random.default_rng(44)
    qs = rng.standard_normal((a.nq, a.dim), dtype=np.float32)
    if a.type in ("f16", "bf16"):
        qs = torch.from_numpy(qs).to(torch.float16 if a.type == "f16" else torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    if a.type == "u8":
        qs = rng.integers(0, 256, (a.nq, a.dim)).astype(np.uint8)
    if a.single:
        c.set_scan_filter(0)
    run = (lambda: c.scan_topk(a.metric if not a.single else 1, qs[0], 20)) if a.single else (lambda: c.scan_topk_batch(a.metric, qs, 20))
    run(); run()
    samples, stop = [], threading.Event()
    th = threading.Thread(target=poll, args=(stop, samples))
    c.set_profiling(True)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < a.seconds:
        run()
        n += 1
    stop.set()
    th.join()
    _, ms, _ = c.profile_mean_ms()
    sclk = sorted(x for x in (num(s.get("sclk")) for s in samples) if x)
    pw = sorted(x for x in (num(s.get("power")) for s in samples) if x)
    med = lambda v: v[len(v) // 2] if v else None
    print(json.dumps({"what": ("single plain scans" if a.single else "%d-query batches" % a.nq) + " %s %dx%d metric %d back to back for %.1f s" % (a.type, a.rows, a.dim, a.metric, a.seconds),
                      "launches": n, "kernel_ms_mean": round(ms, 4), "sclk_MHz": {"median": med(sclk), "min": sclk[0] if sclk else None, "max": sclk[-1] if sclk else None, "samples": len(sclk)},
                      "power_W": {"median": med(pw), "max": pw[-1] if pw else None}, "first_raw_sample": samples[0] if samples else None}))
    c.close()


if __name__ == "__main__":
    main()
