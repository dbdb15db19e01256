#!/bin/bash
# PMC passes over one batched kernel run (own passes, kernel-trace only - see MI355X_MICROARCH.md for the counter rules).
#   tools/tools_profile_batch_pmc.sh <type> <dim> <metric> <outdir>
export TMPDIR=/tmp
T=${1:-u8}; D=${2:-768}; M=${3:-3}; OUT=${4:-gpurun_out/pmc_batch}
mkdir -p $OUT
CMD="python /root/repo/tools/tools_batch_bench.py --type $T --dim $D --nq 1024 --metric $M --reps 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -f csv -d $OUT/p$i -- $CMD > /dev/null 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vg_batch" not in k or "merge" in k: continue
        name = k.split("(")[0][-60:]
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in tot.items():
    print(k)
    for c in sorted(d): print("   %-28s %.4g" % (c, d[c]))
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_WAVE_CYCLES"):
        # GRBM_GUI_ACTIVE counts per XCD (8), the SQ_* counters per wavefront / SIMD: 1024 SIMDs x active cycles = 128 x GRBM_GUI_ACTIVE
        print("   -> MFMA pipe busy %.1f %% of (1024 SIMDs x active cycles); %.1f busy cycles per MFMA instruction; waiting %.0f %% / waiting to issue %.0f %% of the wavefront cycles"
              % (100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128.0 * d["GRBM_GUI_ACTIVE"]), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(d.get("SQ_INSTS_MFMA", 1), 1),
                 100.0 * d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"], 100.0 * d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]))
PY
