#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (tools/gpu_pmc_multi.sh: gpurun_out/pmc_<tag>_<n>/**/*counter_collection.csv)
into one paragraph per kernel: wave life, active / waiting shares, instruction mix, LDS and MFMA busy time.
    python tools/sq_summary.py gpurun_out/pmc_r04sq_1 gpurun_out/pmc_r04sq_2 gpurun_out/pmc_r04sq_3"""
import collections
import csv
import glob
import os
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "boxattn" in k:
                agg[k.split("(")[0][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over the waves; SQ_LDS_* and "
      "SQ_VALU_MFMA_BUSY_CYCLES cycles summed over CUs / SIMDs; averages per dispatch")
for k, d in agg.items():
    g = lambda c: sum(d[c]) / len(d[c]) if d.get(c) else 0.0
    wc, waves = g("SQ_WAVE_CYCLES"), g("SQ_WAVES")
    if not wc or not waves:
        continue
    instr = sum(g(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM",
                               "SQ_INSTS_BRANCH"))
    print(k)
    print("    waves %d  wave life %d qc  active %.1f%%  s_waitcnt %.1f%%  issue-wait %.1f%%  (waiting to issue LDS %.1f%%)"
          % (waves, wc / waves, 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc,
             100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc))
    print("    instr %.2fM = VALU %.2fM (MFMA %.2fM) + SALU %.2fM + LDS %.2fM + VMEM %.2fM + SMEM %.2fM + BRANCH %.2fM"
          "  -> issue %.1f us at 1 instr / 4 cycles / SIMD"
          % (instr / 1e6, g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_MFMA") / 1e6, g("SQ_INSTS_SALU") / 1e6,
             g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_VMEM") / 1e6, g("SQ_INSTS_SMEM") / 1e6,
             g("SQ_INSTS_BRANCH") / 1e6, instr * 4 / 1024 / 2.4e3))
    print("    LDS busy %.2fM cycles (%.1f us per CU), bank conflicts %.2fM, MFMA busy %.2fM cycles (%.1f us per SIMD)"
          % (g("SQ_LDS_IDX_ACTIVE") / 1e6, g("SQ_LDS_IDX_ACTIVE") / 256 / 2.4e3, g("SQ_LDS_BANK_CONFLICT") / 1e6,
             g("SQ_VALU_MFMA_BUSY_CYCLES") / 1e6, g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / 2.4e3))
