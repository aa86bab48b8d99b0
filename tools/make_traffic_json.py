#!/usr/bin/env python3
"""gpurun_out/traffic_<tag>.json (tools/gpu_traffic.sh) -> profiles/hbm_traffic.json.

HBM bytes per launch from the L2 memory-side counters, as MI355X_MICROARCH.md section HBM
prescribes: FETCH_SIZE and WRITE_SIZE collected in separate PMC passes, unit KiB, and on
gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes -> doubled.  (Infinity-Cache hits are
included in these counters, so this is fabric traffic, an upper bound on DRAM traffic.)
usage: python tools/make_traffic_json.py C2/bf16/model=gpurun_out/traffic_bf16_model.json ..."""
import json
import os
import sys

SLOT = {"fwd2_kernel": "fwd", "fwd_dense_kernel": "fwd", "fwd_fast_kernel": "fwd", "pointgrad2_kernel": "bwd_points",
        "bwd_fast_kernel": "bwd_points", "binned_accumulate_kernel": "bwd_accumulate",
        "binned_accumulate_f32_kernel": "bwd_accumulate", "binned_accumulate_tr_kernel": "bwd_accumulate",
        "binned_accumulate_split_kernel": "bwd_accumulate", "pointgrad_dense_kernel": "bwd_points",
        "fwd_dense_f32_kernel": "fwd", "pointgrad_dense_f32_kernel": "bwd_points"}
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "profiles", "hbm_traffic.json")
res = json.load(open(out_path)) if os.path.exists(out_path) else {}
for arg in sys.argv[1:]:
    key, path = arg.split("=")
    entry = {}
    for kname, v in json.load(open(path)).items():
        base = kname.split("::")[-1].split("<")[0]
        if base in SLOT and "FETCH_SIZE_KB_avg" in v and "WRITE_SIZE_KB_avg" in v:
            entry[SLOT[base]] = int((2 * v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"]) * 1024)
            entry[SLOT[base] + "_raw"] = {"FETCH_SIZE_KiB": v["FETCH_SIZE_KB_avg"],
                                          "WRITE_SIZE_KiB": v["WRITE_SIZE_KB_avg"]}
    res[key] = entry
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1)[:1500])
