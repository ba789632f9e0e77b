#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/sq_counters.sh to one JSON: per kernel, the mean of
every counter over the profiled launches, plus wave-instructions per output byte and per token."""
import csv
import glob
import json
import sys

root, tag = sys.argv[1], sys.argv[2]
agg = {}
for f in glob.glob("%s/pmc_%s_*/*/*counter_collection.csv" % (root, tag)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not k.startswith("otz_"):
            continue
        agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {}
OUT_BYTES, TOKENS = 10000 * 65536, None
for k, cs in agg.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    insts = sum(d.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
    if insts:
        d["wave_insts_total"] = insts
        d["wave_insts_per_output_byte"] = insts / OUT_BYTES
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
    out[k] = d
json.dump(out, sys.stdout, indent=1, sort_keys=True)
