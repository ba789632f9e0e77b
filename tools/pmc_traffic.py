#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with
--kernel-trace only) of `bench.py` into profiles/<round>/pmc_traffic.json.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream at 64 bytes, i.e.
reports half the bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming
stores.  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch."""
import csv
import glob
import json
import sys


def mean_counter(pattern, kernel, counter):
    vals = []
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == kernel and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s under %s" % (counter, kernel, pattern))
    return sum(vals) / len(vals), len(vals)


def main():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_build_id
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    kernels = sys.argv[4].split("+") if len(sys.argv) > 4 else ["otz_index_kernel", "otz_apply_kernel", "otz_out_crc_kernel"]
    f = w = 0.0
    per = {}
    nf = nw = 0
    for kernel in kernels:  # the DEFLATE path is three launches (index, apply, CRC pass): their traffic adds up
        fk, nf = mean_counter(fetch_dir + "/*/*counter_collection.csv", kernel, "FETCH_SIZE")
        wk, nw = mean_counter(write_dir + "/*/*counter_collection.csv", kernel, "WRITE_SIZE")
        per[kernel] = {"FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB": wk, "traffic_bytes": (2 * fk + wk) * 1024}
        f += fk
        w += wk
    rep = {"kernel": "+".join(kernels), "per_kernel": per, "kernel_build_id": kernel_build_id(),
           "launches_averaged": [nf, nw], "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w,
           "fetch_bytes_corrected_x2": 2 * f * 1024, "write_bytes": w * 1024,
           "traffic_bytes_per_launch": (2 * f + w) * 1024,
           "workload": {"entries": 10000, "entry_bytes": 65536, "method": 8},
           "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py "
                      "--steps 3 --warmup 1 --no-cpu-baseline"}
    json.dump(rep, open(out, "w"), indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
