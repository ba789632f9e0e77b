#!/usr/bin/env python3
"""Loop census of one kernel in a hipcc -save-temps .s file: for every backward branch,
the instruction mix of the loop body (VALU / SALU / LDS / VMEM / waitcnt / lane spills)."""
import re
import sys


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] <= i:
            loops.append((labels[m.group(2)], i, m.group(2)))
    loops.sort()
    for a, b, lab in loops:
        mix = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "wait": 0, "wrlane": 0, "rdlane": 0, "branch": 0, "other": 0}
        for l in body[a:b + 1]:
            t = l.strip().split()
            if not t or t[0].startswith(";") or t[0].startswith(".") or t[0].endswith(":"):
                continue
            op = t[0]
            if op.startswith("v_writelane"):
                mix["wrlane"] += 1
            elif op.startswith("v_readlane"):
                mix["rdlane"] += 1
            elif op.startswith("v_"):
                mix["valu"] += 1
            elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
                mix["wait"] += 1
            elif op.startswith("s_cbranch") or op.startswith("s_branch"):
                mix["branch"] += 1
            elif op.startswith("s_"):
                mix["salu"] += 1
            elif op.startswith("ds_"):
                mix["lds"] += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
                mix["vmem"] += 1
            else:
                mix["other"] += 1
        tot = sum(mix.values())
        print("%-12s lines %5d-%5d  n=%4d  %s" % (lab, start + a + 1, start + b + 1, tot, " ".join("%s=%d" % kv for kv in mix.items() if kv[1])))


main()
