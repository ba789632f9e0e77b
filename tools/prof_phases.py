#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle breakdown of otz_apply_kernel (work-item 0's view of
each workgroup), via ozg_debug_set_profile_buffer.  Not part of the timed benchmark."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import otezip_amd as oz
from otezip_amd import corpus, _native as N

NAMES = ["res_busy#", "tb_rank/st_items", "tables", "stage", "far_go#", "tb_link", "tb_canon/st_load", "place", "flush", "tb_root", "resolve", "res_go#",
         "rounds#", "B_iters#", "blocks#", "total"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    oz.init(0)
    L = oz.lib()
    arc, ents = corpus.build_archive(n, size)
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents]
    et, dst_len = oz.make_entries(tab)
    dev = torch.device("cuda", 0)
    d_src = torch.frombuffer(bytearray(arc), dtype=torch.uint8).to(dev)
    d_dst = torch.empty(dst_len + 64, dtype=torch.uint8, device=dev)
    d_ent = torch.frombuffer(bytearray(bytes(et)), dtype=torch.uint8).to(dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_prof = torch.zeros(n * 32, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream()

    def run():
        N.check(L.ozg_decode_batch_device_ex(d_src.data_ptr(), d_src.numel(), d_dst.data_ptr(), d_dst.numel(),
                                             d_ent.data_ptr(), d_res.data_ptr(), n, s.cuda_stream, 1))
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    run()
    b.record()
    torch.cuda.synchronize()
    print("plain launch: %.3f ms for %d entries" % (a.elapsed_time(b), n))
    L.ozg_debug_set_profile_buffer(d_prof.data_ptr())
    run()
    torch.cuda.synchronize()
    L.ozg_debug_set_profile_buffer(None)
    pall = d_prof.cpu().numpy().reshape(n, 32).astype(np.float64)
    p = pall[:, :16]
    k1 = pall[:, 16:]
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.int32).reshape(n, 4)
    print("status ok:", int((res[:, 0] == 1).sum()), "/", n)
    tot = p[:, 15].mean()
    for i, name in enumerate(NAMES):
        if name == "-":
            continue
        m = p[:, i].mean()
        if name.endswith("#"):
            print("%-9s %10.1f" % (name, m))
        else:
            print("%-9s %10.0f cyc  %5.1f%%" % (name, m, 100 * m / tot))
    print("---- otz_index_kernel (one wavefront per entry)")
    K1 = ["hdr+lens", "tables", "slots", "walkA", "fixpoint", "accept", "rounds#", "fix_iters#", "rewalks#", "blocks#"]
    tot1 = k1[:, 15].mean()
    for i, name in enumerate(K1):
        m = k1[:, i].mean()
        if name.endswith("#"):
            print("%-10s %10.1f" % (name, m))
        else:
            print("%-10s %10.0f cyc  %5.1f%%" % (name, m, 100 * m / tot1))
    print("%-10s %10.0f cyc" % ("total", tot1))
    t1 = np.sort(k1[:, 15])
    print("index total per entry: p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % tuple(t1[int(q * (n - 1))] for q in (0.1, 0.5, 0.9, 0.99, 1.0)))
    t2 = np.sort(p[:, 15])
    print("apply total per entry: p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % tuple(t2[int(q * (n - 1))] for q in (0.1, 0.5, 0.9, 0.99, 1.0)))
    if k1[:, 13].any():  # (experiment builds -DOTZ_K1_EXP=...: trips of the walk loops per entry)
        print("stepsA#    %10.1f\nstepsB#    %10.1f" % (k1[:, 13].mean(), k1[:, 14].mean()))
        print("prologue: header bytes staged %.0f, header + code-length code %.0f, code lengths %.0f cyc" % (k1[:, 10].mean(), k1[:, 11].mean(), k1[:, 12].mean()))
    oz.shutdown()


if __name__ == "__main__":
    main()
