#!/usr/bin/env python3
"""Diagnostic: when and where the wavefronts of otz_index_kernel (profile build) ran -- per-entry start / end
clocks and HW_ID -- to see how many are resident per CU over the kernel's lifetime."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import otezip_amd as oz
from otezip_amd import corpus, _native as N


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    oz.init(0)
    L = oz.lib()
    arc, ents = corpus.build_archive(n, 65536)
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
    L.ozg_debug_set_profile_buffer(d_prof.data_ptr())
    run()
    torch.cuda.synchronize()
    L.ozg_debug_set_profile_buffer(None)
    p = d_prof.cpu().numpy().reshape(n, 32)[:, 16:].astype(np.uint64)
    start, end, where = p[:, 10].astype(np.int64), p[:, 11].astype(np.int64), p[:, 12]
    dur = end - start
    hw = (where & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((where >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    simd = (hw >> 4) & 3
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("entries %d, distinct CUs %d, distinct XCCs %d" % (n, len(np.unique(key)), len(np.unique(xcc))))
    print("wave duration (cycles): min %d p10 %d median %d p90 %d max %d" % tuple(np.percentile(dur, [0, 10, 50, 90, 100]).astype(int)))
    occ = []
    for k in np.unique(key):
        m = key == k
        t0 = start[m].min()
        span = end[m].max() - t0
        occ.append(dur[m].sum() / span)
    occ = np.array(occ)
    print("per CU: sum of wave durations / (last end - first start): min %.1f median %.1f max %.1f" % (occ.min(), np.median(occ), occ.max()))
    spans = np.array([end[key == k].max() - start[key == k].min() for k in np.unique(key)])
    print("per CU span (cycles): min %d median %d max %d" % (spans.min(), np.median(spans), spans.max()))
    for k in np.unique(key)[:3]:
        m = key == k
        t0 = start[m].min()
        span = end[m].max() - t0
        res = []
        for x in np.linspace(0, span, 17)[:-1]:
            res.append(int(((start[m] - t0 <= x) & (x < end[m] - t0)).sum()))
        order = np.argsort(start[m])
        print("CU key %d: %d waves, resident at 16 evenly spaced times %s; SIMD counts %s" % (k, m.sum(), res, np.bincount(simd[m], minlength=4).tolist()))
        print("   starts (kcycles):", ((start[m][order] - t0) // 1000).tolist())
        print("   durations (kcycles):", (dur[m][order] // 1000).tolist())
    slow = dur > np.percentile(dur, 90)
    for name, col in (("spans", 6), ("fix_iters", 7), ("rewalks", 8), ("walkA cyc", 3), ("fixpoint cyc", 4), ("accept cyc", 5), ("hdr cyc", 0), ("tables cyc", 1)):
        v = p[:, col].astype(np.float64)
        print("%-13s all %10.1f   slowest 10%% %10.1f" % (name, v.mean(), v[slow].mean()))
    per_cu = np.bincount(np.unique(key, return_inverse=True)[1])
    print("waves per CU: min %d median %d max %d" % (per_cu.min(), np.median(per_cu), per_cu.max()))
    oz.shutdown()


if __name__ == "__main__":
    main()
