#!/usr/bin/env python3
"""Diagnostic: encode throughput and the per-phase shader-cycle breakdown of otz_encode_kernel
(lane 0 of each wavefront), via ozg_debug_set_profile_buffer.  Not part of the timed benchmark.
usage: prof_encode.py [buffers] [bytes]"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import otezip_amd as oz
from otezip_amd import corpus, _native as N

NAMES = ["load+hash", "measure", "select", "tok+hist", "litlen code", "dist code", "rle+cl+cost", "header", "emit"]


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    bufs = corpus.text_buffers(nb, size, first=50000)
    oz.init(0)
    L = oz.lib()
    dev = torch.device("cuda", 0)
    ents = (N.OzgEncEntry * nb)()
    src = bytearray()
    off = 0
    for i, b in enumerate(bufs):
        ents[i].src_off = len(src)
        src += b
        ents[i].src_size = len(b)
        ents[i].method = 8
        cap = int(L.ozg_compress_bound(8, len(b)))
        ents[i].dst_cap = cap
        ents[i].dst_off = off
        off += (cap + 15) & ~15
    d_src = torch.frombuffer(src + bytearray(64), dtype=torch.uint8).to(dev)
    d_dst = torch.empty(off + 64, dtype=torch.uint8, device=dev)
    d_ent = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(dev)
    d_res = torch.zeros(nb * 16, dtype=torch.uint8, device=dev)
    d_prof = torch.zeros(nb * 32, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream()

    def enc():
        N.check(L.ozg_encode_batch_device(d_src.data_ptr(), d_src.numel(), d_dst.data_ptr(), d_dst.numel(),
                                          d_ent.data_ptr(), d_res.data_ptr(), nb, s.cuda_stream))
    enc()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        enc()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint32).reshape(-1, 4)
    host = d_dst.cpu().numpy()
    assert (res[:, 0] == 1).all()
    for i in range(0, nb, max(1, nb // 20)):
        assert zlib.decompress(host[ents[i].dst_off:ents[i].dst_off + res[i, 1]].tobytes(), -15) == bufs[i]
    inb = sum(len(x) for x in bufs)
    print("encode: %.3f ms for %d x %d B = %.1f GiB/s in, ratio %.3f" % (ms, nb, size, inb / ms / 1e-3 / 2**30, inb / int(res[:, 1].sum())))
    L.ozg_debug_set_profile_buffer(d_prof.data_ptr())
    enc()
    torch.cuda.synchronize()
    L.ozg_debug_set_profile_buffer(None)
    p = d_prof.cpu().numpy().reshape(nb, 32).astype(np.float64)
    tot = p[:, 15].mean()
    for i, name in enumerate(NAMES):
        print("%-12s %10.0f cyc  %5.1f%%" % (name, p[:, i].mean(), 100 * p[:, i].mean() / tot))
    print("%-12s %10.0f cyc" % ("total", tot))
    oz.shutdown()


if __name__ == "__main__":
    main()
