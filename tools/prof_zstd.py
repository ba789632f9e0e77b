#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle breakdown of otz_zstd_kernel (lane 0's view of each wavefront) on real
Zstandard frames, via ozg_debug_set_profile_buffer.  Not part of any timed measurement."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import otezip_amd as oz
from otezip_amd import corpus
from otezip_amd import corpus, _native as N
from tools import zstd_corpus

NAMES = ["literals", "seq tables", "seq decode", "seq execute", "staging", "xxh64"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    arc, ents = zstd_corpus.build_zstd_real(n, size)
    oz.init(0)
    L = oz.lib()
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
                                             d_ent.data_ptr(), d_res.data_ptr(), n, s.cuda_stream, oz.BATCH_NO_DEFLATE))
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    run()
    b.record()
    torch.cuda.synchronize()
    print("plain launch: %.3f ms for %d entries of %d B" % (a.elapsed_time(b), n, size))
    L.ozg_debug_set_profile_buffer(d_prof.data_ptr())
    run()
    torch.cuda.synchronize()
    L.ozg_debug_set_profile_buffer(None)
    p = d_prof.cpu().numpy().reshape(n, 32).astype(np.float64)
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.int32).reshape(n, 4)
    print("status ok:", int((res[:, 0] == 1).sum()), "/", n)
    tot = p[:, 15].mean()
    for i, name in enumerate(NAMES):
        print("%-12s %12.0f cyc  %5.1f%%" % (name, p[:, i].mean(), 100 * p[:, i].mean() / tot))
    print("%-12s %12.0f cyc" % ("total", tot))
    oz.shutdown()


if __name__ == "__main__":
    main()
