#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md / profiles (not the headline bench):
  * DEFLATE encode, 10 000 x 64 KiB buffers resident in HBM (BASELINE configs[4] shape)
  * private-zstd framing decode, 10 000 x 256 KiB (configs[2] shape, scaled by --zstd-entries)
  * lzfse + BROT framing decode, 64 KiB payloads (configs[3] shape)
  * host-buffer (PCIe-inclusive) DEFLATE decode through ozg_decode_batch_host"""
import ctypes as C
import json
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import otezip_amd as oz
from otezip_amd import corpus
from otezip_amd import corpus, _native as N
from tools import zstd_corpus


def timed(fn, steps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e-3


def decode_resident(arc, ents, flags):
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents]
    et, dst_len = oz.make_entries(tab)
    dev = torch.device("cuda", 0)
    d_src = torch.frombuffer(bytearray(arc), dtype=torch.uint8).to(dev)
    d_dst = torch.empty(dst_len + 64, dtype=torch.uint8, device=dev)
    d_ent = torch.frombuffer(bytearray(bytes(et)), dtype=torch.uint8).to(dev)
    d_res = torch.zeros(len(tab) * 16, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream()
    L = oz.lib()

    def run():
        N.check(L.ozg_decode_batch_device_ex(d_src.data_ptr(), d_src.numel(), d_dst.data_ptr(), d_dst.numel(),
                                             d_ent.data_ptr(), d_res.data_ptr(), len(tab), s.cuda_stream, flags))
    t = timed(run)
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint32).reshape(-1, 4)
    assert (res[:, 0] == 1).all() and all(res[i, 2] == e["crc32"] for i, e in enumerate(ents))
    comp = sum(e["comp_size"] for e in ents)
    out = sum(e["uncomp_size"] for e in ents)
    return dict(ms=t * 1e3, out_GiBps=out / t / 2**30, algo_GBps=(comp + out) / t / 1e9, frac=(comp + out) / t / 8e12)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    report = {}
    # archives first (forks), GPU afterwards
    arc_z, ents_z = corpus.build_archive(max(n // 4, 100), 262144, methods=(93,))
    arc_m, ents_m = corpus.build_archive(n, 65536, methods=(100, 97))
    arc_d, ents_d = corpus.build_archive(min(n, 4000), 65536, methods=(8,))
    bufs = corpus.text_buffers(min(n, 10000), 65536, first=50000)
    arc_rz = ents_rz = None
    try:  # real Zstandard frames (libzstd of the image makes them): configs[2] shape, and 64 KiB entries like configs[1]
        arc_rz, ents_rz = zstd_corpus.build_zstd_real(max(n // 4, 100), 262144)
        arc_rz64, ents_rz64 = zstd_corpus.build_zstd_real(n, 65536)
    except Exception as ex:  # no libzstd: nothing to measure
        print("real zstd corpus unavailable: %r" % (ex,), file=sys.stderr)
    arc_br = arc_lz = None
    try:  # the other two real formats: Brotli streams and LZMA1 (one work-item per stream: thousands of streams per launch)
        arc_br, ents_br = zstd_corpus.build_brotli_real(n, 65536)
        arc_lz, ents_lz = zstd_corpus.build_lzma_real(n, 65536)
        arc_lf, ents_lf = zstd_corpus.build_lzfse_real(min(n, 4000), 65536)
    except Exception as ex:
        print("real Brotli / LZMA corpus unavailable: %r" % (ex,), file=sys.stderr)
    oz.init(0)
    L = oz.lib()
    dev = torch.device("cuda", 0)
    report["zstd93_256KiB_decode"] = decode_resident(arc_z, ents_z, oz.BATCH_NO_DEFLATE | oz.BATCH_NO_LZMA | oz.BATCH_NO_BROTLI | oz.BATCH_NO_LZFSE)
    report["lzfse100_brot97_64KiB_decode"] = decode_resident(arc_m, ents_m, oz.BATCH_NO_DEFLATE | oz.BATCH_NO_ZSTD | oz.BATCH_NO_LZMA)
    if arc_rz is not None:
        report["zstd_rfc8878_256KiB_decode"] = dict(decode_resident(arc_rz, ents_rz, oz.BATCH_NO_DEFLATE | oz.BATCH_NO_LZMA | oz.BATCH_NO_BROTLI | oz.BATCH_NO_LZFSE), entries=len(ents_rz),
                                                      note="real Zstandard frames, libzstd level 3 with content checksum")
        report["zstd_rfc8878_64KiB_decode"] = dict(decode_resident(arc_rz64, ents_rz64, oz.BATCH_NO_DEFLATE | oz.BATCH_NO_LZMA | oz.BATCH_NO_BROTLI | oz.BATCH_NO_LZFSE), entries=len(ents_rz64),
                                                     note="real Zstandard frames, libzstd level 3 with content checksum")
    if arc_br is not None:
        report["brotli_rfc7932_64KiB_decode"] = dict(decode_resident(arc_br, ents_br, oz.BATCH_NO_DEFLATE | oz.BATCH_NO_ZSTD | oz.BATCH_NO_LZMA | oz.BATCH_NO_LZFSE),
                                                     entries=len(ents_br), note="real Brotli streams, libbrotlienc quality 5, window 22")
        report["lzma1_appnote_64KiB_decode"] = dict(decode_resident(arc_lz, ents_lz, oz.BATCH_NO_DEFLATE | oz.BATCH_NO_ZSTD | oz.BATCH_NO_BROTLI | oz.BATCH_NO_LZFSE),
                                                    entries=len(ents_lz), note="APPNOTE 5.8.8 LZMA1, liblzma preset 1, lc3 lp0 pb2")
        report["lzfse_model_64KiB_decode"] = dict(decode_resident(arc_lf, ents_lf, oz.BATCH_NO_DEFLATE | oz.BATCH_NO_ZSTD | oz.BATCH_NO_LZMA | oz.BATCH_NO_BROTLI),
                                                  entries=len(ents_lf), note="LZFSE streams (bvx2 blocks) written by tools/lzfse_model.py: parity unpinned")
    # encode, resident
    nb = len(bufs)
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
    s = torch.cuda.current_stream()

    def enc():
        N.check(L.ozg_encode_batch_device(d_src.data_ptr(), d_src.numel(), d_dst.data_ptr(), d_dst.numel(),
                                          d_ent.data_ptr(), d_res.data_ptr(), nb, s.cuda_stream))
    t = timed(enc, steps=3)
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint32).reshape(-1, 4)
    host = d_dst.cpu().numpy()
    assert (res[:, 0] == 1).all()
    for i in range(0, nb, max(1, nb // 50)):
        assert zlib.decompress(host[ents[i].dst_off:ents[i].dst_off + res[i, 1]].tobytes(), -15) == bufs[i]
    comp = int(res[:, 1].sum())
    inb = sum(len(b) for b in bufs)
    report["deflate_encode_64KiB"] = dict(ms=t * 1e3, in_GiBps=inb / t / 2**30, ratio=inb / comp,
                                          algo_GBps=(inb + comp) / t / 1e9, frac=(inb + comp) / t / 8e12, buffers=nb)
    # PCIe-inclusive decode through host buffers: only the C call is timed (buffers exist and are touched)
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents_d]
    et, dst_len = oz.make_entries(tab)
    nt = len(tab)
    out_bytes = sum(e["uncomp_size"] for e in ents_d)
    h_src = np.frombuffer(bytearray(arc_d) + bytearray(64), dtype=np.uint8)
    h_dst = np.ones(dst_len + 64, dtype=np.uint8)
    h_res = (N.OzgResult * nt)()

    def wall(fn, reps=3):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    def pageable():
        N.check(L.ozg_decode_batch_host(h_src.ctypes.data, len(arc_d), h_dst.ctypes.data, dst_len, et, h_res, nt))
    dt = wall(pageable)
    assert all(h_res[i].status == 1 for i in range(nt))
    report["deflate_decode_host_pageable"] = dict(
        ms=dt * 1e3, out_GiBps=out_bytes / dt / 2**30, entries=nt,
        note="ozg_decode_batch_host, pageable caller buffers: H2D + kernels + D2H, synchronous")
    # the same through pinned buffers and the asynchronous job interface (what zip_api.c drives)
    L.ozg_malloc_pinned.restype = C.c_void_p
    p_src = L.ozg_malloc_pinned(C.c_size_t(len(arc_d) + 64))
    p_dst = L.ozg_malloc_pinned(C.c_size_t(dst_len + 64))
    p_res = L.ozg_malloc_pinned(C.c_size_t(nt * 16))
    C.memmove(p_src, h_src.ctypes.data, len(arc_d))

    def pinned():
        job = C.c_void_p()
        N.check(L.ozg_decode_submit(C.c_void_p(p_src), len(arc_d), C.c_void_p(p_dst), dst_len, et, C.c_void_p(p_res), nt, C.byref(job)))
        N.check(L.ozg_job_wait(job))
    dt = wall(pinned)
    r = np.frombuffer((C.c_uint8 * (nt * 16)).from_address(p_res), dtype=np.int32).reshape(-1, 4)
    assert (r[:, 0] == 1).all()
    report["deflate_decode_host_pinned"] = dict(
        ms=dt * 1e3, out_GiBps=out_bytes / dt / 2**30, entries=nt,
        note="ozg_decode_submit + ozg_job_wait on pinned buffers (one batch, nothing overlapped)")
    for q in (p_src, p_dst, p_res):
        L.ozg_free_pinned(C.c_void_p(q))
    print(json.dumps(report))
    oz.shutdown()


if __name__ == "__main__":
    main()
