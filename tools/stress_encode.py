#!/usr/bin/env python3
"""Randomised encoder stress on a GPU box (not a pytest): N seeded buffers of every kind -- text,
binary, runs, periodic, skewed byte frequencies, incompressible; 0 B .. 400 KB -- through
ozg_encode_batch_host; every DEFLATE payload must inflate (host zlib) to its input, every CRC must
match, and the GPU decoder must read the whole batch back.  Usage: stress_encode.py [n] [seed]"""
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import otezip_amd as oz  # noqa: E402
from otezip_amd import corpus  # noqa: E402


def payload(r):
    kind = r.randrange(7)
    n = r.choice([0, 1, 3, 4, 5, 63, 64, 65, 255, 256, 257, 258, 259, 1000, 4096, 65535, 65536, 65537, 131072, 400000]) \
        if r.random() < 0.4 else r.randrange(0, 100000)
    if kind == 0:
        return corpus.gen_text(r.randrange(1 << 30), n)
    if kind == 1:
        return bytes(r.getrandbits(8) for _ in range(min(n, 30000)))
    if kind == 2:
        unit = bytes(r.getrandbits(8) for _ in range(r.randint(1, 300)))
        return (unit * (n // len(unit) + 1))[:n]
    if kind == 3:
        out = bytearray()
        while len(out) < n:
            out += bytes([r.randrange(3)]) * r.randint(1, 600)
        return bytes(out[:n])
    if kind == 4:
        return bytes(min(255, int(r.expovariate(0.08))) for _ in range(min(n, 40000)))
    if kind == 5:
        t = corpus.gen_text(r.randrange(1 << 30), max(n // 3, 1))
        return (t + t[::-1] + t)[:n]
    return bytes(n)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    r = random.Random(seed)
    bufs = [payload(r) for _ in range(n)]
    oz.init(0)
    out = oz.encode_batch_host(bufs, 8)
    tin = tout = nd = 0
    items, blob = [], bytearray()
    for i, (b, (method, pay, crc)) in enumerate(zip(bufs, out)):
        assert crc == zlib.crc32(b) & 0xFFFFFFFF, i
        if method == 8:
            d = zlib.decompressobj(-15)
            assert d.decompress(pay) + d.flush() == b and d.eof, i
            nd += 1
        else:
            assert method == 0 and pay == b, i
        tin += len(b)
        tout += len(pay)
        items.append((method, len(blob), len(pay), len(b), crc))
        blob += pay
    dec = oz.decode_batch_host(bytes(blob) + b"\0" * 16, items)
    for i, b in enumerate(bufs):
        assert dec.ok(i) and dec.data(i) == b and dec.results[i].flags == 0, i
    print("ENCODE OK: %d buffers, %d deflated, %d stored, %.1f MB -> %.1f MB" % (n, nd, n - nd, tin / 1e6, tout / 1e6))
    oz.shutdown()


if __name__ == "__main__":
    main()
