#!/usr/bin/env python3
"""Randomised parity stress on a GPU box (not part of the timed benchmark, not a pytest):
N seeded DEFLATE entries -- every zlib level and strategy, text / binary / repetitive /
incompressible payloads of 0 B .. 300 KB, multi-block streams, truncated and bit-flipped
copies -- decoded through ozg_decode_batch_host and compared with the oracle exactly as
tests/test_gpu_parity.py does.  Usage: python3 tools/stress_parity.py [n_cases] [seed]"""
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import otezip_amd as oz  # noqa: E402
from otezip_amd import corpus  # noqa: E402
from test_gpu_parity import assert_parity, run_gpu  # noqa: E402


def payload(r):
    kind = r.randrange(6)
    n = r.choice([0, 1, 2, 5, 17, 100, 1000, 4096, 20000, 65536, 70000, 150000, 300000]) if r.random() < 0.5 \
        else r.randrange(0, 90000)
    if kind == 0:
        return corpus.gen_text(r.randrange(1 << 30), n)
    if kind == 1:
        return bytes(r.getrandbits(8) for _ in range(min(n, 40000)))
    if kind == 2:
        unit = bytes(r.getrandbits(8) for _ in range(r.randint(1, 40)))
        return (unit * (n // len(unit) + 1))[:n]
    if kind == 3:
        return bytes(r.choice(b"ab \n") for _ in range(min(n, 60000)))
    if kind == 4:
        w = [bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(r.randint(2, 9))) for _ in range(200)]
        return b" ".join(r.choice(w) for _ in range(n // 5 + 1))[:n]
    return bytes([r.randrange(4)] * r.randint(1, 600)) * (n // 600 + 1)


def encode(r, d):
    level = r.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
    strategy = r.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
    c = zlib.compressobj(level, zlib.DEFLATED, -15, r.choice([1, 8, 9]), strategy)
    out = b""
    step = r.choice([0, 0, 700, 5000, 33000])
    if step:
        for i in range(0, len(d), step):
            out += c.compress(d[i:i + step]) + c.flush(r.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))
    else:
        out += c.compress(d)
    return out + c.flush()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    r = random.Random(seed)
    items = []
    for _ in range(n):
        d = payload(r)
        c = encode(r, d)
        crc = zlib.crc32(d) & 0xFFFFFFFF
        items.append((8, c, len(d), crc, 1))
        roll = r.random()
        if roll < 0.15 and len(c) > 2:
            items.append((8, c[:r.randrange(1, len(c))], len(d), crc, 1))           # truncated
        elif roll < 0.35 and c:
            b = bytearray(c)
            for _ in range(r.randint(1, 3)):
                b[r.randrange(len(b))] ^= 1 << r.randrange(8)
            items.append((8, bytes(b), len(d), crc, 1))                              # bit flips
        elif roll < 0.42:
            items.append((8, c, max(0, len(d) + r.choice([-1, 1, 50, -50])), crc, 1))  # wrong declared size
    oz.init(0)
    n_ok = n_f2 = 0
    for i in range(0, len(items), 256):
        part = items[i:i + 256]
        b = run_gpu(oz, part)
        a, f = assert_parity(part, b, "stress[%d:]" % i)
        n_ok += a
        n_f2 += f
        print("checked %d / %d entries, %d accepted so far" % (min(i + 256, len(items)), len(items), n_ok), flush=True)
    print("PARITY OK: %d entries, %d accepted, %d reference false-rejects repaired" % (len(items), n_ok, n_f2))


if __name__ == "__main__":
    main()
