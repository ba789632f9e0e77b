"""Seeded parity cases shared by the emulation tests (CPU) and the GPU tests.
Every case is (method, comp_bytes, uncomp_size, dir_crc, entry_flags)."""
import os
import random
import struct
import zlib

from helpers import raw_deflate
from oracle import pyoracle as O


def _rng(seed):
    return random.Random(seed)


def small_text(r, n, alphabet=b"abcde fgh"):
    return bytes(r.choice(alphabet) for _ in range(n))


def word_text(r, n):
    words = [bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(r.randint(2, 10))) for _ in range(400)]
    out = bytearray()
    while len(out) < n:
        out += words[min(int(r.paretovariate(1.0)) - 1, 399)] + r.choice([b" ", b" ", b" ", b", ", b". ", b"]]\n[["])
    return bytes(out[:n])


def crc(d):
    return zlib.crc32(d) & 0xFFFFFFFF


def deflate_valid(seed=1, sizes=(0, 1, 2, 6, 54, 100, 1000, 4096, 20000)):
    """Valid raw-DEFLATE streams: levels 0/1/6/9, fixed, huffman-only, rle, multi-block."""
    r = _rng(seed)
    out = []
    for n in sizes:
        for kw in (dict(), dict(level=1), dict(level=9), dict(level=0), dict(strategy=zlib.Z_FIXED),
                   dict(strategy=zlib.Z_HUFFMAN_ONLY), dict(strategy=zlib.Z_RLE), dict(flush_every=777)):
            d = word_text(r, n) if r.random() < 0.6 else small_text(r, n)
            out.append((8, raw_deflate(d, **kw), len(d), crc(d), 0))
    for d in (os.urandom(5000), b"A" * 30000, (b"abc" * 9000)[:25000], bytes(range(256)) * 20):
        out.append((8, raw_deflate(d), len(d), crc(d), 0))
    return out


def deflate_tiny_f2(seed=2, n=60):
    """Small low-entropy streams: a good share end with a short EOB code and hit the
    reference's premature Z_BUF_ERROR."""
    r = _rng(seed)
    out = []
    for _ in range(n):
        d = small_text(r, r.randint(0, 80), b"abc")
        out.append((8, raw_deflate(d, 6, r.choice([0, zlib.Z_FIXED])), len(d), crc(d), 0))
    return out


def deflate_edge(seed=3):
    """Wrong sizes, wrong CRCs, truncation, trailing bytes, empty input, zip-bomb guard."""
    r = _rng(seed)
    out = []
    base = []
    for n in (1, 100, 3000, 20000):
        d = word_text(r, n)
        for kw in (dict(), dict(strategy=zlib.Z_FIXED), dict(level=0), dict(flush_every=700)):
            base.append((d, raw_deflate(d, **kw)))
    for d, c in base:
        out.append((8, c, len(d), crc(d) ^ 5, 0))                    # CRC mismatch: warning, still readable
        out.append((8, c, len(d) - 1, crc(d), 0))                    # stream overflows the entry
        out.append((8, c, max(len(d) - 300, 0), 0, 0))
        out.append((8, c, len(d) + 37, crc(d + b"\0" * 37), 0))      # F12: short stream, zero-filled tail
        out.append((8, c[:-1], len(d), crc(d), 0))                   # truncated
        out.append((8, c[:len(c) // 2], len(d), crc(d), 0))
        out.append((8, c + b"\x00\x55", len(d), crc(d), 0))          # trailing bytes
    out.append((8, b"", 0, 0, 0))
    out.append((8, b"", 10, 0, 0))
    big = b"\0" * 1_200_000
    cb = raw_deflate(big)
    out.append((8, cb, len(big), crc(big), 0))                       # ratio ~1000 but inside the 1 MiB slack: fine
    out.append((8, cb[:20], 5_000_000, 0, 0))                        # claims > comp*1000 + 1 MiB: zip-bomb guard
    out.append((8, cb[:20], 5_000_000, 0, 1))                        # guard overridden: now it is just truncated
    out.append((0, b"x" * 20, 5_000_000, 0, 0))                      # the guard comes before the method switch
    return out


def deflate_fuzz(seed=4, n=120):
    r = _rng(seed)
    base = []
    for m in (100, 3000, 20000):
        d = word_text(r, m)
        for kw in (dict(), dict(strategy=zlib.Z_FIXED), dict(flush_every=900)):
            base.append((d, raw_deflate(d, **kw)))
    out = []
    for _ in range(n):
        d, c = r.choice(base)
        b = bytearray(c)
        for _ in range(r.randint(1, 3)):
            b[r.randrange(len(b))] ^= 1 << r.randrange(8)
        out.append((8, bytes(b), len(d), crc(d), 0))
    for _ in range(n // 3):
        out.append((8, bytes(r.getrandbits(8) for _ in range(r.randint(1, 200))), r.randint(0, 3000), 0, 0))
    return out


def deflate_large(seed=5):
    """Entries of several rounds and more than one 32 KiB window (what the fallback routine's LDS ring flushes
    in mid-entry, and what the apply kernel writes out round by round), multi-round blocks."""
    r = _rng(seed)
    out = []
    for n in (65536, 65537, 100000, 300000):
        d = word_text(r, n)
        out.append((8, raw_deflate(d), len(d), crc(d), 0))
    d = os.urandom(150000)
    out.append((8, raw_deflate(d), len(d), crc(d), 0))               # stored blocks
    d = b"Z" * 400000
    out.append((8, raw_deflate(d), len(d), crc(d), 0))               # 258-byte matches at distance 1
    return out


def deflate_ring_wrap(seed=7, ring=57344):
    """Overlapping matches (distance < copy chunk) whose first period straddles the end of the
    fallback routine's output ring (OTZ_RING bytes: the copy has to split at the wrap without
    losing the period) -- for the apply kernel, which keeps a linear round buffer, the same streams put
    such matches next to a round's start, where the source is partly behind the buffer."""
    r = _rng(seed)
    out = []
    for k, period in ((1, 7), (2, 3), (5, 9), (1, 2), (3, 15), (0, 1), (8, 13)):
        d = word_text(r, ring - k) + bytes(r.sample(range(1, 255), period)) * 23 + word_text(r, 500)
        out.append((8, raw_deflate(d), len(d), crc(d), 0))
    return out


def deflate_far_matches(seed=8, n=6):
    """Matches whose source lies far back (up to the full 32 KiB window): the apply kernel keeps only
    the last few KiB before the round in LDS and reads older bytes from the entry's own output in
    global memory -- at every alignment, with lengths 3..258, and next to fresh literals."""
    r = _rng(seed)
    out = []
    for k in range(n):
        d = bytearray(word_text(r, 30000 + 997 * k))
        while len(d) < 90000 + 5000 * k:
            back = r.randrange(1, min(len(d), 32768 - 300))
            ln = r.choice((3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100, 257, 258))
            at = len(d) - back
            d += d[at:at + ln] if back >= ln else (d[at:] * (ln // back + 1))[:ln]
            d += bytes(r.randrange(256) for _ in range(r.randrange(0, 4)))
        d = bytes(d)
        out.append((8, raw_deflate(d), len(d), crc(d), 0))
    return out


def containers(seed=6):
    """STORE and the private framings of methods 93/100/97/14, good and bad."""
    r = _rng(seed)
    out = []
    frames = {93: O.frame_zstd, 100: O.frame_lzfse, 97: O.frame_brotli, 14: O.frame_lzma_tokens}
    for size in (0, 1, 5, 15, 16, 17, 100, 4095, 4096, 4097, 65535, 65536, 70001, 262144):
        d = word_text(r, size) if size < 100000 else os.urandom(size)
        out.append((0, d, len(d), crc(d), 0))
        for m, fr in frames.items():
            c = fr(d)
            out.append((m, c, len(d), crc(d), 0))
            if size in (5, 4097, 70001):
                out.append((m, c, len(d), crc(d) ^ 1, 0))
                out.append((m, c, len(d) + 3, crc(d), 0))
                out.append((m, c, max(len(d) - 3, 0), crc(d), 0))
                out.append((m, c[:-1], len(d), crc(d), 0))
                out.append((m, c + b"xyz", len(d), crc(d), 0))
                for _ in range(6):
                    b = bytearray(c)
                    b[r.randrange(min(len(b), 24))] ^= 1 << r.randrange(8)
                    out.append((m, bytes(b), len(d), crc(d), 0))
    p = os.urandom(1000)
    z = b"\x28\xb5\x2f\xfd\x70"
    out += [
        (0, p, 999, crc(p), 0),                                                   # STORE size mismatch
        (93, O.frame_zstd(p, magic=b"\xfd\x2f\xb5\x28"), 1000, crc(p), 0),       # the reference ENcoder's magic: rejected (F5)
        (93, z + b"\x05" + struct.pack("<H", 1000) + p, 1000, crc(p), 0),        # "compressed" type-2 block == memcpy
        (93, z + b"\x04" + struct.pack("<H", 1000) + p, 1000, crc(p), 0),        # no last block
        (93, z + b"\x05\x00\x00", 0, 0, 0),                                       # empty type-2 block
        (93, z + b"\x03" + struct.pack("<H", 1) + b"x", 1, 0, 0),                # RLE block type
        (93, z + b"\x01" + struct.pack("<H", 1000) + p + b"junk", 1000, crc(p), 0),
        (97, b"XXXX\x01" + b"\0" * 8, 0, 0, 0),                                   # zero length field, bad magic (F7 oddity)
        (97, b"XXXX\x01" + b"\0" * 8 + b"abcd", 5, 0, 0),
        (14, O.frame_lzma_tokens(b"helloworld", hdr_size=0), 10, crc(b"helloworld"), 0),
        (14, O.frame_lzma_tokens(b"helloworld", hdr_size=11), 10, crc(b"helloworld"), 0),
        (14, O.frame_lzma_tokens(b"helloworld", hdr_size=2**64 - 1), 10, crc(b"helloworld"), 0),
        (14, O.frame_lzma_tokens(b"")[:13], 0, 0, 0),
        (14, O.frame_lzma_tokens(b"a" * 65537), 65537, crc(b"a" * 65537), 0),    # beyond the reference's 64 KiB staging (F8)
        (14, b"\x5d\0\0\0\x01" + struct.pack("<Q", 0) + b"\x01\x00", 0, 0, 0),
        (12, p, 1000, crc(p), 0),                                                 # unsupported method
    ]
    return out
