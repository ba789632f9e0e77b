#!/usr/bin/env python3
"""gen_golden.py -- writes tests/golden/entries.json from outputs of the REFERENCE itself.

Run in the build container only (needs oracle/_ref/libotezip_ref.so, built by oracle/Makefile
from /root/reference).  Each fixture is data: a compressed payload, the directory fields, the
reference's own verdict for zip_fopen_index on a one-entry archive holding it (NULL or not),
and the SHA-256 of the bytes it returned.  For valid DEFLATE streams the reference
false-rejects the expected bytes come from the host's zlib instead and the fixture
says so.  Inputs are deterministic (fixed seeds)."""
import base64
import hashlib
import json
import os
import random
import struct
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyoracle as O  # noqa: E402


def raw(data, level=6, strategy=0, flush_every=None):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    if flush_every:
        out = b""
        for i in range(0, len(data), flush_every):
            out += c.compress(data[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH)
        return out + c.flush()
    return c.compress(data) + c.flush()


def text(r, n):
    words = [bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(r.randint(2, 10))) for _ in range(300)]
    out = bytearray()
    while len(out) < n:
        out += words[min(int(r.paretovariate(1.0)) - 1, 299)] + r.choice([b" ", b" ", b", ", b". "])
    return bytes(out[:n])


def main():
    O.build()
    assert O.have_ref(), "reference build missing"
    r = random.Random(20261003)
    fx = []

    def add(name, method, comp, usize, crc, flags=0, note=""):
        ref = O.ref_extract_entry(method, comp, usize, crc, ignore_zipbomb=flags & 1)
        orc = O.extract_entry(method, comp, usize, crc, ignore_zipbomb=flags & 1)
        undefined = bool(orc["verdict"] & O.V_NO_PARITY)
        ref_ok = ref is not None
        correct_ok = orc["correct_rc"] == 0
        if not undefined:
            assert (orc["rc"] == 0) == ref_ok, name
            if ref_ok:
                assert ref == orc["out"], name
        sha = None
        if correct_ok:
            out = ref if ref_ok else orc["out"]
            if not ref_ok and method == 8:
                # F2: the independent decoder is the witness
                want = zlib.decompress(comp, -15)
                assert out[:len(want)] == want, name
            sha = hashlib.sha256(out).hexdigest()
        fx.append(dict(name=name, method=method, comp=base64.b64encode(comp).decode(), uncomp_size=usize,
                       crc32=crc & 0xFFFFFFFF, flags=flags, ref_ok=ref_ok, correct_ok=correct_ok, sha256=sha,
                       undefined=undefined, note=note))

    # raw DEFLATE, the streams SURVEY 8(c)(iii) lists
    for n in (0, 1, 6, 54, 100, 1024, 4096):
        for tag, kw in (("l1", dict(level=1)), ("l6", dict()), ("l9", dict(level=9)), ("fixed", dict(strategy=zlib.Z_FIXED)),
                        ("huff", dict(strategy=zlib.Z_HUFFMAN_ONLY)), ("stored", dict(level=0)),
                        ("multi", dict(flush_every=max(n // 3, 1)))):
            d = text(r, n)
            add("deflate_%d_%s" % (n, tag), 8, raw(d, **kw), n, zlib.crc32(d))
    for tag, kw in (("l6", dict()), ("fixed", dict(strategy=zlib.Z_FIXED)), ("multi", dict(flush_every=20000))):
        d = text(r, 65536)
        add("deflate_65536_%s" % tag, 8, raw(d, **kw), 65536, zlib.crc32(d))
    # the reference's premature Z_BUF_ERROR (F2)
    k = 0
    while k < 6:
        d = bytes(r.choice(b"abc") for _ in range(r.randint(1, 40)))
        c = raw(d, 6, r.choice([0, zlib.Z_FIXED]))
        if O.extract_entry(8, c, len(d), zlib.crc32(d))["verdict"] & O.V_REF_FALSE_REJECT:
            add("deflate_f2_%d" % k, 8, c, len(d), zlib.crc32(d), note="valid stream, reference returns NULL (F2)")
            k += 1
    d = text(r, 3000)
    c = raw(d)
    add("deflate_short_stream", 8, raw(d[:1000]), 3000, zlib.crc32(d[:1000] + b"\0" * 2000), note="F12 zero-filled tail")
    add("deflate_overflow", 8, c, 2999, zlib.crc32(d))
    add("deflate_truncated", 8, c[:-2], 3000, zlib.crc32(d))
    add("deflate_crc_mismatch", 8, c, 3000, zlib.crc32(d) ^ 1, note="readable; reference only warns")
    add("deflate_empty_input", 8, b"", 0, 0)
    bomb = b"\0" * 1_100_000
    add("deflate_ratio_1000_inside_slack", 8, raw(bomb), len(bomb), zlib.crc32(bomb))
    add("deflate_zipbomb_guard", 8, raw(bomb)[:20], 5_000_000, 0, note="claims uncomp > comp*1000 + 1 MiB: rejected unread")
    add("deflate_zipbomb_guard_ignored", 8, raw(bomb)[:20], 5_000_000, 0, flags=1, note="guard off: truncated stream")
    # STORE and the private framings, 4 KiB payloads
    p = text(r, 4096)
    add("store_4k", 0, p, 4096, zlib.crc32(p))
    add("store_size_mismatch", 0, p, 4095, zlib.crc32(p))
    add("zstd93_le_magic", 93, O.frame_zstd(p), 4096, zlib.crc32(p))
    add("zstd93_be_magic", 93, O.frame_zstd(p, magic=b"\xfd\x2f\xb5\x28"), 4096, zlib.crc32(p),
        note="what the reference's own encoder writes; its decoder rejects it (F5)")
    add("zstd93_type2", 93, b"\x28\xb5\x2f\xfd\x70\x05" + struct.pack("<H", 4096) + p, 4096, zlib.crc32(p))
    add("zstd93_rle_block", 93, b"\x28\xb5\x2f\xfd\x70\x03\x01\x00x", 1, 0)
    add("zstd93_multi_block", 93, O.frame_zstd(p, block=1000), 4096, zlib.crc32(p))
    add("lzfse100", 100, O.frame_lzfse(p), 4096, zlib.crc32(p))
    add("lzfse100_empty", 100, O.frame_lzfse(b""), 0, 0, note="len 0 is an error in the reference")
    add("brotli97", 97, O.frame_brotli(p), 4096, zlib.crc32(p))
    bad = bytearray(O.frame_brotli(p))
    bad[100] ^= 1
    add("brotli97_container_crc", 97, bytes(bad), 4096, zlib.crc32(p))
    add("brotli97_zero_len_bad_magic", 97, b"XXXX\x01" + b"\0" * 8, 0, 0, note="accepted by the reference (F7)")
    # LZMA-token streams; the large ones are run-heavy so that the fixtures stay small
    for n in (1, 100):
        d = text(r, n)
        add("lzma14_%d" % n, 14, O.frame_lzma_tokens(d), n, zlib.crc32(d))
    for n in (65535, 65536, 65537):
        d = b"".join(bytes([65 + (i % 26)]) * 250 for i in range(n // 250 + 1))[:n - 7] + b"tail..!"
        add("lzma14_%d" % n, 14, O.frame_lzma_tokens(d), n, zlib.crc32(d),
            note="reference stages at most 65536 bytes (F8)" if n == 65537 else "")
    add("method_unsupported", 12, p, 4096, zlib.crc32(p))

    # reference DEFLATE encoder: only its output SIZE is meaningful (its bytes are invalid)
    from otezip_amd import corpus
    enc_sizes = []
    for seed, size in [(0, 65536), (1, 65536), (2, 65536), (3, 65536), (7, 20000), (8, 3000), (9, 100)]:
        d = corpus.gen_text(seed, size)
        ret, outb = O.ref_codec_encode(8, d, len(d) + len(d) // 8 + 75)
        assert ret == O.Z_STREAM_END
        enc_sizes.append(dict(seed=seed, size=size, ref_deflate_size=len(outb), input_crc32=zlib.crc32(d) & 0xFFFFFFFF))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "entries.json")
    with open(out, "w") as f:
        json.dump(dict(generator="oracle/gen_golden.py", reference="trufae/otezip @ /root/reference (2026-03-13)",
                       empty_zip_hex="504b0506" + "00" * 18, entries=fx,
                       ref_deflate_sizes=enc_sizes), f, indent=0)
    print("wrote %s: %d fixtures, %d bytes" % (out, len(fx), os.path.getsize(out)))


if __name__ == "__main__":
    main()
