#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: writes tests/golden/zstd_frames.json -- Zstandard frames produced by the image's
libzstd 1.4.8 (tools/zstdlib.py), each with the SHA-256 of its content, and the list of format features
every frame exercises (found by inspect_frame below, which parses block / literal / sequence headers).
The fixtures pin oracle/otz_zstd_rfc.c and the GPU decoder on machines where no libzstd is present.

    python3 oracle/gen_zstd_golden.py        (needs libzstd.so.1; the output is committed)
"""
import hashlib
import json
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import zstdlib as Z  # noqa: E402
from otezip_amd import corpus  # noqa: E402


def inspect_frame(fr):
    """Features of the frames in `fr`: block types, literal section types / stream counts, sequence table modes."""
    feats = set()
    ip = 0
    while ip < len(fr):
        magic = struct.unpack_from("<I", fr, ip)[0]
        if magic & 0xFFFFFFF0 == 0x184D2A50:
            feats.add("skippable")
            ip += 8 + struct.unpack_from("<I", fr, ip + 4)[0]
            continue
        assert magic == 0xFD2FB528
        fhd = fr[ip + 4]
        h = ip + 5
        single = (fhd >> 5) & 1
        if not single:
            h += 1
            feats.add("window_descriptor")
        else:
            feats.add("single_segment")
        h += (0, 1, 2, 4)[fhd & 3]
        fl = fhd >> 6
        fcs_len = (1 if single else 0) if fl == 0 else (2, 4, 8)[fl - 1]
        feats.add("fcs%d" % fcs_len)
        h += fcs_len
        if fhd & 4:
            feats.add("checksum")
        nblocks = 0
        while True:
            bh = fr[h] | (fr[h + 1] << 8) | (fr[h + 2] << 16)
            h += 3
            last, typ, bs = bh & 1, (bh >> 1) & 3, bh >> 3
            nblocks += 1
            feats.add(("block_raw", "block_rle", "block_compressed")[typ])
            if typ == 0:
                h += bs
            elif typ == 1:
                h += 1
            else:
                p = h
                b0 = fr[p]
                lt, sf = b0 & 3, (b0 >> 2) & 3
                feats.add(("lit_raw", "lit_rle", "lit_huffman", "lit_treeless")[lt])
                if lt < 2:
                    hdr = (1, 2, 1, 3)[sf]
                    regen = (b0 >> 3) if hdr == 1 else ((b0 >> 4) | (fr[p + 1] << 4) | ((fr[p + 2] << 12) if hdr == 3 else 0))
                    used = hdr + (regen if lt == 0 else 1)
                else:
                    if sf < 2:
                        v = b0 | (fr[p + 1] << 8) | (fr[p + 2] << 16)
                        comp, hdr = (v >> 14) & 0x3FF, 3
                    elif sf == 2:
                        v = struct.unpack_from("<I", fr, p)[0]
                        comp, hdr = v >> 18, 4
                    else:
                        v = struct.unpack_from("<I", fr, p)[0] | (fr[p + 4] << 32)
                        comp, hdr = v >> 22, 5
                    feats.add("lit_1stream" if sf == 0 else "lit_4streams")
                    if lt == 2:
                        feats.add("huf_weights_fse" if fr[p + hdr] < 128 else "huf_weights_direct")
                    used = hdr + comp
                q = p + used
                nseq = fr[q]
                q += 1
                if nseq >= 128:
                    if nseq == 255:
                        nseq = fr[q] + (fr[q + 1] << 8) + 0x7F00
                        q += 2
                    else:
                        nseq = ((nseq - 128) << 8) + fr[q]
                        q += 1
                if nseq == 0:
                    feats.add("no_sequences")
                else:
                    modes = fr[q]
                    for name, sh in (("ll", 6), ("of", 4), ("ml", 2)):
                        feats.add("%s_%s" % (name, ("predefined", "rle", "fse", "repeat")[(modes >> sh) & 3]))
                    if nseq > 64:
                        feats.add("more_than_64_sequences")
                h += bs
            if last:
                break
        if nblocks > 1:
            feats.add("multi_block")
        if fhd & 4:
            h += 4
        ip = h
    return sorted(feats)


def cases():
    rng = random.Random(20261004)
    text = corpus.gen_text(11, 400000)
    rnd = bytes(rng.randrange(256) for _ in range(6000))
    yield "empty", b"", dict(level=3)
    yield "one_byte", b"z", dict(level=3)
    yield "tiny_text", text[:60], dict(level=3)
    yield "text_200_l1", text[:200], dict(level=1)
    yield "zeros_1000", bytes(1000), dict(level=3)
    yield "zeros_300k", bytes(300000), dict(level=3)
    yield "random_6000", rnd, dict(level=3)
    yield "text_1k_l1", text[:1000], dict(level=1)
    yield "text_1k_l19", text[:1000], dict(level=19)
    yield "text_5k", text[:5000], dict(level=3)
    yield "text_5k_nochecksum", text[:5000], dict(level=3, checksum=False)
    yield "text_5k_nosize", text[:5000], dict(level=3, content_size=False)
    yield "text_20k_l9", text[:20000], dict(level=9)
    yield "text_24k_fast", text[:24000], dict(level=-3)
    yield "period7", (b"abcdefg" * 3000), dict(level=3)
    yield "period3_minmatch3", (b"xyz" * 2000 + text[:500]) * 3, dict(level=6, minmatch=3)
    yield "mixed", text[:3000] + rnd[:2000] + bytes(1500) + text[1000:4000], dict(level=5)
    yield "windowlog10", text[:30000], dict(level=3, windowlog=10)
    # several blocks (> 128 KiB of content): repeat / treeless modes show up here
    base = bytearray(text[:6000])
    mut = bytearray()
    while len(mut) < 300000: # a text repeated with a changed letter every ~150 bytes: compresses 30:1, three blocks
        for k in range(0, len(base), 150):
            base[k + rng.randrange(150) if k + 150 <= len(base) else k] = rng.choice(b"abcdefghijklmnopqrstuvwxyz ")
        mut += base
    yield "mutated_300k", bytes(mut[:300000]), dict(level=3)
    yield "mutated_300k_l1", bytes(mut[:300000]), dict(level=1)
    yield "mutated_300k_l12", bytes(mut[:300000]), dict(level=12)
    # the "repeat" modes of the three sequence tables (found by search: these seeds / levels make libzstd 1.4.8 use them)
    for seed, kw in ((2, dict(level=15)), (3, dict(level=12, windowlog=15)), (0, dict(level=19, windowlog=17))):
        r2 = random.Random(seed)
        b2 = bytearray(text[seed * 5000:seed * 5000 + 6000])
        m2 = bytearray()
        while len(m2) < 300000:
            for k in range(0, len(b2) - 150, 150):
                b2[k + r2.randrange(150)] = r2.choice(b"abcdefghijklmnopqrstuvwxyz ")
            m2 += b2
        yield "mutated_seed%d" % seed, bytes(m2[:300000]), kw
    yield "lowent_40k", bytes(rng.choice(b"aaaaabbbcd") for _ in range(40000)), dict(level=3)
    yield "bytes_ramp", bytes((i * 7 + (i >> 8)) & 255 for i in range(150000)), dict(level=4)


def craft_frames():
    """Frames libzstd's ENCODER does not produce but its decoder (and the format) accepts, written by hand:
    a compressed block without sequences, RLE literals, a Huffman code given by direct weights."""
    def frame(blocks, size):
        # single segment, 1-byte content size (size < 256), no checksum
        out = struct.pack("<IBB", 0xFD2FB528, 0x20, size)
        for k, (typ, body) in enumerate(blocks):
            bh = (1 if k == len(blocks) - 1 else 0) | (typ << 1) | (len(body) << 3)
            out += bytes([bh & 255, (bh >> 8) & 255, (bh >> 16) & 255]) + body
        return out
    # 1. compressed block: raw literals "hello, literal world" and zero sequences
    lit = b"hello, literal world"
    yield "crafted_no_sequences", frame([(2, bytes([(len(lit) << 3) | 0]) + lit + b"\x00")], len(lit)), lit
    # 2. compressed block: RLE literals (27 x 'r'), zero sequences
    yield "crafted_lit_rle", frame([(2, bytes([(27 << 3) | 1]) + b"r" + b"\x00")], 27), b"r" * 27
    # 3. Huffman literals with DIRECT weights, one stream.  Alphabet a,b,c,d with weights 3,2,1,(1 implied):
    #    max bits 3; codes (weight 1 first, natural order): c=000, d=001, b=01, a=1  (number of bits = 4 - weight)
    msg = b"abacabadabacaba" * 3 + b"dcba"
    code = {ord("c"): ("000"), ord("d"): ("001"), ord("b"): ("01"), ord("a"): ("1")}
    # the bit string as the decoder consumes it (first symbol first, most significant bit of each code first) ...
    consumed = "".join(code[ch] for ch in msg)
    # ... which is the stream read from its end: last byte's highest set bit is the marker, then consumed bits
    # follow downwards.  Lay them out: stream_bits[high .. 0] = "1" + consumed
    allbits = "1" + consumed
    pad = (-len(allbits)) % 8
    allbits = "0" * pad + allbits           # most significant bit of the LAST byte first
    by = [int(allbits[i:i + 8], 2) for i in range(0, len(allbits), 8)]
    stream = bytes(reversed(by))            # byte 0 holds the bits read last
    # weights for symbols 0 .. 'c' (the last present symbol 'd' is implied): direct representation allows <= 128
    # listed weights, so shift the alphabet down: use symbols 0,1,2,3 instead of letters
    msg2 = bytes({ord("a"): 0, ord("b"): 1, ord("c"): 2, ord("d"): 3}[c] for c in msg)
    weights = [3, 2, 1]                      # symbols 0,1,2; symbol 3 implied (weight 1)
    hdr = bytes([127 + len(weights), (weights[0] << 4) | weights[1], (weights[2] << 4)])
    comp = hdr + stream
    regen = len(msg2)
    lh = (2) | (0 << 2) | (regen << 4) | (len(comp) << 14)   # Huffman, 1 stream, 10-bit sizes
    body = bytes([lh & 255, (lh >> 8) & 255, (lh >> 16) & 255]) + comp + b"\x00"
    yield "crafted_huf_direct_weights", frame([(2, body)], regen), msg2


def main():
    out = {"generator": "oracle/gen_zstd_golden.py", "libzstd": Z.version(), "frames": []}
    all_feats = set()
    for name, data, kw in cases():
        fr = Z.compress(data, **kw)
        ok, back = Z.decompress(fr, len(data))
        assert ok and back == data
        feats = inspect_frame(fr)
        all_feats |= set(feats)
        if len(fr) > 20000:
            raise SystemExit("fixture %s too large: %d" % (name, len(fr)))
        out["frames"].append({"name": name, "frame": fr.hex(), "size": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                              "features": feats})
    # composed cases: a skippable frame in front, two frames back to back
    a = Z.compress(b"first frame " * 40, level=3)
    b = Z.compress(corpus.gen_text(5, 3000), level=3)
    skip = struct.pack("<II", 0x184D2A53, 7) + b"skipme!"
    for name, fr, data in (("skippable_then_frame", skip + b, corpus.gen_text(5, 3000)),
                           ("two_frames", a + b, b"first frame " * 40 + corpus.gen_text(5, 3000)),
                           ("frame_skippable_frame", a + skip + b, b"first frame " * 40 + corpus.gen_text(5, 3000))):
        ok, back = Z.decompress(fr, len(data))
        assert ok and back == data, name
        feats = inspect_frame(fr)
        all_feats |= set(feats)
        out["frames"].append({"name": name, "frame": fr.hex(), "size": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                              "features": feats})
    for name, fr, data in craft_frames():
        ok, back = Z.decompress(fr, len(data))
        assert ok and back == data, "libzstd rejects the hand-written frame " + name
        feats = inspect_frame(fr)
        all_feats |= set(feats)
        out["frames"].append({"name": name, "frame": fr.hex(), "size": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                              "features": feats, "crafted": True})
    out["features_covered"] = sorted(all_feats)
    path = os.path.join(ROOT, "tests", "golden", "zstd_frames.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote %s: %d frames, %d bytes; features: %s" % (path, len(out["frames"]), os.path.getsize(path), ", ".join(sorted(all_feats))))


if __name__ == "__main__":
    main()
