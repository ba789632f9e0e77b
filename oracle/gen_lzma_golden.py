#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: writes tests/golden/lzma_streams.json -- ZIP method-14 payloads as APPNOTE 5.8.8 defines
them (version, properties size, properties, raw LZMA1 stream), the streams written by liblzma through
Python's lzma module (FORMAT_RAW / FILTER_LZMA1), each with the SHA-256 of its content.  The reference has
no LZMA decoder to restate (ref src/lib/lzma-dec.inc.c:102-155 is a private token format); liblzma is the
third-party implementation the GPU decoder (otezip_amd/csrc/otz_lzma.h) is pinned against.

    python3 oracle/gen_lzma_golden.py      (the output is committed)
"""
import hashlib
import json
import lzma
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from otezip_amd import corpus  # noqa: E402


def zip_lzma(data, lc=3, lp=0, pb=2, dict_size=1 << 20, preset=6):
    """A method-14 payload: 09 14 | 05 00 | properties | raw LZMA1 stream (liblzma ends it with the end marker)."""
    f = [{"id": lzma.FILTER_LZMA1, "preset": preset, "lc": lc, "lp": lp, "pb": pb, "dict_size": dict_size}]
    raw = lzma.compress(data, format=lzma.FORMAT_RAW, filters=f)
    back = lzma.decompress(raw, format=lzma.FORMAT_RAW, filters=f)
    assert back == data
    return b"\x09\x14\x05\x00" + bytes([(pb * 5 + lp) * 9 + lc]) + struct.pack("<I", dict_size) + raw


def cases():
    rng = random.Random(20261004)
    text = corpus.gen_text(21, 120000)
    yield "empty", b"", {}
    yield "one_byte", b"q", {}
    yield "tiny", text[:40], {}
    yield "zeros_5000", bytes(5000), {}
    yield "random_3000", bytes(rng.randrange(256) for _ in range(3000)), {}
    yield "text_2k", text[:2000], {}
    yield "text_2k_lc0_lp2_pb0", text[:2000], dict(lc=0, lp=2, pb=0)
    yield "text_2k_lc4_pb4", text[:2000], dict(lc=4, lp=0, pb=4, preset=1)
    yield "text_2k_lc2_lp2", text[:2000], dict(lc=2, lp=2, pb=2, preset=9)
    yield "text_20k", text[:20000], dict(preset=9)
    yield "period5", b"abcde" * 4000, {}
    yield "reps", (text[:300] + b"--" + text[100:400] + b"++") * 30, dict(preset=9)
    yield "ramp", bytes((i * 7 + (i >> 8)) & 255 for i in range(30000)), dict(dict_size=4096)
    yield "mixed", text[:3000] + bytes(rng.randrange(256) for _ in range(1500)) + bytes(2000) + text[500:3500], {}


def main():
    out = {"generator": "oracle/gen_lzma_golden.py", "liblzma": "python lzma module (liblzma)", "streams": []}
    for name, data, kw in cases():
        blob = zip_lzma(data, **kw)
        assert len(blob) < 16000, name
        out["streams"].append({"name": name, "payload": blob.hex(), "size": len(data), "sha256": hashlib.sha256(data).hexdigest()})
    path = os.path.join(ROOT, "tests", "golden", "lzma_streams.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote %s: %d streams, %d bytes" % (path, len(out["streams"]), os.path.getsize(path)))


if __name__ == "__main__":
    main()
