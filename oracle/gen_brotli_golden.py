#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: writes tests/golden/brotli_streams.json -- Brotli streams (RFC 7932) written by the
image's libbrotlienc 1.0.9 at several qualities, window sizes and modes, plus a few streams put together by
hand for what the encoder never emits (metadata meta-blocks, an empty last meta-block behind data, the
smallest window), each with the SHA-256 of its content as libbrotlidec 1.0.9 decodes it.  The reference has
no Brotli decoder to restate (ref src/lib/brotli.inc.c:57-80 is a "BROT" container around stored bytes);
libbrotli is the third-party implementation the GPU decoder (otezip_amd/csrc/otz_brotli.h) is pinned against.

    python3 oracle/gen_brotli_golden.py      (the output is committed)
"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from otezip_amd import corpus  # noqa: E402
from tools import brotlilib as B  # noqa: E402

# prose and markup full of words of the static dictionary, in the casings and with the affixes its transforms make
WEB = (b"<!DOCTYPE html><html><head><meta charset=\"utf-8\"><title>The International Information Government</title>"
       b"<link rel=\"stylesheet\" type=\"text/css\" href=\"style.css\"></head><body><div class=\"container\">"
       b"<p>The development of the university and the department of education in the United States. </p>"
       b"<p>However, the following information is available: International, NATIONAL, Government's, "
       b"something, Something, SOMETHING, between the different countries. Because of this, there are many "
       b"important questions about the history of the world.</p><a href=\"http://www.example.com/index.html\">"
       b"Click here</a> for more information about the company and its products and services.</div></body></html>\n")


class Bits:
    """LSB-first bit writer (RFC 7932 section 1.5.2)."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, n):
        self.acc |= v << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def done(self):
        self.align()
        return bytes(self.out)


def crafted():
    """Streams the encoder does not write.  Each is checked below against libbrotlidec like every other."""
    # WBITS = 16 ('0'), metadata meta-block of 5 bytes, uncompressed meta-block, metadata of 0 bytes, ISLAST+ISLASTEMPTY
    w = Bits()
    w.put(0, 1)
    w.put(0, 1); w.put(3, 2); w.put(0, 1); w.put(1, 2); w.put(4, 8); w.align()  # noqa: E702  MSKIPLEN-1 = 4
    w.out += b"META!"
    payload = b"stored bytes behind a metadata block"
    w.put(0, 1); w.put(0, 2); w.put(len(payload) - 1, 16); w.put(1, 1); w.align()  # noqa: E702  MNIBBLES 4, uncompressed
    w.out += payload
    w.put(0, 1); w.put(3, 2); w.put(0, 1); w.put(0, 2); w.align()  # noqa: E702  empty metadata
    w.put(1, 1); w.put(1, 1)  # noqa: E702
    yield "crafted_metadata_uncompressed", w.done(), payload
    # WBITS = 10 (the smallest window: '1', '000', '010'), two uncompressed meta-blocks
    w = Bits()
    w.put(1, 1); w.put(0, 3); w.put(2, 3)  # noqa: E702
    payload = bytes(range(256)) * 9
    for part in (payload[:1500], payload[1500:]):
        w.put(0, 1); w.put(0, 2); w.put(len(part) - 1, 16); w.put(1, 1); w.align()  # noqa: E702
        w.out += part
    w.put(1, 1); w.put(1, 1)  # noqa: E702
    yield "crafted_wbits10_two_stored", w.done(), payload
    yield "crafted_empty", b"\x06", b""


def cases():
    rng = random.Random(20261004)
    text = corpus.gen_text(23, 200000)
    noise = bytes(rng.randrange(256) for _ in range(6000))
    yield "empty_q5", b"", dict(quality=5)
    yield "one_byte", b"q", dict(quality=9)
    yield "tiny_q0", text[:40], dict(quality=0)
    yield "tiny_q11", text[:40], dict(quality=11)
    yield "zeros", bytes(5000), dict(quality=6)
    yield "noise_q1", noise, dict(quality=1)  # uncompressed meta-blocks
    yield "web_q11_text", WEB * 3, dict(quality=11, mode=1)  # static dictionary words and transforms
    yield "web_q5", WEB, dict(quality=5, lgwin=18)
    yield "web_q2_w10", WEB * 4, dict(quality=2, lgwin=10)
    yield "text_q0", text[:6000], dict(quality=0)
    yield "text_q1", text[:6000], dict(quality=1)
    yield "text_q3", text[:8000], dict(quality=3, lgwin=16)
    yield "text_q4", text[:8000], dict(quality=4)
    yield "text_q6", text[:12000], dict(quality=6, lgwin=12)  # matches further back than the window
    yield "text_q9", text[:12000], dict(quality=9, lgwin=24)
    yield "text_q10", text[:9000], dict(quality=10)  # many literal and distance trees, context maps
    yield "text_q11", text[:9000], dict(quality=11, mode=1)
    yield "period3", b"xyz" * 9000, dict(quality=7)
    yield "ramp_q11", bytes((i * 7 + (i >> 8)) & 255 for i in range(20000)), dict(quality=11)
    yield "mixed_q8", text[:3000] + noise[:1500] + bytes(2000) + WEB + text[500:3500], dict(quality=8)
    yield "mixed_q11", text[:3000] + noise[:1500] + bytes(2000) + WEB + text[500:3500], dict(quality=11)
    yield "utf8_q11", ("Zürich Ünïcödé — текст на русском языке и ελληνικά; 日本語のテキスト。 " * 60).encode(), dict(quality=11, mode=1)


def main():
    out = {"generator": "oracle/gen_brotli_golden.py", "libbrotli": B.version(), "streams": []}
    todo = [(n, B.compress(d, **kw), d) for n, d, kw in cases()] + list(crafted())
    for name, blob, data in todo:
        back = B.decompress(blob, len(data))
        assert back == data, name  # libbrotlidec reads it, and reads what was meant
        assert len(blob) < 16000, (name, len(blob))
        out["streams"].append({"name": name, "payload": blob.hex(), "size": len(data), "sha256": hashlib.sha256(data).hexdigest()})
    path = os.path.join(ROOT, "tests", "golden", "brotli_streams.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote %s: %d streams, %d bytes" % (path, len(out["streams"]), os.path.getsize(path)))


if __name__ == "__main__":
    main()
