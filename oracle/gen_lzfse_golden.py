#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: writes tests/golden/lzfse_streams.json -- LZFSE streams written by the Python MODEL of the
format (tools/lzfse_model.py), each with the SHA-256 of its content.

PARITY UNPINNED: no implementation of LZFSE exists on the image and the reference has none either (its method 100 is a
private framing: ref src/lib/lzfse.inc.c:225-257).  These fixtures pin the GPU decoder (otezip_amd/csrc/otz_lzfse.h)
to the model, not to Apple's library.

    python3 oracle/gen_lzfse_golden.py      (the output is committed)
"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from otezip_amd import corpus  # noqa: E402
from tools import lzfse_model as M  # noqa: E402


def cases():
    rng = random.Random(20261004)
    text = corpus.gen_text(29, 200000)
    noise = bytes(rng.randrange(256) for _ in range(3000))
    yield "empty_raw", b"", "raw"
    yield "empty_lzfse", b"", "lzfse"
    yield "one_byte_lzvn", b"q", "lzvn"
    yield "tiny_lzvn", text[:40], "lzvn"
    yield "text_lzvn", text[:9000], "lzvn"
    yield "period3_lzvn", b"xyz" * 4000, "lzvn"
    yield "noise_raw", noise, "raw"
    yield "noise_lzfse", noise, "lzfse"
    yield "zeros_lzfse", bytes(20000), "lzfse"
    yield "text_lzfse", text[:12000], "lzfse"
    yield "text_lzfse_v1", text[:6000], "lzfse_v1"
    yield "ab_lzfse", bytes(rng.choice(b"ab") for _ in range(9000)), "lzfse"
    yield "mixed_blocks", text[:15000], "mixed"
    yield "ramp_lzvn", bytes((i * 7 + (i >> 8)) & 255 for i in range(12000)), "lzvn"


def main():
    out = {"generator": "oracle/gen_lzfse_golden.py", "pinned_to": "tools/lzfse_model.py (a model of the format: PARITY UNPINNED)", "streams": []}
    for name, data, kind in cases():
        blob = M.encode(data, kind)
        assert M.decode(blob) == data, name
        assert len(blob) < 16000, (name, len(blob))
        out["streams"].append({"name": name, "payload": blob.hex(), "size": len(data), "sha256": hashlib.sha256(data).hexdigest()})
    path = os.path.join(ROOT, "tests", "golden", "lzfse_streams.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote %s: %d streams, %d bytes" % (path, len(out["streams"]), os.path.getsize(path)))


if __name__ == "__main__":
    main()
