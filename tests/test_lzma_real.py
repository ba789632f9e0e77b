"""LZMA1 under ZIP method 14 as APPNOTE 5.8.8 defines it -- SURVEY.md section 8, row f4.

The reference's method 14 is a private token format (src/lib/lzma-dec.inc.c:102-155); the real format has no
reference code.  It is pinned to liblzma: tests/golden/lzma_streams.json (written by liblzma through
oracle/gen_lzma_golden.py, SHA-256 of the content) and, live, Python's lzma module.  The emulated kernel body
(tests/simt_emu) runs everywhere; the -m gpu tests go through the C ABI."""
import hashlib
import json
import lzma
import os
import random
import sys
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from oracle.gen_lzma_golden import zip_lzma  # noqa: E402
from helpers import ST_OK, emu_decode  # noqa: E402

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "lzma_streams.json")))


def item(blob, data):
    return (14, blob, len(data), zlib.crc32(data) & 0xFFFFFFFF, 0)


def golden():
    out = []
    for g in GOLDEN["streams"]:
        blob = bytes.fromhex(g["payload"])
        lc_lp_pb = blob[4]
        f = [{"id": lzma.FILTER_LZMA1, "lc": lc_lp_pb % 9, "lp": lc_lp_pb // 9 % 5, "pb": lc_lp_pb // 45,
              "dict_size": int.from_bytes(blob[5:9], "little")}]
        data = lzma.decompress(blob[9:], format=lzma.FORMAT_RAW, filters=f)  # liblzma reads its own stream
        assert len(data) == g["size"] and hashlib.sha256(data).hexdigest() == g["sha256"], g["name"]
        out.append((g["name"], blob, data))
    return out


def sweep(seed):
    from otezip_amd import corpus
    rng = random.Random(seed)
    text = corpus.gen_text(seed, 200000)
    datas = [b"", b"z", text[:33], text[:5000], text[:65536], bytes(9000), bytes(rng.randrange(256) for _ in range(7000)),
             b"xyz" * 7000, text[:3000] + bytes(rng.randrange(256) for _ in range(3000)) + text[:3000], text]
    out = []
    for d in datas:
        for kw in ({}, dict(lc=0, lp=0, pb=0, preset=1), dict(lc=1, lp=3, pb=1), dict(lc=4, lp=0, pb=4, preset=9, dict_size=4096)):
            out.append((d, kw))
    return out


def check(items, results, datas):
    for (m, blob, n, crc, _), (r, out, guard), d in zip(items, results, datas):
        assert r.status == ST_OK, (len(d), r.status)
        assert out == d and r.total_out == len(d) and r.crc32 == crc and r.flags == 0
        assert set(guard[-16:]) <= {0xCD}, "wrote past the entry's region"


def test_emulated_kernel_decodes_the_golden_streams():
    g = golden()
    items = [item(b, d) for _, b, d in g]
    check(items, emu_decode(items), [d for _, _, d in g])


def test_emulated_kernel_against_liblzma():
    sw = sweep(3)[::3]
    items = [item(zip_lzma(d, **kw), d) for d, kw in sw]
    check(items, emu_decode(items), [d for d, _ in sw])


def test_emulated_kernel_failures_and_routing():
    """Wrong sizes, damaged streams, and the reference's private token format, which shares the method id."""
    _, blob, data = next(g for g in golden() if g[0] == "text_2k")
    crc = zlib.crc32(data) & 0xFFFFFFFF
    pdata = b"AAAAAAAAAAAAAAAAAAAABBBBBBBBBBBBBBBBBBBBBBBBCCCCCC private tokens " * 20
    priv = O.frame_lzma_tokens(pdata)
    want = O.extract_entry(14, priv, len(pdata), zlib.crc32(pdata) & 0xFFFFFFFF)
    cut = blob[:len(blob) // 2]
    flip = bytearray(blob)
    flip[30] ^= 0x10
    items = [(14, blob, len(data) + 7, crc, 0), (14, blob, len(data) - 7, crc, 0), (14, cut, len(data), crc, 0),
             (14, bytes(flip), len(data), crc, 0), (14, blob, len(data), crc ^ 5, 0), item(priv, pdata)]
    res = emu_decode(items)
    assert all(res[k][0].status != ST_OK for k in (0, 1, 2))
    # a damaged stream either breaks or decodes to other bytes: then the CRC says so
    assert res[3][0].status != ST_OK or res[3][0].flags & 4
    assert res[4][0].status == ST_OK and res[4][0].flags & 4 and res[4][1] == data
    assert (res[5][0].status == ST_OK) == (want["correct_rc"] == 0) and (want["correct_rc"] != 0 or res[5][1] == want["out"])


@pytest.fixture(scope="module")
def oz():
    import otezip_amd as oz
    oz.init(0)
    yield oz
    oz.shutdown()


def _batch(oz, triples):
    blob, tab = bytearray(), []
    for payload, usize, crc in triples:
        blob += b"\0" * 3
        tab.append((14, len(blob), len(payload), usize, crc))
        blob += payload
    return oz.decode_batch_host(bytes(blob), tab)


@pytest.mark.gpu
def test_gpu_golden_and_liblzma_sweep(oz):
    g = golden()
    b = _batch(oz, [(blob, len(d), zlib.crc32(d) & 0xFFFFFFFF) for _, blob, d in g])
    for k, (name, _, d) in enumerate(g):
        assert b.ok(k) and b.data(k) == d and b.results[k].flags == 0, (name, b.status(k))
    sw = sweep(8)
    b = _batch(oz, [(zip_lzma(d, **kw), len(d), zlib.crc32(d) & 0xFFFFFFFF) for d, kw in sw])
    for k, (d, kw) in enumerate(sw):
        assert b.ok(k) and b.data(k) == d, (len(d), kw, b.status(k))


@pytest.mark.gpu
def test_gpu_many_entries_mixed_with_the_private_format(oz):
    """4 000 x 16 KiB LZMA1 entries (one work-item each) interleaved with entries of the reference's private
    token format under the same method id: the sniffing keeps them apart, every CRC-32 is checked on the device."""
    from otezip_amd import corpus
    n = 4000
    triples, want = [], []
    for i in range(n):
        if i % 5 == 4:
            d = bytes([65 + i % 7]) * 300 + b"tail %d" % i
            triples.append((O.frame_lzma_tokens(d), len(d), zlib.crc32(d) & 0xFFFFFFFF))
        else:
            d = corpus.gen_text(i, 16384)
            triples.append((zip_lzma(d, preset=1), len(d), zlib.crc32(d) & 0xFFFFFFFF))
        want.append(d)
    b = _batch(oz, triples)
    bad = [k for k in range(n) if not b.ok(k) or b.results[k].crc32 != triples[k][2] or b.results[k].flags]
    assert not bad, bad[:10]
    for k in range(0, n, 37):
        assert b.data(k) == want[k]
