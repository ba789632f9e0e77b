"""Brotli (RFC 7932) under ZIP method 97 -- SURVEY.md section 8, row f4.

The reference's method 97 is a "BROT" container around stored bytes (src/lib/brotli.inc.c:57-80); the real
format has no reference code.  It is pinned to libbrotli 1.0.9: tests/golden/brotli_streams.json (written by
libbrotlienc through oracle/gen_brotli_golden.py, SHA-256 of the content as libbrotlidec decodes it) and, live,
the image's libbrotlienc / libbrotlidec through ctypes (tools/brotlilib.py).  The emulated kernel body
(tests/simt_emu) runs everywhere; the -m gpu tests go through the C ABI."""
import hashlib
import json
import os
import random
import sys
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from tools import brotlilib as B  # noqa: E402
from helpers import ST_OK, emu_decode  # noqa: E402

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "brotli_streams.json")))
live = pytest.mark.skipif(not B.available(), reason="libbrotli is not on this machine")


def item(blob, data):
    return (97, blob, len(data), zlib.crc32(data) & 0xFFFFFFFF, 0)


def golden():
    """(name, stream, content): the content is not stored -- its length and SHA-256 are; where libbrotlidec is
    present it decodes its own stream, elsewhere the kernel's output is checked against the hash alone."""
    out = []
    for g in GOLDEN["streams"]:
        blob = bytes.fromhex(g["payload"])
        data = B.decompress(blob, g["size"]) if B.available() else None
        if data is not None:
            assert len(data) == g["size"] and hashlib.sha256(data).hexdigest() == g["sha256"], g["name"]
        out.append((g["name"], blob, data, g))
    return out


def sweep(seed, big=True):
    from otezip_amd import corpus
    from oracle.gen_brotli_golden import WEB
    rng = random.Random(seed)
    text = corpus.gen_text(seed, 300000)
    datas = [b"", b"z", text[:33], text[:5000], text[:65536], bytes(9000), bytes(rng.randrange(256) for _ in range(7000)),
             b"xyz" * 7000, WEB * 5, text[:3000] + bytes(rng.randrange(256) for _ in range(3000)) + text[:3000]]
    if big:
        datas.append(text)  # meta-blocks longer than 64 KiB: five MLEN nibbles
    out = []
    for d in datas:
        for kw in (dict(quality=0), dict(quality=2, lgwin=10), dict(quality=5, lgwin=16), dict(quality=9, lgwin=24),
                   dict(quality=11, mode=1)):
            out.append((d, kw))
    return out


def check(items, results, datas):
    for (m, blob, n, crc, _), (r, out, guard), d in zip(items, results, datas):
        assert r.status == ST_OK, (len(d), r.status)
        assert out == d and r.total_out == len(d) and r.crc32 == crc and r.flags == 0
        assert set(guard[-16:]) <= {0xCD}, "wrote past the entry's region"


def test_emulated_kernel_decodes_the_golden_streams():
    g = golden()
    items = [(97, blob, meta["size"], 0, 0) for _, blob, _, meta in g]
    for (name, _, data, meta), (r, out, guard) in zip(g, emu_decode(items)):
        assert r.status == ST_OK and r.total_out == meta["size"], (name, r.status)
        assert hashlib.sha256(out).hexdigest() == meta["sha256"], name
        assert r.crc32 == zlib.crc32(out) & 0xFFFFFFFF
        assert data is None or out == data
        assert set(guard[-16:]) <= {0xCD}, "wrote past the entry's region"


@live
def test_emulated_kernel_against_libbrotli():
    sw = sweep(3, big=False)[::2]
    items = [item(B.compress(d, **kw), d) for d, kw in sw]
    check(items, emu_decode(items), [d for d, _ in sw])


@live
def test_emulated_kernel_agrees_with_libbrotlidec_on_damaged_streams():
    """Every prefix of a small stream, and single-bit damage all over a larger one: where libbrotlidec rejects,
    the entry must not come out OK; where it still decodes to the expected size, the bytes must be the same
    (the CRC-32 then says whether they are the right ones)."""
    from oracle.gen_brotli_golden import WEB
    d = WEB[:400]
    blob = B.compress(d, quality=11, mode=1)
    variants = [blob[:k] for k in range(len(blob))]
    big = B.compress(WEB * 2 + bytes(range(200)), quality=6)
    rng = random.Random(9)
    d2 = WEB * 2 + bytes(range(200))
    for _ in range(150):
        b = bytearray(big)
        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        variants.append(bytes(b))
    sizes = [len(d)] * len(blob) + [len(d2)] * 150
    items = [(97, v, n, 0, 0) for v, n in zip(variants, sizes)]
    res = emu_decode(items)
    n_ok = 0
    for v, n, (r, out, _) in zip(variants, sizes, res):
        want = B.decompress(v, n)
        if want is not None and len(want) == n:
            assert r.status == ST_OK and out == want, (len(v), r.status)
            n_ok += 1
        else:
            assert r.status != ST_OK, len(v)
    assert n_ok > 0  # some damage only changes bytes


def test_emulated_kernel_failures_and_routing():
    """Wrong sizes, a wrong CRC, and the reference's "BROT" container, which shares the method id."""
    name, blob, _, meta = next(g for g in golden() if g[0] == "text_q4")
    n = meta["size"]
    pdata = b"the reference stores the bytes behind its own header " * 20
    priv = O.frame_brotli(pdata)
    want = O.extract_entry(97, priv, len(pdata), zlib.crc32(pdata) & 0xFFFFFFFF)
    r0 = emu_decode([(97, blob, n, 0, 0)])[0]
    crc = r0[0].crc32
    items = [(97, blob, n + 7, crc, 0), (97, blob, n - 7, crc, 0), (97, blob, n, crc ^ 5, 0), item(priv, pdata),
             (97, b"BROT" + blob, n, crc, 0)]
    res = emu_decode(items)
    assert all(res[k][0].status != ST_OK for k in (0, 1, 4))
    assert res[2][0].status == ST_OK and res[2][0].flags & 4 and res[2][1] == r0[1]
    assert (res[3][0].status == ST_OK) == (want["correct_rc"] == 0) and (want["correct_rc"] != 0 or res[3][1] == want["out"])


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
        tab.append((97, len(blob), len(payload), usize, crc))
        blob += payload
    return oz.decode_batch_host(bytes(blob), tab)


@pytest.mark.gpu
def test_gpu_golden_and_libbrotli_sweep(oz):
    g = golden()
    b = _batch(oz, [(blob, meta["size"], 0) for _, blob, _, meta in g])
    for k, (name, _, _, meta) in enumerate(g):
        assert b.ok(k), (name, b.status(k))
        assert hashlib.sha256(b.data(k)).hexdigest() == meta["sha256"], name
        assert b.results[k].crc32 == zlib.crc32(b.data(k)) & 0xFFFFFFFF
    if not B.available():
        return
    sw = sweep(8)
    b = _batch(oz, [(B.compress(d, **kw), len(d), zlib.crc32(d) & 0xFFFFFFFF) for d, kw in sw])
    for k, (d, kw) in enumerate(sw):
        assert b.ok(k) and b.data(k) == d and b.results[k].flags == 0, (len(d), kw, b.status(k))


@pytest.mark.gpu
@live
def test_gpu_many_entries_mixed_with_the_private_container(oz):
    """3 000 x 16 KiB Brotli entries (one work-item each) interleaved with entries in the reference's "BROT"
    container under the same method id: the sniffing keeps them apart, every CRC-32 is checked on the device."""
    from otezip_amd import corpus
    n = 3000
    triples, want = [], []
    for i in range(n):
        d = corpus.gen_text(i, 16384)
        if i % 5 == 4:
            triples.append((O.frame_brotli(d), len(d), zlib.crc32(d) & 0xFFFFFFFF))
        else:
            triples.append((B.compress(d, quality=(1, 5, 9)[i % 3], lgwin=18), len(d), zlib.crc32(d) & 0xFFFFFFFF))
        want.append(d)
    b = _batch(oz, triples)
    bad = [k for k in range(n) if not b.ok(k) or b.results[k].crc32 != triples[k][2] or b.results[k].flags]
    assert not bad, bad[:10]
    for k in range(0, n, 97):
        assert b.data(k) == want[k]
