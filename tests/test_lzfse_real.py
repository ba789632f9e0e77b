"""LZFSE streams (blocks bvx- / bvxn / bvx1 / bvx2 / bvx$) under ZIP method 100 -- SURVEY.md section 8, row f4.

PARITY UNPINNED.  The reference's method 100 is a private framing (src/lib/lzfse.inc.c:225-257); the real format has
no reference code, and nothing on the image implements it.  The GPU decoder (otezip_amd/csrc/otz_lzfse.h) is tested
against a Python model of the format (tools/lzfse_model.py: encoder + decoder, written separately from the kernel):
the streams the model writes (tests/golden/lzfse_streams.json and live sweeps) and the same accept / reject verdict
as the model's decoder on damaged streams.  That shows the kernel does what the model says -- not that the model is
Apple's library.  The emulated kernel body (tests/simt_emu) runs everywhere; the -m gpu tests go through the C ABI."""
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
from tools import lzfse_model as M  # noqa: E402
from helpers import ST_OK, emu_decode  # noqa: E402

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "lzfse_streams.json")))


def item(blob, data):
    return (100, blob, len(data), zlib.crc32(data) & 0xFFFFFFFF, 0)


def golden():
    out = []
    for g in GOLDEN["streams"]:
        blob = bytes.fromhex(g["payload"])
        data = M.decode(blob)
        assert len(data) == g["size"] and hashlib.sha256(data).hexdigest() == g["sha256"], g["name"]
        out.append((g["name"], blob, data))
    return out


def sweep(seed):
    from otezip_amd import corpus
    rng = random.Random(seed)
    text = corpus.gen_text(seed, 300000)
    datas = [b"", b"z", text[:33], text[:5000], text[:65536], bytes(9000), bytes(rng.randrange(256) for _ in range(7000)),
             b"xyz" * 7000, text[:3000] + bytes(rng.randrange(256) for _ in range(3000)) + text[:3000], text[:150000]]
    return [(d, k) for d in datas for k in ("raw", "lzvn", "lzfse", "lzfse_v1", "mixed")]


def check(items, results, datas):
    for (m, blob, n, crc, _), (r, out, guard), d in zip(items, results, datas):
        assert r.status == ST_OK, (len(d), r.status)
        assert out == d and r.total_out == len(d) and r.crc32 == crc and r.flags == 0
        assert set(guard[-16:]) <= {0xCD}, "wrote past the entry's region"


def test_emulated_kernel_decodes_the_golden_streams():
    g = golden()
    items = [item(b, d) for _, b, d in g]
    check(items, emu_decode(items), [d for _, _, d in g])


def test_emulated_kernel_against_the_model():
    sw = sweep(3)[::2]
    items = [item(M.encode(d, k), d) for d, k in sw]
    check(items, emu_decode(items), [d for d, _ in sw])


def test_emulated_kernel_agrees_with_the_model_on_damaged_streams():
    """Every prefix of a small stream and single-bit damage all over larger ones: where the model's decoder rejects,
    the entry must not come out OK; where it still decodes to the expected size, the bytes must be the same."""
    from otezip_amd import corpus
    d = corpus.gen_text(7, 600)
    variants, sizes = [], []
    for kind in ("lzvn", "lzfse"):
        blob = M.encode(d, kind)
        variants += [blob[:k] for k in range(len(blob))]
        sizes += [len(d)] * len(blob)
    rng = random.Random(9)
    d2 = corpus.gen_text(8, 4000) + bytes(range(200))
    for kind in ("lzvn", "lzfse", "lzfse_v1", "mixed"):
        big = M.encode(d2, kind)
        for _ in range(60):
            b = bytearray(big)
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            variants.append(bytes(b))
            sizes.append(len(d2))
    res = emu_decode([(100, v, n, 0, 0) for v, n in zip(variants, sizes)])
    n_ok = 0
    for v, n, (r, out, _) in zip(variants, sizes, res):
        try:
            want = M.decode(v)
        except (M.FormatError, IndexError, KeyError, ValueError, struct_error()):
            want = None
        if want is not None and len(want) == n:
            assert r.status == ST_OK and out == want, (len(v), r.status)
            n_ok += 1
        else:
            assert r.status != ST_OK, len(v)
    assert n_ok > 0  # some damage only changes bytes


def struct_error():
    import struct
    return struct.error


def test_emulated_kernel_failures_and_routing():
    """Wrong sizes, a wrong CRC, and the reference's private framing, which shares the method id."""
    name, blob, data = next(g for g in golden() if g[0] == "text_lzfse")
    n, crc = len(data), zlib.crc32(data) & 0xFFFFFFFF
    pdata = b"the reference stores the bytes behind a five-byte header " * 20
    priv = O.frame_lzfse(pdata)
    want = O.extract_entry(100, priv, len(pdata), zlib.crc32(pdata) & 0xFFFFFFFF)
    items = [(100, blob, n + 7, crc, 0), (100, blob, n - 7, crc, 0), (100, blob, n, crc ^ 5, 0), item(priv, pdata)]
    res = emu_decode(items)
    assert all(res[k][0].status != ST_OK for k in (0, 1))
    assert res[2][0].status == ST_OK and res[2][0].flags & 4 and res[2][1] == data
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
        tab.append((100, len(blob), len(payload), usize, crc))
        blob += payload
    return oz.decode_batch_host(bytes(blob), tab)


@pytest.mark.gpu
def test_gpu_golden_and_model_sweep(oz):
    g = golden()
    b = _batch(oz, [(blob, len(d), zlib.crc32(d) & 0xFFFFFFFF) for _, blob, d in g])
    for k, (name, _, d) in enumerate(g):
        assert b.ok(k) and b.data(k) == d and b.results[k].flags == 0, (name, b.status(k))
    sw = sweep(8)
    b = _batch(oz, [(M.encode(d, k), len(d), zlib.crc32(d) & 0xFFFFFFFF) for d, k in sw])
    for k, (d, kind) in enumerate(sw):
        assert b.ok(k) and b.data(k) == d, (len(d), kind, b.status(k))


@pytest.mark.gpu
def test_gpu_many_entries_mixed_with_the_private_framing(oz):
    """2 000 x 8 KiB LZFSE entries (one work-item each) interleaved with entries in the reference's 06 framing under
    the same method id: the sniffing keeps them apart, every CRC-32 is checked on the device."""
    from otezip_amd import corpus
    n = 2000
    triples, want = [], []
    for i in range(n):
        d = corpus.gen_text(i, 8192)
        if i % 5 == 4:
            triples.append((O.frame_lzfse(d), len(d), zlib.crc32(d) & 0xFFFFFFFF))
        else:
            triples.append((M.encode(d, ("lzfse", "lzvn", "mixed")[i % 3]), len(d), zlib.crc32(d) & 0xFFFFFFFF))
        want.append(d)
    b = _batch(oz, triples)
    bad = [k for k in range(n) if not b.ok(k) or b.results[k].crc32 != triples[k][2] or b.results[k].flags]
    assert not bad, bad[:10]
    for k in range(0, n, 97):
        assert b.data(k) == want[k]
