"""Zstandard frames (RFC 8878) under ZIP method 93 -- SURVEY.md section 8, row f4.

The reference has no Zstandard decoder (src/lib/zstd.inc.c:172-182, :689-692), so parity is pinned
elsewhere: (1) tests/golden/zstd_frames.json, frames written by the image's libzstd 1.4.8 (and three
hand-written ones that libzstd's decoder accepts) with the SHA-256 of their content; (2) the XXH64
content checksum inside the frames; (3) libzstd itself, live, where the shared library is present
(it is part of the image); (4) oracle/otz_zstd_rfc.c, a CPU decoder written from the RFC.
The emulated kernels (tests/simt_emu) run everywhere; the -m gpu tests go through the C ABI."""
import ctypes as C
import hashlib
import json
import os
import random
import struct
import sys
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from tools import zstdlib as Z  # noqa: E402
from helpers import ST_OK, emu_decode  # noqa: E402

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "zstd_frames.json")))
needs_libzstd = pytest.mark.skipif(not Z.available(), reason="libzstd.so.1 not present")


def rfc_decode(frame, cap):
    L = O.lib()
    L.otz_zstd_rfc_decode.restype = C.c_int
    L.otz_zstd_rfc_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    out = C.create_string_buffer(max(cap, 1))
    n, fl = C.c_size_t(0), C.c_uint32(0)
    rc = L.otz_zstd_rfc_decode(frame, len(frame), out, cap, C.byref(n), C.byref(fl))
    return rc, out.raw[:n.value], fl.value


def golden_items():
    return [(g["name"], bytes.fromhex(g["frame"]), g["size"], g["sha256"]) for g in GOLDEN["frames"]]


def item(frame, data):
    return (93, frame, len(data), zlib.crc32(data) & 0xFFFFFFFF, 0)


def sweep(seed=7, big=True):
    """(data, libzstd parameters): sizes from 0 to several blocks, entropy from constant to random."""
    from otezip_amd import corpus
    rng = random.Random(seed)
    text = corpus.gen_text(seed, 300000)
    datas = [b"", b"x", text[:17], text[:300], text[:4000], text[:70000], bytes(5000), bytes(rng.randrange(256) for _ in range(9000)),
             (b"0123456789" * 3000), bytes(rng.choice(b"abcd") for _ in range(30000)), text[:2000] + bytes(rng.randrange(256) for _ in range(2000)) + text[:2000]]
    if big:
        datas += [text[:262144], text, (text[:50000] * 7)[:300000], bytes((i * 13 + (i >> 9)) & 255 for i in range(200000))]
    out = []
    for d in datas:
        for kw in (dict(level=1), dict(level=3), dict(level=7, checksum=False), dict(level=19), dict(level=-5, content_size=False),
                   dict(level=5, minmatch=3, windowlog=12)):
            out.append((d, kw))
    return out


# ------------------------------------------------------------------ the checker is checked first

def test_golden_covers_the_format():
    need = {"block_raw", "block_rle", "block_compressed", "lit_raw", "lit_rle", "lit_huffman", "lit_treeless", "lit_1stream", "lit_4streams",
            "huf_weights_direct", "huf_weights_fse", "no_sequences", "more_than_64_sequences", "multi_block", "checksum", "skippable",
            "window_descriptor", "single_segment"}
    for t in ("ll", "of", "ml"):
        need |= {"%s_%s" % (t, m) for m in ("predefined", "rle", "fse", "repeat")}
    assert need <= set(GOLDEN["features_covered"]), need - set(GOLDEN["features_covered"])


def test_rfc_decoder_against_golden_frames():
    for name, fr, size, sha in golden_items():
        rc, out, fl = rfc_decode(fr, size)
        assert rc == 0 and len(out) == size and hashlib.sha256(out).hexdigest() == sha, name
        rc, _, _ = rfc_decode(fr, size - 1) if size else (-5, b"", 0)
        assert rc != 0, name + ": must not fit a smaller buffer"


@needs_libzstd
def test_rfc_decoder_against_libzstd():
    for data, kw in sweep():
        fr = Z.compress(data, **kw)
        rc, out, fl = rfc_decode(fr, len(data))
        assert rc == 0 and out == data, (len(data), kw)
        assert bool(fl & 1) == kw.get("checksum", True)


@needs_libzstd
def test_rfc_decoder_rejects_what_libzstd_rejects():
    """Truncations and bit flips: the same verdict as libzstd, and the same bytes when both accept
    (a flip may land in a place that does not matter, e.g. the unused header bit)."""
    from otezip_amd import corpus
    rng = random.Random(11)
    data = corpus.gen_text(4, 20000)
    fr = Z.compress(data, level=3)
    trials = [fr[:k] for k in (3, 4, 5, 9, 20, len(fr) // 2, len(fr) - 5, len(fr) - 1)]  # (an EMPTY input is "no frame": libzstd returns 0 bytes, the entry decoder wants a frame)
    for _ in range(300):
        b = bytearray(fr)
        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        trials.append(bytes(b))
    agree = 0
    for t in trials:
        ok, back = Z.decompress(t, len(data))
        rc, out, _ = rfc_decode(t, len(data))
        assert (rc == 0) == ok, (len(t), rc, ok)
        if ok:
            assert out == back
            agree += 1
    assert agree < len(trials)


# ------------------------------------------------------------------ the kernel bodies, emulated

def check(items, results, datas):
    for (m, fr, n, crc, _), (r, out, guard), d in zip(items, results, datas):
        assert r.status == ST_OK, (len(d), r.status)
        assert out == d and r.total_out == len(d) and r.crc32 == crc and r.flags == 0
        assert set(guard[-16:]) <= {0xCD}, "wrote past the entry's region"


def test_emulated_kernel_decodes_the_golden_frames():
    gi = golden_items()
    datas = []
    items = []
    for name, fr, size, sha in gi:
        rc, out, _ = rfc_decode(fr, size)
        assert rc == 0
        datas.append(out)
        items.append(item(fr, out))
    check(items, emu_decode(items), datas)


@needs_libzstd
def test_emulated_kernel_against_libzstd():
    sw = sweep(seed=9, big=False)[::2]
    items = [item(Z.compress(d, **kw), d) for d, kw in sw]
    check(items, emu_decode(items), [d for d, _ in sw])


def test_emulated_kernel_failures_and_routing():
    """Wrong lengths, corrupt frames, and the private framing that shares the magic."""
    name, fr, size, sha = next(g for g in golden_items() if g[0] == "text_5k")
    rc, data, _ = rfc_decode(fr, size)
    crc = zlib.crc32(data) & 0xFFFFFFFF
    bad_ck = bytearray(fr)
    bad_ck[-1] ^= 0x40  # the content checksum
    priv = O.frame_zstd(b"private framing payload " * 50)  # fifth byte 0x70: unused bit set
    priv0 = priv[:4] + b"\x00" + priv[5:]  # the reference skips the fifth byte unread: any value is accepted
    pdata = b"private framing payload " * 50
    items = [(93, fr, size + 16, crc, 0), (93, fr, size - 16, crc, 0), (93, fr[:len(fr) // 2], size, crc, 0), (93, bytes(bad_ck), size, crc, 0),
             (93, fr, size, crc ^ 1, 0), item(priv, pdata), item(priv0, pdata)]
    res = emu_decode(items)
    assert res[0][0].status != ST_OK and res[1][0].status != ST_OK  # total_out != uncomp_size (ref otezip.c:555)
    assert res[2][0].status != ST_OK and res[3][0].status != ST_OK
    assert res[4][0].status == ST_OK and res[4][0].flags & 4 and res[4][1] == data  # CRC mismatch is a flag, as everywhere
    for k in (5, 6):
        assert res[k][0].status == ST_OK and res[k][1] == pdata, k
        want = O.extract_entry(93, items[k][1], len(pdata), items[k][3])
        assert want["correct_rc"] == 0 and want["out"] == pdata


def test_lds_budget_of_the_zstd_kernel():
    from helpers import emu_lib
    L = emu_lib()
    L.emu_lds_bytes_zstd.restype = C.c_uint32
    assert L.emu_lds_bytes_zstd() <= 160 * 1024 // 13  # at least 13 wavefronts per CU


# ------------------------------------------------------------------ the GPU

@pytest.fixture(scope="module")
def oz():
    import otezip_amd as oz
    oz.init(0)
    yield oz
    oz.shutdown()


@pytest.mark.gpu
def test_gpu_decodes_golden_and_crafted_frames(oz):
    items, datas = [], []
    for name, fr, size, sha in golden_items():
        rc, out, _ = rfc_decode(fr, size)
        assert rc == 0
        datas.append(out)
        items.append(item(fr, out)[:4])
    b = oz.decode_batch_host(b"".join(b"\0" * 5 + it[1] for it in items), [(93, sum(5 + len(x[1]) for x in items[:k]) + 5, len(it[1]), it[2], it[3])
                                                                            for k, it in enumerate(items)])
    for k, d in enumerate(datas):
        assert b.ok(k), (GOLDEN["frames"][k]["name"], b.status(k))
        assert b.data(k) == d and b.results[k].crc32 == zlib.crc32(d) & 0xFFFFFFFF and b.results[k].flags == 0
        assert hashlib.sha256(b.data(k)).hexdigest() == GOLDEN["frames"][k]["sha256"]


def _batch(oz, pairs):
    blob, tab = bytearray(), []
    for fr, usize, crc in pairs:
        blob += b"\0" * 3
        tab.append((93, len(blob), len(fr), usize, crc))
        blob += fr
    return oz.decode_batch_host(bytes(blob), tab)


@pytest.mark.gpu
@needs_libzstd
def test_gpu_against_libzstd_sweep_and_corruptions(oz):
    sw = sweep(seed=21)
    frames = [Z.compress(d, **kw) for d, kw in sw]
    b = _batch(oz, [(fr, len(d), zlib.crc32(d) & 0xFFFFFFFF) for fr, (d, _) in zip(frames, sw)])
    for k, (d, kw) in enumerate(sw):
        assert b.ok(k) and b.data(k) == d, (len(d), kw, b.status(k))
    # corrupt frames: accepted exactly when libzstd accepts them, with libzstd's bytes
    from otezip_amd import corpus
    rng = random.Random(5)
    data = corpus.gen_text(8, 30000)
    fr = Z.compress(data, level=4)
    trials = [fr[:k] for k in (4, 5, 9, len(fr) // 2, len(fr) - 1)]
    for _ in range(200):
        t = bytearray(fr)
        t[rng.randrange(len(t))] ^= 1 << rng.randrange(8)
        trials.append(bytes(t))
    b = _batch(oz, [(t, len(data), zlib.crc32(data) & 0xFFFFFFFF) for t in trials])
    for k, t in enumerate(trials):
        ok, back = Z.decompress(t, len(data))
        if ok and len(back) == len(data):
            assert b.ok(k) and b.data(k) == back, k
        elif t[4:5] and (t[4] & 0x10):
            pass  # the flip set the unused header bit: that is the private framing's signature, the other decoder judges it
        else:
            assert not b.ok(k), k


@pytest.mark.gpu
@needs_libzstd
def test_gpu_config2_shape_real_zstd_full_count(oz):
    """BASELINE configs[2] with REAL Zstandard frames: 10 000 x 256 KiB enwik-style entries, libzstd level 3 with
    content checksums (2.5 GiB of output).  Every entry's CRC-32 and length are checked on the device, the
    frames' XXH64 checksums inside the kernel, and every 97th entry byte for byte."""
    import multiprocessing as mp
    from otezip_amd import corpus
    n, size = 10000, 262144
    with mp.get_context("fork").Pool(min(mp.cpu_count(), 16)) as pool:
        made = pool.map(_make_zstd, [(i, size) for i in range(n)], chunksize=64)
    blob, tab = bytearray(), []
    for fr, crc in made:
        tab.append((93, len(blob), len(fr), size, crc))
        blob += fr
    ratio = n * size / len(blob)
    b = oz.decode_batch_host(bytes(blob), tab)
    bad = [k for k in range(n) if not b.ok(k) or b.results[k].crc32 != tab[k][4] or b.results[k].flags]
    assert not bad, bad[:10]
    for k in range(0, n, 97):
        assert b.data(k) == corpus.gen_text(k, size)
    rec = os.path.join(ROOT, "gpurun_out")
    os.makedirs(rec, exist_ok=True)
    with open(os.path.join(rec, "zstd_real_full_count.txt"), "a") as f:
        f.write("10000 x 256 KiB real Zstandard frames (libzstd %s level 3, ratio %.2f): all decoded, CRC-32 and XXH64 verified\n" % (Z.version(), ratio))


def _make_zstd(a):
    from otezip_amd import corpus
    i, size = a
    d = corpus.gen_text(i, size)
    return Z.compress(d, level=3), zlib.crc32(d) & 0xFFFFFFFF
