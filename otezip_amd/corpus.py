"""Synthetic corpora and in-memory ZIP32 archives for the benchmark and the
full-size tests.

enwik8 is not available offline, so "enwik-style" text is generated: a fixed
30 000-word vocabulary of random lowercase words (2-10 letters), words drawn
Zipf(1.0), separators ' ' / ', ' / '. ' / ']]\\n[[' at 85/7/5/3 %, one RNG stream
per entry seeded with the entry index.  Payloads are compressed with the host's
zlib (level 6, raw DEFLATE) -- a valid encoder, unlike the reference's own
 -- and laid out as a real ZIP32 archive image (local headers,
payloads, central directory, EOCD) with the field layout the reference parses
(ref src/lib/otezip.c:360-378, :429-437).
"""
import multiprocessing as mp
import os
import struct
import zlib

import numpy as np

VOCAB_WORDS = 30000
_SEPS = [b" ", b", ", b". ", b"]]\n[["]
_SEP_P = np.array([0.85, 0.07, 0.05, 0.03])

_vocab = None


def vocab():
    """(table[V+4, 10] uint8, lengths[V+4], zipf_cdf[V]); rows V..V+3 are the separators."""
    global _vocab
    if _vocab is None:
        rng = np.random.default_rng(0x07E21B)
        lens = rng.integers(2, 11, size=VOCAB_WORDS)
        tab = np.zeros((VOCAB_WORDS + 4, 10), dtype=np.uint8)
        # letter frequencies roughly English so that literals are not uniform
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        w = 1.0 / np.arange(1, 27) ** 0.8
        w /= w.sum()
        tab[:VOCAB_WORDS] = letters[rng.choice(26, size=(VOCAB_WORDS, 10), p=w)]
        lens = np.concatenate([lens, [len(s) for s in _SEPS]])
        for k, s in enumerate(_SEPS):
            tab[VOCAB_WORDS + k, :len(s)] = np.frombuffer(s, dtype=np.uint8)
        p = 1.0 / np.arange(1, VOCAB_WORDS + 1)
        cdf = np.cumsum(p / p.sum())
        _vocab = (tab, lens.astype(np.int64), cdf)
    return _vocab


def gen_text(seed, size):
    """`size` bytes of enwik-style text, deterministic in (seed, size)."""
    tab, lens, cdf = vocab()
    rng = np.random.default_rng([0xC0FFEE, int(seed)])
    out = np.empty(0, dtype=np.uint8)
    need = size
    parts = []
    while need > 0:
        k = need // 5 + 16
        words = np.searchsorted(cdf, rng.random(k), side="right").clip(0, VOCAB_WORDS - 1)
        seps = VOCAB_WORDS + rng.choice(4, size=k, p=_SEP_P)
        toks = np.empty(2 * k, dtype=np.int64)
        toks[0::2] = words
        toks[1::2] = seps
        l = lens[toks]
        total = int(l.sum())
        start = np.cumsum(l) - l
        tok_of_byte = np.repeat(np.arange(2 * k), l)
        j = np.arange(total) - np.repeat(start, l)
        chunk = tab[toks[tok_of_byte], j]
        parts.append(chunk[:need])
        need -= min(need, total)
    if not parts:
        return b""
    out = np.concatenate(parts) if len(parts) > 1 else parts[0]
    return out.tobytes()


# --- the reference's private framings (what its DEcoders accept) ---

def frame_zstd_private(payload, block=65535):
    """LE magic 28 B5 2F FD, one ignored byte, raw blocks [hdr][len LE16], empty last block
    (ref src/lib/zstd.inc.c:495-498, :554-560, :404-410)."""
    out = bytearray(b"\x28\xb5\x2f\xfd\x70")
    for i in range(0, len(payload), block):
        b = payload[i:i + block]
        out += b"\x00" + struct.pack("<H", len(b)) + b
    out += b"\x01\x00\x00"
    return bytes(out)


def frame_lzfse_private(payload):
    """0x06 + LE32 length + raw bytes (ref src/lib/lzfse.inc.c:179-190)."""
    return b"\x06" + struct.pack("<I", len(payload)) + payload


def frame_brotli_private(payload):
    """'BROT' 01 + LE64 length + LE32 CRC-32 + raw bytes (ref src/lib/brotli.inc.c:41-55)."""
    return b"BROT\x01" + struct.pack("<QI", len(payload), zlib.crc32(payload) & 0xFFFFFFFF) + payload


def raw_deflate(data, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, 0)
    return c.compress(data) + c.flush()


_ENCODERS = {0: lambda d: d, 8: raw_deflate, 93: frame_zstd_private, 100: frame_lzfse_private,
             97: frame_brotli_private}


def _make_one(args):
    idx, size, method = args
    data = gen_text(idx, size)
    return idx, method, _ENCODERS[method](data), len(data), zlib.crc32(data) & 0xFFFFFFFF


def _cache_path(tag):
    """Generated corpora are kept on local disk (OTEZIP_BENCH_CACHE, default /tmp/otezip_amd_cache): a
    tool that runs under rocprofv3 must not fork a worker pool (the profiler has initialised the GPU
    in that process), so profiling scripts generate in an unprofiled step and the profiled one loads."""
    d = os.environ.get("OTEZIP_BENCH_CACHE", "/tmp/otezip_amd_cache")
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        return None
    return os.path.join(d, tag + ".pkl")


def _cached(tag, make, min_items, n):
    import pickle
    path = _cache_path(tag) if n >= min_items else None
    if path and os.path.exists(path):
        try:
            with open(path, "rb") as f:
                return pickle.load(f)
        except Exception:
            pass
    val = make()
    if path:
        try:
            tmp = "%s.tmp%d" % (path, os.getpid())
            with open(tmp, "wb") as f:
                pickle.dump(val, f, protocol=4)
            os.replace(tmp, path)
        except OSError:
            pass
    return val


def build_archive(n, size, methods=(8,), workers=None, first=0):
    """Generate n entries (entry i uses methods[i % len(methods)]) and serialise a ZIP32
    archive image.  Returns (archive_bytes, entries) with entries =
    [dict(name, method, data_off, comp_size, uncomp_size, crc32)].  Large corpora are cached on disk."""
    tag = "arc_v3_n%d_s%d_m%s_f%d" % (n, size, "-".join(str(m) for m in methods), first)
    return _cached(tag, lambda: _build_archive(n, size, methods, workers, first), 1000, n)


def _default_workers():
    """OTEZIP_CORPUS_NO_FORK=1 (what the tools set before a rocprofv3 pass): a corpus that is not in the cache is
    generated in this process, slowly, instead of by a forked pool -- a process the profiler has GPU-initialised must not fork."""
    if os.environ.get("OTEZIP_CORPUS_NO_FORK", "0") not in ("", "0"):
        return 1
    return min(mp.cpu_count(), 16)


def _build_archive(n, size, methods, workers, first):
    jobs = [(first + i, size, methods[i % len(methods)]) for i in range(n)]
    if workers is None:
        workers = _default_workers()
    if workers > 1 and n >= 64:
        with mp.get_context("fork").Pool(workers) as pool:
            made = pool.map(_make_one, jobs, chunksize=max(1, n // (workers * 8)))
    else:
        made = [_make_one(j) for j in jobs]
    return _serialise(made)


def _serialise(made):
    n = len(made)
    out = bytearray()
    cd = bytearray()
    entries = []
    for idx, method, comp, usize, crc in made:
        name = ("e%06d.txt" % idx).encode()
        ofs = len(out)
        out += struct.pack("<IHHHHHIIIHH", 0x04034B50, 20, 0, method, 0, 0x21, crc, len(comp), usize, len(name), 0)
        out += name
        data_off = len(out)
        out += comp
        cd += struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 0x031E, 20, 0, method, 0, 0x21, crc, len(comp), usize,
                          len(name), 0, 0, 0, 0, 0o100644 << 16, ofs)
        cd += name
        entries.append(dict(name=name.decode(), method=method, data_off=data_off, comp_size=len(comp),
                            uncomp_size=usize, crc32=crc))
    cd_ofs = len(out)
    out += cd
    out += struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, n & 0xFFFF, n & 0xFFFF, len(cd), cd_ofs, 0)
    return bytes(out), entries




def _text_one(args):
    return gen_text(*args)


def text_buffers(n, size, first=0, workers=None):
    """The n uncompressed buffers of the corpus (what build_archive compresses), for the encode side."""
    return _cached("txt_v3_n%d_s%d_f%d" % (n, size, first), lambda: _text_buffers(n, size, first, workers), 1000, n)


def _text_buffers(n, size, first, workers):
    jobs = [(first + i, size) for i in range(n)]
    if workers is None:
        workers = _default_workers()
    if workers > 1 and n >= 64:
        with mp.get_context("fork").Pool(workers) as pool:
            return pool.map(_text_one, jobs, chunksize=max(1, n // (workers * 8)))
    return [_text_one(j) for j in jobs]

