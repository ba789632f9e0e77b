"""Measurement / test infrastructure (never imported by the product): archive images whose entries hold the
REAL formats -- Zstandard frames (method 93, the image's libzstd through tools/zstdlib.py), Brotli streams
(method 97, libbrotlienc through tools/brotlilib.py), LZMA1 per APPNOTE 5.8.8 (method 14, Python's lzma)."""
import multiprocessing as mp
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otezip_amd import corpus
from tools import zstdlib as Z


def _zstd_one(args):
    idx, size, level = args
    data = corpus.gen_text(idx, size)
    return idx, 93, Z.compress(data, level=level), len(data), zlib.crc32(data) & 0xFFFFFFFF


def build_zstd_real(n, size, level=3, first=0, workers=None):
    """(archive_bytes, entries) like corpus.build_archive, the payloads libzstd frames with content checksums."""
    def make():
        jobs = [(first + i, size, level) for i in range(n)]
        w = workers if workers is not None else min(mp.cpu_count(), 16)
        if w > 1 and n >= 64:
            with mp.get_context("fork").Pool(w) as pool:
                made = pool.map(_zstd_one, jobs, chunksize=max(1, n // (w * 8)))
        else:
            made = [_zstd_one(j) for j in jobs]
        return corpus._serialise(made)
    return corpus._cached("zstdreal_v1_n%d_s%d_l%d_f%d" % (n, size, level, first), make, 1000, n)


def _brotli_one(args):
    from tools import brotlilib as B
    idx, size, quality = args
    data = corpus.gen_text(idx, size)
    return idx, 97, B.compress(data, quality=quality, lgwin=22), len(data), zlib.crc32(data) & 0xFFFFFFFF


def _lzma_one(args):
    import lzma
    import struct
    idx, size, preset = args
    data = corpus.gen_text(idx, size)
    f = [{"id": lzma.FILTER_LZMA1, "preset": preset, "lc": 3, "lp": 0, "pb": 2, "dict_size": 1 << 20}]
    raw = lzma.compress(data, format=lzma.FORMAT_RAW, filters=f)
    payload = b"\x09\x14\x05\x00" + bytes([(2 * 5 + 0) * 9 + 3]) + struct.pack("<I", 1 << 20) + raw
    return idx, 14, payload, len(data), zlib.crc32(data) & 0xFFFFFFFF


def _lzfse_one(args):
    from tools import lzfse_model as M
    idx, size, kind = args
    data = corpus.gen_text(idx, size)
    return idx, 100, M.encode(data, ("lzfse", "lzvn")[kind]), len(data), zlib.crc32(data) & 0xFFFFFFFF


def _build(tag, one, n, size, knob, first, workers):
    def make():
        jobs = [(first + i, size, knob) for i in range(n)]
        w = workers if workers is not None else min(mp.cpu_count(), 16)
        if w > 1 and n >= 64:
            with mp.get_context("fork").Pool(w) as pool:
                made = pool.map(one, jobs, chunksize=max(1, n // (w * 8)))
        else:
            made = [one(j) for j in jobs]
        return corpus._serialise(made)
    return corpus._cached("%s_v1_n%d_s%d_k%d_f%d" % (tag, n, size, knob, first), make, 1000, n)


def build_brotli_real(n, size, quality=5, first=0, workers=None):
    """(archive_bytes, entries): method-97 payloads are Brotli streams (RFC 7932) as libbrotlienc writes them."""
    return _build("brotlireal", _brotli_one, n, size, quality, first, workers)


def build_lzma_real(n, size, preset=1, first=0, workers=None):
    """(archive_bytes, entries): method-14 payloads are APPNOTE 5.8.8 headers + raw LZMA1 streams from liblzma."""
    return _build("lzmareal", _lzma_one, n, size, preset, first, workers)


def build_lzfse_real(n, size, kind=0, first=0, workers=None):
    """(archive_bytes, entries): method-100 payloads are LZFSE streams as tools/lzfse_model.py writes them (kind 0:
    bvx2 blocks, 1: LZVN) -- a model of the format, parity unpinned."""
    return _build("lzfsereal", _lzfse_one, n, size, kind, first, workers)
