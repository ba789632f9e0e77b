"""Measurement / test infrastructure (never imported by the product): archive images whose method-93 entries
hold REAL Zstandard frames, written by the image's libzstd through tools/zstdlib.py."""
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
