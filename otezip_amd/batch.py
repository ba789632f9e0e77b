"""Batch descriptors for the ozg_* ABI: the Python mirror of what the C host
builds from the central directory after zip_open (ref src/lib/otezip.c:347-393)."""
import ctypes as C

from . import _native as N


def make_entries(items, align=16):
    """items: iterable of (method, src_off, comp_size, uncomp_size, crc32[, flags]).
    Lays the outputs out back to back on 16-byte boundaries.
    Returns (ctypes array of OzgEntry, dst_len)."""
    items = list(items)
    ents = (N.OzgEntry * max(len(items), 1))()
    off = 0
    for i, it in enumerate(items):
        method, src_off, csize, usize, crc = it[:5]
        e = ents[i]
        e.method, e.src_off, e.comp_size, e.uncomp_size = method, src_off, csize, usize
        e.crc32 = crc & 0xFFFFFFFF
        e.flags = it[5] if len(it) > 5 else 0
        e.dst_off = off
        off += (usize + align - 1) & ~(align - 1)
    return ents, max(off, 16)


class Batch:
    """Result of one decode: per-entry status words plus the output arena."""

    def __init__(self, entries, results, dst, n):
        self.entries, self.results, self.dst, self.n = entries, results, dst, n

    def status(self, i):
        return self.results[i].status

    def ok(self, i):
        return self.results[i].status == N.ST_OK

    def data(self, i):
        e = self.entries[i]
        return bytes(self.dst[e.dst_off:e.dst_off + e.uncomp_size])


def decode_batch_host(src, items):
    """ozg_decode_batch_host on a bytes-like `src`; items as in make_entries."""
    items = list(items)
    ents, dst_len = make_entries(items)
    n = len(items)
    src = bytes(src)
    dst = C.create_string_buffer(dst_len)
    res = (N.OzgResult * max(n, 1))()
    N.check(N.lib().ozg_decode_batch_host(src, len(src), dst, dst_len, ents, res, n), "ozg_decode_batch_host")
    return Batch(ents, res, dst.raw, n)
