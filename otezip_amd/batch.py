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


def encode_batch_host(buffers, method=8):
    """ozg_encode_batch_host over a list of bytes objects (what zip_file_add does per entry,
    batched).  Returns [(method_written, payload_bytes, crc32_of_input)]."""
    bufs = [bytes(b) for b in buffers]
    n = len(bufs)
    ents = (N.OzgEncEntry * max(n, 1))()
    src = bytearray()
    off = 0
    L = N.lib()
    for i, b in enumerate(bufs):
        ents[i].src_off = len(src)
        src += b
        ents[i].src_size = len(b)
        ents[i].method = method
        cap = int(L.ozg_compress_bound(method, len(b)))
        ents[i].dst_cap = cap
        ents[i].dst_off = off
        off += (cap + 15) & ~15
    src += b"\0" * 16
    dst = C.create_string_buffer(off + 16)
    res = (N.OzgEncResult * max(n, 1))()
    N.check(L.ozg_encode_batch_host(bytes(src), len(src), dst, off + 16, ents, res, n), "ozg_encode_batch_host")
    out = []
    for i in range(n):
        if res[i].status != N.ST_OK:
            raise N.NativeError("encode entry %d failed: %d" % (i, res[i].status))
        out.append((res[i].method, dst.raw[ents[i].dst_off:ents[i].dst_off + res[i].comp_size], res[i].crc32))
    return out
