"""Shared helpers for the test-suite (test infrastructure, not product code)."""
import ctypes as C
import os
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle as O  # noqa: E402


class OzgEntry(C.Structure):
    # include/otezip_gpu.h: struct ozg_entry (32 bytes)
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("comp_size", C.c_uint32),
                ("uncomp_size", C.c_uint32), ("crc32", C.c_uint32), ("method", C.c_uint16), ("flags", C.c_uint16)]


class OzgResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("total_out", C.c_uint32), ("crc32", C.c_uint32), ("flags", C.c_uint32)]


ST_OK, ST_NEED_OUT, ST_DATA, ST_BUF, ST_SIZE, ST_BOMB, ST_METHOD, ST_INTERNAL = 1, 0, -3, -5, -20, -21, -22, -30
F_REF_FALSE_REJECT, F_SHORT_STREAM, F_CRC_MISMATCH = 1, 2, 4


def raw_deflate(data, level=6, strategy=0, flush_every=None):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    if flush_every:
        out = b""
        for i in range(0, len(data), flush_every):
            out += c.compress(data[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH)
        return out + c.flush()
    return c.compress(data) + c.flush()


def layout(items):
    """items: [(method, comp_bytes, uncomp_size, crc, flags)] -> (src_bytes, entries array, dst_len)."""
    n = len(items)
    ents = (OzgEntry * max(n, 1))()
    src = bytearray()
    dst_off = 0
    for i, (method, comp, usize, crc, flags) in enumerate(items):
        # deliberately odd source offsets: payloads in a ZIP are unaligned
        src += b"\xEE" * ((7 * i + 3) % 13)
        ents[i].src_off = len(src)
        src += comp
        ents[i].dst_off = dst_off
        ents[i].comp_size = len(comp)
        ents[i].uncomp_size = usize
        ents[i].crc32 = crc & 0xFFFFFFFF
        ents[i].method = method
        ents[i].flags = flags
        dst_off += (usize + 15) & ~15
        dst_off += 16  # guard gap
    return bytes(src) + b"\xEE" * 32, ents, dst_off + 16


_emu = None
_emu_blob = None


def emu_lib():
    """CPU emulation of the kernel bodies (tests/simt_emu)."""
    global _emu
    if _emu is None:
        d = os.path.join(ROOT, "tests", "simt_emu")
        name = "libotz_emu.so"
        if os.environ.get("OTZ_EMU_ASAN"):   # sanitizer run: LD_PRELOAD=$(gcc -print-file-name=libasan.so) OTZ_EMU_ASAN=1 pytest ...
            name = "libotz_emu_asan.so"
            subprocess.check_call(["make", "-s", "-C", d, os.path.join(d, name)])
        else:
            subprocess.check_call(["make", "-s", "-C", d])
        L = C.CDLL(os.path.join(d, name))
        L.emu_decode_batch.restype = C.c_int
        L.emu_decode_batch.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(OzgEntry),
                                       C.POINTER(OzgResult), C.c_uint32, C.c_uint64, C.c_uint64]
        L.emu_decode_batch_mode.restype = C.c_int
        L.emu_decode_batch_mode.argtypes = L.emu_decode_batch.argtypes + [C.c_int, C.c_uint32]
        L.emu_lds_bytes_index.restype = C.c_uint32
        L.emu_lds_bytes_inflate.restype = C.c_uint32
        L.emu_lds_bytes_copy.restype = C.c_uint32
        L.emu_lds_bytes_apply.restype = C.c_uint32
        global _emu_blob  # the RFC 7932 constant data the Brotli decoder reads (the library links it in)
        _emu_blob = C.create_string_buffer(open(os.path.join(ROOT, "otezip_amd", "data", "brotli_rfc7932.bin"), "rb").read())
        L.emu_set_brotli_blob(_emu_blob)
        _emu = L
    return _emu


def emu_decode(items, ratio=1000, slack=1 << 20, mode=0, region_pct=100):
    """Run items through the emulated kernels. Returns [(OzgResult, out_bytes)].
    mode 0: product path (index + apply kernels); 1: the workgroup-per-entry routine alone;
    2: product path without arena regions (forces the in-kernel fallback)."""
    src, ents, dst_len = layout(items)
    dst = C.create_string_buffer(b"\xCD" * dst_len, dst_len)
    res = (OzgResult * max(len(items), 1))()
    emu_lib().emu_decode_batch_mode(src, len(src), dst, dst_len, ents, res, len(items), ratio, slack, mode, region_pct)
    out = []
    for i, it in enumerate(items):
        o = ents[i].dst_off
        out.append((res[i], dst.raw[o:o + it[2]], dst.raw[o + it[2]:o + ((it[2] + 15) & ~15) + 16]))
    return out


def check_against_oracle(items, results, what=""):
    """The parity assertion: same accept/reject as the oracle (F2 repaired), identical
    bytes, identical CRC, flags consistent."""
    for (method, comp, usize, crc, flags), (r, out, guard) in zip(items, results):
        o = O.extract_entry(method, comp, usize, crc, ignore_zipbomb=flags & 1)
        tag = "%s method=%d comp=%d usize=%d" % (what, method, len(comp), usize)
        assert r.status != -999, "no result written: " + tag
        if o["verdict"] & O.V_NO_PARITY:
            assert r.status != ST_OK or True  # no parity claimed (reference reads indeterminate memory)
            continue
        ok = r.status == ST_OK
        assert ok == (o["correct_rc"] == 0), "%s: status %d vs oracle rc %d/%d verdict %d" % (
            tag, r.status, o["rc"], o["correct_rc"], o["verdict"])
        if ok:
            assert out == o["out"], tag + ": bytes differ"
            assert r.crc32 == o["crc"], tag + ": crc differs"
            assert bool(r.flags & F_CRC_MISMATCH) == bool(o["verdict"] & O.V_CRC_MISMATCH), tag
            assert bool(r.flags & F_REF_FALSE_REJECT) == bool(o["verdict"] & O.V_REF_FALSE_REJECT), tag + ": F2 flag"
            assert bool(r.flags & F_SHORT_STREAM) == bool(o["verdict"] & O.V_SHORT_STREAM), tag + ": F12 flag"
        assert set(guard[-16:]) <= {0xCD}, tag + ": wrote past the entry's region"


class OzgEncEntry(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("src_size", C.c_uint32), ("dst_cap", C.c_uint32),
                ("method", C.c_uint16), ("level", C.c_uint16), ("reserved", C.c_uint32)]


class OzgEncResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("comp_size", C.c_uint32), ("crc32", C.c_uint32), ("method", C.c_uint16),
                ("flags", C.c_uint16)]


def enc_layout(bufs, method=8):
    """[(bytes)] -> (src, entries, dst_len) for the encode ABI."""
    n = len(bufs)
    ents = (OzgEncEntry * max(n, 1))()
    src = bytearray()
    dst_off = 0
    for i, b in enumerate(bufs):
        src += b"\x11" * ((5 * i + 1) % 7)
        ents[i].src_off = len(src)
        src += b
        ents[i].src_size = len(b)
        cap = max(len(b) + len(b) // 8 + 75, 16)
        ents[i].dst_cap = cap
        ents[i].dst_off = dst_off
        ents[i].method = method if not isinstance(method, (list, tuple)) else method[i]
        dst_off += (cap + 15) & ~15
    return bytes(src) + b"\x11" * 16, ents, dst_off + 16


def emu_encode(bufs, method=8):
    src, ents, dst_len = enc_layout(bufs, method)
    dst = C.create_string_buffer(b"\xCD" * dst_len, dst_len)
    res = (OzgEncResult * max(len(bufs), 1))()
    L = emu_lib()
    L.emu_encode_batch.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(OzgEncEntry), C.POINTER(OzgEncResult), C.c_uint32]
    L.emu_encode_batch(src, dst, ents, res, len(bufs))
    return [(res[i], dst.raw[ents[i].dst_off:ents[i].dst_off + res[i].comp_size]) for i in range(len(bufs))]
