#!/usr/bin/env python3
"""Writes otezip_amd/data/brotli_rfc7932.bin: the constant data of RFC 7932 that a Brotli decoder needs --
the static dictionary (Appendix A: 122 784 bytes, SHA-256 20e42eb1...), the word-list geometry (NDBITS,
section 8), the 121 word transforms with their prefix/suffix strings (Appendix B) and the literal context
lookup tables (section 7.1).  The RFC gives them as data; they are read here from the image's
libbrotlicommon (1.0.9), which exports them, and checked against the RFC's own checksum of the dictionary.

Layout (little endian): u32 magic 'OBRD', u32 version, then u32 offset/size pairs of the sections
    0 dictionary bytes          1 size_bits_by_length u8[32]     2 offsets_by_length u32[32]
    3 context lookup u8[2048]   4 transforms u8[121][3]          5 prefix_suffix bytes
    6 prefix_suffix_map u16[]
followed by the sections, each 16-byte aligned."""
import ctypes as C
import hashlib
import os
import struct
import sys


class Dict(C.Structure):
    _fields_ = [("size_bits_by_length", C.c_uint8 * 32), ("offsets_by_length", C.c_uint32 * 32),
                ("data_size", C.c_size_t), ("data", C.POINTER(C.c_uint8))]


class Transforms(C.Structure):
    _fields_ = [("prefix_suffix_size", C.c_uint16), ("prefix_suffix", C.POINTER(C.c_uint8)),
                ("prefix_suffix_map", C.POINTER(C.c_uint16)), ("num_transforms", C.c_uint32),
                ("transforms", C.POINTER(C.c_uint8)), ("params", C.POINTER(C.c_uint8)),
                ("cut_off", C.c_int16 * 10)]


def main(out):
    lib = C.CDLL("libbrotlicommon.so.1")
    lib.BrotliGetDictionary.restype = C.POINTER(Dict)
    lib.BrotliGetTransforms.restype = C.POINTER(Transforms)
    d = lib.BrotliGetDictionary().contents
    t = lib.BrotliGetTransforms().contents
    data = bytes(d.data[:d.data_size])
    assert len(data) == 122784
    assert hashlib.sha256(data).hexdigest() == "20e42eb1b511c21806d4d227d07e5dd06877d8ce7b3a817f378f313653f35c70"
    assert t.num_transforms == 121
    ps = bytes(t.prefix_suffix[:t.prefix_suffix_size])
    tr = bytes(t.transforms[:3 * t.num_transforms])
    n_map = max(max(tr[0::3]), max(tr[2::3])) + 1
    psm = [t.prefix_suffix_map[i] for i in range(n_map)]
    assert max(tr[1::3]) <= 20  # RFC 7932 transform types only (no shift transforms)
    ctx = bytes((C.c_uint8 * 2048).in_dll(lib, "_kBrotliContextLookupTable"))
    sections = [data, bytes(d.size_bits_by_length), struct.pack("<32I", *d.offsets_by_length), ctx, tr, ps,
                struct.pack("<%dH" % n_map, *psm)]
    head = 8 + 8 * len(sections)
    off = (head + 15) & ~15
    table, body = b"", b""
    for s in sections:
        table += struct.pack("<II", off, len(s))
        pad = (-len(s)) & 15
        body += s + b"\0" * pad
        off += len(s) + pad
    blob = b"OBRD" + struct.pack("<I", 1) + table
    blob += b"\0" * (((head + 15) & ~15) - len(blob)) + body
    with open(out, "wb") as f:
        f.write(blob)
    print(out, len(blob), "bytes;", n_map, "prefix/suffix strings; sha256", hashlib.sha256(blob).hexdigest()[:16])


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "otezip_amd", "data", "brotli_rfc7932.bin"))
