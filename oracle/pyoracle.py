"""ctypes bindings for the CPU oracle (oracle/libotz_oracle.so) and, when it was
built, for the reference itself (oracle/_ref/libotezip_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (otezip_amd) never imports it.
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libotz_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libotezip_ref.so")
REF_BENCH = os.path.join(HERE, "_ref", "ref_bench")

Z_OK, Z_STREAM_END, Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR = 0, 1, -2, -3, -4, -5
Z_FINISH = 4
M_STORE, M_DEFLATE, M_LZMA, M_ZSTD, M_BROTLI, M_LZFSE = 0, 8, 14, 93, 97, 100
V_REF_FALSE_REJECT, V_REF_UNDEFINED, V_SHORT_STREAM, V_CRC_MISMATCH, V_OVERSUBSCRIBED = 1, 2, 4, 8, 16
V_NO_PARITY = V_REF_UNDEFINED | V_OVERSUBSCRIBED  # malformed input: the new build only has to fail cleanly


def build(force=False):
    """Compile the oracle (and the reference when /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "otz_oracle.c"))):
        subprocess.check_call(["make", "-s", "-C", HERE, os.path.join(HERE, "libotz_oracle.so")])
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class CodecResult(C.Structure):
    _fields_ = [("ret", C.c_int), ("correct_ret", C.c_int), ("total_in", C.c_uint32),
                ("total_out", C.c_uint32), ("verdict", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        L.otz_crc32.restype = C.c_uint32
        L.otz_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        for name in ("otz_inflate_raw", "otz_zstd_decode", "otz_lzfse_decode", "otz_brotli_decode", "otz_lzma_decode"):
            f = getattr(L, name)
            f.restype = CodecResult
            f.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.otz_extract_entry.restype = C.c_int
        L.otz_extract_entry.argtypes = [C.c_uint16, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                        C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        for name in ("otz_zstd_encode", "otz_lzfse_encode", "otz_brotli_encode", "otz_lzma_encode"):
            if hasattr(L, name):
                f = getattr(L, name)
                f.restype = C.c_uint64
                f.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        if hasattr(L, "otz_compress_bound"):
            L.otz_compress_bound.restype = C.c_uint64
            L.otz_compress_bound.argtypes = [C.c_uint16, C.c_uint64]
        L.otz_compress_data_private.restype = C.c_uint64
        L.otz_compress_data_private.argtypes = [C.c_uint16, C.c_char_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint16)]
        _lib = L
    return _lib


def compress_data_private(method, data):
    """Oracle of otezip_compress_data for methods 93/100/97/14 -> (method_written, payload)."""
    data = bytes(data)
    cap = max(len(data), int(lib().otz_compress_bound(method, len(data)))) + 16
    out = C.create_string_buffer(cap)
    m = C.c_uint16(0)
    n = lib().otz_compress_data_private(method, data, len(data), out, C.byref(m))
    return m.value, out.raw[:n]


def ref_compress_data(method, data):
    """What the reference's zip_file_add writes for one buffer: (method, payload), read back from
    an archive it created (default_method poked the way its CLI does, src/main.c:190)."""
    import zipfile
    R = ref()

    class Zip(C.Structure):
        _fields_ = [("fp", C.c_void_p), ("entries", C.c_void_p), ("n_entries", C.c_uint64), ("mode", C.c_int),
                    ("next_index", C.c_uint64), ("default_method", C.c_uint16)]
    R.zip_open.restype = C.POINTER(Zip)
    R.zip_source_buffer.restype = C.c_void_p
    R.zip_source_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int]
    R.zip_file_add.restype = C.c_int64
    R.zip_file_add.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    R.zip_close.argtypes = [C.c_void_p]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.zip")
        err = C.c_int(0)
        za = R.zip_open(path.encode(), 1 | 8, C.byref(err))
        za.contents.default_method = method
        data = bytes(data)
        keep = C.create_string_buffer(data, max(len(data), 1))
        src = R.zip_source_buffer(za, keep, len(data), 0)
        assert R.zip_file_add(za, b"e", src, 0) == 0
        R.zip_close(za)
        R.zip_open.restype = C.c_void_p
        raw = open(path, "rb").read()
    m, csize = struct.unpack_from("<H", raw, 8)[0], struct.unpack_from("<I", raw, 18)[0]
    nlen = struct.unpack_from("<H", raw, 26)[0]
    return m, raw[30 + nlen:30 + nlen + csize]


def crc32(data, crc=0):
    return lib().otz_crc32(crc, bytes(data), len(data))


_DECODERS = {M_DEFLATE: "otz_inflate_raw", M_ZSTD: "otz_zstd_decode", M_LZFSE: "otz_lzfse_decode",
             M_BROTLI: "otz_brotli_decode", M_LZMA: "otz_lzma_decode"}


def codec_decode(method, data, out_cap):
    """Run one oracle codec. Returns (CodecResult, bytes(out_cap))."""
    data = bytes(data)
    out = C.create_string_buffer(max(out_cap, 1))
    r = getattr(lib(), _DECODERS[method])(data, len(data), out, out_cap)
    return r, out.raw[:out_cap]


def extract_entry(method, comp, uncomp_size, dir_crc, strict_crc=0, ignore_zipbomb=0,
                  max_ratio=1000, max_slack=1 << 20):
    """Oracle of otezip_extract_entry. Returns dict(rc, correct_rc, out, crc, verdict)."""
    comp = bytes(comp)
    out = C.create_string_buffer(max(uncomp_size, 1))
    crc = C.c_uint32(0)
    verdict = C.c_uint32(0)
    crc_rc = C.c_int(0)
    rc = lib().otz_extract_entry(method, comp, len(comp), uncomp_size, dir_crc, strict_crc, ignore_zipbomb,
                                 max_ratio, max_slack, out, C.byref(crc), C.byref(verdict), C.byref(crc_rc))
    return dict(rc=rc, correct_rc=crc_rc.value, out=out.raw[:uncomp_size], crc=crc.value, verdict=verdict.value)


# --------------------------------------------------------------------------- reference


class ZStream(C.Structure):
    # ref: src/include/otezip/zstream.h:14-33
    _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_uint), ("total_in", C.c_ulong),
                ("next_out", C.c_void_p), ("avail_out", C.c_uint), ("total_out", C.c_ulong),
                ("msg", C.c_char_p), ("state", C.c_void_p), ("zalloc", C.c_void_p), ("zfree", C.c_void_p),
                ("opaque", C.c_void_p), ("data_type", C.c_int), ("adler", C.c_ulong), ("reserved", C.c_ulong)]


class RefZipFile(C.Structure):
    # ref: src/include/otezip/zip.h:100-104
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("size", C.c_uint32), ("pos", C.c_uint64)]


_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.zip_open.restype = C.c_void_p
        R.zip_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        R.zip_close.argtypes = [C.c_void_p]
        R.zip_get_num_files.restype = C.c_uint64
        R.zip_get_num_files.argtypes = [C.c_void_p]
        R.zip_fopen_index.restype = C.POINTER(RefZipFile)
        R.zip_fopen_index.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        R.zip_fclose.argtypes = [C.POINTER(RefZipFile)]
        _ref = R
    return _ref


_REF_DEC = {M_DEFLATE: ("inflateInit2", "inflate", "inflateEnd"),
            M_ZSTD: ("zstdDecompressInit", "zstdDecompress", "zstdDecompressEnd"),
            M_LZFSE: ("lzfseDecompressInit", "lzfseDecompress", "lzfseDecompressEnd"),
            M_BROTLI: ("brotliDecompressInit", "brotliDecompress", "brotliDecompressEnd"),
            M_LZMA: ("lzmaDecompressInit", "lzmaDecompress", "lzmaDecompressEnd")}
_REF_ENC = {M_DEFLATE: ("deflateInit2", "deflate", "deflateEnd"),
            M_ZSTD: ("zstdInit", "zstdCompress", "zstdEnd"),
            M_LZFSE: ("lzfseInit", "lzfseCompress", "lzfseEnd"),
            M_BROTLI: ("brotliInit", "brotliCompress", "brotliEnd"),
            M_LZMA: ("lzmaInit", "lzmaCompress", "lzmaEnd")}


def ref_codec_decode(method, data, out_cap):
    """Drive the reference codec exactly as src/lib/otezip.c:503-522 (and the
    sibling branches) does: zeroed output, one call with Z_FINISH.
    Returns (ret, total_in, total_out, out_bytes)."""
    R = ref()
    init, run, end = _REF_DEC[method]
    data = bytes(data)
    inb = C.create_string_buffer(data, max(len(data), 1))
    out = C.create_string_buffer(max(out_cap, 1))
    s = ZStream()
    s.next_in = C.cast(inb, C.c_void_p)
    s.avail_in = len(data)
    s.next_out = C.cast(out, C.c_void_p)
    s.avail_out = out_cap
    if method == M_DEFLATE:
        rc = getattr(R, init)(C.byref(s), -15)
    else:
        rc = getattr(R, init)(C.byref(s))
    assert rc == Z_OK
    ret = getattr(R, run)(C.byref(s), Z_FINISH)
    tin, tout = s.total_in, s.total_out
    getattr(R, end)(C.byref(s))
    return ret, tin, tout, out.raw[:out_cap]


def ref_codec_encode(method, data, out_cap):
    """Drive the reference encoder as otezip_compress_data does
    (src/lib/otezip.c:826-843, :866-892, :937-959, :981-1004, :1025-1062).
    Returns (ret, bytes_out)."""
    R = ref()
    init, run, end = _REF_ENC[method]
    data = bytes(data)
    inb = C.create_string_buffer(data, max(len(data), 1))
    out = C.create_string_buffer(max(out_cap, 1))
    s = ZStream()
    s.next_in = C.cast(inb, C.c_void_p)
    s.avail_in = len(data)
    s.next_out = C.cast(out, C.c_void_p)
    s.avail_out = out_cap
    if method == M_DEFLATE:
        rc = R.deflateInit2(C.byref(s), -1, 8, -15, 8, 0)
    elif method == M_ZSTD:
        rc = R.zstdInit(C.byref(s), 1)  # this codec's Z_DEFAULT_COMPRESSION is 1 (src/lib/zstd.inc.c:60)
    else:
        rc = getattr(R, init)(C.byref(s), -1)
    assert rc == Z_OK
    ret = Z_OK
    if method == M_ZSTD:
        while ret == Z_OK:
            ret = R.zstdCompress(C.byref(s), Z_FINISH)
    else:
        ret = getattr(R, run)(C.byref(s), Z_FINISH)
    tout = s.total_out
    getattr(R, end)(C.byref(s))
    return ret, out.raw[:tout]


def ref_set_globals(verify_crc=0, ignore_zipbomb=0, ratio=1000, slack=1 << 20):
    R = ref()
    C.c_int.in_dll(R, "otezip_verify_crc").value = verify_crc
    C.c_int.in_dll(R, "otezip_ignore_zipbomb").value = ignore_zipbomb
    C.c_uint64.in_dll(R, "otezip_max_expansion_ratio").value = ratio
    C.c_uint64.in_dll(R, "otezip_max_expansion_slack").value = slack


def ref_extract_archive(path):
    """zip_open + zip_fopen_index over every entry with the reference.
    Returns a list with bytes or None (NULL) per entry."""
    R = ref()
    err = C.c_int(0)
    za = R.zip_open(path.encode(), 0, C.byref(err))
    if not za:
        raise RuntimeError("reference zip_open failed: %d" % err.value)
    res = []
    for i in range(R.zip_get_num_files(za)):
        zf = R.zip_fopen_index(za, i, 0)
        if not zf:
            res.append(None)
            continue
        res.append(C.string_at(zf.contents.data, zf.contents.size))
        R.zip_fclose(zf)
    R.zip_close(za)
    return res


# --------------------------------------------------------------------------- crafted archives


def craft_zip(entries):
    """Serialise [(name, method, payload_bytes, uncomp_size, crc32)] into a ZIP32
    archive with the field layout the reference parses (LFH/CDH/EOCD,
    ref: src/lib/otezip.c:360-378, :429-437)."""
    out = bytearray()
    cd = bytearray()
    for name, method, payload, usize, crc in entries:
        nb = name.encode()
        ofs = len(out)
        out += struct.pack("<IHHHHHIIIHH", 0x04034B50, 20, 0, method, 0, 0x21, crc, len(payload), usize, len(nb), 0)
        out += nb
        out += payload
        cd += struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 0x031E, 20, 0, method, 0, 0x21, crc, len(payload), usize,
                          len(nb), 0, 0, 0, 0, 0o100644 << 16, ofs)
        cd += nb
    cd_ofs = len(out)
    out += cd
    out += struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, len(entries), len(entries), len(cd), cd_ofs, 0)
    return bytes(out)


def ref_extract_entry(method, comp, uncomp_size, dir_crc, **globals_):
    """The reference's zip_fopen_index on a one-entry crafted archive."""
    ref_set_globals(**globals_)
    with tempfile.NamedTemporaryFile(suffix=".zip", delete=False) as f:
        f.write(craft_zip([("e", method, bytes(comp), uncomp_size, dir_crc)]))
        path = f.name
    try:
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(2)
        os.dup2(devnull, 2)  # the reference prints CRC warnings to stderr
        try:
            return ref_extract_archive(path)[0]
        finally:
            os.dup2(saved, 2)
            os.close(saved)
            os.close(devnull)
    finally:
        os.unlink(path)
        ref_set_globals()


# --------------------------------------------------------------------------- private framings (inputs)


def frame_zstd(payload, magic=b"\x28\xb5\x2f\xfd", block=65535):
    """The hybrid the reference DEcoder accepts: LE magic, one
    ignored byte, raw blocks [hdr:1][len:LE16], then an empty last block."""
    out = bytearray(magic + b"\x70")
    for i in range(0, len(payload), block):
        b = payload[i:i + block]
        out += bytes([0]) + struct.pack("<H", len(b)) + b
    out += b"\x01\x00\x00"
    return bytes(out)


def frame_lzfse(payload):
    return b"\x06" + struct.pack("<I", len(payload)) + payload


def frame_brotli(payload):
    import zlib
    return b"BROT\x01" + struct.pack("<Q", len(payload)) + struct.pack("<I", zlib.crc32(payload) & 0xFFFFFFFF) + payload


def frame_lzma_tokens(payload, hdr_size=None):
    """Literal-run / RLE tokens in the reference's private method-14 format."""
    out = bytearray(b"\x5d\x00\x00\x00\x01" + struct.pack("<Q", len(payload) if hdr_size is None else hdr_size))
    i = 0
    n = len(payload)
    while i < n:
        j = i
        while j < n and payload[j] == payload[i] and j - i < 255:
            j += 1
        if j - i >= 4:
            out += bytes([0, payload[i], j - i])
            i = j
        else:
            k = min(n - i, 3)
            out += bytes([1, k]) + payload[i:i + k]
            i += k
    return bytes(out)
