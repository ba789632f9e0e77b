"""TEST INFRASTRUCTURE: the image's own libzstd (1.4.8, /usr/lib/x86_64-linux-gnu/libzstd.so.1, no headers)
through ctypes -- a third-party Zstandard implementation to pin the RFC 8878 decoders against
(oracle/otz_zstd_rfc.c, the GPU path under ZIP method 93).  None of this is reachable from the product."""
import ctypes as C
import ctypes.util

_L = None

# ZSTD_cParameter values (zstd.h, stable API of 1.4.x)
C_LEVEL, C_WINDOWLOG, C_HASHLOG, C_CHAINLOG, C_SEARCHLOG, C_MINMATCH, C_TARGETLEN, C_STRATEGY = 100, 101, 102, 103, 104, 105, 106, 107
C_CONTENTSIZE, C_CHECKSUM, C_DICTID = 200, 201, 202


def lib():
    global _L
    if _L is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        try:
            L = C.CDLL(name)
        except OSError:
            return None
        L.ZSTD_compressBound.restype = C.c_size_t
        L.ZSTD_compressBound.argtypes = [C.c_size_t]
        L.ZSTD_isError.restype = C.c_uint
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.ZSTD_decompress.restype = C.c_size_t
        L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.ZSTD_createCCtx.restype = C.c_void_p
        L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        L.ZSTD_CCtx_setParameter.restype = C.c_size_t
        L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ZSTD_compress2.restype = C.c_size_t
        L.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.ZSTD_versionNumber.restype = C.c_uint
        _L = L
    return _L


def available():
    return lib() is not None


def version():
    v = lib().ZSTD_versionNumber()
    return "%d.%d.%d" % (v // 10000, v // 100 % 100, v % 100)


def compress(data, level=3, checksum=True, content_size=True, **params):
    """One frame.  params: windowlog=, minmatch=, strategy=, ... (advanced parameters by lower-case name)."""
    L = lib()
    cctx = L.ZSTD_createCCtx()
    try:
        def setp(p, v):
            r = L.ZSTD_CCtx_setParameter(cctx, p, v)
            if L.ZSTD_isError(r):
                raise ValueError("ZSTD_CCtx_setParameter(%d, %d) failed" % (p, v))
        setp(C_LEVEL, level)
        setp(C_CHECKSUM, 1 if checksum else 0)
        setp(C_CONTENTSIZE, 1 if content_size else 0)
        names = {"windowlog": C_WINDOWLOG, "hashlog": C_HASHLOG, "chainlog": C_CHAINLOG, "searchlog": C_SEARCHLOG,
                 "minmatch": C_MINMATCH, "targetlen": C_TARGETLEN, "strategy": C_STRATEGY}
        for k, v in params.items():
            setp(names[k], v)
        cap = L.ZSTD_compressBound(len(data))
        out = C.create_string_buffer(cap)
        n = L.ZSTD_compress2(cctx, out, cap, data, len(data))
        if L.ZSTD_isError(n):
            raise ValueError("ZSTD_compress2 failed")
        return out.raw[:n]
    finally:
        L.ZSTD_freeCCtx(cctx)


def decompress(frame, cap):
    """(ok, bytes) as libzstd sees the frame(s)."""
    L = lib()
    out = C.create_string_buffer(max(cap, 1))
    n = L.ZSTD_decompress(out, cap, frame, len(frame))
    if L.ZSTD_isError(n):
        return False, b""
    return True, out.raw[:n]
