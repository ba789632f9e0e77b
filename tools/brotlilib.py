"""ctypes view of the image's libbrotli (1.0.9): the independent implementation the Brotli kernel is pinned
to (tests, fixture generator, corpus builder).  Test/tool infrastructure: the product never loads it."""
import ctypes as C

_enc = _dec = None


def _libs():
    global _enc, _dec
    if _enc is None:
        _enc = C.CDLL("libbrotlienc.so.1")
        _dec = C.CDLL("libbrotlidec.so.1")
        _enc.BrotliEncoderCompress.restype = C.c_int
        _enc.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
        _enc.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
        _enc.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
        _dec.BrotliDecoderDecompress.restype = C.c_int
        _dec.BrotliDecoderDecompress.argtypes = [C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
    return _enc, _dec


def available():
    try:
        _libs()
        return True
    except OSError:
        return False


def version():
    enc, dec = _libs()
    return "%x" % dec.BrotliDecoderVersion()


def compress(data, quality=5, lgwin=22, mode=0):
    """mode: 0 generic, 1 text (UTF-8), 2 font."""
    enc, _ = _libs()
    cap = enc.BrotliEncoderMaxCompressedSize(len(data)) or (len(data) + 1024)
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    if not enc.BrotliEncoderCompress(quality, lgwin, mode, len(data), data, C.byref(n), out):
        raise RuntimeError("BrotliEncoderCompress failed")
    return out.raw[:n.value]


def decompress(comp, cap):
    """Returns the decoded bytes, or None when libbrotlidec rejects the stream (or it needs more than cap bytes)."""
    _, dec = _libs()
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t(cap)
    rc = dec.BrotliDecoderDecompress(len(comp), comp, C.byref(n), out)
    return out.raw[:n.value] if rc == 1 else None
