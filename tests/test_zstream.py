"""The DEFLATE codec seam (inflateInit2/inflate/inflateEnd, deflateInit2/deflate/deflateEnd over
z_stream; ref src/lib/deflate.inc.c:89-97) driven the way the reference's own callers drive it
(ref src/lib/otezip.c:503-522, :826-843): whole input, whole output buffer, one call with Z_FINISH.
Behind the symbols is the GPU batch path; the checker is the oracle (and zlib for the encoder)."""
import ctypes as C
import os
import zlib

import pytest

import cases
from oracle import pyoracle as O

Z_FINISH = 4
SYMS = ["inflateInit2", "inflateInit2_", "inflate", "inflateEnd", "deflateInit2", "deflateInit2_", "deflate", "deflateEnd"]


def _lib():
    import otezip_amd as oz
    if not os.path.exists(oz.lib_path()):
        oz.build_native()
    return oz.lib()


def _stream(data, out_cap):
    inb = C.create_string_buffer(bytes(data), max(len(data), 1))
    out = C.create_string_buffer(b"\xCD" * max(out_cap, 1), max(out_cap, 1))
    s = O.ZStream()
    s.next_in = C.cast(inb, C.c_void_p)
    s.avail_in = len(data)
    s.next_out = C.cast(out, C.c_void_p)
    s.avail_out = out_cap
    return s, inb, out


def gpu_inflate(L, data, out_cap):
    s, inb, out = _stream(data, out_cap)
    assert L.inflateInit2(C.byref(s), -15) == 0
    ret = L.inflate(C.byref(s), Z_FINISH)
    tin, tout, left = s.total_in, s.total_out, s.avail_out
    again = L.inflate(C.byref(s), Z_FINISH) if ret == 1 else None
    assert L.inflateEnd(C.byref(s)) == 0
    return ret, tin, tout, left, out.raw[:out_cap], again


def gpu_deflate(L, data, out_cap):
    s, inb, out = _stream(data, out_cap)
    assert L.deflateInit2(C.byref(s), 6, 8, -15, 8, 0) == 0
    ret = L.deflate(C.byref(s), Z_FINISH)
    tout = s.total_out
    assert L.deflateEnd(C.byref(s)) == 0
    return ret, s.total_in, out.raw[:tout]


def test_seam_symbols_are_exported_and_fail_loudly_without_a_gpu():
    import torch
    L = _lib()
    for name in SYMS:
        assert hasattr(L, name), name
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    s, inb, out = _stream(b"\x03\x00", 16)
    assert L.inflateInit2(C.byref(s), -15) == 0
    assert L.inflate(C.byref(s), Z_FINISH) == -2 and s.msg   # Z_STREAM_ERROR: no CPU codec behind the seam
    assert s.total_out == 0 and out.raw[:4] == b"\xCD" * 4
    assert L.inflateEnd(C.byref(s)) == 0 and L.inflateEnd(C.byref(s)) == -2
    assert L.inflate(None, Z_FINISH) == -2 and L.deflateInit2(C.byref(s), 6, 9, -15, 8, 0) == -2


@pytest.mark.gpu
def test_inflate_matches_the_oracle(gpu_lib):
    L = gpu_lib.lib()
    n_ok = n_bad = n_short = 0
    for method, comp, usize, crc, flags in cases.deflate_valid() + cases.deflate_edge() + cases.deflate_fuzz(n=40):
        if method != 8 or usize == 0:
            continue
        r, ref_out = O.codec_decode(8, comp, usize)
        if r.verdict & O.V_NO_PARITY:
            continue
        ret, tin, tout, left, out, again = gpu_inflate(L, comp, usize)
        if r.correct_ret == 1:         # Z_STREAM_END (after the F2 repair)
            assert ret == 1 and tout == r.total_out and out[:tout] == ref_out[:tout], (len(comp), usize)
            assert tin == len(comp) and left == usize - tout and again == 1
            assert out[tout:] == b"\xCD" * (usize - tout)      # nothing written past the stream's output
            n_ok += 1
            n_short += tout < usize
        elif r.correct_ret == 0:       # reference: Z_OK with the buffer full; one-shot seam: buffer too small
            assert ret == -5 and tout == 0 and out == b"\xCD" * usize
        else:
            assert ret < 0, (ret, r.correct_ret)
            n_bad += 1
    assert n_ok > 10 and n_bad > 5


@pytest.mark.gpu
def test_deflate_writes_valid_streams(gpu_lib):
    from otezip_amd import corpus
    L = gpu_lib.lib()
    for data in (b"", b"x", b"abcabcabcabc" * 50, corpus.gen_text(3, 65536), os.urandom(5000), b"\0" * 100000):
        cap = len(data) + len(data) // 8 + 75        # compressBound, ref src/lib/deflate.inc.c:72-77
        ret, tin, comp = gpu_deflate(L, data, cap)
        assert ret == 1 and tin == len(data)
        assert zlib.decompress(comp, -15) == data      # an independent inflate reads it
        r2, _, tout, _, out, _ = gpu_inflate(L, comp, max(len(data), 1))
        assert r2 == 1 and tout == len(data) and out[:tout] == data
        if len(data) >= 600 and data[:3] == b"abc":
            assert len(comp) < len(data) // 4
    ret, tin, comp = gpu_deflate(L, os.urandom(4096), 64)   # output buffer too small
    assert ret == -5 and comp == b""
