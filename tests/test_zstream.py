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


def gpu_inflate(L, data, out_cap, wbits=-15):
    s, inb, out = _stream(data, out_cap)
    assert L.inflateInit2(C.byref(s), wbits) == 0
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
    n_ok = n_bad = n_short = n_full = 0
    for method, comp, usize, crc, flags in cases.deflate_valid() + cases.deflate_edge() + cases.deflate_fuzz(n=40):
        if method != 8 or usize == 0:
            continue
        r, ref_out = O.codec_decode(8, comp, usize)
        if r.verdict & O.V_NO_PARITY:
            continue
        ret, tin, tout, left, out, again = gpu_inflate(L, comp, usize)
        assert ret == r.correct_ret, (ret, r.correct_ret, len(comp), usize)   # the reference's code (after the F2 repair)
        if r.correct_ret == 1:         # Z_STREAM_END
            assert tout == r.total_out and out[:tout] == ref_out[:tout], (len(comp), usize)
            assert tin == r.total_in and left == usize - tout and again == 1   # trailing bytes are not consumed
            assert out[tout:] == b"\xCD" * (usize - tout)      # nothing written past the stream's output
            n_ok += 1
            n_short += tout < usize
        elif r.correct_ret == 0:       # Z_OK: the buffer is full, the token that did not fit has been read
            assert tout == r.total_out == usize and out == ref_out[:usize] and tin == r.total_in and left == 0
            n_full += 1
        else:
            n_bad += 1
    assert n_ok > 10 and n_bad > 5 and n_full > 5


@pytest.mark.gpu
def test_inflate_honours_window_bits_like_the_reference(gpu_lib):
    """zlib / gzip / auto-detected framing (ref deflate-dec.inc.c:452-477, :555-584): the header is
    skipped; the trailer, which the reference leaves unread, is verified when the input
    holds it and counted in total_in (documented extension)."""
    import gzip
    L = gpu_lib.lib()
    data = cases.word_text(__import__("random").Random(5), 30000)
    raw = cases.raw_deflate(data)
    zl = zlib.compress(data, 6)
    gz = gzip.compress(data, 6, mtime=0)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz_named = b"\x1f\x8b\x08\x18" + b"\0" * 6 + b"name.txt\0" + b"a comment\0" + co.compress(data)[10:] + co.flush()
    for wbits, blob, hdr, trl in ((15, zl, 2, 4), (31, gz, 10, 8), (47, gz, 10, 8), (47, zl, 2, 4), (31, gz_named, 10 + 9 + 10, 8), (-15, raw, 0, 0)):
        ret, tin, tout, left, out, again = gpu_inflate(L, blob, len(data) + 10, wbits)
        assert ret == 1 and out[:tout] == data and tout == len(data), (wbits, ret)
        r, _ = O.codec_decode(8, blob[hdr:], len(data) + 10)
        assert tin == hdr + r.total_in + trl == len(blob), (wbits, tin, hdr, r.total_in)
        if trl:
            # a trailer cut short is not looked at (as in the reference); a wrong one is a data error
            ret, tin, tout, _, out, _ = gpu_inflate(L, blob[:-2], len(data) + 10, wbits)
            assert ret == 1 and out[:tout] == data and tin == hdr + r.total_in
            for k in (1, trl):
                bad = bytearray(blob)
                bad[-k] ^= 0x20
                assert gpu_inflate(L, bytes(bad), len(data) + 10, wbits)[0] == -3, (wbits, k)
    # the wrong framing is a data error, as in the reference
    for wbits, blob in ((31, zl), (15, gz), (47, b"\x00\x01junk")):
        assert gpu_inflate(L, blob, 100, wbits)[0] == -3


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


# ---- the four private container codecs behind the same seam, checked against the reference BUILD
# (oracle/_ref, compiled from /root/reference by oracle/Makefile) where it is present, else
# against the oracle's restatement of the same writers / readers.

CODECS = {93: "zstd", 100: "lzfse", 97: "brotli", 14: "lzma"}
ALL_SYMS = SYMS + [c + s for c in CODECS.values() for s in (
    "Init", "Compress", "End", "DecompressInit", "Decompress", "DecompressEnd", "CompressInit2", "CompressInit2_",
    "DecompressInit2", "DecompressInit2_")] + ["lzfse_compress", "lzfse_decompress"]


def test_every_reference_codec_symbol_is_exported():
    L = _lib()
    for name in ALL_SYMS:
        assert hasattr(L, name), name
    if O.have_ref():   # the reference's own export list, minus its zip_* API (covered in test_abi / test_zip_api)
        out = __import__("subprocess").run(["nm", "-D", "--defined-only", O.REF_SO], capture_output=True, text=True).stdout
        ref_syms = {l.split()[-1] for l in out.splitlines() if " T " in l}
        missing = {s for s in ref_syms if not hasattr(L, s)}
        assert not missing, missing


def gpu_codec(L, name, decode, data, out_cap):
    s, inb, out = _stream(data, out_cap)
    init = getattr(L, name + ("DecompressInit" if decode else "Init"))
    run = getattr(L, name + ("Decompress" if decode else "Compress"))
    end = getattr(L, name + ("DecompressEnd" if decode else "End"))
    assert (init(C.byref(s)) if decode else init(C.byref(s), -1)) == 0
    ret = run(C.byref(s), Z_FINISH)
    tin, tout = s.total_in, s.total_out
    assert end(C.byref(s)) == 0
    return ret, tin, tout, out.raw[:out_cap]


def _expected_encode(method, data, cap):
    if O.have_ref():
        return O.ref_codec_encode(method, data, cap)
    if method == 93:      # the encoder's own magic order and end marker (ref zstd.inc.c:254-414)
        enc = b"\xfd\x2f\xb5\x28\x70" + b"".join(
            b"\x00" + len(data[i:i + 65535]).to_bytes(2, "little") + data[i:i + 65535] for i in range(0, len(data), 65535)) + b"\x01\x00\x00"
    else:
        enc = {100: O.frame_lzfse, 97: O.frame_brotli, 14: O.frame_lzma_tokens}[method](data)
    return (1, enc) if len(enc) <= cap else (-5, b"")


@pytest.mark.gpu
def test_container_writers_are_byte_exact(gpu_lib):
    from otezip_amd import corpus
    L = gpu_lib.lib()
    bufs = [b"", b"a", b"hello hello hello", corpus.gen_text(9, 70000), os.urandom(3000), b"\0" * 200000, b"ab" * 40000]
    for method, name in CODECS.items():
        for d in bufs:
            if method == 14 and len(d) > 65536 and not O.have_ref():
                continue
            cap = 2 * len(d) + 1100
            want_ret, want = _expected_encode(method, d, cap)
            ret, tin, tout, out = gpu_codec(L, name, False, d, cap)
            assert (ret == 1) == (want_ret == 1), (name, len(d), ret, want_ret)
            if ret == 1:
                assert out[:tout] == want and tin == len(d), (name, len(d))
        ret, _, tout, _ = gpu_codec(L, name, False, os.urandom(5000), 100)     # no room
        assert ret == -5 and tout == 0


@pytest.mark.gpu
def test_container_readers_match_the_reference(gpu_lib):
    L = gpu_lib.lib()
    n_ok = n_bad = 0
    for method, comp, usize, crc, flags in cases.containers():
        if method not in CODECS or usize > 300000:
            continue
        cap = usize + 64                                   # a codec caller's buffer is rarely exact
        r, ref_out = O.codec_decode(method, comp, cap)
        ret, tin, tout, out = gpu_codec(L, CODECS[method], True, comp, cap)
        if r.ret == 1:
            assert ret == 1 and tout == r.total_out and out[:tout] == ref_out[:tout], (method, len(comp), usize)
            n_ok += 1
        else:
            assert ret != 1, (method, len(comp), usize, r.ret)
            n_bad += 1
    assert n_ok >= 20 and n_bad >= 4
    d = b"round trip through the buffer API " * 100
    comp = C.create_string_buffer(len(d) + 64)
    n = L.lzfse_compress(d, len(d), comp, len(comp))
    assert n == len(d) + 5
    back = C.create_string_buffer(len(d))
    assert L.lzfse_decompress(comp, n, back, len(back)) == len(d) and back.raw == d
    assert L.lzfse_decompress(comp, n, back, 10) == 0


@pytest.mark.gpu
def test_deflate_frames_like_zlib_for_every_window_bits(gpu_lib):
    """deflateInit2 honours windowBits the way zlib does (the reference refuses > 15, which is why its own
    CLI's -g fails): raw, zlib framing with the Adler-32 taken on the device, gzip framing with
    CRC-32 and ISIZE.  Python's zlib / gzip read all three."""
    import gzip
    from otezip_amd import corpus
    L = gpu_lib.lib()
    for data in (b"", b"a", corpus.gen_text(3, 70000), bytes(range(256)) * 300, b"\xff" * 5000):
        for wbits in (-15, 15, 9, 31, 25):
            s, inb, out = _stream(data, len(data) + len(data) // 8 + 200)
            assert L.deflateInit2(C.byref(s), 6, 8, wbits, 8, 0) == 0
            assert L.deflate(C.byref(s), Z_FINISH) == 1
            blob = out.raw[:s.total_out]
            assert s.total_in == len(data)
            assert L.deflateEnd(C.byref(s)) == 0
            if wbits < 0:
                assert zlib.decompress(blob, -15) == data
            elif wbits < 16:
                assert zlib.decompress(blob) == data and s.adler == zlib.adler32(data)
            else:
                assert gzip.decompress(blob) == data and s.adler == zlib.crc32(data)
    s, inb, out = _stream(b"x", 64)
    for bad in (0, 7, 16, 23, 32, 47, -7, -16):
        assert L.deflateInit2(C.byref(s), 6, 8, bad, 8, 0) == -2, bad
    # no room for the framing itself
    s, inb, out = _stream(b"x" * 100, 9)
    assert L.deflateInit2(C.byref(s), 6, 8, 31, 8, 0) == 0 and L.deflate(C.byref(s), Z_FINISH) == -5
    L.deflateEnd(C.byref(s))
    # The one observable difference from the reference, pinned: windowBits 8..15 (what the deflateInit() macro
    # passes) gives the RFC 1950 framing here and a raw stream there (ref deflate-enc.inc.c:207-215);
    # OTEZIP_DEFLATE_RAW_8_15=1 restores the reference's reading for callers that frame the bytes themselves.
    data = corpus.gen_text(7, 5000)
    s, inb, out = _stream(data, len(data) + 300)
    assert L.deflateInit2(C.byref(s), 6, 8, 15, 8, 0) == 0 and L.deflate(C.byref(s), Z_FINISH) == 1
    framed = out.raw[:s.total_out]
    L.deflateEnd(C.byref(s))
    assert framed[0] & 0x0F == 8 and (framed[0] * 256 + framed[1]) % 31 == 0 and zlib.decompress(framed) == data
    os.environ["OTEZIP_DEFLATE_RAW_8_15"] = "1"
    try:
        s, inb, out = _stream(data, len(data) + 300)
        assert L.deflateInit2(C.byref(s), 6, 8, 15, 8, 0) == 0 and L.deflate(C.byref(s), Z_FINISH) == 1
        raw = out.raw[:s.total_out]
        L.deflateEnd(C.byref(s))
    finally:
        del os.environ["OTEZIP_DEFLATE_RAW_8_15"]
    assert zlib.decompress(raw, -15) == data and len(raw) == len(framed) - 6


@pytest.mark.gpu
def test_inflate_goes_on_when_the_caller_makes_room(gpu_lib):
    """The protocol of the reference's own CLI (src/main.c:612-650): Z_OK with the buffer full, Z_BUF_ERROR
    while there is no room, and after the caller has grown its buffer the stream goes on -- here until a
    gzip member that expands 16 x has been read through a buffer that starts at a twentieth of its size."""
    import gzip
    from otezip_amd import corpus
    L = gpu_lib.lib()
    data = (corpus.gen_text(9, 4000) * 40) + bytes(70000) + corpus.gen_text(10, 30000)
    blob = gzip.compress(data, 6, mtime=0)
    assert len(data) > 10 * len(blob)
    inb = C.create_string_buffer(blob, len(blob))
    alloc = len(data) // 20
    buf = C.create_string_buffer(alloc)
    s = O.ZStream()
    s.next_in = C.cast(inb, C.c_void_p)
    s.avail_in = len(blob)
    s.next_out = C.cast(buf, C.c_void_p)
    s.avail_out = alloc
    assert L.inflateInit2(C.byref(s), 15 + 32) == 0
    calls = {0: 0, 1: 0, -5: 0}
    got = b""
    for _ in range(200):
        ret = L.inflate(C.byref(s), 0)
        calls[ret] = calls.get(ret, 0) + 1
        if ret == 1:
            break
        if ret == -5:
            assert s.avail_out == 0
            produced = s.total_out
            got = buf.raw[:produced]  # (what realloc would carry over)
            alloc *= 2
            buf = C.create_string_buffer(got, alloc)
            s.next_out = C.cast(C.addressof(buf) + produced, C.c_void_p)
            s.avail_out = alloc - produced
            continue
        assert ret == 0 and s.avail_out == 0, ret
    assert ret == 1 and s.total_out == len(data) and buf.raw[:len(data)] == data
    assert s.total_in == len(blob) and s.avail_in == 0
    assert calls[0] >= 3 and calls[-5] >= 3
    assert L.inflateEnd(C.byref(s)) == 0


@pytest.mark.gpu
def test_real_brotli_lzma1_and_lzfse_streams_through_the_seam(gpu_lib):
    """brotliDecompress on a Brotli stream (RFC 7932), lzmaDecompress on an APPNOTE 5.8.8 LZMA1 stream with an
    end marker, lzfseDecompress on an LZFSE stream: such streams do not say how long their output is, the caller's buffer is the room
    (OZG_EF_SIZE_IS_CAP) and total_out says where the stream ended.  Checked against libbrotli / liblzma."""
    import lzma
    import struct
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from otezip_amd import corpus
    from tools import brotlilib as B
    L = gpu_lib.lib()
    datas = [b"", b"q", corpus.gen_text(4, 3000), corpus.gen_text(5, 70000), bytes(5000), b"xyz" * 3000]
    for d in datas:
        if B.available():
            for q in (1, 5, 11):
                comp = B.compress(d, quality=q)
                ret, tin, tout, out = gpu_codec(L, "brotli", True, comp, len(d) + 777)
                assert ret == 1 and tout == len(d) and out[:tout] == d, ("brotli", len(d), q, ret, tout)
        f = [{"id": lzma.FILTER_LZMA1, "preset": 6, "lc": 3, "lp": 0, "pb": 2, "dict_size": 1 << 20}]
        raw = lzma.compress(d, format=lzma.FORMAT_RAW, filters=f)  # (liblzma ends a raw stream with the end marker)
        comp = b"\x09\x14\x05\x00" + bytes([(2 * 5 + 0) * 9 + 3]) + struct.pack("<I", 1 << 20) + raw
        ret, tin, tout, out = gpu_codec(L, "lzma", True, comp, len(d) + 777)
        assert ret == 1 and tout == len(d) and out[:tout] == d, ("lzma", len(d), ret, tout)
        from tools import lzfse_model as M
        for kind in ("lzvn", "lzfse", "mixed"):   # (LZFSE: checked against the model of the format, parity unpinned)
            ret, tin, tout, out = gpu_codec(L, "lzfse", True, M.encode(d, kind), len(d) + 777)
            assert ret == 1 and tout == len(d) and out[:tout] == d, ("lzfse", kind, len(d), ret, tout)
    # a valid stream that needs more room than there is: Z_BUF_ERROR, nothing consumed or written, so that a caller
    # that grows its buffer and calls again (the reference's CLI: ref src/main.c:612-650) gets it then -- not the
    # Z_DATA_ERROR of the reference's container reader, which these bytes would get if they were handed on to it
    d = datas[3]
    big = []
    if B.available():
        big.append(("brotli", B.compress(d, quality=5)))
    raw = lzma.compress(d, format=lzma.FORMAT_RAW, filters=f)
    big.append(("lzma", b"\x09\x14\x05\x00" + bytes([(2 * 5 + 0) * 9 + 3]) + struct.pack("<I", 1 << 20) + raw))
    big.append(("lzfse", M.encode(d, "mixed")))
    for name, comp in big:
        ret, tin, tout, _ = gpu_codec(L, name, True, comp, 1000)
        assert ret == -5 and tin == 0 and tout == 0, (name, ret, tin, tout)
        ret, tin, tout, out = gpu_codec(L, name, True, comp, len(d))
        assert ret == 1 and out[:tout] == d, (name, ret, tout)
