"""Pins the oracle to the reference itself (oracle/_ref/libotezip_ref.so, compiled by
oracle/Makefile from the sources under /root/reference).  Skipped where that build is absent
(it is git-ignored and only exists in the build container / travels with gpurun)."""
import os
import random
import zlib

import pytest

import cases
from helpers import O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="reference build (oracle/_ref) not present")


def _cmp_codec(method, comp, cap):
    r, out = O.codec_decode(method, comp, cap)
    ret, tin, tout, rout = O.ref_codec_decode(method, comp, cap)
    if r.verdict & O.V_REF_UNDEFINED:
        return 0
    if r.verdict & O.V_REF_FALSE_REJECT:
        assert ret == O.Z_BUF_ERROR and r.correct_ret == O.Z_STREAM_END
        return 1
    assert (r.ret, r.total_in, r.total_out) == (ret, tin, tout), (method, len(comp), cap)
    assert out[:tout] == rout[:tout]
    return 0


def test_inflate_matches_reference_on_valid_and_damaged_streams():
    rnd = random.Random(99)
    f2 = 0
    for method, comp, usize, crc, flags in cases.deflate_valid(sizes=(0, 1, 6, 54, 1000, 20000)) + cases.deflate_tiny_f2():
        f2 += _cmp_codec(8, comp, usize)
        if usize:
            _cmp_codec(8, comp, usize - 1)
            _cmp_codec(8, comp, usize + 7)
            _cmp_codec(8, comp[:-1], usize)
            _cmp_codec(8, comp + b"\0\0", usize)
    assert f2 > 0
    for method, comp, usize, crc, flags in cases.deflate_fuzz(n=300):
        _cmp_codec(8, comp, usize)
    for _ in range(500):
        _cmp_codec(8, os.urandom(rnd.randint(1, 64)), 300)


def test_container_codecs_match_reference():
    for method, comp, usize, crc, flags in cases.containers():
        if method in (93, 100, 97, 14):
            _cmp_codec(method, comp, usize)
            _cmp_codec(method, comp, 0)


def test_extract_entry_matches_zip_fopen_index():
    """Through the reference's own zip_open/zip_fopen_index on one-entry archives."""
    items = cases.containers() + cases.deflate_edge() + cases.deflate_tiny_f2(n=30)
    for method, comp, usize, crc, flags in items[::2]:
        for strict in (0, 1):
            o = O.extract_entry(method, comp, usize, crc, strict_crc=strict, ignore_zipbomb=flags & 1)
            if o["verdict"] & O.V_REF_UNDEFINED:  # over-subscribed sets ARE compared: the oracle restates the first-match search
                continue
            r = O.ref_extract_entry(method, comp, usize, crc, verify_crc=strict, ignore_zipbomb=flags & 1)
            assert (o["rc"] == 0) == (r is not None), (method, len(comp), usize, strict)
            if r is not None:
                assert r == o["out"]


def test_private_writers_match_reference_zip_file_add():
    """otezip_compress_data for methods 93/100/97/14 (incl. the "not smaller -> STORE" rule and the
    chunk-space checks of the LZMA-token writer) against archives the reference itself writes."""
    import test_encode
    for m in (93, 100, 97, 14):
        for d in [b""] + test_encode._private_cases():
            assert O.compress_data_private(m, d) == O.ref_compress_data(m, d), (m, len(d))
