"""The kernel BODIES (otezip_amd/csrc/*.h, the exact source hipcc compiles) executed on CPU
threads through tests/simt_emu, against the oracle.  This is how the parallel-decode logic is
exercised -- and run under sanitizers -- on machines without a GPU; the GPU parity tests
(test_gpu_parity.py) run the same cases through the real library."""
import pytest

import cases
from helpers import check_against_oracle, emu_decode, emu_lib


def _run(items):
    res = emu_decode(items)
    check_against_oracle(items, res)
    return res


def test_lds_budget_allows_two_workgroups_per_cu():
    assert emu_lib().emu_lds_bytes_inflate() <= 80 * 1024
    assert emu_lib().emu_lds_bytes_apply() <= 18 * 1280  # seven apply workgroups per CU (LDS comes in 1280-byte granules)
    assert emu_lib().emu_lds_bytes_copy() <= 16 * 1024
    assert emu_lib().emu_lds_bytes_index() <= 6 * 1280  # twenty index wavefronts per CU: LDS comes in 1280-byte granules


def test_workgroup_per_entry_routine_alone():
    """The self-contained decode of otz_inflate.h (what entries take whose index overflows its arena region)."""
    items = cases.deflate_valid(sizes=(0, 6, 1000, 20000)) + cases.deflate_tiny_f2(n=12) + cases.deflate_edge()[:30]
    check_against_oracle(items, emu_decode(items, mode=1))


def test_index_overflow_takes_the_fallback():
    """No arena region at all, and regions far too small: the index kernel must say so and the
    entries must still decode (through the overflow list)."""
    items = cases.deflate_valid(sizes=(6, 1000, 20000))[:20] + [cases.deflate_large()[0]]
    check_against_oracle(items, emu_decode(items, mode=2))
    check_against_oracle(items, emu_decode(items, region_pct=12))
    check_against_oracle(items, emu_decode(items, region_pct=20))


def test_deflate_valid_streams():
    _run(cases.deflate_valid(sizes=(0, 1, 6, 54, 1000, 4096)))


def test_deflate_reference_false_rejects_are_repaired_and_flagged():
    res = _run(cases.deflate_tiny_f2(n=40))
    assert any(r.status == 1 and r.flags & 1 for r, _, _ in res)


def test_deflate_edge_cases():
    _run([c for c in cases.deflate_edge() if c[2] < 100000])


def test_deflate_fuzz_never_hangs_or_overruns():
    _run(cases.deflate_fuzz(n=60))


def test_deflate_entries_of_many_rounds():
    items = cases.deflate_large()
    _run([items[0], items[2]])  # 64 KiB and 100 000 B of text: several rounds, the window slides


def test_deflate_overlapping_matches_at_buffer_boundaries():
    _run(cases.deflate_ring_wrap())


def test_deflate_matches_reaching_behind_the_lds_window():
    _run(cases.deflate_far_matches(n=3))


def test_containers():
    items = [c for c in cases.containers() if c[2] <= 70001]
    _run(items)


def test_stream_end_position_and_full_buffer_semantics():
    """What the z_stream seam needs from the kernels: the exact end of the stream (the reference's
    total_in is the number of bytes its bit reader had loaded: ref deflate-dec.inc.c:44-62), and -- when
    the stream produces more than the buffer holds -- the buffer filled exactly as the reference leaves it
    when it returns Z_OK (:521-545, :693-760), with total_in behind the token that did not fit."""
    import ctypes as C
    import random
    from helpers import O, layout, OzgResult, emu_lib, ST_OK, ST_NEED_OUT
    r = random.Random(11)
    items = []
    for d, kw in [(cases.word_text(r, 3000), {}), (cases.word_text(r, 20000), {}), (cases.small_text(r, 500), dict(strategy=2)),
                  (b"Q" * 5000, {}), (cases.word_text(r, 4000), dict(level=0)), (cases.word_text(r, 9000), dict(flush_every=700))]:
        c = cases.raw_deflate(d, **kw)
        for cap in (len(d), len(d) - 1, len(d) // 2, 7, len(d) + 20, max(len(d) - 258, 1)):
            items.append((8, c + b"trailing", cap, 0, 0))
    src, ents, dst_len = layout(items)
    dst = C.create_string_buffer(b"\xCD" * dst_len, dst_len)
    res = (OzgResult * len(items))()
    ends = (C.c_uint64 * len(items))()
    L = emu_lib()
    L.emu_set_end_bits(ends)
    try:
        L.emu_decode_batch_mode(src, len(src), dst, dst_len, ents, res, len(items), 1000, 1 << 20, 0, 100)
    finally:
        L.emu_set_end_bits(None)
    n_ok = n_full = 0
    for i, (m, comp, cap, crc, fl) in enumerate(items):
        o, ref_out = O.codec_decode(8, comp, cap)
        if o.verdict & O.V_NO_PARITY:
            continue
        out = dst.raw[ents[i].dst_off:ents[i].dst_off + cap]
        if o.correct_ret == 1:
            assert res[i].status == ST_OK and res[i].total_out == o.total_out and out[:o.total_out] == ref_out[:o.total_out]
            assert (ends[i] + 7) // 8 == o.total_in, (i, ends[i], o.total_in)
            n_ok += 1
        elif o.correct_ret == 0:
            assert res[i].status == ST_NEED_OUT and res[i].total_out == cap == o.total_out, (i, res[i].status, res[i].total_out)
            assert out == ref_out[:cap]
            assert (ends[i] + 7) // 8 == o.total_in, (i, ends[i], o.total_in)
            n_full += 1
    assert n_ok >= 10 and n_full >= 15
