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
    assert emu_lib().emu_lds_bytes_copy() <= 16 * 1024
    assert emu_lib().emu_lds_bytes_index() <= 10 * 1024  # sixteen index wavefronts per CU


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


def test_deflate_beyond_the_lds_ring():
    items = cases.deflate_large()
    _run([items[0], items[2]])  # 64 KiB and 100 000 B of text: several rounds, two flushes


def test_deflate_overlapping_matches_across_the_ring_end():
    _run(cases.deflate_ring_wrap())


def test_containers():
    items = [c for c in cases.containers() if c[2] <= 70001]
    _run(items)
