"""The oracle against the committed golden fixtures (tests/golden/entries.json, produced by
oracle/gen_golden.py from outputs of the reference itself).  Runs anywhere: no GPU, no
/root/reference."""
import golden
from helpers import O


def test_fixture_file_is_small_and_well_formed():
    g = golden.load()
    assert g["generator"] == "oracle/gen_golden.py"
    assert len(g["entries"]) >= 60
    assert bytes.fromhex(g["empty_zip_hex"]) == b"PK\x05\x06" + b"\0" * 18  # ref test/unit/test_empty_zip.c:10-19


def test_oracle_reproduces_reference_verdicts_and_bytes():
    n_f2 = 0
    for method, comp, usize, crc, flags, fx in golden.load_items():
        o = O.extract_entry(method, comp, usize, crc, ignore_zipbomb=flags & 1)
        if fx["undefined"]:
            assert o["verdict"] & O.V_REF_UNDEFINED
            continue
        assert (o["rc"] == 0) == fx["ref_ok"], fx["name"]              # what zip_fopen_index returns in the reference
        assert (o["correct_rc"] == 0) == fx["correct_ok"], fx["name"]  # with F2 repaired
        if fx["correct_ok"]:
            assert golden.sha(o["out"]) == fx["sha256"], fx["name"]
        if fx["correct_ok"] and not fx["ref_ok"]:
            assert o["verdict"] & O.V_REF_FALSE_REJECT, fx["name"]
            n_f2 += 1
    assert n_f2 >= 6


def test_crc32_known_answers():
    assert O.crc32(b"") == 0
    assert O.crc32(b"123456789") == 0xCBF43926           # CRC-32/ISO-HDLC check value
    assert O.crc32(b"789", O.crc32(b"123456")) == 0xCBF43926  # chaining through the crc argument
