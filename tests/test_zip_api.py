"""The drop-in boundary: the reference's libzip-subset C API served by libotezip_amd.so.
A plain C program (tests/c/zip_roundtrip.c) is compiled against include/otezip/zip.h and linked
with the library, exactly as a caller of the reference would be."""
import os
import subprocess
import zipfile
import zlib

import pytest

import golden
from helpers import O, ROOT


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    import otezip_amd as oz
    if not os.path.exists(oz.lib_path()):
        oz.build_native()
    out = str(tmp_path_factory.mktemp("bin") / "zip_roundtrip")
    libdir = os.path.dirname(oz.lib_path())
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "zip_roundtrip.c"), "-o", out,
                           "-L", libdir, "-lotezip_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def run(prog, *args):
    p = subprocess.run([prog] + list(args), capture_output=True, text=True, timeout=600)
    return p.returncode, p.stdout, p.stderr


def make_config1_archive(path):
    """BASELINE configs[0] shape: 100 entries, four tiny texts x 25 names, STORE + DEFLATE mix
    (python's zipfile writes method 8 where it helps; the reference's own encoder cannot, F3)."""
    texts = [b"hello\n", b"world\n", b"This is a test file for otezip compression checks.\n",
             b"test content test content test content test content !\n"]
    want = []
    with zipfile.ZipFile(path, "w") as z:
        for i in range(100):
            data = texts[i % 4] * (1 + (i // 4) % 3)
            comp = zipfile.ZIP_DEFLATED if (i % 4) >= 2 else zipfile.ZIP_STORED
            z.writestr(zipfile.ZipInfo("dir/f%03d.txt" % i), data, compress_type=comp)
            want.append(data)
    return want


def test_empty_archive_opens_with_zero_entries(prog, tmp_path):
    p = tmp_path / "empty.zip"
    p.write_bytes(bytes.fromhex(golden.load()["empty_zip_hex"]))  # ref test/unit/test_empty_zip.c:10-19
    rc, out, _ = run(prog, "list", str(p))
    assert rc == 0 and out.split() == ["0"]


def test_directory_listing_matches_the_archive(prog, tmp_path):
    p = str(tmp_path / "c1.zip")
    want = make_config1_archive(p)
    rc, out, _ = run(prog, "list", p)
    assert rc == 0
    lines = out.strip().split("\n")
    assert lines[0] == "100"
    with zipfile.ZipFile(p) as z:
        for line, info, data in zip(lines[1:], z.infolist(), want):
            name, method, csize, usize, crc = line.split()
            assert name == info.filename and int(method) == info.compress_type
            assert int(csize) == info.compress_size and int(usize) == len(data)
            assert int(crc, 16) == zlib.crc32(data) & 0xFFFFFFFF


def test_not_a_zip_is_rejected(prog, tmp_path):
    p = tmp_path / "junk.bin"
    p.write_bytes(b"this is not an archive" * 10)
    rc, out, _ = run(prog, "list", str(p))
    assert rc == 1 and out.startswith("open-failed 21")  # ZIP_ER_INCONS, as the reference


def test_without_gpu_extract_fails_loudly(prog, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = str(tmp_path / "c1.zip")
    make_config1_archive(p)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    rc, out, err = run(prog, "extract", p, str(out_dir))
    assert all(line.endswith("NULL") for line in out.strip().split("\n"))
    assert "no CPU fallback" in err or "no HIP device" in err


@pytest.mark.gpu
def test_config1_extract_is_bit_exact(prog, tmp_path):
    p = str(tmp_path / "c1.zip")
    want = make_config1_archive(p)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    rc, out, err = run(prog, "extract", p, str(out_dir))
    assert rc == 0, err
    for i, data in enumerate(want):
        assert (out_dir / str(i)).read_bytes() == data
        # and the same as the oracle's view of the entry
    assert "NULL" not in out


@pytest.mark.gpu
def test_crafted_archive_of_every_method(prog, tmp_path):
    items = [it for it in golden.load_items() if not it[5]["undefined"]]
    arc = O.craft_zip([("e%03d" % i, it[0], it[1], it[2], it[3]) for i, it in enumerate(items)])
    p = tmp_path / "all.zip"
    p.write_bytes(arc)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    rc, out, err = run(prog, "extract", str(p), str(out_dir))
    assert rc == 0, err
    lines = out.strip().split("\n")
    for i, (it, line) in enumerate(zip(items, lines)):
        fx = it[5]
        if fx["flags"] & 1:
            continue  # otezip_ignore_zipbomb is a process-wide switch the C driver does not set
        assert line.endswith("NULL") == (not fx["correct_ok"]), fx["name"]
        if fx["name"] == "deflate_zipbomb_guard":
            assert "zipbomb" in err  # the reference's message, ref otezip.c:459
        if fx["correct_ok"]:
            assert golden.sha((out_dir / str(i)).read_bytes()) == fx["sha256"], fx["name"]
    assert "CRC mismatch for" in err  # the reference's warning text for deflate_crc_mismatch


@pytest.mark.gpu
def test_add_then_set_compression_then_read_back(prog, tmp_path):
    """ref test/unit/test_set_file_compression.c -- fails on the reference, passes here."""
    p = str(tmp_path / "w.zip")
    rc, out, err = run(prog, "setcomp", p)
    assert rc == 0, (rc, err)
    with zipfile.ZipFile(p) as z:  # and an independent reader accepts the archive
        assert z.read("hello.txt") == b"A" * 4096 and z.read("empty") == b""
        assert z.getinfo("hello.txt").compress_type == zipfile.ZIP_DEFLATED   # written by the GPU encoder
        assert z.getinfo("hello.txt").compress_size < 100
