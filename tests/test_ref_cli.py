"""Drop-in boundary check with the reference's own caller: oracle/_ref/otezip_cli_amd is the reference's
CLI (src/main.c), compiled unchanged against include/otezip/*.h of THIS repo and linked with
libotezip_amd.so (oracle/Makefile; built in the build container, the binary travels to the GPU box).
If a struct layout, a constant, a prototype or an ownership rule of the API differed from the
reference's, this program would not build, or would not produce the right files."""
import os
import subprocess
import zipfile

import pytest

from helpers import ROOT

CLI = os.path.join(ROOT, "oracle", "_ref", "otezip_cli_amd")


def test_reference_cli_links_against_the_product_library():
    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/otezip_cli_amd not built (needs /root/reference: build container only)")
    out = subprocess.run(["ldd", CLI], capture_output=True, text=True).stdout
    assert "libotezip_amd.so" in out
    syms = subprocess.run(["nm", "-D", "--undefined-only", CLI], capture_output=True, text=True).stdout
    for name in ("zip_open", "zip_fopen_index", "zip_fclose", "zip_file_add", "zip_close", "inflate", "deflate"):
        assert name in syms, name


def _config1(path):
    texts = [b"hello\n", b"world\n", b"This is a test file for otezip compression checks.\n",
             b"test content test content test content test content !\n"]
    want = {}
    with zipfile.ZipFile(path, "w") as z:
        for i in range(100):
            data = texts[i % 4] * (1 + (i // 4) % 3)
            comp = zipfile.ZIP_DEFLATED if (i % 4) >= 2 else zipfile.ZIP_STORED
            z.writestr(zipfile.ZipInfo("f%03d.txt" % i), data, compress_type=comp)
            want["f%03d.txt" % i] = data
    return want


@pytest.mark.gpu
def test_reference_cli_lists_and_extracts_config0(tmp_path):
    """BASELINE configs[0]: the 100-entry STORE + DEFLATE archive through `mzip -l` and `mzip -x`."""
    if not os.path.exists(CLI):
        pytest.fail("oracle/_ref/otezip_cli_amd did not travel with the snapshot")
    arc = str(tmp_path / "c0.zip")
    want = _config1(arc)
    p = subprocess.run([CLI, "-l", arc], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    for name in want:
        assert name in p.stdout
    out = tmp_path / "x"
    out.mkdir()
    p = subprocess.run([CLI, "-x", arc], capture_output=True, text=True, cwd=str(out), timeout=300)
    assert p.returncode == 0, p.stderr
    for name, data in want.items():
        assert (out / name).read_bytes() == data, name


@pytest.mark.gpu
def test_reference_cli_creates_an_archive_other_readers_accept(tmp_path):
    """`mzip -c out.zip files -z deflate`: zip_file_add + default_method through the reference's CLI;
    python's zipfile must read the result (the reference's own encoder writes invalid streams)."""
    if not os.path.exists(CLI):
        pytest.fail("oracle/_ref/otezip_cli_amd did not travel with the snapshot")
    src = tmp_path / "in"
    src.mkdir()
    files = {"a.txt": b"alpha " * 2000, "b.bin": bytes(range(256)) * 40, "c.txt": b"short\n"}
    for n, d in files.items():
        (src / n).write_bytes(d)
    arc = str(tmp_path / "new.zip")
    p = subprocess.run([CLI, "-c", arc] + sorted(files) + ["-z", "deflate"], capture_output=True, text=True, cwd=str(src), timeout=300)
    assert p.returncode == 0, (p.stdout, p.stderr)
    with zipfile.ZipFile(arc) as z:
        assert z.testzip() is None
        for n, d in files.items():
            assert z.read(n) == d
        assert z.getinfo("a.txt").compress_type == zipfile.ZIP_DEFLATED


@pytest.mark.gpu
def test_reference_cli_gzip_and_gunzip(tmp_path):
    """`mzip -g file` and `mzip -d file.gz` (ref src/main.c:590-832).  With the reference's own library -g always
    fails (deflateInit2 refuses windowBits 31); here it writes a gzip member the system's gzip
    accepts, and -d reads what the system's gzip wrote -- including a member that expands far beyond the
    4 x its CLI first allocates, so that its grow-and-call-again loop (main.c:612-650) runs."""
    import shutil
    if not os.path.exists(CLI):
        pytest.fail("oracle/_ref/otezip_cli_amd did not travel with the snapshot")
    if not shutil.which("gzip"):
        pytest.skip("no system gzip")
    from otezip_amd import corpus
    text = corpus.gen_text(12, 300000)
    sparse = (corpus.gen_text(13, 3000) * 50) + bytes(400000)  # expands ~40 x
    for name, data in (("t.txt", text), ("s.bin", sparse), ("tiny", b"x")):
        src = tmp_path / name
        src.write_bytes(data)
        p = subprocess.run([CLI, "-g", str(src), str(src) + ".gz"], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, (p.stdout, p.stderr)
        assert subprocess.run(["gzip", "-t", str(src) + ".gz"], capture_output=True).returncode == 0
        assert subprocess.run(["gzip", "-dc", str(src) + ".gz"], capture_output=True).stdout == data
        # the other way round: the system's gzip writes, the reference's CLI on this library reads
        sysgz = tmp_path / (name + ".sys.gz")
        sysgz.write_bytes(subprocess.run(["gzip", "-c", str(src)], capture_output=True).stdout)
        out = tmp_path / (name + ".back")
        p = subprocess.run([CLI, "-d", str(sysgz), str(out)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, (p.stdout, p.stderr)
        assert out.read_bytes() == data
