"""The drop-in boundary: the reference's libzip-subset C API served by libotezip_amd.so.
A plain C program (tests/c/zip_roundtrip.c) is compiled against include/otezip/zip.h and linked
with the library, exactly as a caller of the reference would be."""
import os
import subprocess
import sys
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
                           "-L", libdir, "-lotezip_amd", "-lpthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def run(prog, *args, env=None):
    p = subprocess.run([prog] + list(args), capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env) if env else None)
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


@pytest.mark.gpu
def test_global_switches_through_the_c_api(prog, tmp_path):
    """otezip_verify_crc = 1 turns the CRC warning into a failure (ref otezip.c:670-679);
    otezip_ignore_zipbomb = 1 lets the guarded entry through to its codec (ref :454)."""
    items = [it for it in golden.load_items() if not it[5]["undefined"]]
    arc = O.craft_zip([("e%03d" % i, it[0], it[1], it[2], it[3]) for i, it in enumerate(items)])
    p = tmp_path / "all.zip"
    p.write_bytes(arc)
    names = [it[5]["name"] for it in items]
    i_crc, i_bomb = names.index("deflate_crc_mismatch"), names.index("deflate_zipbomb_guard")
    res = {}
    for verify, ignore in ((0, 0), (1, 0), (0, 1)):
        out_dir = tmp_path / ("o%d%d" % (verify, ignore))
        out_dir.mkdir()
        rc, out, err = run(prog, "extract2", str(p), str(out_dir), str(verify), str(ignore))
        assert rc == 0, err
        res[(verify, ignore)] = (out.strip().split("\n"), err)
    assert not res[(0, 0)][0][i_crc].endswith("NULL") and "CRC mismatch for" in res[(0, 0)][1]
    assert res[(1, 0)][0][i_crc].endswith("NULL")
    assert res[(0, 0)][0][i_bomb].endswith("NULL") and "zipbomb" in res[(0, 0)][1]
    # with the guard off the entry reaches inflate, which fails on its truncated stream -- but not as a bomb
    assert "zipbomb" not in res[(0, 1)][1]
    for k in range(len(items)):
        if k not in (i_crc, i_bomb) and not (items[k][5]["flags"] & 1):
            assert res[(0, 0)][0][k] == res[(1, 0)][0][k] == res[(0, 1)][0][k], names[k]


@pytest.mark.gpu
def test_ref_strict_mimics_the_reference_false_reject(prog, tmp_path):
    """OTEZIP_REF_STRICT=1: streams the reference's inflate rejects by mistake are refused too."""
    import cases
    f2 = [c for c in cases.deflate_tiny_f2(n=60)
          if O.extract_entry(*c[:4])["verdict"] & O.V_REF_FALSE_REJECT][:5]
    assert f2
    arc = O.craft_zip([("f%d" % i, c[0], c[1], c[2], c[3]) for i, c in enumerate(f2)])
    p = tmp_path / "f2.zip"
    p.write_bytes(arc)
    for strict in (False, True):
        out_dir = tmp_path / ("s%d" % strict)
        out_dir.mkdir()
        env = dict(os.environ, OTEZIP_REF_STRICT="1") if strict else os.environ
        q = subprocess.run([prog, "extract", str(p), str(out_dir)], capture_output=True, text=True, env=env, timeout=600)
        lines = q.stdout.strip().split("\n")
        assert all(line.endswith("NULL") == strict for line in lines), (strict, lines)


@pytest.mark.gpu
def test_replace_append_and_reverse_iteration(prog, tmp_path):
    p = str(tmp_path / "r.zip")
    rc, out, err = run(prog, "replace", p)
    assert rc == 0, (rc, err)
    with zipfile.ZipFile(p) as z:
        assert z.read("a") == b"a" * 3000 and z.read("b") == b"B" * 7000 and z.read("c") == b"c" * 100
    # a larger archive, last entry first: every window is a miss, every byte must still be right
    p2 = str(tmp_path / "c1.zip")
    want = make_config1_archive(p2)
    out_dir = tmp_path / "rev"
    out_dir.mkdir()
    rc, out, err = run(prog, "extract2", p2, str(out_dir), "0", "0", "rev")
    assert rc == 0 and "NULL" not in out, err
    for i, data in enumerate(want):
        assert (out_dir / str(i)).read_bytes() == data


@pytest.mark.gpu
def test_bulk_extract_through_the_api_full_count(prog, tmp_path):
    """BASELINE configs[1] at its full count through the drop-in API: 10 000 x 64 KiB DEFLATE entries read
    from disk with zip_open / zip_fopen_index / zip_fclose (pipelined windows, pinned staging).
    Every 97th entry's bytes are checked through their CRC; the rate goes to gpurun_out/."""
    from otezip_amd import corpus
    arc, ents = corpus.build_archive(10000, 65536)
    p = tmp_path / "big.zip"
    p.write_bytes(arc)
    rc, out, err = run(prog, "bench-extract", str(p), "97", "3")
    assert rc == 0, err
    lines = out.strip().split("\n")
    totals = [l.split() for l in lines if l.startswith("TOTAL")]
    assert len(totals) == 3
    for total in totals:
        assert int(total[1]) == 10000 * 65536 and int(total[3]) == 0
    n_checked = 0
    for line in lines:
        if line.startswith("TOTAL"):
            continue
        i, crc, size = line.split()
        assert int(crc, 16) == ents[int(i)]["crc32"] and int(size) == 65536
        n_checked += 1
    assert n_checked == 3 * len(range(0, 10000, 97))
    rec = os.path.join(ROOT, "gpurun_out")
    os.makedirs(rec, exist_ok=True)
    with open(os.path.join(rec, "api_extract_rate.txt"), "a") as f:
        for k, total in enumerate(totals):
            f.write("bench-extract 10000x65536 pass %d (%s): %.3f s, %.2f GiB/s out (zip_open..zip_close, archive in "
                    "the page cache)\n" % (k, "cold: creates the GPU context, pins the staging buffers" if k == 0 else "warm",
                                           float(total[2]), int(total[1]) / float(total[2]) / 2**30))


@pytest.mark.gpu
def test_bulk_add_with_set_file_compression_full_count(prog, tmp_path):
    """BASELINE configs[4]: 10 000 x 64 KiB buffers through zip_file_add with default method STORE, DEFLATE
    selected afterwards with zip_set_file_compression, one zip_close.  Python's zipfile and the GPU reader
    must both return the inputs; every entry must be method 8."""
    from otezip_amd import corpus
    n, size = 10000, 65536
    bufs = corpus.text_buffers(n, size)
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        for b in bufs:
            f.write(b)
    p = str(tmp_path / "w.zip")
    rc, out, err = run(prog, "bench-add", p, str(inp), str(n), str(size), "3")
    assert rc == 0, (rc, err)
    passes = [float(l.split()[2]) for l in out.strip().split("\n") if l.startswith("TOTAL")]
    assert len(passes) == 3
    secs = passes[0]
    comp_total = 0
    with zipfile.ZipFile(p) as z:
        infos = z.infolist()
        assert len(infos) == n
        for i in range(0, n, 53):
            assert infos[i].compress_type == zipfile.ZIP_DEFLATED
            assert z.read(infos[i]) == bufs[i], i
        comp_total = sum(i.compress_size for i in infos)
        assert all(i.compress_type == zipfile.ZIP_DEFLATED for i in infos)
    rc, out2, err = run(prog, "bench-extract", p, "101")
    assert rc == 0, err
    for line in [l for l in out2.strip().split("\n") if not l.startswith("TOTAL")]:
        i, crc, sz = line.split()
        assert int(crc, 16) == zlib.crc32(bufs[int(i)]) & 0xFFFFFFFF
    with open(os.path.join(ROOT, "gpurun_out", "api_add_rate.txt"), "a") as f:
        for k, t in enumerate(passes):
            f.write("bench-add %dx%d pass %d (%s): %.3f s, %.2f GiB/s in, ratio %.3f (zip_open..zip_close incl. file write)\n" %
                    (n, size, k, "cold: creates the GPU context, pins the staging buffers" if k == 0 else "warm", t,
                     n * size / t / 2**30, n * size / comp_total))


@pytest.mark.gpu
@pytest.mark.parametrize("stat_every", [0, 700])
def test_ragged_adds_cross_encode_batches(prog, tmp_path, stat_every):
    """The writer encodes its pending sources in batches of 32 MiB / 16 384 entries: 40 000 tiny entries,
    an entry larger than a batch, entries that end exactly on a batch boundary, and (second run) stats in
    between that force partial flushes.  Python's zipfile must read everything back."""
    import random
    from otezip_amd import corpus
    r = random.Random(4)
    sizes = [r.choice([0, 1, 5, 60, 700]) for _ in range(40000)] + [40 << 20, 3, (32 << 20) - 64, 64, 1 << 20, 0, 65536]
    blob = corpus.gen_text(77, 4 << 20)
    inp = tmp_path / "in.bin"
    off, bufs = 0, []
    with open(inp, "wb") as f:
        for sz in sizes:
            b = (blob[off % 1000:] + blob * (sz // len(blob) + 1))[:sz]
            off += 37
            bufs.append(b)
            f.write(b)
    sz_file = tmp_path / "sizes.txt"
    sz_file.write_text("\n".join(map(str, sizes)) + "\n")
    p = str(tmp_path / "r.zip")
    rc, out, err = run(prog, "add-list", p, str(inp), str(sz_file), str(stat_every))
    assert rc == 0, (rc, err)
    with zipfile.ZipFile(p) as z:
        infos = z.infolist()
        assert len(infos) == len(sizes)
        for i in list(range(0, 40000, 997)) + list(range(40000, len(sizes))):
            assert infos[i].file_size == sizes[i]
            assert z.read(infos[i]) == bufs[i], i
        assert infos[40000].compress_type == zipfile.ZIP_DEFLATED and infos[40000].compress_size < sizes[40000] // 2


@pytest.mark.gpu
def test_config3_full_count_mixed_archive_through_zip_open(prog, tmp_path):
    """BASELINE configs[3]: a 50 000-entry archive alternating the private lzfse (100) and BROT (97) framings,
    64 KiB payloads (3.3 GB), read through zip_open / zip_fopen_index: ~50 pipelined windows.  Every 211th
    entry is checked through its CRC; per-codec rates are recorded."""
    from otezip_amd import corpus
    arc, ents = corpus.build_archive(50000, 65536, methods=(100, 97))
    p = tmp_path / "mixed.zip"
    p.write_bytes(arc)
    del arc
    rc, out, err = run(prog, "bench-extract", str(p), "211", "2")
    assert rc == 0, err
    lines = out.strip().split("\n")
    totals = [l.split() for l in lines if l.startswith("TOTAL")]
    assert len(totals) == 2 and all(int(t[1]) == 50000 * 65536 and int(t[3]) == 0 for t in totals)
    for line in lines:
        if not line.startswith("TOTAL"):
            i, crc, size = line.split()
            assert int(crc, 16) == ents[int(i)]["crc32"] and int(size) == 65536
    with open(os.path.join(ROOT, "gpurun_out", "api_extract_rate.txt"), "a") as f:
        for k, t in enumerate(totals):
            f.write("bench-extract 50000x65536 methods 100+97, pass %d: %.3f s, %.2f GiB/s out\n" %
                    (k, float(t[2]), int(t[1]) / float(t[2]) / 2**30))
    # per-codec rates (what configs[3] asks for): the same payloads, one codec per archive
    os.unlink(p)
    for method, name in ((100, "lzfse framing (100)"), (97, "BROT framing (97)")):
        arc, ents = corpus.build_archive(25000, 65536, methods=(method,))
        p.write_bytes(arc)
        del arc
        rc, out, err = run(prog, "bench-extract", str(p), "499", "2")
        assert rc == 0, err
        totals = [l.split() for l in out.strip().split("\n") if l.startswith("TOTAL")]
        assert len(totals) == 2 and all(int(t[1]) == 25000 * 65536 and int(t[3]) == 0 for t in totals)
        with open(os.path.join(ROOT, "gpurun_out", "api_extract_rate.txt"), "a") as f:
            f.write("bench-extract 25000x65536 %s, warm pass: %.3f s, %.2f GiB/s out\n" %
                    (name, float(totals[1][2]), int(totals[1][1]) / float(totals[1][2]) / 2**30))
        os.unlink(p)


@pytest.mark.gpu
def test_entry_larger_than_a_window(prog, tmp_path):
    """One STORE entry of 60 MB (beyond the 48 MB a window normally reads) between small ones: it gets a
    window of its own (the reference accepts entries up to 2 GiB, ref otezip.c:102)."""
    import random
    r = random.Random(9)
    big = bytes(r.getrandbits(8) for _ in range(1 << 16)) * 960   # 60 MiB
    p = str(tmp_path / "big.zip")
    with zipfile.ZipFile(p, "w") as z:
        z.writestr("small0", b"x" * 1000, compress_type=zipfile.ZIP_DEFLATED)
        z.writestr("big", big, compress_type=zipfile.ZIP_STORED)
        z.writestr("small1", b"y" * 3000, compress_type=zipfile.ZIP_DEFLATED)
    rc, out, err = run(prog, "bench-extract", p, "1")
    assert rc == 0, err
    lines = out.strip().split("\n")
    want = [b"x" * 1000, big, b"y" * 3000]
    for line, data in zip(lines, want):
        i, crc, size = line.split()
        assert int(size) == len(data) and int(crc, 16) == zlib.crc32(data) & 0xFFFFFFFF


# ---------------------------------------------------------------- several contexts, several archives

TWO_CTX = {"OTEZIP_GPU_DEVICE_MAP": "0,0", "OTEZIP_GPU_DEVICES": "2"}


@pytest.mark.gpu
def test_two_device_contexts_give_the_results_of_one(prog, tmp_path):
    """The library's multi-GPU path (OTEZIP_GPU_DEVICES: the windows of an archive and the encode batches of
    a flush are dealt to the devices in turn, one read-ahead window / encode flight per device) on a box
    with ONE GPU: OTEZIP_GPU_DEVICE_MAP=0,0 makes logical devices 0 and 1 two independent contexts
    (arena, job slots, streams, staging) on the same card.  Reading and writing must give exactly what
    one device gives."""
    from otezip_amd import corpus
    n, size = 6000, 65536
    arc, ents = corpus.build_archive(n, size)
    p = tmp_path / "r.zip"
    p.write_bytes(arc)
    rc1, out1, err1 = run(prog, "bench-extract", str(p), "1")
    rc2, out2, err2 = run(prog, "bench-extract", str(p), "1", env=TWO_CTX)
    assert rc1 == 0 and rc2 == 0, (err1, err2)
    l1 = [l for l in out1.strip().split("\n") if not l.startswith("TOTAL")]
    l2 = [l for l in out2.strip().split("\n") if not l.startswith("TOTAL")]
    assert len(l1) == n and l1 == l2
    for line in l1[::37]:
        i, crc, sz = line.split()
        assert int(crc, 16) == ents[int(i)]["crc32"] and int(sz) == size
    # that both contexts really took windows: the trace names the device of every submit (OTEZIP_TRACE)
    rc3, _, err3 = run(prog, "bench-extract", str(p), "0", env=dict(TWO_CTX, OTEZIP_TRACE="1"))
    assert rc3 == 0
    assert "devices ready: 2 from 0" in err3
    # writing: 3 000 buffers -> several encode batches, dealt to the two contexts
    bufs = corpus.text_buffers(3000, size)
    inp = tmp_path / "in.bin"
    inp.write_bytes(b"".join(bufs))
    pa, pb = str(tmp_path / "w1.zip"), str(tmp_path / "w2.zip")
    rc, _, err = run(prog, "bench-add", pa, str(inp), "3000", str(size))
    assert rc == 0, err
    rc, _, err = run(prog, "bench-add", pb, str(inp), "3000", str(size), env=TWO_CTX)
    assert rc == 0, err
    with zipfile.ZipFile(pa) as za, zipfile.ZipFile(pb) as zb:
        ia, ib = za.infolist(), zb.infolist()
        assert len(ia) == len(ib) == 3000
        for a, b in zip(ia, ib):
            assert (a.filename, a.CRC, a.compress_size, a.file_size, a.compress_type) == \
                   (b.filename, b.CRC, b.compress_size, b.file_size, b.compress_type)
        for k in range(0, 3000, 41):
            assert zb.read(ib[k]) == bufs[k]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [0, 1])
def test_many_archives_walked_at_once(prog, tmp_path, threads):
    """More handles walking their windows than a device has staged-batch slots: a full slot table is
    back-pressure (the read-ahead is skipped, the window somebody waits for takes the synchronous call),
    never a failed zip_fopen_index.  Interleaved on one thread, and one thread per handle."""
    from otezip_amd import corpus
    n, size = 1500, 65536
    arc, _ = corpus.build_archive(n, size)
    p = tmp_path / "m.zip"
    p.write_bytes(arc)
    k = 12
    rc, out, err = run(prog, "multi", str(p), str(k), str(threads))
    assert rc == 0, (out, err[-2000:])
    total, bad = out.split()[1:3]
    assert int(total) == k * n * size and int(bad) == 0
    assert "GPU decode failed" not in err


@pytest.mark.gpu
def test_failed_flush_leaves_only_what_was_written(prog, tmp_path):
    """A deferred add whose batch cannot be encoded (here: no usable device) must not stay in the
    directory: zip_close returns -1 and the archive lists exactly the entries it holds (none), as with the
    reference, which fails inside zip_file_add and never records such an entry."""
    p = str(tmp_path / "f.zip")
    rc, out, err = run(prog, "add-fail", p, "5", env={"OTEZIP_GPU_DEVICE": "99"})
    assert rc == 0, (rc, out, err)
    assert out.split()[1] == "-1"
    with zipfile.ZipFile(p) as z:
        assert z.namelist() == []
    rc, out, err = run(prog, "list", p)
    assert rc == 0 and out.split()[0] == "0"


@pytest.mark.gpu
def test_every_method_and_both_kinds_of_payload_in_one_archive(prog, tmp_path):
    """One archive whose entries mix STORE, DEFLATE and -- under each of the method ids 93, 14, 97, 100 -- BOTH what
    the reference writes there (its private framings) and the real format other writers put there (Zstandard frames,
    LZMA1 per APPNOTE, Brotli streams, LZFSE streams), walked entry by entry through zip_fopen_index: every entry
    comes out with the CRC-32 of its content, whichever decoder the sniffing picked."""
    import lzma
    import struct
    sys.path.insert(0, ROOT)
    from otezip_amd import corpus
    from oracle import pyoracle as O
    from tools import brotlilib as B, lzfse_model as LF, zstdlib as Z
    made, want = [], []

    def add(method, payload, data):
        made.append((len(made), method, payload, len(data), zlib.crc32(data) & 0xFFFFFFFF))
        want.append(data)
    for i in range(240):
        d = corpus.gen_text(1000 + i, 2000 + 37 * i)
        k = i % 12
        if k == 0:
            add(0, d, d)
        elif k == 1:
            add(8, zlib.compress(d, 6)[2:-4], d)
        elif k == 2:
            add(93, O.frame_zstd(d), d)
        elif k == 3:
            add(93, Z.compress(d, level=3) if Z.available() else O.frame_zstd(d), d)
        elif k == 4:
            add(14, O.frame_lzma_tokens(d), d)
        elif k == 5:
            f = [{"id": lzma.FILTER_LZMA1, "preset": 1, "lc": 3, "lp": 0, "pb": 2, "dict_size": 1 << 16}]
            raw = lzma.compress(d, format=lzma.FORMAT_RAW, filters=f)
            add(14, b"\x09\x14\x05\x00" + bytes([(2 * 5 + 0) * 9 + 3]) + struct.pack("<I", 1 << 16) + raw, d)
        elif k == 6:
            add(97, O.frame_brotli(d), d)
        elif k == 7:
            add(97, B.compress(d, quality=5) if B.available() else O.frame_brotli(d), d)
        elif k == 8:
            add(100, O.frame_lzfse(d), d)
        elif k == 9:
            add(100, LF.encode(d, "lzfse"), d)
        elif k == 10:
            add(100, LF.encode(d, "lzvn"), d)
        else:
            add(8, zlib.compress(d, 1)[2:-4], d)
    arc, _ = corpus._serialise(made)
    p = tmp_path / "mixed.zip"
    p.write_bytes(arc)
    rc, out, err = run(prog, "bench-extract", str(p), "1")
    assert rc == 0, err[-400:]
    lines = [l.split() for l in out.strip().split("\n") if not l.startswith("TOTAL")]
    assert len(lines) == len(want)
    for (i, crc, size), d in zip(lines, want):
        assert int(size) == len(d) and int(crc, 16) == zlib.crc32(d) & 0xFFFFFFFF, (i, made[int(i)][1])
