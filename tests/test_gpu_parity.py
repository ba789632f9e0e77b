"""Parity tests proper: the HIP path, called through the C ABI (ozg_decode_batch_host /
ozg_decode_batch_device), against the CPU oracle on the same seeded inputs, against the
committed golden fixtures, and -- at BASELINE.json's full size -- through size-independent
properties.  Bar: bit-exact output bytes, identical accept/reject (with the reference's F2
false-reject repaired and flagged, SURVEY parity rule 2), identical CRC-32."""
import os
import sys
import zlib

import pytest

import cases
from helpers import O, F_CRC_MISMATCH, F_REF_FALSE_REJECT, F_SHORT_STREAM, ST_OK

pytestmark = pytest.mark.gpu


def run_gpu(oz, items):
    src = bytearray()
    tab = []
    for i, (method, comp, usize, crc, flags) in enumerate(items):
        src += b"\xEE" * ((7 * i + 3) % 13)  # payloads in an archive are not aligned
        tab.append((method, len(src), len(comp), usize, crc, flags))
        src += comp
    src += b"\xEE" * 32
    return oz.decode_batch_host(bytes(src), tab)


def assert_parity(items, b, what):
    n_ok = n_f2 = 0
    for i, (method, comp, usize, crc, flags) in enumerate(items):
        o = O.extract_entry(method, comp, usize, crc, ignore_zipbomb=flags & 1)
        tag = "%s[%d] method=%d comp=%d usize=%d" % (what, i, method, len(comp), usize)
        if o["verdict"] & O.V_NO_PARITY:
            continue  # reference reads indeterminate memory here: no parity claimed (rule 4)
        r = b.results[i]
        assert (r.status == ST_OK) == (o["correct_rc"] == 0), "%s: status %d, oracle rc %d/%d" % (
            tag, r.status, o["rc"], o["correct_rc"])
        if r.status == ST_OK:
            n_ok += 1
            assert b.data(i) == o["out"], tag + ": bytes differ"
            assert r.crc32 == o["crc"], tag + ": crc differs"
            assert bool(r.flags & F_CRC_MISMATCH) == bool(o["verdict"] & O.V_CRC_MISMATCH), tag
            assert bool(r.flags & F_REF_FALSE_REJECT) == bool(o["verdict"] & O.V_REF_FALSE_REJECT), tag
            assert bool(r.flags & F_SHORT_STREAM) == bool(o["verdict"] & O.V_SHORT_STREAM), tag
            n_f2 += bool(r.flags & F_REF_FALSE_REJECT)
    return n_ok, n_f2


@pytest.mark.parametrize("name", ["deflate_valid", "deflate_tiny_f2", "deflate_edge", "deflate_fuzz", "deflate_large", "deflate_ring_wrap", "deflate_far_matches",
                                  "containers"])
def test_parity_vs_oracle(gpu_lib, name):
    items = getattr(cases, name)()
    b = run_gpu(gpu_lib, items)
    n_ok, n_f2 = assert_parity(items, b, name)
    assert n_ok > 0
    if name == "deflate_tiny_f2":
        assert n_f2 > 0, "corpus should contain streams the reference false-rejects"


def test_golden_fixtures(gpu_lib):
    import golden
    items = golden.load_items()
    b = run_gpu(gpu_lib, [it[:5] for it in items])
    for i, it in enumerate(items):
        exp = it[5]
        r = b.results[i]
        if exp["undefined"]:
            continue
        assert (r.status == ST_OK) == exp["correct_ok"], (exp["name"], r.status)
        if r.status == ST_OK:
            assert golden.sha(b.data(i)) == exp["sha256"], exp["name"]
            assert bool(r.flags & F_REF_FALSE_REJECT) == (exp["ref_ok"] is False), exp["name"]


def test_full_size_properties(gpu_lib):
    """BASELINE configs[1] shape at a size the GPU does in milliseconds: 2 000 x 64 KiB entries.
    Every entry's CRC-32 (computed on the GPU from the LDS-resident output) must equal the
    directory CRC computed by the host's zlib; a sample is compared byte for byte with an
    independent inflate; decoding twice is idempotent."""
    from otezip_amd import corpus
    arc, ents = corpus.build_archive(2000, 65536)
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents]
    b1 = gpu_lib.decode_batch_host(arc, tab)
    b2 = gpu_lib.decode_batch_host(arc, tab)
    for i, e in enumerate(ents):
        r = b1.results[i]
        assert r.status == ST_OK and r.flags & F_CRC_MISMATCH == 0 and r.crc32 == e["crc32"], i
        assert b2.results[i].crc32 == r.crc32
    for i in range(0, len(ents), 97):
        e = ents[i]
        assert b1.data(i) == zlib.decompress(arc[e["data_off"]:e["data_off"] + e["comp_size"]], -15)
    assert b1.dst == b2.dst


def test_mixed_methods_archive(gpu_lib):
    """configs[3]/[4] shape, small: alternating private-zstd / lzfse / brotli containers + STORE."""
    from otezip_amd import corpus
    arc, ents = corpus.build_archive(400, 70001, methods=(93, 100, 97, 0, 8))
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents]
    b = gpu_lib.decode_batch_host(arc, tab)
    for i, e in enumerate(ents):
        assert b.results[i].status == ST_OK and b.results[i].crc32 == e["crc32"], (i, e["method"], b.results[i].status)
        assert zlib.crc32(b.data(i)) & 0xFFFFFFFF == e["crc32"]


def test_device_resident_api(gpu_lib):
    """ozg_decode_batch_device with torch-owned HBM buffers on torch's current stream."""
    import ctypes as C
    import numpy as np
    import torch
    from otezip_amd import corpus, _native as N
    arc, ents = corpus.build_archive(300, 65536)
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents]
    et, dst_len = gpu_lib.make_entries(tab)
    dev = torch.device("cuda", 0)
    d_src = torch.frombuffer(bytearray(arc), dtype=torch.uint8).to(dev)
    d_dst = torch.empty(dst_len, dtype=torch.uint8, device=dev)
    d_ent = torch.frombuffer(bytearray(bytes(et)), dtype=torch.uint8).to(dev)
    d_res = torch.zeros(len(tab) * 16, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream()
    N.check(gpu_lib.lib().ozg_decode_batch_device(d_src.data_ptr(), d_src.numel(), d_dst.data_ptr(), d_dst.numel(),
                                                  d_ent.data_ptr(), d_res.data_ptr(), len(tab), s.cuda_stream))
    torch.cuda.synchronize()
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint32).reshape(-1, 4)
    host = d_dst.cpu().numpy()
    for i, e in enumerate(ents):
        assert res[i, 0] == ST_OK and res[i, 2] == e["crc32"]
        assert zlib.crc32(host[et[i].dst_off:et[i].dst_off + e["uncomp_size"]].tobytes()) & 0xFFFFFFFF == e["crc32"]


def test_device_resident_batch_of_every_method_is_sniffed_on_the_device(gpu_lib):
    """Payloads that are only in HBM (ozg_decode_batch_device, no batch flags): the host cannot look at their first
    bytes, so what the batch holds is counted on the device -- a real-format kernel none of whose streams is there
    gets no launch and no scratch, the others scratch for what there is -- and the results are those of the staged
    path.  The reference's framings under 93 / 100 / 97 / 14, DEFLATE, STORE, and (where the image has the
    libraries) real Zstandard / Brotli / LZMA1 streams, interleaved."""
    import lzma
    import struct
    import numpy as np
    import torch
    from otezip_amd import corpus, _native as N
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools import brotlilib as B
    from tools import zstdlib as Z
    src, tab = bytearray(), []
    datas = [corpus.gen_text(100 + i, 3000 + 977 * i) for i in range(24)]
    for i, d in enumerate(datas):
        kind = i % 8
        if kind == 0:
            m, comp = 8, corpus.raw_deflate(d)
        elif kind == 1:
            m, comp = 0, d
        elif kind == 2:
            m, comp = 93, corpus.frame_zstd_private(d)
        elif kind == 3:
            m, comp = 100, corpus.frame_lzfse_private(d)
        elif kind == 4:
            m, comp = 97, corpus.frame_brotli_private(d)
        elif kind == 5 and B.available():
            m, comp = 97, B.compress(d, quality=5)
        elif kind == 6 and Z.available():
            m, comp = 93, Z.compress(d, 3)
        elif kind == 7:
            f = [{"id": lzma.FILTER_LZMA1, "preset": 6, "lc": 3, "lp": 0, "pb": 2, "dict_size": 1 << 20}]
            m, comp = 14, b"\x09\x14\x05\x00" + bytes([(2 * 5 + 0) * 9 + 3]) + struct.pack("<I", 1 << 20) + lzma.compress(d, format=lzma.FORMAT_RAW, filters=f)
        else:
            m, comp = 8, corpus.raw_deflate(d)
        src += b"\x00" * (i % 5)
        tab.append((m, len(src), len(comp), len(d), zlib.crc32(d) & 0xFFFFFFFF))
        src += comp
    for only_framings in (False, True):  # (second pass: no real stream at all -- none of the four kernels has work)
        t = [x for k, x in enumerate(tab) if not (only_framings and k % 8 in (5, 6, 7))]
        et, dst_len = gpu_lib.make_entries(t)
        dev = torch.device("cuda", 0)
        d_src = torch.frombuffer(bytearray(src), dtype=torch.uint8).to(dev)
        d_dst = torch.zeros(dst_len, dtype=torch.uint8, device=dev)
        d_ent = torch.frombuffer(bytearray(bytes(et)), dtype=torch.uint8).to(dev)
        d_res = torch.zeros(len(t) * 16, dtype=torch.uint8, device=dev)
        N.check(gpu_lib.lib().ozg_decode_batch_device(d_src.data_ptr(), d_src.numel(), d_dst.data_ptr(), d_dst.numel(),
                                                      d_ent.data_ptr(), d_res.data_ptr(), len(t), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint32).reshape(-1, 4)
        host = d_dst.cpu().numpy()
        staged = gpu_lib.decode_batch_host(bytes(src), t)
        for i, (m, off, cs, us, crc) in enumerate(t):
            assert res[i, 0] == ST_OK and res[i, 2] == crc, (i, m, int(res[i, 0]))
            assert host[et[i].dst_off:et[i].dst_off + us].tobytes() == staged.data(i) and staged.ok(i), (i, m)


def test_device_path_rejects_descriptors_outside_the_buffers(gpu_lib):
    """A table built from an untrusted central directory must not turn into an out-of-bounds access in
    HBM: every kernel checks src_off/comp_size/dst_off/uncomp_size against src_len/dst_len
    (overflow-safe) and fails the entry with OZG_ST_INTERNAL; its neighbours still decode."""
    import numpy as np
    import torch
    from otezip_amd import corpus, _native as N
    arc, ents = corpus.build_archive(64, 8192, methods=(8, 0, 93, 100))
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents]
    et, dst_len = gpu_lib.make_entries(tab)
    bad = {3: ("src_off", len(arc) - 5), 4: ("src_off", 2**63), 5: ("comp_size", 2**32 - 1), 9: ("dst_off", dst_len),
           10: ("dst_off", 2**64 - 16), 11: ("uncomp_size", 2**32 - 1), 17: ("dst_off", et[17].dst_off + 8),
           18: ("src_off", 2**64 - 1), 22: ("uncomp_size", dst_len + 1)}
    for i, (field, v) in bad.items():
        setattr(et[i], field, v)
    dev = torch.device("cuda", 0)
    d_src = torch.frombuffer(bytearray(arc), dtype=torch.uint8).to(dev)
    guard = 1 << 20
    d_dst = torch.full((dst_len + 2 * guard,), 0xCD, dtype=torch.uint8, device=dev)
    d_ent = torch.frombuffer(bytearray(bytes(et)), dtype=torch.uint8).to(dev)
    d_res = torch.zeros(len(tab) * 16, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream()
    N.check(gpu_lib.lib().ozg_decode_batch_device(d_src.data_ptr(), d_src.numel(), d_dst.data_ptr() + guard, dst_len,
                                                  d_ent.data_ptr(), d_res.data_ptr(), len(tab), s.cuda_stream))
    torch.cuda.synchronize()
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.int32).reshape(-1, 4)
    host = d_dst.cpu().numpy()
    assert (host[:guard] == 0xCD).all() and (host[guard + dst_len:] == 0xCD).all()   # nothing written outside dst
    for i, e in enumerate(ents):
        if i in bad:
            assert res[i, 0] == -30, (i, bad[i], res[i, 0])   # OZG_ST_INTERNAL
        else:
            assert res[i, 0] == ST_OK and np.uint32(res[i, 2]) == e["crc32"], i
    # the encode side checks its descriptors too
    import ctypes as C
    L = gpu_lib.lib()
    bufs = [corpus.gen_text(i, 5000) for i in range(8)]
    ee = (N.OzgEncEntry * 8)()
    src = b"".join(bufs)
    off = 0
    for i, b in enumerate(bufs):
        ee[i].src_off, ee[i].src_size, ee[i].method = 5000 * i, 5000, 8
        ee[i].dst_cap = int(L.ozg_compress_bound(8, 5000))
        ee[i].dst_off = off
        off += (ee[i].dst_cap + 15) & ~15
    ee[2].src_off = 2**62
    ee[5].dst_off = off + 4096
    d_s = torch.frombuffer(bytearray(src), dtype=torch.uint8).to(dev)
    d_o = torch.zeros(off + 64, dtype=torch.uint8, device=dev)
    d_e = torch.frombuffer(bytearray(bytes(ee)), dtype=torch.uint8).to(dev)
    d_r = torch.zeros(8 * 16, dtype=torch.uint8, device=dev)
    N.check(L.ozg_encode_batch_device(d_s.data_ptr(), len(src), d_o.data_ptr(), off, d_e.data_ptr(), d_r.data_ptr(), 8, s.cuda_stream))
    torch.cuda.synchronize()
    er = np.frombuffer(d_r.cpu().numpy().tobytes(), dtype=np.int32).reshape(-1, 4)
    assert [int(x) for x in er[:, 0]] == [1, 1, -30, 1, 1, -30, 1, 1]


def test_crc32_batch_device(gpu_lib):
    """ozg_crc32_batch_device = otezip_crc32 (ref src/lib/crc32.inc.c:40-47) over N ranges of a device buffer."""
    import numpy as np
    import torch
    from otezip_amd import _native as N
    rng = np.random.default_rng(3)
    buf = rng.integers(0, 256, size=3_000_000, dtype=np.uint8)
    lens = [0, 1, 3, 15, 16, 17, 255, 4095, 4096, 4097, 65536, 1_000_003, 5]
    offs, o = [], 7
    for n in lens:
        offs.append(o)
        o += n + 11
    dev = torch.device("cuda", 0)
    d_buf = torch.from_numpy(buf).to(dev)
    d_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    d_len = torch.tensor(lens, dtype=torch.int32, device=dev)
    d_crc = torch.zeros(len(lens), dtype=torch.int32, device=dev)
    N.check(gpu_lib.lib().ozg_crc32_batch_device(d_buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), d_crc.data_ptr(),
                                                len(lens), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = d_crc.cpu().numpy().astype(np.uint32)
    for g, a, n in zip(got, offs, lens):
        assert int(g) == zlib.crc32(buf[a:a + n].tobytes()) & 0xFFFFFFFF, n


def test_config1_full_count_every_entry_byte_for_byte(gpu_lib):
    """BASELINE configs[1] at its full count, 10 000 x 64 KiB DEFLATE: EVERY entry compared byte for byte with
    an independent inflate (host zlib), every 250th also with the oracle."""
    from otezip_amd import corpus
    arc, ents = corpus.build_archive(10000, 65536)
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents]
    b = gpu_lib.decode_batch_host(arc, tab)
    for i, e in enumerate(ents):
        r = b.results[i]
        assert r.status == ST_OK and r.flags == 0 and r.crc32 == e["crc32"], i
        comp = arc[e["data_off"]:e["data_off"] + e["comp_size"]]
        assert b.data(i) == zlib.decompress(comp, -15), i
        if i % 250 == 0:
            o = O.extract_entry(8, comp, e["uncomp_size"], e["crc32"])
            assert o["correct_rc"] == 0 and b.data(i) == o["out"]


def test_config2_full_count_private_zstd_framing(gpu_lib):
    """BASELINE configs[2]: 10 000 x 256 KiB entries in the private method-93 framing (2.6 GB), one GPU.
    CRC of every entry against the directory, every 500th entry byte for byte against the oracle."""
    from otezip_amd import corpus
    arc, ents = corpus.build_archive(10000, 262144, methods=(93,))
    tab = [(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in ents]
    b = gpu_lib.decode_batch_host(arc, tab)
    for i, e in enumerate(ents):
        r = b.results[i]
        assert r.status == ST_OK and r.flags == 0 and r.crc32 == e["crc32"] and r.total_out == 262144, i
        if i % 500 == 0:
            o = O.extract_entry(93, arc[e["data_off"]:e["data_off"] + e["comp_size"]], e["uncomp_size"], e["crc32"])
            assert o["correct_rc"] == 0 and b.data(i) == o["out"]
