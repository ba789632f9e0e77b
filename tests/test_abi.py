"""The C-ABI library: builds for gfx950, loads without a GPU, exports every symbol the public
header declares, has the documented struct layouts, and refuses to run without a device
(there is no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import otezip_amd as oz
    if not os.path.exists(oz.lib_path()):
        oz.build_native()
    return oz


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ozg|zip|otezip)_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(native):
    lib = native.lib()
    names = _declared("otezip_gpu.h")
    assert "ozg_decode_batch_device" in names and "ozg_decode_batch_host" in names
    for name in names:
        assert hasattr(lib, name), "libotezip_amd.so does not export " + name
    from otezip_amd import _native
    assert set(names) <= set(_native.ABI), "python mirror is missing: %s" % (set(names) - set(_native.ABI))


def test_struct_layouts_match_header(native):
    from otezip_amd import _native as N
    assert C.sizeof(N.OzgEntry) == 32 and C.sizeof(N.OzgResult) == 16
    assert N.OzgEntry.src_off.offset == 0 and N.OzgEntry.dst_off.offset == 8
    assert N.OzgEntry.comp_size.offset == 16 and N.OzgEntry.uncomp_size.offset == 20
    assert N.OzgEntry.crc32.offset == 24 and N.OzgEntry.method.offset == 28 and N.OzgEntry.flags.offset == 30


def test_code_object_targets_gfx950(native):
    # llvm-objdump --offloading drops the extracted code objects next to its input: point
    # it at a throw-away copy
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        copy = shutil.copy(native.lib_path(), tmp)
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", copy],
                             capture_output=True, text=True, cwd=tmp).stdout
    if not out.strip():
        pytest.skip("llvm-objdump --offloading unavailable")
    assert "gfx950" in out


def _kernel_resources(name):
    rep = os.path.join(ROOT, "build", "otz_kernels.resources.txt")
    if not os.path.exists(rep):
        pytest.skip("no resource report (library was not built in this tree)")
    txt = open(rep).read()
    m = re.search(r"Function Name: %s\b.*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                  r"Occupancy \[waves/SIMD\]: (\d+)" % name, txt, flags=re.S)
    assert m, name + " missing from the resource report"
    return tuple(map(int, m.groups()))


def test_apply_kernel_keeps_seven_workgroups_per_cu(native):
    """Seven 256-work-item workgroups per CU = 7 waves per SIMD = at most 72 VGPRs (and at most 18 granules of
    LDS each: tests/test_emu_parity.py).  Wavefronts per CU are what the kernel's time follows.
    No scratch segment (see the index kernel's test below).  The build writes the compiler's resource report
    next to the object file."""
    vgprs, scratch, occ = _kernel_resources("otz_apply_kernel")
    assert vgprs <= 72 and scratch == 0 and occ >= 7, (vgprs, scratch, occ)


def test_index_kernel_keeps_five_waves_per_simd(native):
    """One wavefront per entry, 20 of them per CU: <= 8 KB of LDS (checked in test_emu_parity) and <= 96
    VGPRs, and no scratch segment: spills cost memory round trips, and on this pool a kernel with a
    scratch segment has hung rocprofv3's --pmc passes."""
    vgprs, scratch, occ = _kernel_resources("otz_index_kernel")
    assert vgprs <= 96 and scratch == 0 and occ >= 5, (vgprs, scratch, occ)


def test_fails_loudly_without_a_gpu(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert native.lib().ozg_device_count() == 0
    with pytest.raises(native.NativeError):
        native.init(0)
    # and nothing decodes without ozg_init
    b = (native.OzgEntry * 1)()
    r = (native.OzgResult * 1)()
    rc = native.lib().ozg_decode_batch_host(b"x", 1, C.create_string_buffer(16), 16, b, r, 1)
    assert rc == -1  # OZG_E_NODEVICE
    assert b"no CPU" in native.lib().ozg_last_error() or b"no HIP" in native.lib().ozg_last_error()


def test_no_product_module_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under otezip_amd/ may import, link or call it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "otezip_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "otz_oracle" not in text and "libotz_oracle" not in text, f
