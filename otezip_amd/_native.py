"""ctypes binding of include/otezip_gpu.h (the ozg_* C ABI)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

ST_OK, ST_NEED_OUT, ST_DATA, ST_BUF, ST_SIZE, ST_BOMB, ST_METHOD, ST_INTERNAL = 1, 0, -3, -5, -20, -21, -22, -30
F_REF_FALSE_REJECT, F_SHORT_STREAM, F_CRC_MISMATCH = 1, 2, 4
METHOD_STORE, METHOD_DEFLATE, METHOD_LZMA, METHOD_ZSTD, METHOD_BROTLI, METHOD_LZFSE = 0, 8, 14, 93, 97, 100
EF_IGNORE_ZIPBOMB, EF_SIZE_IS_CAP = 1, 2  # ozg_entry.flags
BATCH_ONLY_DEFLATE, BATCH_NO_DEFLATE, BATCH_NO_ZSTD, BATCH_NO_LZMA, BATCH_NO_BROTLI, BATCH_NO_LZFSE = 1, 2, 4, 8, 16, 32


class NativeError(RuntimeError):
    pass


class OzgEntry(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("comp_size", C.c_uint32),
                ("uncomp_size", C.c_uint32), ("crc32", C.c_uint32), ("method", C.c_uint16), ("flags", C.c_uint16)]


class OzgResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("total_out", C.c_uint32), ("crc32", C.c_uint32), ("flags", C.c_uint32)]


class OzgEncEntry(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("src_size", C.c_uint32), ("dst_cap", C.c_uint32),
                ("method", C.c_uint16), ("level", C.c_uint16), ("reserved", C.c_uint32)]


class OzgEncResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("comp_size", C.c_uint32), ("crc32", C.c_uint32), ("method", C.c_uint16),
                ("flags", C.c_uint16)]


class OzgLimits(C.Structure):
    _fields_ = [("max_expansion_ratio", C.c_uint64), ("max_expansion_slack", C.c_uint64)]


def lib_path():
    return os.path.join(HERE, "libotezip_amd.so")


def build_native(force=False):
    """hipcc --offload-arch=gfx950 build of the in-tree shared library."""
    args = ["make", "-s", "-C", os.path.join(HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return lib_path()


_lib = None
_inited = False

# every symbol include/otezip_gpu.h declares: (restype, argtypes)
_VP, _U64, _U32 = C.c_void_p, C.c_uint64, C.c_uint32
ABI = {
    "ozg_init": (C.c_int, [C.c_int]),
    "ozg_use_device": (C.c_int, [C.c_int]),
    "ozg_current_device": (C.c_int, []),
    "ozg_decode_submit": (C.c_int, [_VP, _U64, _VP, _U64, C.POINTER(OzgEntry), _VP, _U32, C.POINTER(_VP)]),
    "ozg_encode_submit": (C.c_int, [_VP, _U64, _VP, _U64, _VP, _VP, _U32, C.POINTER(_VP)]),
    "ozg_checksum_host": (C.c_int, [C.c_int, _VP, _U64, C.POINTER(C.c_uint32)]),
    "ozg_encode_submit_packed": (C.c_int, [_VP, _U64, _VP, _U64, _VP, _VP, _U32, C.POINTER(_VP)]),
    "ozg_job_done": (C.c_int, [_VP]),
    "ozg_job_wait": (C.c_int, [_VP]),
    "ozg_shutdown": (None, []),
    "ozg_last_error": (C.c_char_p, []),
    "ozg_device_count": (C.c_int, []),
    "ozg_set_limits": (None, [C.POINTER(OzgLimits)]),
    "ozg_decode_batch_device": (C.c_int, [_VP, _U64, _VP, _U64, _VP, _VP, _U32, _VP]),
    "ozg_decode_batch_device_ex": (C.c_int, [_VP, _U64, _VP, _U64, _VP, _VP, _U32, _VP, _U32]),
    "ozg_decode_batch_host": (C.c_int, [_VP, _U64, _VP, _U64, C.POINTER(OzgEntry), C.POINTER(OzgResult), _U32]),
    "ozg_decode_batch_host_ex": (C.c_int, [_VP, _U64, _VP, _U64, C.POINTER(OzgEntry), C.POINTER(OzgResult), _U32, _VP]),
    "ozg_compress_bound": (C.c_uint64, [C.c_uint16, _U64]),
    "ozg_encode_batch_device": (C.c_int, [_VP, _U64, _VP, _U64, _VP, _VP, _U32, _VP]),
    "ozg_encode_batch_host": (C.c_int, [_VP, _U64, _VP, _U64, C.POINTER(OzgEncEntry), C.POINTER(OzgEncResult), _U32]),
    "ozg_crc32_batch_device": (C.c_int, [_VP, _VP, _VP, _VP, _U32, _VP]),
    "ozg_malloc": (_VP, [C.c_size_t]),
    "ozg_free": (None, [_VP]),
    "ozg_malloc_pinned": (_VP, [C.c_size_t]),
    "ozg_free_pinned": (None, [_VP]),
    "ozg_memcpy_h2d": (C.c_int, [_VP, _VP, C.c_size_t, _VP]),
    "ozg_memcpy_d2h": (C.c_int, [_VP, _VP, C.c_size_t, _VP]),
    "ozg_stream_sync": (C.c_int, [_VP]),
    "ozg_debug_set_profile_buffer": (None, [_VP]),
    "ozg_kernel_name_inflate": (C.c_char_p, []),
    "ozg_kernel_name_index": (C.c_char_p, []),
    "ozg_kernel_name_copy": (C.c_char_p, []),
}


def lib():
    """Load the shared library (no GPU needed for this). Raises if it is absent."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise NativeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and looks it up by
        # file name, so it has to be loaded BEFORE this library's DT_NEEDED entry is resolved
        # (then both share torch's copy).  Plain C hosts never load torch and use /opt/rocm.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(p)
        for name, (res, args) in ABI.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc, what="ozg call"):
    if rc != 0:
        raise NativeError("%s failed (%d): %s" % (what, rc, lib().ozg_last_error().decode()))


def init(device=0):
    """ozg_init: selects the HIP device. Raises NativeError when there is none."""
    global _inited
    check(lib().ozg_init(device), "ozg_init(%d)" % device)
    _inited = True


def shutdown():
    global _inited
    if _inited:
        lib().ozg_shutdown()
        _inited = False
