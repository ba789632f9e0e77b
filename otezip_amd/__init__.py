"""otezip_amd -- MI355X (gfx950) native implementation of otezip's per-entry
compress/decompress hot path, behind the reference's libzip-subset C API.

The product is the C-ABI shared library ``otezip_amd/libotezip_amd.so`` (HIP
kernels + ``ozg_*`` batch ABI + ``zip_*`` host API).  This Python package is a
thin ctypes mirror of that ABI for tests, the benchmark and torch-side plumbing
(device buffers, streams, torch.distributed); it contains no codec logic and no
CPU fallback -- without the library or without a GPU it raises.
"""
from ._native import (  # noqa: F401
    OzgEntry, OzgResult, NativeError, lib, lib_path, build_native, init, shutdown,
    ST_OK, ST_NEED_OUT, ST_DATA, ST_BUF, ST_SIZE, ST_BOMB, ST_METHOD, ST_INTERNAL,
    F_REF_FALSE_REJECT, F_SHORT_STREAM, F_CRC_MISMATCH,
    METHOD_STORE, METHOD_DEFLATE, METHOD_LZMA, METHOD_ZSTD, METHOD_BROTLI, METHOD_LZFSE,
    BATCH_ONLY_DEFLATE, BATCH_NO_DEFLATE, BATCH_NO_ZSTD, BATCH_NO_LZMA, BATCH_NO_BROTLI, BATCH_NO_LZFSE,
)
from .batch import Batch, decode_batch_host, encode_batch_host, make_entries  # noqa: F401

__version__ = "0.1.0"
