#!/usr/bin/env python3
"""Generate (and cache on local disk) every synthetic corpus the measurement tools use.  Run it
UNPROFILED before any rocprofv3 pass: corpus generation forks a worker pool, and a process the
profiler has already GPU-initialised must not fork (tools/profile_round.sh, tools/gpu_quick.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otezip_amd import corpus
from tools import zstd_corpus


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    import bench
    bench.load_shard(0, 10000, 65536, 8, os.environ.get("OTEZIP_BENCH_CACHE", "/tmp/otezip_amd_cache"))
    corpus.build_archive(max(n // 4, 100), 262144, methods=(93,))
    corpus.build_archive(n, 65536, methods=(100, 97))
    corpus.build_archive(min(n, 4000), 65536, methods=(8,))
    for k in (2000, 4000, 10000):
        corpus.build_archive(k, 65536)
    corpus.text_buffers(min(n, 10000), 65536, first=50000)  # (exactly what tools/bench_secondary.py asks for with this n)
    if n != 10000:
        corpus.text_buffers(10000, 65536, first=50000)
    try:
        zstd_corpus.build_zstd_real(max(n // 4, 100), 262144)
        zstd_corpus.build_zstd_real(n, 65536)
    except Exception as ex:
        print("real zstd corpus unavailable: %r" % (ex,))
    try:
        zstd_corpus.build_brotli_real(n, 65536)
        zstd_corpus.build_lzma_real(n, 65536)
        zstd_corpus.build_lzfse_real(min(n, 4000), 65536)
    except Exception as ex:
        print("real Brotli / LZMA corpus unavailable: %r" % (ex,))
    print("corpora cached")


if __name__ == "__main__":
    main()
