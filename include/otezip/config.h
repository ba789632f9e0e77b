/* otezip/config.h -- build-time switches and ZIP method ids of otezip-amd.
 * Same macro names and values as the reference's config.h
 * (ref src/include/otezip/config.h:16-40) so that callers compile unchanged. */
#ifndef OTEZIP_CONFIG_H
#define OTEZIP_CONFIG_H

#define OTEZIP_VERSION_MAJOR 0
#define OTEZIP_VERSION_MINOR 4
#define OTEZIP_VERSION_PATCH 8
#define OTEZIP_STRINGIFY(x) #x
#define OTEZIP_TOSTRING(x) OTEZIP_STRINGIFY (x)
#define OTEZIP_VERSION OTEZIP_TOSTRING (OTEZIP_VERSION_MAJOR) "." OTEZIP_TOSTRING (OTEZIP_VERSION_MINOR) "." OTEZIP_TOSTRING (OTEZIP_VERSION_PATCH)

/* every codec of the reference is served by the GPU library */
#define OTEZIP_ENABLE_STORE 1
#define OTEZIP_ENABLE_DEFLATE 1
#define OTEZIP_ENABLE_ZSTD 1
#define OTEZIP_ENABLE_LZFSE 1
#define OTEZIP_ENABLE_LZMA 1
#define OTEZIP_ENABLE_BROTLI 1

/* ZIP method ids (APPNOTE values; 100 is the reference's private choice for lzfse) */
#define OTEZIP_METHOD_STORE 0
#define OTEZIP_METHOD_DEFLATE 8
#define OTEZIP_METHOD_LZMA 14
#define OTEZIP_METHOD_ZSTD 93
#define OTEZIP_METHOD_LZ4 94
#define OTEZIP_METHOD_BROTLI 97
#define OTEZIP_METHOD_LZFSE 100

#define OTEZIP_LZMA_HEADER_SIZE 13
#define OTEZIP_LZMA_OVERHEAD_RATIO 8

#endif
