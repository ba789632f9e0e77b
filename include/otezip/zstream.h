/* otezip/zstream.h -- the zlib-shaped stream record the reference's codec entry
 * points take (ref src/include/otezip/zstream.h:14-33).  Field order, types and
 * therefore layout are identical to system zlib's z_stream. */
#ifndef OTEZIP_ZSTREAM_H
#define OTEZIP_ZSTREAM_H

#include <stdint.h>

typedef unsigned char Bytef;
typedef unsigned int uInt;
typedef unsigned long uLong;
typedef void *voidpf;

typedef struct {
	const Bytef *next_in;
	uInt avail_in;
	uLong total_in;
	Bytef *next_out;
	uInt avail_out;
	uLong total_out;
	const char *msg;
	void *state;
	void *zalloc;
	void *zfree;
	voidpf opaque;
	int data_type;
	uLong adler;
	uLong reserved;
} z_stream;
typedef z_stream *z_streamp;

/* Return codes and flush values (same numbers as zlib; ref src/include/otezip/zlib.h:40-63). */
#define Z_OK 0
#define Z_STREAM_END 1
#define Z_STREAM_ERROR (-2)
#define Z_DATA_ERROR (-3)
#define Z_MEM_ERROR (-4)
#define Z_BUF_ERROR (-5)
#define Z_NO_FLUSH 0
#define Z_FINISH 4
#define Z_DEFLATED 8
#define MAX_WBITS 15

/* The DEFLATE codec seam (ref src/lib/deflate.inc.c:89-97).  One call of inflate() /
 * deflate() processes next_in[0, avail_in) into next_out[0, avail_out) on the GPU batch path
 * (a batch of one); see otezip_amd/csrc/zstream_api.c for the contract.  The container
 * codecs' z_stream entry points (zstd*, lzfse*, brotli*, lzma*) are not exported yet:
 * those methods are reached through zip_fopen_index / zip_file_add. */
#ifdef __cplusplus
extern "C" {
#endif
int inflateInit2(z_stream *strm, int windowBits);
int inflateInit2_(z_stream *strm, int windowBits, const char *version, int stream_size);
int inflate(z_stream *strm, int flush);
int inflateEnd(z_stream *strm);
int deflateInit2(z_stream *strm, int level, int method, int windowBits, int memLevel, int strategy);
int deflateInit2_(z_stream *strm, int level, int method, int windowBits, int memLevel, int strategy, const char *version, int stream_size);
int deflate(z_stream *strm, int flush);
int deflateEnd(z_stream *strm);
#ifdef __cplusplus
}
#endif

#endif
