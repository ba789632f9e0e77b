/* otezip/zstream.h -- the zlib-shaped stream record the reference's codec entry
 * points take (ref src/include/otezip/zstream.h:14-33).  Field order, types and
 * therefore layout are identical to system zlib's z_stream. */
#ifndef OTEZIP_ZSTREAM_H
#define OTEZIP_ZSTREAM_H

#include <stddef.h>
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

/* The codec seam (ref src/lib/deflate.inc.c:89-97, zstd.inc.c:135-148, lzfse.inc.c:46-47 and
 * :272-372, brotli.inc.c:82-214, lzma-enc.inc.c:92-98, lzma-dec.inc.c:84-90).  One call of
 * inflate() / deflate() / <codec>Compress() / <codec>Decompress() processes next_in[0, avail_in)
 * into next_out[0, avail_out) on the GPU batch path (a batch of one); see
 * otezip_amd/csrc/zstream_api.c for the contract. */
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

#define OTEZIP_CODEC_DECLS(name) \
	int name##Init(z_stream *strm, int level); \
	int name##Compress(z_stream *strm, int flush); \
	int name##End(z_stream *strm); \
	int name##DecompressInit(z_stream *strm); \
	int name##Decompress(z_stream *strm, int flush); \
	int name##DecompressEnd(z_stream *strm); \
	int name##CompressInit2(z_stream *strm, int level, int windowBits, int memLevel, int strategy); \
	int name##CompressInit2_(z_stream *strm, int level, int windowBits, int memLevel, int strategy, const char *version, int stream_size); \
	int name##DecompressInit2(z_stream *strm, int windowBits); \
	int name##DecompressInit2_(z_stream *strm, int windowBits, const char *version, int stream_size);
OTEZIP_CODEC_DECLS (zstd)
OTEZIP_CODEC_DECLS (lzfse)
OTEZIP_CODEC_DECLS (brotli)
OTEZIP_CODEC_DECLS (lzma)
size_t lzfse_compress(const void *in, size_t in_sz, void *out, size_t out_cap);
size_t lzfse_decompress(const void *in, size_t in_sz, void *out, size_t out_cap);
#ifdef __cplusplus
}
#endif

#endif
