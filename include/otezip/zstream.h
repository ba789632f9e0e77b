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

#endif
