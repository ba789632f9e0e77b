/* otezip/zip.h -- the libzip-subset C API of otezip, served by otezip-amd.
 *
 * Drop-in for the reference header (ref src/include/otezip/zip.h): the same
 * function names, argument meaning and error behaviour (:192-235), the same
 * constants (:129-170) and -- because callers read them directly -- the same
 * struct layouts (:78-110, :173-182).  Link against libotezip_amd.so instead of
 * compiling the reference's otezip.c.
 *
 * What is different underneath: zip_fopen_index() does not decode one entry on
 * the CPU.  The first request for an entry decodes a whole window of
 * consecutive entries in one GPU batch (ozg_decode_batch_host, see
 * otezip_gpu.h); later requests inside the window are served from it.  There
 * is no CPU decoder in this library: without a HIP device zip_fopen_index()
 * returns NULL and says why on stderr.
 */
#ifndef OTEZIP_H_
#define OTEZIP_H_

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "./config.h"

/* unaligned little-endian helpers used by callers of the reference header */
static inline uint16_t otezip_read_le16(const uint8_t *p) { return (uint16_t) (p[0] | (p[1] << 8)); }
static inline uint32_t otezip_read_le32(const uint8_t *p) {
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline uint64_t otezip_read_le64(const uint8_t *p) {
	return (uint64_t)otezip_read_le32 (p) | ((uint64_t)otezip_read_le32 (p + 4) << 32);
}
static inline void otezip_write_le16(uint8_t *p, uint16_t v) {
	p[0] = (uint8_t)v;
	p[1] = (uint8_t) (v >> 8);
}
static inline void otezip_write_le32(uint8_t *p, uint32_t v) {
	otezip_write_le16 (p, (uint16_t)v);
	otezip_write_le16 (p + 2, (uint16_t) (v >> 16));
}
static inline void otezip_write_le64(uint8_t *p, uint64_t v) {
	otezip_write_le32 (p, (uint32_t)v);
	otezip_write_le32 (p + 4, (uint32_t) (v >> 32));
}

typedef uint64_t zip_uint64_t;
typedef int64_t zip_int64_t;
typedef int zip_flags_t;
typedef int32_t zip_int32_t;
typedef uint32_t zip_uint32_t;
typedef uint16_t zip_uint16_t;
typedef uint8_t zip_uint8_t;

/* one central-directory entry (layout: ref zip.h:78-88) */
struct otezip_entry {
	char *name;
	uint32_t local_hdr_ofs;
	uint32_t comp_size;
	uint32_t uncomp_size;
	uint16_t method;
	uint32_t crc32;
	uint16_t file_time;
	uint16_t file_date;
	uint32_t external_attr;
};

/* archive handle; the CLI of the reference reads entries[] and writes
 * default_method directly, so this prefix is public (layout: ref zip.h:91-98) */
struct zip {
	FILE *fp;
	struct otezip_entry *entries;
	zip_uint64_t n_entries;
	int mode; /* 0 read-only, 1 write */
	zip_uint64_t next_index;
	uint16_t default_method;
};

/* an opened entry: the whole uncompressed file (layout: ref zip.h:100-104) */
struct zip_file {
	uint8_t *data;
	uint32_t size;
	zip_uint64_t pos;
};

struct zip_source {
	const void *buf;
	zip_uint64_t len;
	int freep;
};

struct otezip_error {
	int zip_err;
	int sys_err;
};

typedef struct zip otezip_archive;
typedef struct zip_file otezip_file;
typedef struct zip_source otezip_src_buf;
typedef struct zip zip_t;
typedef struct zip_file zip_file_t;
typedef struct zip_source zip_source_t;
typedef struct otezip_error zip_error_t;

#ifndef ZIP_RDONLY
#define ZIP_RDONLY 0
#endif
#ifndef ZIP_CREATE
#define ZIP_CREATE 1
#endif
#ifndef ZIP_EXCL
#define ZIP_EXCL 2
#endif
#ifndef ZIP_TRUNCATE
#define ZIP_TRUNCATE 8
#endif

#define ZIP_CM_STORE 0
#define ZIP_CM_DEFLATE 8
#define ZIP_UINT64_MAX ((zip_uint64_t)-1)

#define ZIP_ER_OK 0
#define ZIP_ER_READ 5
#define ZIP_ER_NOENT 9
#define ZIP_ER_EXISTS 10
#define ZIP_ER_OPEN 11
#define ZIP_ER_INVAL 18
#define ZIP_ER_NOZIP 19
#define ZIP_ER_INCONS 21
#define ZIP_ER_RDONLY 25

#define ZIP_STAT_NAME 0x0001u
#define ZIP_STAT_INDEX 0x0002u
#define ZIP_STAT_SIZE 0x0004u
#define ZIP_STAT_COMP_SIZE 0x0008u
#define ZIP_STAT_MTIME 0x0010u
#define ZIP_STAT_CRC 0x0020u
#define ZIP_STAT_COMP_METHOD 0x0040u

struct zip_stat {
	zip_uint64_t valid;
	const char *name;
	zip_uint64_t index;
	zip_uint64_t size;
	zip_uint64_t comp_size;
	time_t mtime;
	zip_uint32_t crc;
	zip_uint16_t comp_method;
};
typedef struct zip_stat zip_stat_t;

#ifdef __cplusplus
extern "C" {
#endif

zip_t *zip_open(const char *path, int flags, int *errorp);
zip_t *zip_open_from_source(zip_source_t *src, int flags, zip_error_t *error);
int zip_close(zip_t *za);

zip_uint64_t zip_get_num_files(zip_t *za);
zip_int64_t zip_name_locate(zip_t *za, const char *fname, zip_flags_t flags);
const char *zip_get_name(zip_t *za, zip_uint64_t index, zip_flags_t flags);

zip_file_t *zip_fopen_index(zip_t *za, zip_uint64_t index, zip_flags_t flags);
int zip_fclose(zip_file_t *zf);
zip_int64_t zip_fread(zip_file_t *zf, void *buf, zip_uint64_t nbytes);

zip_source_t *zip_source_buffer(zip_t *za, const void *data, zip_uint64_t len, int freep);
zip_source_t *zip_source_buffer_create(const void *data, zip_uint64_t len, int freep, zip_error_t *error);
void zip_source_free(zip_source_t *src);
zip_int64_t zip_file_add(zip_t *za, const char *name, zip_source_t *src, zip_flags_t flags);
int zip_file_replace(zip_t *za, zip_uint64_t index, zip_source_t *src, zip_flags_t flags);
int zip_replace(zip_t *za, zip_uint64_t index, zip_source_t *src);
zip_int64_t zip_add(zip_t *za, const char *name, zip_source_t *src);
int zip_set_file_compression(zip_t *za, zip_uint64_t index, zip_int32_t comp, zip_uint32_t comp_flags);

int zip_stat(zip_t *za, const char *fname, zip_flags_t flags, zip_stat_t *st);
int zip_stat_index(zip_t *za, zip_uint64_t index, zip_flags_t flags, zip_stat_t *st);
void zip_stat_init(zip_stat_t *st);

int otezip_method_from_string(const char *method_name);

#ifdef __cplusplus
}
#endif

/* runtime switches, same names and defaults as the reference (ref otezip.c:157-166) */
extern int otezip_verify_crc;
extern uint64_t otezip_max_expansion_ratio;
extern uint64_t otezip_max_expansion_slack;
extern int otezip_ignore_zipbomb;

#endif /* OTEZIP_H_ */
