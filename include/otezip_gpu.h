/* otezip_gpu.h -- C ABI between plain-C host code and the HIP (gfx950) codec
 * kernels of otezip-amd.  Plain pointers and sizes only; no C++ / torch types.
 *
 * What it replaces.  otezip decodes one entry per call inside
 * otezip_extract_entry() (ref src/lib/otezip.c:399-684): read the compressed
 * bytes, switch on the ZIP method, run the codec with Z_FINISH into a zeroed
 * buffer of uncomp_size bytes, CRC-32 the result.  ozg_decode_batch_* is that
 * function from ":479 switch(method)" to ":679 CRC check", for N independent
 * entries in one launch.  ozg_encode_batch_* is otezip_compress_data()
 * (ref src/lib/otezip.c:788-1076) plus the CRC of zip_file_add (:1124).
 *
 * Entry points a binding needs (the reference is C, so the "FFI" is a plain
 * declaration + link against libotezip_amd.so):
 *   ozg_init / ozg_shutdown / ozg_last_error
 *   ozg_decode_batch_device   device-resident src/dst/tables, async on a stream
 *   ozg_decode_batch_device_ex  same, with a method-mix hint
 *   ozg_decode_batch_host     host buffers; stages over PCIe and waits
 *   ozg_decode_submit / ozg_job_wait   the same, asynchronous (pinned buffers, own stream)
 *   ozg_use_device            several GPUs from one process
 *   ozg_encode_batch_device / _host   the encode side, same conventions
 *   ozg_crc32_batch_device    otezip_crc32 (ref src/lib/crc32.inc.c:40-47) x N
 *   ozg_malloc / ozg_free / ozg_memcpy_* / ozg_stream_sync   so that host C
 *                             code needs no HIP headers
 */
#ifndef OTEZIP_GPU_H
#define OTEZIP_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* library-level return codes */
#define OZG_OK 0
#define OZG_E_NODEVICE (-1) /* no HIP device / runtime: there is NO CPU fallback */
#define OZG_E_ARG (-2)
#define OZG_E_HIP (-3) /* a HIP call failed; see ozg_last_error() */
#define OZG_E_NOMEM (-4)

/* per-entry status: the zlib-style code the reference codec returns for the
 * entry (ref src/lib/deflate.inc.c:35-43), with the premature Z_BUF_ERROR of
 * the reference's inflate repaired (parity rule 2). */
#define OZG_ST_OK 1          /* Z_STREAM_END: dst[0..uncomp_size) is the entry */
#define OZG_ST_NEED_OUT 0    /* Z_OK: stream produces more than uncomp_size bytes */
#define OZG_ST_DATA (-3)     /* Z_DATA_ERROR: malformed stream */
#define OZG_ST_BUF (-5)      /* Z_BUF_ERROR: stream truncated */
#define OZG_ST_SIZE (-20)    /* codec ended but total_out != uncomp_size (ref :555, :584, :625, :654),
                                or STORE with comp_size != uncomp_size (:482) */
#define OZG_ST_BOMB (-21)    /* expansion guard, ref :454-462 */
#define OZG_ST_METHOD (-22)  /* unsupported method, ref :662-665 */
#define OZG_ST_INTERNAL (-30)

/* result flags */
#define OZG_F_REF_FALSE_REJECT 1u /* valid DEFLATE stream the reference itself rejects (F2) */
#define OZG_F_SHORT_STREAM 2u     /* DEFLATE ended early; tail zero-filled as the reference does (F12) */
#define OZG_F_CRC_MISMATCH 4u     /* crc32 != expected (ref :669) */

/* entry flags (in) */
#define OZG_EF_IGNORE_ZIPBOMB 1u /* mirrors the global otezip_ignore_zipbomb */
#define OZG_EF_SIZE_IS_CAP 2u   /* uncomp_size is the room, not the size: a Brotli stream (RFC 7932) or an end-marked LZMA1
                                 * stream that ends before it is complete, total_out says where (the codec seam's calls:
                                 * such a stream does not carry its length; a ZIP entry always does) */

#define OZG_METHOD_STORE 0
#define OZG_METHOD_DEFLATE 8
#define OZG_METHOD_LZMA 14
#define OZG_METHOD_ZSTD 93
#define OZG_METHOD_BROTLI 97
#define OZG_METHOD_LZFSE 100

/* One archive entry to decode: the fields of struct otezip_entry
 * (ref src/include/otezip/zip.h:78-88) the decode path reads, with file
 * offsets replaced by offsets into the caller's src/dst buffers. 32 bytes. */
typedef struct ozg_entry {
	uint64_t src_off;     /* compressed payload: src + src_off, any alignment */
	uint64_t dst_off;     /* output: dst + dst_off, MUST be a multiple of 16 */
	uint32_t comp_size;
	uint32_t uncomp_size; /* output capacity == the avail_out the reference passes (:507) */
	uint32_t crc32;       /* directory CRC-32 to compare with */
	uint16_t method;
	uint16_t flags;       /* OZG_EF_* */
} ozg_entry;

/* 16 bytes */
typedef struct ozg_result {
	int32_t status;     /* OZG_ST_* */
	uint32_t total_out; /* bytes the codec produced */
	uint32_t crc32;     /* CRC-32 of dst[0..uncomp_size) when status == OZG_ST_OK */
	uint32_t flags;     /* OZG_F_* */
} ozg_result;

/* Tunables mirrored from the reference's globals (ref src/lib/otezip.c:157-166). */
typedef struct ozg_limits {
	uint64_t max_expansion_ratio; /* default 1000 */
	uint64_t max_expansion_slack; /* default 1 MiB */
} ozg_limits;

/* One context per device.  ozg_init(d) prepares device d and makes it the calling thread's
 * current one; a host that drives several GPUs of a node calls it once per device and then
 * picks the device of the following calls with ozg_use_device(d) (per thread).  ozg_shutdown
 * releases every context.  Device numbers are LOGICAL: with OTEZIP_GPU_DEVICE_MAP="0,0,1" in the
 * environment logical devices 0 and 1 are two independent contexts on physical GPU 0 and logical 2 is
 * one on GPU 1 (ozg_device_count() then returns 3); without it logical = physical.  A thread that has
 * not chosen starts on the device of the process's first successful ozg_init. */
int ozg_init(int device);
int ozg_use_device(int device);
int ozg_current_device(void);
void ozg_shutdown(void);
const char *ozg_last_error(void);
int ozg_device_count(void);
void ozg_set_limits(const ozg_limits *lim);

/* All pointers are device pointers; the call only enqueues work on `stream`
 * (a hipStream_t passed as void*, NULL = default stream).  dst regions of
 * different entries must not overlap; each needs uncomp_size bytes rounded up
 * to 16.  Entries whose status != OZG_ST_OK leave their dst region undefined. */
int ozg_decode_batch_device(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_entry *d_entries, ozg_result *d_results, uint32_t n, void *stream);

/* Same, with a hint about the method mix so that a launch can be skipped when
 * the caller (who parsed the central directory) knows it is not needed. */
#define OZG_BATCH_ONLY_DEFLATE 1u /* every entry is method 8: skip the copy kernel */
#define OZG_BATCH_NO_DEFLATE 2u   /* no entry is method 8: skip the inflate kernel */
#define OZG_BATCH_NO_LZMA 8u      /* no entry is method 14: skip the LZMA1 kernel (a method-14 payload is sniffed: an LZMA1
                                   * stream behind the APPNOTE 5.8.8 header, or the reference's private token format) */
#define OZG_BATCH_NO_ZSTD 4u      /* no entry is method 93: skip the Zstandard kernels (a method-93 payload is sniffed:
                                   * a Zstandard frame per RFC 8878, or the reference's private raw-block framing) */
#define OZG_BATCH_NO_BROTLI 16u   /* no entry is method 97: skip the Brotli kernel (a method-97 payload is sniffed: the
                                   * reference's "BROT" container, or a Brotli stream per RFC 7932) */
#define OZG_BATCH_NO_LZFSE 32u    /* no entry is method 100: skip the LZFSE kernel (a method-100 payload is sniffed: the
                                   * reference's 06 framing, or an LZFSE stream of bvx blocks -- parity unpinned) */
int ozg_decode_batch_device_ex(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_entry *d_entries, ozg_result *d_results, uint32_t n, void *stream, uint32_t batch_flags);

/* Same with host buffers: copies src and the tables to the device, runs the
 * batch, copies dst and results back, and waits. */
int ozg_decode_batch_host(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *results, uint32_t n);

/* ozg_decode_batch_host plus, for DEFLATE entries, where each stream ended: end_bits[i] = the bit
 * position behind the final end-of-block code (status OZG_ST_OK), or behind the token that did
 * not fit (OZG_ST_NEED_OUT: dst then holds uncomp_size bytes, as the reference's inflate leaves
 * its buffer when it returns Z_OK); 0 otherwise and for other methods.  This is what the z_stream
 * seam derives total_in from.  end_bits may be NULL. */
int ozg_decode_batch_host_ex(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *results, uint32_t n, uint64_t *end_bits);

/* Asynchronous form of ozg_decode_batch_host for an ingest / egress pipeline: enqueues the H2D
 * copies, the kernels and the D2H copies on a stream of its own and returns at once.  h_src,
 * h_dst and h_results should come from ozg_malloc_pinned (pageable memory works, but then the
 * copies are not asynchronous); they must stay valid until the job has been waited for; the
 * descriptor table is copied.  At most eight jobs per device are in flight (OZG_E_NOMEM
 * beyond that: back-pressure, not failure).  ozg_job_wait blocks until outputs and results are in host
 * memory and releases the job; ozg_job_done polls (1: a wait would not block) and never releases. */
typedef struct ozg_job ozg_job;
int ozg_decode_submit(const void *h_src, uint64_t src_len, void *h_dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *h_results, uint32_t n, ozg_job **job);
int ozg_job_done(ozg_job *job);
int ozg_job_wait(ozg_job *job);

/* ---- encode: otezip_compress_data (ref src/lib/otezip.c:788-1076) + the CRC of zip_file_add (:1124) ---- */

/* One buffer to compress. 32 bytes. */
typedef struct ozg_enc_entry {
	uint64_t src_off;  /* input: src + src_off, any alignment */
	uint64_t dst_off;  /* output: dst + dst_off, MUST be a multiple of 16 */
	uint32_t src_size;
	uint32_t dst_cap;  /* >= ozg_compress_bound(method, src_size) */
	uint16_t method;   /* requested method (what za->default_method holds) */
	uint16_t level;    /* ignored: the reference always runs its default level (ref :826) */
	uint32_t reserved; /* 0, or OZG_ENC_F_* */
} ozg_enc_entry;
/* codec-level call (the <codec>Compress() of the z_stream seam): always write the requested
 * method -- DEFLATE, or the reference's private 93 / 100 / 97 / 14 framings, byte-exact --
 * never fall back to STORE; OZG_ST_BUF if dst_cap is too small. */
#define OZG_ENC_F_CODEC_ONLY 1u
/* report the Adler-32 of the input in ozg_enc_result.crc32 instead of its CRC-32 (the zlib framing of the
 * z_stream seam, RFC 1950, ends with it) */
#define OZG_ENC_F_ADLER32 2u

/* 16 bytes */
typedef struct ozg_enc_result {
	int32_t status;     /* OZG_ST_OK, or OZG_ST_INTERNAL on a bad descriptor */
	uint32_t comp_size; /* bytes written at dst + dst_off */
	uint32_t crc32;     /* CRC-32 of the INPUT buffer */
	uint16_t method;    /* method actually written: the requested one, or STORE after the
	                       reference's "not smaller -> STORE" rule (ref :846-850 etc.) */
	uint16_t flags;
} ozg_enc_result;

/* output space the reference allocates per method (ref :819, :860, :931, :975, :1019),
 * never less than src_size so that the STORE fallback fits */
uint64_t ozg_compress_bound(uint16_t method, uint64_t src_size);

int ozg_encode_batch_device(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_enc_entry *d_entries, ozg_enc_result *d_results, uint32_t n, void *stream);
int ozg_encode_batch_host(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_enc_entry *entries, ozg_enc_result *results, uint32_t n);
/* The encode side of the asynchronous pipeline (ozg_job_wait / ozg_job_done as for decode; what
 * zip_close's batched writer drives): one job = H2D of the sources, otezip_compress_data
 * (ref src/lib/otezip.c:788-1076) + CRC-32 (:1124) for every buffer, D2H.  Pinned buffers. */
int ozg_encode_submit(const void *h_src, uint64_t src_len, void *h_dst, uint64_t dst_len,
	const ozg_enc_entry *entries, ozg_enc_result *h_results, uint32_t n, ozg_job **job);
/* The same, with the outputs PACKED on the device before they cross PCIe: after ozg_job_wait the
 * payload of entry i starts at the sum over j < i of ((h_results[j].comp_size + 15) & ~15) (entries whose
 * status is not OZG_ST_OK take no room) instead of at dst_off[i].  dst_off / dst_cap still describe the
 * device-side slots; h_dst needs room for the packed bytes only (at most dst_len).  What the batched
 * writer of zip_close uses: the worst-case slot layout is 2-3 x the bytes that matter.  The payload's copy
 * to the host can only be issued once the sizes are back: ozg_job_wait does that, and so does an ozg_job_done
 * poll that finds them back (it then reports 0 until that copy has finished too). */
int ozg_encode_submit_packed(const void *h_src, uint64_t src_len, void *h_dst, uint64_t dst_len,
	const ozg_enc_entry *entries, ozg_enc_result *h_results, uint32_t n, ozg_job **job);

/* Checksum of a host buffer on the device (one call = H2D, one kernel, 4 bytes back): kind 0 = CRC-32,
 * 1 = Adler-32.  What inflate() of the z_stream seam verifies a zlib / gzip trailer with. */
int ozg_checksum_host(int kind, const void *buf, uint64_t len, uint32_t *out);

/* crc[i] = CRC-32 of d_buf[off[i] .. off[i]+len[i]) (all device pointers). */
int ozg_crc32_batch_device(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t *d_crc,
	uint32_t n, void *stream);

/* thin wrappers so a C host needs no HIP headers */
void *ozg_malloc(size_t bytes);
void ozg_free(void *d_ptr);
void *ozg_malloc_pinned(size_t bytes);
void ozg_free_pinned(void *h_ptr);
int ozg_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int ozg_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int ozg_stream_sync(void *stream);

/* Diagnostics only: d_buf = n x 32 uint64 device words which the DEFLATE kernels fill with
 * per-phase shader-cycle counts for every entry of the following batches (16 words of the
 * apply kernel, then 16 of the index kernel); NULL = off. */
void ozg_debug_set_profile_buffer(void *d_buf);

/* Names of the kernels a profiler will show, for tooling. */
const char *ozg_kernel_name_inflate(void); /* second half of the DEFLATE decode (index -> bytes) */
const char *ozg_kernel_name_index(void);   /* first half (token chain -> index) */
const char *ozg_kernel_name_copy(void);

#ifdef __cplusplus
}
#endif
#endif
