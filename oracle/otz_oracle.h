/* otz_oracle.h -- CPU restatement of the otezip per-entry codec hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The shipped path is the HIP
 * library under otezip_amd/csrc, which never links or calls this code.
 *
 * Parity status: PINNED.  Every function here is validated bit-for-bit against
 * the reference itself (oracle/_ref/libotezip_ref.so, built by oracle/Makefile
 * from the sources under /root/reference, never copied) by
 * oracle/gen_golden.py and tests/test_oracle_vs_ref.py, and against the golden
 * fixtures committed under tests/golden/ (which carry the reference's own
 * verdicts and output hashes).  The reference ships no known-answer vectors of
 * its own; the 22-byte empty archive literal of
 * test/unit/test_empty_zip.c:10-19 is the only byte fixture it holds and it is
 * covered by tests/test_zip_api.py.
 *
 * All "ref:" citations are relative to /root/reference/.
 */
#ifndef OTZ_ORACLE_H
#define OTZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* zlib-style return codes (ref: src/lib/deflate.inc.c:35-43) */
#define OTZ_Z_OK 0
#define OTZ_Z_STREAM_END 1
#define OTZ_Z_STREAM_ERROR (-2)
#define OTZ_Z_DATA_ERROR (-3)
#define OTZ_Z_MEM_ERROR (-4)
#define OTZ_Z_BUF_ERROR (-5)

/* ZIP method ids (ref: src/include/otezip/config.h:28-35) */
#define OTZ_M_STORE 0
#define OTZ_M_DEFLATE 8
#define OTZ_M_LZMA 14
#define OTZ_M_ZSTD 93
#define OTZ_M_BROTLI 97
#define OTZ_M_LZFSE 100

/* Verdict bits reported next to the reference-faithful return code. */
#define OTZ_V_REF_FALSE_REJECT 1u /* stream is valid and fully decoded, but the
                                     reference returns Z_BUF_ERROR */
#define OTZ_V_REF_UNDEFINED 2u    /* the reference reads indeterminate memory /
                                     indexes out of bounds here: no parity
                                     claimed, oracle reports Z_DATA_ERROR */
#define OTZ_V_SHORT_STREAM 4u     /* accepted with total_out < uncomp_size:
                                     zero-filled tail */
#define OTZ_V_CRC_MISMATCH 8u     /* computed CRC != directory CRC */
#define OTZ_V_OVERSUBSCRIBED 16u  /* a Huffman code set in the stream violates the Kraft inequality:
                                     malformed input that the reference still "decodes" through its
                                     first-match symbol search (ref deflate-dec.inc.c:86-119 has no
                                     check); only accept/reject is comparable (parity rule 4) */

typedef struct {
	int ret;            /* what the reference codec returns (Z_* code) */
	int correct_ret;    /* same, with the F2 false-reject repaired: what an
	                       RFC 1951-correct inflate returns on this stream */
	uint32_t total_in;  /* reference's strm.total_in at return */
	uint32_t total_out; /* bytes produced */
	uint32_t verdict;   /* OTZ_V_* bits */
} otz_codec_result;

/* ref: src/lib/crc32.inc.c:40-47 */
uint32_t otz_crc32(uint32_t crc, const void *buf, size_t len);

/* ref: src/lib/deflate-dec.inc.c:452-518 (inflateInit2 with windowBits=-15),
 * :547-831 (inflate, single call, flush=Z_FINISH) */
otz_codec_result otz_inflate_raw(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap);

/* ref: src/lib/zstd.inc.c:479-705 (single call, Z_FINISH) */
otz_codec_result otz_zstd_decode(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap);
/* ref: src/lib/lzfse.inc.c:225-257, :314-338 */
otz_codec_result otz_lzfse_decode(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap);
/* ref: src/lib/brotli.inc.c:57-80, :152-180 */
otz_codec_result otz_brotli_decode(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap);
/* ref: src/lib/lzma-dec.inc.c:102-155, :216-308 */
otz_codec_result otz_lzma_decode(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap);

/* Per-entry decode driver, from the point where the compressed bytes are in
 * memory: ref src/lib/otezip.c:454-462 (zip-bomb guard), :479-684 (method
 * switch + CRC).  out must hold uncomp_size bytes (>=1 allocated).
 * Returns 0 (entry readable, out[0..uncomp_size) valid) or -1 (reference
 * returns NULL from zip_fopen_index).
 * correct_rc receives the same with parity rule 2 applied (F2 repaired).
 * strict_crc mirrors the global otezip_verify_crc; ignore_zipbomb mirrors
 * otezip_ignore_zipbomb; ratio/slack the two expansion globals. */
int otz_extract_entry(uint16_t method, const uint8_t *comp, uint32_t comp_size, uint32_t uncomp_size, uint32_t dir_crc,
	int strict_crc, int ignore_zipbomb, uint64_t max_ratio, uint64_t max_slack,
	uint8_t *out, uint32_t *out_crc, uint32_t *verdict, int *correct_rc);

/* ---- encode side (ref: src/lib/otezip.c:788-1076 and the codecs' writers) ---- */

/* Bound the reference allocates per method (ref: src/lib/otezip.c:819, :860,
 * :931, :975, :1019; compressBound src/lib/deflate.inc.c:72-77). 0 = unknown method */
uint64_t otz_compress_bound(uint16_t method, uint64_t in_size);

/* Private framings the reference emits; byte-exact restatements.
 * Each returns bytes written, or 0 when the reference codec fails. */
/* ref: src/lib/zstd.inc.c:254-414 driven as src/lib/otezip.c:878-881 */
uint64_t otz_zstd_encode(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap);
/* ref: src/lib/lzfse.inc.c:179-219 */
uint64_t otz_lzfse_encode(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap);
/* ref: src/lib/brotli.inc.c:41-55 */
uint64_t otz_brotli_encode(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap);
/* ref: src/lib/lzma-enc.inc.c:110-161, :166-322 */
uint64_t otz_lzma_encode(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap);

/* otezip_compress_data for methods 93/100/97/14 (ref src/lib/otezip.c:855-1071): payload bytes
 * the reference writes and the method it ends up with after its "not smaller -> STORE" rule.
 * out must hold max(n, otz_compress_bound(method, n)) bytes. Returns the payload size. */
uint64_t otz_compress_data_private(uint16_t method, const uint8_t *in, uint64_t n, uint8_t *out, uint16_t *method_out);

/* The reference's DEFLATE encoder emits undecodable bytes; only its output SIZE is
 * comparable (parity rule 5).  Those sizes are taken from the reference build itself and
 * committed as fixtures (tests/golden/entries.json "ref_deflate_sizes"); the encoder is not
 * restated here. */

/* Zstandard per RFC 8878 (otz_zstd_rfc.c): every frame of src (skippable frames skipped, content
 * checksums verified).  Returns 0, -3 (corrupt / dictionary) or -5 (output does not fit);
 * *flags bit 0: a frame carried a content checksum.  No reference code exists for this
 * (src/lib/zstd.inc.c:172-182, :689-692); pinned against libzstd 1.4.8 of the image. */
int otz_zstd_rfc_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len, uint32_t *flags);
uint64_t otz_xxh64(const uint8_t *p, size_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
