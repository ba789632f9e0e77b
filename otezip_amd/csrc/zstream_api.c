/* zstream_api.c -- the codec seam ("Seam B"): every codec entry point the reference exports
 * over the zlib-shaped z_stream record, 50 symbols --
 *   inflateInit2[_] / inflate / inflateEnd, deflateInit2[_] / deflate / deflateEnd
 *     (ref src/lib/deflate.inc.c:89-97, src/include/otezip/zlib.h:91-106; decoder
 *     src/lib/deflate-dec.inc.c:547-843, encoder src/lib/deflate-enc.inc.c:262-541),
 *   {zstd,lzfse,brotli,lzma}{Init,Compress,End,DecompressInit,Decompress,DecompressEnd,
 *     CompressInit2[_],DecompressInit2[_]} (ref zstd.inc.c:135-148, lzfse.inc.c:272-372,
 *     brotli.inc.c:82-214, lzma-enc.inc.c:92-98, lzma-dec.inc.c:84-90),
 *   lzfse_compress / lzfse_decompress (ref lzfse.inc.c:46-47).
 *
 * The reference's own callers (otezip_extract_entry, otezip_compress_data, ref
 * src/lib/otezip.c:503-522, :819-1062) hand a codec the whole input and the whole output
 * buffer in one call.  That is the contract implemented here: ONE call of inflate() /
 * deflate() / <codec>Compress() / <codec>Decompress() runs the GPU batch path (a batch of one)
 * over next_in[0, avail_in) into next_out[0, avail_out).  There is no CPU codec behind these
 * symbols: without a GPU they return Z_STREAM_ERROR.  The only byte work on the host is framing:
 * zlib / gzip headers and trailers, and reading a container's length field(s) so that the batch
 * entry carries the exact size.
 *
 * inflate() follows the reference in what a caller can see: windowBits selects raw / zlib / gzip /
 * auto-detected framing and the header is skipped the way the reference skips it (ref
 * deflate-dec.inc.c:361-443, :452-477, :555-584); at Z_STREAM_END total_in counts exactly the bytes
 * the reference's bit reader would have loaded (the kernels report the bit position behind the
 * final end-of-block code); a stream that produces more than avail_out returns Z_OK with the
 * buffer FULL (ref :521-545, :693-760, :805-830), the next call with no room returns Z_BUF_ERROR,
 * and once the caller has made room (the reference's own CLI doubles its buffer, src/main.c:612-650)
 * the stream goes on where the output stopped.  The batch decoder has no resumable state, so "goes
 * on" is: the stream is decoded again from its first byte into a buffer that holds everything
 * produced so far plus the new room, and the new part is handed out.  That needs the caller's input
 * buffer to stay in place between the calls, which it does in every caller of the reference.
 *
 * Beyond the reference (documented extensions):
 *  - inflate verifies a zlib (Adler-32) or gzip (CRC-32, ISIZE) trailer when the input holds it, and
 *    counts it in total_in; a wrong trailer is Z_DATA_ERROR.  The reference never reads trailers.
 *  - deflateInit2 honours windowBits like zlib: -8..-15 raw, 8..15 zlib framing, 24..31 gzip framing
 *    (the reference refuses everything above 15, deflate-enc.inc.c:207-211, which is why its own
 *    CLI's -g always fails).  The checksums are taken on the device.
 *
 * Differences from the reference that a caller can observe:
 *  - on an error return (Z_DATA_ERROR, Z_BUF_ERROR) nothing is consumed and nothing written;
 *  - a stream fed in pieces (input arriving across calls) is not supported: the first call needs the
 *    whole compressed stream;
 *  - inflate: parity rules of DESIGN.md apply (the reference's premature Z_BUF_ERROR on
 *    some valid streams is repaired; malformed input fails cleanly);
 *  - deflate: the level is ignored and the output is VALID DEFLATE, dynamic-Huffman blocks (the
 *    reference's is not valid); only sizes are comparable;
 *  - zstdCompress finishes in one call (the reference wants to be called until it stops
 *    returning Z_OK, ref otezip.c:878-881); the bytes are the same;
 *  - zstdDecompress also accepts Zstandard frames per RFC 8878 that carry their content size.
 * The container writers are byte-exact, including the reference encoder's own magic byte order
 * for method 93 (which its decoder rejects). */
#include <stdlib.h>
#include <string.h>

#include "../../include/otezip/zstream.h"
#include "../../include/otezip_gpu.h"

enum { OZS_WRAP_NONE, OZS_WRAP_ZLIB, OZS_WRAP_GZIP, OZS_WRAP_AUTO };

struct ozs_state {
	int decode;
	int finished;
	int method;
	int wrap;        /* inflate: framing selected by windowBits */
	int header_done;
	int wrap_seen;   /* inflate, after the header: OZS_WRAP_ZLIB / _GZIP / _NONE (what AUTO turned out to be) */
	int stalled;     /* inflate returned Z_OK with a full buffer: the next call with room decodes again */
	const unsigned char *in0; /* inflate: first byte of the DEFLATE stream (behind the header) */
	uLong in0_len;            /* bytes the caller had from there on */
	uLong in0_used;           /* of those, what next_in has been advanced over */
	uLong out_made;           /* output bytes handed out since in0 */
};

static int ozs_device(void) {
	const char *d = getenv ("OTEZIP_GPU_DEVICE");
	return d ? atoi (d) : 0;
}

static int ozs_init(z_stream *strm, int decode, int method) {
	if (!strm) {
		return Z_STREAM_ERROR;
	}
	struct ozs_state *st = (struct ozs_state *)calloc (1, sizeof (*st));
	if (!st) {
		return Z_MEM_ERROR;
	}
	st->decode = decode;
	st->method = method;
	strm->state = st;
	strm->total_in = strm->total_out = 0;
	strm->msg = NULL;
	return Z_OK;
}

static int ozs_end(z_stream *strm) {
	if (!strm || !strm->state) {
		return Z_STREAM_ERROR;
	}
	free (strm->state);
	strm->state = NULL;
	return Z_OK;
}

int inflateInit2(z_stream *strm, int windowBits) {
	/* ref deflate-dec.inc.c:452-477: < 0 raw, 8..15 zlib, +16 gzip, +32 auto-detect (the window
	 * size itself does not matter here: the kernels always keep the full 32 KiB) */
	int rc = ozs_init (strm, 1, OZG_METHOD_DEFLATE);
	if (rc == Z_OK) {
		struct ozs_state *st = (struct ozs_state *)strm->state;
		st->wrap = windowBits < 0 ? OZS_WRAP_NONE : windowBits >= 40 ? OZS_WRAP_AUTO : windowBits >= 24 ? OZS_WRAP_GZIP : OZS_WRAP_ZLIB;
	}
	return rc;
}

/* ref skip_gzip_header, deflate-dec.inc.c:361-417: bytes to skip, or -1 */
static int ozs_skip_gzip(const unsigned char *buf, size_t len) {
	if (len < 10 || buf[0] != 0x1f || buf[1] != 0x8b || buf[2] != 8) {
		return -1;
	}
	const unsigned flags = buf[3];
	size_t pos = 10;
	if (flags & 0x04) { /* FEXTRA */
		if (pos + 2 > len) {
			return -1;
		}
		pos += 2u + (buf[pos] | ((size_t)buf[pos + 1] << 8));
		if (pos > len) {
			return -1;
		}
	}
	for (unsigned f = 0x08; f <= 0x10; f <<= 1) { /* FNAME, FCOMMENT: zero-terminated */
		if (flags & f) {
			while (pos < len && buf[pos] != 0) {
				pos++;
			}
			if (pos >= len) {
				return -1;
			}
			pos++;
		}
	}
	if (flags & 0x02) { /* FHCRC */
		pos += 2;
		if (pos > len) {
			return -1;
		}
	}
	return (int)pos;
}

/* ref skip_zlib_header, :420-443 */
static int ozs_skip_zlib(const unsigned char *buf, size_t len) {
	if (len < 2 || (buf[0] & 0x0f) != 8 || (buf[0] * 256u + buf[1]) % 31u != 0) {
		return -1;
	}
	size_t pos = 2;
	if (buf[1] & 0x20) { /* preset dictionary id */
		pos += 4;
		if (pos > len) {
			return -1;
		}
	}
	return (int)pos;
}

int inflateInit2_(z_stream *strm, int windowBits, const char *version, int stream_size) {
	(void)version;
	(void)stream_size;
	return inflateInit2 (strm, windowBits);
}

int inflateEnd(z_stream *strm) {
	return ozs_end (strm);
}

static void ozs_advance(z_stream *strm, uInt in_used, uInt out_made) {
	strm->next_in += in_used;
	strm->avail_in -= in_used;
	strm->total_in += in_used;
	strm->next_out += out_made;
	strm->avail_out -= out_made;
	strm->total_out += out_made;
}

/* Payload length a private container announces, read on the host so that the batch entry can
 * carry the exact output size (the batch path is built for zip entries, whose size is known):
 * -1 = malformed / cannot tell (the GPU decoder then reports the error itself). */
static long long ozs_container_len(int method, const unsigned char *p, uInt n) {
	unsigned long long v = 0;
	if (method == OZG_METHOD_LZFSE) { /* 0x06, LE32 length (ref lzfse.inc.c:225-257) */
		if (n < 5) {
			return -1;
		}
		return (long long)((unsigned long long)p[1] | ((unsigned long long)p[2] << 8) | ((unsigned long long)p[3] << 16) | ((unsigned long long)p[4] << 24));
	}
	if (method == OZG_METHOD_BROTLI) { /* LE64 at 5 (ref brotli.inc.c:57-80) */
		if (n < 13) {
			return -1;
		}
		for (int i = 7; i >= 0; i--) {
			v = (v << 8) | p[5 + i];
		}
		return v > 0x7FFFFFFFull ? -1 : (long long)v;
	}
	if (method == OZG_METHOD_ZSTD) { /* magic, descriptor, [type|last, LE16 length, payload]* (ref zstd.inc.c:479-705) */
		uInt at = 5;
		if (n < 5) {
			return -1;
		}
		for (;;) {
			if (n - at < 3) {
				return -1;
			}
			unsigned b0 = p[at], len = p[at + 1] | ((unsigned)p[at + 2] << 8);
			at += 3;
			if (((b0 >> 1) & 3u) == 1u || ((b0 >> 1) & 3u) == 3u) {
				return -1;
			}
			if (n - at < len) {
				return -1;
			}
			v += len;
			at += len;
			if (b0 & 1u) {
				return (long long)v;
			}
		}
	}
	return -1;
}

/* a Zstandard frame's content size, when the header carries it (RFC 8878 3.1.1.1); -1 otherwise */
static long long ozs_zstd_content_size(const unsigned char *p, uInt n) {
	if (n < 6 || p[0] != 0x28 || p[1] != 0xB5 || p[2] != 0x2F || p[3] != 0xFD || (p[4] & 0x18)) {
		return -1; /* (bit 4 set: the reference's private framing; bit 3: reserved) */
	}
	const unsigned fhd = p[4], single = (fhd >> 5) & 1u, flag = fhd >> 6, did = fhd & 3u;
	uInt at = 5 + (single ? 0u : 1u) + (did == 3u ? 4u : did);
	const unsigned len = flag == 0 ? single : flag == 1 ? 2u : flag == 2 ? 4u : 8u;
	if (len == 0 || at + len > n) {
		return -1;
	}
	unsigned long long v = 0;
	for (unsigned k = 0; k < len; k++) {
		v |= (unsigned long long)p[at + k] << (8 * k);
	}
	if (len == 2) {
		v += 256;
	}
	return v > 0x7FFFFFFFull ? -1 : (long long)v;
}

/* the trailer behind a framed DEFLATE stream, if the input holds it: 0 absent, > 0 its length (verified), -1 wrong */
static int ozs_check_trailer(const struct ozs_state *st, const unsigned char *t, uLong avail, const unsigned char *out, uLong n, uint32_t crc) {
	if (st->wrap_seen == OZS_WRAP_GZIP) {
		if (avail < 8) {
			return 0; /* truncated: the reference never looks */
		}
		const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
		const uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
		return want == crc && isize == (uint32_t)n ? 8 : -1;
	}
	if (st->wrap_seen == OZS_WRAP_ZLIB) {
		if (avail < 4) {
			return 0;
		}
		uint32_t ad = 0;
		if (ozg_checksum_host (1, out, n, &ad) != OZG_OK) {
			return -1;
		}
		const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | (uint32_t)t[3];
		return want == ad ? 4 : -1;
	}
	return 0;
}

static int ozs_decode(z_stream *strm) {
	if (!strm || !strm->state || !strm->next_out || (!strm->next_in && strm->avail_in)) {
		return Z_STREAM_ERROR;
	}
	struct ozs_state *st = (struct ozs_state *)strm->state;
	if (!st->decode) {
		return Z_STREAM_ERROR;
	}
	if (st->finished) {
		return Z_STREAM_END;
	}
	if (st->stalled && strm->avail_out == 0) {
		return Z_BUF_ERROR; /* no room, no progress (ref: the caller grows its buffer and calls again, main.c:612-650) */
	}
	if (st->method == OZG_METHOD_DEFLATE && !st->header_done) {
		/* ref deflate-dec.inc.c:555-584 */
		int skip = 0;
		st->wrap_seen = OZS_WRAP_NONE;
		if (st->wrap != OZS_WRAP_NONE) {
			if (st->wrap == OZS_WRAP_GZIP || (st->wrap == OZS_WRAP_AUTO && strm->avail_in >= 2 && strm->next_in[0] == 0x1f && strm->next_in[1] == 0x8b)) {
				skip = ozs_skip_gzip (strm->next_in, strm->avail_in);
				st->wrap_seen = OZS_WRAP_GZIP;
			} else {
				skip = ozs_skip_zlib (strm->next_in, strm->avail_in);
				st->wrap_seen = OZS_WRAP_ZLIB;
			}
			if (skip < 0) {
				return Z_DATA_ERROR;
			}
		}
		strm->next_in += skip;
		strm->avail_in -= (uInt)skip;
		strm->total_in += (uLong)skip;
		st->header_done = 1;
	}
	if (!st->in0) {
		st->in0 = strm->next_in;
		st->in0_len = strm->avail_in;
		st->in0_used = 0;
		st->out_made = 0;
	}
	if (ozg_init (ozs_device ()) != OZG_OK) {
		strm->msg = ozg_last_error ();
		return Z_STREAM_ERROR;
	}
	/* the stream is decoded from its first byte: everything handed out so far, plus the room there is now */
	const unsigned char *in = st->in0;
	const uLong in_len = st->in0_len;
	if (in_len > 0xFFFFFFFFul || st->out_made + (uLong)strm->avail_out > 0xFFFFFFF0ul) {
		return Z_MEM_ERROR;
	}
	uInt want = (uInt) (st->out_made + strm->avail_out);
	/* (method 14 needs no look-ahead: its token reader stops by itself and the batch kernel
	 * reports what it wrote even when that is not the size the entry announced) */
	/* A Brotli stream (RFC 7932) and an LZMA1 stream behind the APPNOTE header do not say how long their
	 * output is: the buffer is the room, the stream's own end (for LZMA1: its end marker) is the end.  What looks
	 * like one is tried as one first; whatever that does not decode takes the reference's road below, with the
	 * reference's verdicts (its readers do not look at the magic bytes: ref brotli.inc.c:154-171, lzma-dec.inc.c:216-232). */
	const int real_brotli = st->method == OZG_METHOD_BROTLI && in_len >= 1 && !(in_len >= 4 && memcmp (in, "BROT", 4) == 0);
	const int real_lzma = st->method == OZG_METHOD_LZMA && in_len >= 14 && in[2] == 5 && in[3] == 0;
	const int real_lzfse = st->method == OZG_METHOD_LZFSE && in_len >= 4 && memcmp (in, "bvx", 3) == 0; /* (its end block says where it ends) */
	int real_needs_room = 0;
	if ((real_brotli || real_lzma || real_lzfse) && st->out_made == 0) {
		ozg_entry e;
		memset (&e, 0, sizeof (e));
		e.comp_size = (uint32_t)in_len;
		e.uncomp_size = strm->avail_out;
		e.method = (uint16_t)st->method;
		e.flags = OZG_EF_IGNORE_ZIPBOMB | OZG_EF_SIZE_IS_CAP;
		ozg_result r;
		memset (&r, 0, sizeof (r));
		const uint64_t out_len = ((uint64_t)e.uncomp_size + 15u) & ~(uint64_t)15u;
		unsigned char *out = (unsigned char *)malloc (out_len ? out_len : 16);
		if (!out) {
			return Z_MEM_ERROR;
		}
		const int rc = ozg_decode_batch_host (in, in_len, out, out_len, &e, &r, 1);
		if (rc == OZG_OK && r.status == OZG_ST_OK && r.total_out <= strm->avail_out) {
			memcpy (strm->next_out, out, r.total_out);
			ozs_advance (strm, strm->avail_in, r.total_out);
			st->finished = 1;
			free (out);
			return Z_STREAM_END;
		}
		free (out);
		/* a stream of the real format that was valid as far as the room went: the caller's buffer is too small.  The
		 * reference's container reader has the first word (it accepts things by its own, looser rules); what it
		 * turns down is then answered Z_BUF_ERROR with nothing consumed or written, so that a caller that grows its
		 * buffer and calls again (the reference's CLI does: ref src/main.c:612-650) gets the stream then -- not the
		 * Z_DATA_ERROR the container reader has for a valid stream of the real format */
		real_needs_room = rc == OZG_OK && r.status == OZG_ST_SIZE;
	}
	if (st->method != OZG_METHOD_DEFLATE && st->method != OZG_METHOD_LZMA) {
		long long plen = st->method == OZG_METHOD_ZSTD ? ozs_zstd_content_size (in, (uInt)in_len) : -1;
		if (plen < 0) {
			plen = ozs_container_len (st->method, in, (uInt)in_len);
		}
		if (plen >= 0) {
			if ((unsigned long long)plen > strm->avail_out) {
				return Z_BUF_ERROR; /* the payload does not fit the caller's buffer */
			}
			want = (uInt)plen;
		}
	}
	ozg_entry e;
	memset (&e, 0, sizeof (e));
	e.comp_size = (uint32_t)in_len;
	e.uncomp_size = want;
	e.method = (uint16_t)st->method;
	e.flags = OZG_EF_IGNORE_ZIPBOMB; /* the ratio guard belongs to the archive reader (ref otezip.c:454-462) */
	ozg_result r;
	memset (&r, 0, sizeof (r));
	const unsigned char dummy = 0;
	uint64_t out_len = ((uint64_t)want + 15u) & ~(uint64_t)15u;
	unsigned char *out = (unsigned char *)malloc (out_len ? out_len : 16);
	if (!out) {
		return Z_MEM_ERROR;
	}
	uint64_t end_bit = 0;
	int rc = ozg_decode_batch_host_ex (in_len ? in : &dummy, in_len, out, out_len, &e, &r, 1, &end_bit);
	if (rc != OZG_OK) {
		free (out);
		strm->msg = ozg_last_error ();
		return rc == OZG_E_NOMEM ? Z_MEM_ERROR : Z_STREAM_ERROR;
	}
	/* input the codec consumed: for DEFLATE the bytes the reference's bit reader would have loaded */
	uLong used = in_len;
	if (st->method == OZG_METHOD_DEFLATE) {
		const uint64_t bytes = (end_bit + 7u) >> 3;
		used = bytes < in_len ? (uLong)bytes : in_len;
	}
	/* what this call adds: the caller's next_in sits in0_used bytes into the stream, out_made bytes are out */
	const uLong in_step = used > st->in0_used ? used - st->in0_used : 0;
	int ret;
	switch (r.status) {
	case OZG_ST_OK: {
		/* end of the stream; a short DEFLATE stream leaves the rest of the buffer untouched */
		int tl = 0;
		if (st->method == OZG_METHOD_DEFLATE) {
			uint32_t crc = r.crc32;
			if (st->wrap_seen == OZS_WRAP_GZIP && r.total_out != want && in_len - used >= 8 &&
				ozg_checksum_host (0, out, r.total_out, &crc) != OZG_OK) {
				/* (a stream shorter than the buffer: the kernel's CRC covers the zero-filled rest too) */
				ret = Z_STREAM_ERROR;
				break;
			}
			tl = ozs_check_trailer (st, in + used, in_len - used, out, r.total_out, crc);
			if (tl < 0) {
				ret = Z_DATA_ERROR;
				break;
			}
		}
		const uLong fresh = r.total_out > st->out_made ? r.total_out - st->out_made : 0;
		memcpy (strm->next_out, out + st->out_made, fresh);
		ozs_advance (strm, (uInt) (in_step + (uLong)tl), (uInt)fresh);
		st->finished = 1;
		ret = Z_STREAM_END;
		break;
	}
	case OZG_ST_NEED_OUT:
		if (st->method == OZG_METHOD_DEFLATE) {
			/* the stream wants more than there is room for: buffer full, Z_OK (ref :805-830) */
			const uLong fresh = r.total_out > st->out_made ? r.total_out - st->out_made : 0;
			memcpy (strm->next_out, out + st->out_made, fresh);
			ozs_advance (strm, (uInt)in_step, (uInt)fresh);
			st->in0_used += in_step;
			st->out_made += fresh;
			st->stalled = 1;
			ret = Z_OK;
			break;
		}
		ret = Z_BUF_ERROR;
		break;
	case OZG_ST_DATA:
		ret = Z_DATA_ERROR;
		break;
	case OZG_ST_SIZE:
		if (st->method == OZG_METHOD_LZMA && r.total_out <= want) {
			/* the token stream ended with fewer bytes than the caller's buffer holds: for a
			 * codec-level call that is simply the end of the stream (ref lzma-dec.inc.c:297-310) */
			memcpy (strm->next_out, out, r.total_out);
			ozs_advance (strm, strm->avail_in, r.total_out);
			st->finished = 1;
			ret = Z_STREAM_END;
			break;
		}
		ret = Z_BUF_ERROR;
		break;
	case OZG_ST_BUF:
		ret = Z_BUF_ERROR;
		break;
	default:
		ret = Z_STREAM_ERROR;
		break;
	}
	free (out);
	if (real_needs_room && ret != Z_STREAM_END && ret != Z_OK) {
		ret = Z_BUF_ERROR;
	}
	return ret;
}

int inflate(z_stream *strm, int flush) {
	(void)flush;
	return ozs_decode (strm);
}

int deflateInit2(z_stream *strm, int level, int method, int windowBits, int memLevel, int strategy) {
	(void)level; /* the reference's caller always ends up at its default level (ref otezip.c:826) */
	(void)memLevel;
	(void)strategy;
	if (method != 8) {
		return Z_STREAM_ERROR;
	}
	/* zlib's convention (the reference accepts -8..-15 and 8..15 and writes raw DEFLATE for both,
	 * ref deflate-enc.inc.c:207-215; +16 = gzip is what its CLI asks for, src/main.c:676) */
	int wrap;
	if (windowBits >= -15 && windowBits <= -8) {
		wrap = OZS_WRAP_NONE;
	} else if (windowBits >= 8 && windowBits <= 15) {
		/* zlib's meaning (RFC 1950 framing, Adler-32 in strm->adler).  The reference writes raw DEFLATE here
		 * too: a caller ported from it that frames the bytes itself (puts them into a ZIP entry, say) asks
		 * for -15 -- or sets OTEZIP_DEFLATE_RAW_8_15=1, which restores the reference's reading of 8..15 */
		const char *raw = getenv ("OTEZIP_DEFLATE_RAW_8_15");
		wrap = raw && raw[0] && raw[0] != '0' ? OZS_WRAP_NONE : OZS_WRAP_ZLIB;
	} else if (windowBits >= 24 && windowBits <= 31) {
		wrap = OZS_WRAP_GZIP;
	} else {
		return Z_STREAM_ERROR;
	}
	int rc = ozs_init (strm, 0, OZG_METHOD_DEFLATE);
	if (rc == Z_OK) {
		((struct ozs_state *)strm->state)->wrap = wrap;
	}
	return rc;
}

int deflateInit2_(z_stream *strm, int level, int method, int windowBits, int memLevel, int strategy, const char *version, int stream_size) {
	(void)version;
	(void)stream_size;
	return deflateInit2 (strm, level, method, windowBits, memLevel, strategy);
}

int deflateEnd(z_stream *strm) {
	return ozs_end (strm);
}

static int ozs_encode(z_stream *strm) {
	if (!strm || !strm->state || !strm->next_out || (!strm->next_in && strm->avail_in)) {
		return Z_STREAM_ERROR;
	}
	struct ozs_state *st = (struct ozs_state *)strm->state;
	if (st->decode) {
		return Z_STREAM_ERROR;
	}
	if (st->finished) {
		return Z_STREAM_END;
	}
	if (ozg_init (ozs_device ()) != OZG_OK) {
		strm->msg = ozg_last_error ();
		return Z_STREAM_ERROR;
	}
	/* framing of the DEFLATE seam (RFC 1950 / RFC 1952): header, raw stream, trailer */
	const int wrap = st->method == OZG_METHOD_DEFLATE ? st->wrap : OZS_WRAP_NONE;
	const uInt head = wrap == OZS_WRAP_GZIP ? 10u : wrap == OZS_WRAP_ZLIB ? 2u : 0u;
	const uInt tail = wrap == OZS_WRAP_GZIP ? 8u : wrap == OZS_WRAP_ZLIB ? 4u : 0u;
	if (strm->avail_out < head + tail) {
		return Z_BUF_ERROR;
	}
	ozg_enc_entry e;
	memset (&e, 0, sizeof (e));
	e.src_size = strm->avail_in;
	e.dst_cap = strm->avail_out - head - tail;
	e.method = (uint16_t)st->method;
	e.reserved = OZG_ENC_F_CODEC_ONLY | (wrap == OZS_WRAP_ZLIB ? OZG_ENC_F_ADLER32 : 0u);
	ozg_enc_result r;
	memset (&r, 0, sizeof (r));
	const unsigned char dummy = 0;
	uint64_t out_len = ((uint64_t)e.dst_cap + 15u) & ~(uint64_t)15u;
	unsigned char *out = (unsigned char *)malloc (out_len ? out_len : 16);
	if (!out) {
		return Z_MEM_ERROR;
	}
	int rc = ozg_encode_batch_host (strm->avail_in ? strm->next_in : &dummy, strm->avail_in, out, out_len, &e, &r, 1);
	if (rc != OZG_OK) {
		free (out);
		strm->msg = ozg_last_error ();
		return rc == OZG_E_NOMEM ? Z_MEM_ERROR : Z_STREAM_ERROR;
	}
	int ret;
	if (r.status == OZG_ST_OK) {
		unsigned char *o = strm->next_out;
		const uInt n_in = strm->avail_in;
		if (wrap == OZS_WRAP_GZIP) {
			static const unsigned char gz[10] = { 0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff }; /* no name, no time, unknown OS */
			memcpy (o, gz, 10);
		} else if (wrap == OZS_WRAP_ZLIB) {
			o[0] = 0x78; /* deflate, 32 KiB window */
			o[1] = 0x9c; /* default level; makes the pair a multiple of 31 */
		}
		memcpy (o + head, out, r.comp_size);
		unsigned char *t = o + head + r.comp_size;
		if (wrap == OZS_WRAP_GZIP) {
			for (int k = 0; k < 4; k++) {
				t[k] = (unsigned char) (r.crc32 >> (8 * k));
				t[4 + k] = (unsigned char) ((uint32_t)n_in >> (8 * k));
			}
		} else if (wrap == OZS_WRAP_ZLIB) {
			for (int k = 0; k < 4; k++) {
				t[k] = (unsigned char) (r.crc32 >> (8 * (3 - k))); /* (the Adler-32 of the input: OZG_ENC_F_ADLER32) */
			}
		}
		ozs_advance (strm, strm->avail_in, head + r.comp_size + tail);
		strm->adler = r.crc32; /* CRC-32 (Adler-32 for the zlib framing) of the input, as zlib reports it */
		st->finished = 1;
		ret = Z_STREAM_END;
	} else {
		ret = r.status == OZG_ST_BUF ? Z_BUF_ERROR : Z_STREAM_ERROR; /* avail_out too small: nothing is written */
	}
	free (out);
	return ret;
}

int deflate(z_stream *strm, int flush) {
	(void)flush;
	return ozs_encode (strm);
}

/* The four private container codecs behind the same calling convention (ref zstd.inc.c:135-148,
 * :206-476, :708-752; lzfse.inc.c:272-372; brotli.inc.c:82-214; lzma-enc.inc.c:92-98, :166-340;
 * lzma-dec.inc.c:84-90, :169-345): <c>Init / <c>Compress / <c>End, <c>DecompressInit /
 * <c>Decompress / <c>DecompressEnd, and the Init2 / Init2_ aliases that ignore their extra
 * arguments just as the reference's do. */
#define OZS_CODEC(name, METHOD) \
	int name##Init(z_stream *strm, int level) { (void)level; return ozs_init (strm, 0, METHOD); } \
	int name##Compress(z_stream *strm, int flush) { (void)flush; return ozs_encode (strm); } \
	int name##End(z_stream *strm) { return ozs_end (strm); } \
	int name##DecompressInit(z_stream *strm) { return ozs_init (strm, 1, METHOD); } \
	int name##Decompress(z_stream *strm, int flush) { (void)flush; return ozs_decode (strm); } \
	int name##DecompressEnd(z_stream *strm) { return ozs_end (strm); } \
	int name##CompressInit2(z_stream *strm, int level, int windowBits, int memLevel, int strategy) { \
		(void)windowBits; (void)memLevel; (void)strategy; return name##Init (strm, level); } \
	int name##CompressInit2_(z_stream *strm, int level, int windowBits, int memLevel, int strategy, const char *version, int stream_size) { \
		(void)windowBits; (void)memLevel; (void)strategy; (void)version; (void)stream_size; return name##Init (strm, level); } \
	int name##DecompressInit2(z_stream *strm, int windowBits) { (void)windowBits; return name##DecompressInit (strm); } \
	int name##DecompressInit2_(z_stream *strm, int windowBits, const char *version, int stream_size) { \
		(void)windowBits; (void)version; (void)stream_size; return name##DecompressInit (strm); }

OZS_CODEC (zstd, OZG_METHOD_ZSTD)
OZS_CODEC (lzfse, OZG_METHOD_LZFSE)
OZS_CODEC (brotli, OZG_METHOD_BROTLI)
OZS_CODEC (lzma, OZG_METHOD_LZMA)

/* The reference's buffer-to-buffer lzfse pair (ref src/lib/lzfse.inc.c:46-47, :196-257):
 * bytes written, 0 on failure. */
size_t lzfse_compress(const void *in, size_t in_sz, void *out, size_t out_cap) {
	z_stream s;
	memset (&s, 0, sizeof (s));
	if (in_sz > 0xFFFFFFFFu || out_cap > 0xFFFFFFFFu || lzfseInit (&s, 0) != Z_OK) {
		return 0;
	}
	s.next_in = (const Bytef *)in;
	s.avail_in = (uInt)in_sz;
	s.next_out = (Bytef *)out;
	s.avail_out = (uInt)out_cap;
	int rc = lzfseCompress (&s, Z_FINISH);
	size_t n = rc == Z_STREAM_END ? (size_t)s.total_out : 0;
	lzfseEnd (&s);
	return n;
}

size_t lzfse_decompress(const void *in, size_t in_sz, void *out, size_t out_cap) {
	z_stream s;
	memset (&s, 0, sizeof (s));
	if (in_sz > 0xFFFFFFFFu || out_cap > 0xFFFFFFFFu || lzfseDecompressInit (&s) != Z_OK) {
		return 0;
	}
	s.next_in = (const Bytef *)in;
	s.avail_in = (uInt)in_sz;
	s.next_out = (Bytef *)out;
	s.avail_out = (uInt)out_cap;
	int rc = lzfseDecompress (&s, Z_FINISH);
	size_t n = rc == Z_STREAM_END ? (size_t)s.total_out : 0;
	lzfseDecompressEnd (&s);
	return n;
}
