/* zstream_api.c -- the codec seam ("Seam B") for DEFLATE: inflateInit2 / inflate /
 * inflateEnd and deflateInit2 / deflate / deflateEnd over the zlib-shaped z_stream record
 * (ref src/lib/deflate.inc.c:89-97, src/include/otezip/zlib.h:91-106, decoder
 * src/lib/deflate-dec.inc.c:547-843, encoder src/lib/deflate-enc.inc.c:262-541).
 *
 * The reference's own callers (otezip_extract_entry, otezip_compress_data, ref
 * src/lib/otezip.c:503-522, :819-850) hand a codec the whole input and the whole output
 * buffer in one call.  That is the contract implemented here: ONE call of inflate() /
 * deflate() runs the GPU batch path (a batch of one) over next_in[0, avail_in) into
 * next_out[0, avail_out).  There is no CPU codec behind these symbols: without a GPU
 * they return Z_STREAM_ERROR.  A stream fed in pieces across several calls is not
 * supported (the second call reports Z_STREAM_END / Z_BUF_ERROR as noted below).
 *
 * Differences from the reference that a caller can observe:
 *  - total_in / next_in / avail_in: all of avail_in is reported consumed on success (the
 *    batch ABI does not return the position of the final end-of-block code);
 *  - inflate with too little output space: Z_BUF_ERROR and no output, where the reference
 *    returns Z_OK with the buffer filled (its callers treat both as failure, ref otezip.c:517);
 *  - inflate: parity rules of DESIGN.md apply (the reference's premature Z_BUF_ERROR on
 *    some valid streams is repaired; malformed input fails cleanly);
 *  - deflate: the level is ignored and the output is a VALID fixed-Huffman stream (the
 *    reference's is not); only sizes are comparable. */
#include <stdlib.h>
#include <string.h>

#include "../../include/otezip/zstream.h"
#include "../../include/otezip_gpu.h"

struct ozs_state {
	int decode;
	int finished;
};

static int ozs_device(void) {
	const char *d = getenv ("OTEZIP_GPU_DEVICE");
	return d ? atoi (d) : 0;
}

static int ozs_init(z_stream *strm, int decode) {
	if (!strm) {
		return Z_STREAM_ERROR;
	}
	struct ozs_state *st = (struct ozs_state *)calloc (1, sizeof (*st));
	if (!st) {
		return Z_MEM_ERROR;
	}
	st->decode = decode;
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
	(void)windowBits; /* raw DEFLATE, 32 KiB window, whatever is asked (ref :12-40 ignores it for raw) */
	return ozs_init (strm, 1);
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

int inflate(z_stream *strm, int flush) {
	(void)flush;
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
	if (ozg_init (ozs_device ()) != OZG_OK) {
		strm->msg = ozg_last_error ();
		return Z_STREAM_ERROR;
	}
	ozg_entry e;
	memset (&e, 0, sizeof (e));
	e.comp_size = strm->avail_in;
	e.uncomp_size = strm->avail_out;
	e.method = OZG_METHOD_DEFLATE;
	e.flags = OZG_EF_IGNORE_ZIPBOMB; /* the ratio guard belongs to the archive reader (ref otezip.c:454-462) */
	ozg_result r;
	memset (&r, 0, sizeof (r));
	const unsigned char dummy = 0;
	uint64_t out_len = ((uint64_t)strm->avail_out + 15u) & ~(uint64_t)15u;
	unsigned char *out = (unsigned char *)malloc (out_len ? out_len : 16);
	if (!out) {
		return Z_MEM_ERROR;
	}
	int rc = ozg_decode_batch_host (strm->avail_in ? strm->next_in : &dummy, strm->avail_in, out, out_len, &e, &r, 1);
	if (rc != OZG_OK) {
		free (out);
		strm->msg = ozg_last_error ();
		return rc == OZG_E_NOMEM ? Z_MEM_ERROR : Z_STREAM_ERROR;
	}
	int ret;
	switch (r.status) {
	case OZG_ST_OK:
		/* end of the final block; a short stream leaves the rest of the buffer untouched */
		memcpy (strm->next_out, out, r.total_out);
		ozs_advance (strm, strm->avail_in, r.total_out);
		st->finished = 1;
		ret = Z_STREAM_END;
		break;
	case OZG_ST_NEED_OUT:
		/* The stream wants more than avail_out bytes.  The reference fills the buffer and
		 * returns Z_OK; the batch path does not materialise a partial last round, so the
		 * one-shot seam reports the buffer as too small and writes nothing. */
		ret = Z_BUF_ERROR;
		break;
	case OZG_ST_DATA:
		ret = Z_DATA_ERROR;
		break;
	case OZG_ST_BUF:
		ret = Z_BUF_ERROR;
		break;
	default:
		ret = Z_STREAM_ERROR;
		break;
	}
	free (out);
	return ret;
}

int deflateInit2(z_stream *strm, int level, int method, int windowBits, int memLevel, int strategy) {
	(void)level; /* the reference's caller always ends up at its default level (ref otezip.c:826) */
	(void)windowBits;
	(void)memLevel;
	(void)strategy;
	if (method != 8) {
		return Z_STREAM_ERROR;
	}
	return ozs_init (strm, 0);
}

int deflateInit2_(z_stream *strm, int level, int method, int windowBits, int memLevel, int strategy, const char *version, int stream_size) {
	(void)version;
	(void)stream_size;
	return deflateInit2 (strm, level, method, windowBits, memLevel, strategy);
}

int deflateEnd(z_stream *strm) {
	return ozs_end (strm);
}

int deflate(z_stream *strm, int flush) {
	(void)flush;
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
	ozg_enc_entry e;
	memset (&e, 0, sizeof (e));
	e.src_size = strm->avail_in;
	e.dst_cap = strm->avail_out;
	e.method = OZG_METHOD_DEFLATE;
	e.reserved = OZG_ENC_F_CODEC_ONLY;
	ozg_enc_result r;
	memset (&r, 0, sizeof (r));
	const unsigned char dummy = 0;
	uint64_t out_len = ((uint64_t)strm->avail_out + 15u) & ~(uint64_t)15u;
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
		memcpy (strm->next_out, out, r.comp_size);
		ozs_advance (strm, strm->avail_in, r.comp_size);
		strm->adler = r.crc32; /* CRC-32 of the input, as the zip writer needs it */
		st->finished = 1;
		ret = Z_STREAM_END;
	} else {
		ret = r.status == OZG_ST_BUF ? Z_BUF_ERROR : Z_STREAM_ERROR; /* avail_out too small: nothing is written */
	}
	free (out);
	return ret;
}
