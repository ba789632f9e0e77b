/* otz_oracle.c -- CPU restatement of the otezip per-entry codec hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see otz_oracle.h).  Parity status: PINNED against
 * the reference built from /root/reference (oracle/_ref) and against
 * tests/golden/.
 *
 * This is a restatement, not a copy: the reference's inflate searches every
 * symbol for every bit (src/lib/deflate-dec.inc.c:671-691); here the same
 * "shortest length first, lowest symbol index first" rule is resolved through
 * a prefix-indexed map, and the z_stream plumbing is replaced by plain
 * cursors.  What is kept exactly is the observable behaviour: return codes,
 * total_in/total_out, the bytes, and the premature Z_BUF_ERROR of SURVEY F2.
 */
#include "otz_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ CRC-32 */

static uint32_t g_crc_tab[256];
static int g_crc_ready;

static void crc_init(void) {
	/* reflected 0xEDB88320, the table of ref src/lib/crc32.inc.c:3-37 */
	for (uint32_t n = 0; n < 256; n++) {
		uint32_t c = n;
		for (int k = 0; k < 8; k++) {
			c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
		}
		g_crc_tab[n] = c;
	}
	g_crc_ready = 1;
}

uint32_t otz_crc32(uint32_t crc, const void *buf, size_t len) {
	if (!g_crc_ready) {
		crc_init ();
	}
	const uint8_t *p = (const uint8_t *)buf;
	crc = ~crc;
	for (size_t i = 0; i < len; i++) {
		crc = g_crc_tab[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
	}
	return ~crc;
}

/* ----------------------------------------------------------------- inflate */

/* One canonical code set as the reference builds it
 * (ref src/lib/deflate-dec.inc.c:86-119): code values are assigned with
 * uint16_t wrap-around and no Kraft check; a symbol can only ever match when
 * its value fits its length.  map[(1<<len)|code] = lowest such symbol. */
typedef struct {
	int16_t map[65536];
} otz_codeset;

static int g_oversubscribed; /* set by codeset_build; read and cleared by otz_inflate_raw */

static int codeset_build(otz_codeset *cs, const uint8_t *lengths, int n) {
	uint16_t bl_count[16];
	uint16_t next_code[16];
	memset (bl_count, 0, sizeof (bl_count));
	for (int i = 0; i < n; i++) {
		if (lengths[i] > 15) {
			return -1;
		}
		if (lengths[i]) {
			bl_count[lengths[i]]++;
		}
	}
	uint16_t code = 0;
	next_code[0] = 0;
	long left = 1;
	for (int bits = 1; bits <= 15; bits++) {
		code = (uint16_t) ((code + bl_count[bits - 1]) << 1);
		next_code[bits] = code;
		left = (left << 1) - bl_count[bits];
		if (left < 0) {
			g_oversubscribed = 1;
			left = 0;
		}
	}
	memset (cs->map, 0xFF, sizeof (cs->map));
	/* ascending symbol order hands out the code values; the first symbol
	 * to claim a (length, value) pair keeps it */
	for (int i = 0; i < n; i++) {
		int l = lengths[i];
		if (!l) {
			continue;
		}
		uint16_t c = next_code[l]++;
		if (c < (1u << l)) {
			uint32_t key = (1u << l) | c;
			if (cs->map[key] < 0) {
				cs->map[key] = (int16_t)i;
			}
		}
	}
	return 0;
}

typedef struct {
	const uint8_t *in;
	uint32_t in_len;
	uint32_t in_pos; /* bytes handed to the bit buffer so far (= total_in) */
	uint32_t bitbuf;
	uint32_t bitcnt;
} otz_bits;

/* ref get_bit, src/lib/deflate-dec.inc.c:44-61: at most one byte buffered */
static int bit1(otz_bits *b) {
	if (b->bitcnt == 0) {
		if (b->in_pos >= b->in_len) {
			return -1;
		}
		b->bitbuf = b->in[b->in_pos++];
		b->bitcnt = 8;
	}
	int bit = (int) (b->bitbuf & 1);
	b->bitbuf >>= 1;
	b->bitcnt--;
	return bit;
}

/* ref get_bits, :64-83 (LSB-first) */
static int bitsn(otz_bits *b, int n) {
	int v = 0;
	for (int i = 0; i < n; i++) {
		int bit = bit1 (b);
		if (bit < 0) {
			return -1;
		}
		v |= bit << i;
	}
	return v;
}

/* ref :669-695 -- code bits arrive MSB-first */
static int decode_sym(otz_bits *b, const otz_codeset *cs) {
	uint32_t key = 1;
	for (int len = 1; len <= 15; len++) {
		int bit = bit1 (b);
		if (bit < 0) {
			return -1; /* out of input */
		}
		key = (key << 1) | (uint32_t)bit;
		int s = cs->map[key];
		if (s >= 0) {
			return s;
		}
	}
	return -2; /* no code of any length matches */
}

static const uint16_t k_len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t k_len_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t k_dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t k_dist_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
static const uint8_t k_cl_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

/* ref read_dynamic_huffman, :122-266.  *undef is set where the reference
 * itself runs off its arrays. */
static int read_dynamic(otz_bits *b, otz_codeset *lit, otz_codeset *dist, int *undef) {
	int hlit = bitsn (b, 5);
	if (hlit < 0) {
		return OTZ_Z_DATA_ERROR;
	}
	hlit += 257;
	int hdist = bitsn (b, 5);
	if (hdist < 0) {
		return OTZ_Z_DATA_ERROR;
	}
	hdist += 1;
	int hclen = bitsn (b, 4);
	if (hclen < 0) {
		return OTZ_Z_DATA_ERROR;
	}
	hclen += 4;
	uint8_t cl_len[19];
	memset (cl_len, 0, sizeof (cl_len));
	for (int i = 0; i < hclen; i++) {
		int l = bitsn (b, 3);
		if (l < 0) {
			return OTZ_Z_DATA_ERROR;
		}
		cl_len[k_cl_order[i]] = (uint8_t)l;
	}
	otz_codeset *cl = (otz_codeset *)malloc (sizeof (otz_codeset));
	if (!cl) {
		return OTZ_Z_MEM_ERROR;
	}
	codeset_build (cl, cl_len, 19);
	enum { LL_MAX = 286 + 32 };
	uint8_t ll[LL_MAX + 2];
	memset (ll, 0, sizeof (ll));
	int index = 0;
	int rc = OTZ_Z_OK;
	while (index < hlit + hdist && index < LL_MAX) {
		int code = decode_sym (b, cl);
		if (code < 0) {
			rc = OTZ_Z_DATA_ERROR;
			break;
		}
		if (code < 16) {
			ll[index++] = (uint8_t)code;
			continue;
		}
		uint8_t val = 0;
		int rep;
		if (code == 16) {
			if (index == 0) {
				rc = OTZ_Z_DATA_ERROR;
				break;
			}
			val = ll[index - 1];
			rep = bitsn (b, 2);
			if (rep < 0) {
				rc = OTZ_Z_DATA_ERROR;
				break;
			}
			rep += 3;
		} else if (code == 17) {
			rep = bitsn (b, 3);
			if (rep < 0) {
				rc = OTZ_Z_DATA_ERROR;
				break;
			}
			rep += 3;
		} else {
			rep = bitsn (b, 7);
			if (rep < 0) {
				rc = OTZ_Z_DATA_ERROR;
				break;
			}
			rep += 11;
		}
		if (index + rep > hlit + hdist || index + rep > LL_MAX) {
			rc = OTZ_Z_DATA_ERROR;
			break;
		}
		for (int i = 0; i < rep; i++) {
			ll[index++] = val;
		}
	}
	free (cl);
	if (rc != OTZ_Z_OK) {
		return rc;
	}
	if (hlit + hdist > LL_MAX) {
		/* reference builds the distance set from beyond ll_lengths[] */
		*undef = 1;
		return OTZ_Z_DATA_ERROR;
	}
	codeset_build (lit, ll, hlit);
	codeset_build (dist, ll + hlit, hdist);
	return OTZ_Z_OK;
}

otz_codec_result otz_inflate_raw(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap) {
	otz_codec_result r;
	memset (&r, 0, sizeof (r));
	otz_bits b = { in, in_len, 0, 0, 0 };
	uint32_t op = 0;
	int final_block = 0;
	int state = 0; /* 0 header, 1 block setup, 2 symbols -- ref :612-803 */
	int btype = 0;
	int ref_done = 0; /* reference has already returned */
	int undef = 0;
	int ret;
	g_oversubscribed = 0;
	otz_codeset *lit = (otz_codeset *)malloc (sizeof (otz_codeset));
	otz_codeset *dist = (otz_codeset *)malloc (sizeof (otz_codeset));
	if (!lit || !dist) {
		free (lit);
		free (dist);
		r.ret = r.correct_ret = OTZ_Z_MEM_ERROR;
		return r;
	}

#define REF_RETURNS(code) \
	do { \
		if (!ref_done) { \
			ref_done = 1; \
			r.ret = (code); \
			r.total_in = b.in_pos; \
			r.total_out = op; \
		} \
	} while (0)

	if (in_len == 0) {
		/* ref :610 loop never runs, :821-830 */
		ret = out_cap == 0 ? OTZ_Z_OK : OTZ_Z_BUF_ERROR;
		goto done;
	}
	for (;;) {
		/* a correct decoder may still hold bits when the last byte has been
		 * loaded; the reference's loop condition is avail_in > 0 (:610) */
		if (b.in_pos >= b.in_len && b.bitcnt == 0 && state != 1) {
			ret = OTZ_Z_BUF_ERROR;
			goto done;
		}
		if (state == 0) {
			final_block = bit1 (&b);
			if (final_block < 0) {
				ret = OTZ_Z_DATA_ERROR;
				goto done;
			}
			btype = bitsn (&b, 2);
			/* ref :620-623: an exhausted get_bits leaves btype = -1 which no
			 * case label matches; the next check then returns Z_BUF_ERROR */
			if (btype < 0) {
				ret = OTZ_Z_BUF_ERROR;
				goto done;
			}
			state = 1;
		} else if (state == 1) {
			if (btype == 0) {
				/* ref read_uncompressed_block, :269-319 */
				b.bitcnt = 0;
				if (b.in_len - b.in_pos < 4) {
					ret = OTZ_Z_DATA_ERROR;
					goto done;
				}
				uint16_t len = (uint16_t) (in[b.in_pos] | (in[b.in_pos + 1] << 8));
				uint16_t nlen = (uint16_t) (in[b.in_pos + 2] | (in[b.in_pos + 3] << 8));
				b.in_pos += 4;
				if (len != (uint16_t) (~nlen)) {
					ret = OTZ_Z_DATA_ERROR;
					goto done;
				}
				if (b.in_len - b.in_pos < len) {
					ret = OTZ_Z_DATA_ERROR;
					goto done;
				}
				if (out_cap - op < len) {
					ret = OTZ_Z_BUF_ERROR;
					goto done;
				}
				memcpy (out + op, in + b.in_pos, len);
				b.in_pos += len;
				op += len;
				state = 0;
			} else if (btype == 1) {
				/* ref init_fixed_huffman, :322-349 */
				uint8_t l[288];
				for (int i = 0; i < 288; i++) {
					l[i] = (uint8_t) (i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8);
				}
				codeset_build (lit, l, 288);
				memset (l, 5, 32);
				codeset_build (dist, l, 32);
				state = 2;
			} else if (btype == 2) {
				int rc = read_dynamic (&b, lit, dist, &undef);
				if (rc != OTZ_Z_OK) {
					ret = OTZ_Z_DATA_ERROR;
					goto done;
				}
				state = 2;
			} else {
				ret = OTZ_Z_DATA_ERROR;
				goto done;
			}
		} else {
			int sym = decode_sym (&b, lit);
			if (sym < 0) {
				ret = OTZ_Z_DATA_ERROR;
				goto done;
			}
			if (sym < 256) {
				if (op >= out_cap) {
					/* ref :700-703 pending literal, Z_OK */
					ret = OTZ_Z_OK;
					goto done;
				}
				out[op++] = (uint8_t)sym;
			} else if (sym == 256) {
				state = 0;
				if (final_block) {
					ret = OTZ_Z_STREAM_END;
					goto done;
				}
			} else if (sym <= 285) {
				int li = sym - 257;
				int length = k_len_base[li];
				if (k_len_extra[li]) {
					int e = bitsn (&b, k_len_extra[li]);
					if (e < 0) {
						ret = OTZ_Z_DATA_ERROR;
						goto done;
					}
					length += e;
				}
				int ds = decode_sym (&b, dist);
				if (ds < 0) {
					ret = OTZ_Z_DATA_ERROR;
					goto done;
				}
				if (ds >= 30) {
					/* ref :766-774 indexes dist_base[30..31] */
					undef = 1;
					ret = OTZ_Z_DATA_ERROR;
					goto done;
				}
				int distance = k_dist_base[ds];
				if (k_dist_extra[ds]) {
					int e = bitsn (&b, k_dist_extra[ds]);
					if (e < 0) {
						ret = OTZ_Z_DATA_ERROR;
						goto done;
					}
					distance += e;
				}
				if ((uint32_t)distance > op) {
					/* ref :524 reads window bytes nobody wrote (malloc'd,
					 * :493): indeterminate, no parity claimed */
					undef = 1;
					ret = OTZ_Z_DATA_ERROR;
					goto done;
				}
				/* ref do_copy_from_window, :521-545 */
				while (length > 0 && op < out_cap) {
					out[op] = out[op - (uint32_t)distance];
					op++;
					length--;
				}
				if (length > 0) {
					ret = OTZ_Z_OK; /* pending copy, needs output space */
					goto done;
				}
			} else {
				ret = OTZ_Z_DATA_ERROR;
				goto done;
			}
		}
		/* ref :805-816 */
		if (final_block && state == 0) {
			ret = OTZ_Z_STREAM_END;
			goto done;
		}
		if (b.in_pos >= b.in_len) {
			REF_RETURNS (OTZ_Z_BUF_ERROR);
		}
	}
done:
	free (lit);
	free (dist);
	r.correct_ret = ret;
	if (!ref_done) {
		r.ret = ret;
		r.total_in = b.in_pos;
		r.total_out = op;
	} else if (ret == OTZ_Z_STREAM_END) {
		r.verdict |= OTZ_V_REF_FALSE_REJECT;
		r.total_in = b.in_pos;
		r.total_out = op;
	}
	if (undef) {
		r.verdict |= OTZ_V_REF_UNDEFINED;
	}
	if (g_oversubscribed) {
		r.verdict |= OTZ_V_OVERSUBSCRIBED;
	}
	return r;
#undef REF_RETURNS
}

/* --------------------------------------------------- private "zstd" framing */

otz_codec_result otz_zstd_decode(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap) {
	otz_codec_result r;
	memset (&r, 0, sizeof (r));
	uint32_t ip = 0, op = 0;
	int ret;
	if (in_len < 5) {
		ret = OTZ_Z_BUF_ERROR; /* ref :490-492 */
		goto done;
	}
	uint32_t magic = (uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24);
	if (magic != 0xFD2FB528u) {
		ret = OTZ_Z_DATA_ERROR; /* ref :495-498 */
		goto done;
	}
	ip = 5;
	ret = OTZ_Z_OK; /* ref :704 when input runs dry without a last block */
	while (ip < in_len) {
		if (in_len - ip < 3) {
			ret = OTZ_Z_BUF_ERROR; /* ref :698-701 */
			break;
		}
		uint8_t h = in[ip];
		int last = h & 1;
		int type = (h >> 1) & 3;
		uint32_t bs = (uint32_t)in[ip + 1] | ((uint32_t)in[ip + 2] << 8);
		if (type == 0 || type == 2) {
			if (in_len - ip - 3 < bs) {
				ret = OTZ_Z_BUF_ERROR; /* ref :570-576, :635-641 (header un-read) */
				break;
			}
			if (type == 2 && bs == 0) {
				ip += 3;
				ret = OTZ_Z_DATA_ERROR; /* ref :189-191 via :647-649 */
				break;
			}
			ip += 3;
			if (out_cap - op >= bs) {
				memcpy (out + op, in + ip, bs);
				op += bs;
				ip += bs;
			} else {
				/* ref :608-632 / :676-683 then :513-548: the part that fits is
				 * handed out and the call returns Z_OK with avail_out == 0 */
				uint32_t fit = out_cap - op;
				memcpy (out + op, in + ip, fit);
				op += fit;
				if (type == 0) {
					ip += bs;
				}
				ret = OTZ_Z_OK;
				break;
			}
			if (last) {
				ret = OTZ_Z_STREAM_END;
				break;
			}
		} else {
			ip += 3;
			ret = OTZ_Z_DATA_ERROR; /* ref :689-692 */
			break;
		}
	}
done:
	r.ret = r.correct_ret = ret;
	r.total_in = ip;
	r.total_out = op;
	return r;
}

/* -------------------------------------------------- private "lzfse" framing */

otz_codec_result otz_lzfse_decode(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap) {
	otz_codec_result r;
	memset (&r, 0, sizeof (r));
	r.ret = r.correct_ret = OTZ_Z_BUF_ERROR; /* ref lzfse__map_result, :263-268 */
	if (in_len == 0 || out_cap == 0) {
		return r; /* ref :245-247 */
	}
	if (in_len < 5 || in[0] != 0x06) {
		return r; /* ref :226-228 */
	}
	uint32_t len = (uint32_t)in[1] | ((uint32_t)in[2] << 8) | ((uint32_t)in[3] << 16) | ((uint32_t)in[4] << 24);
	/* ref :231: "len + 5 > src_sz" is evaluated in uint32_t, so it wraps */
	if ((uint32_t) (len + 5u) > in_len || len > out_cap) {
		return r;
	}
	if ((uint64_t)len + 5u > in_len) {
		/* wrapped: the reference memcpy's far past its input */
		r.verdict |= OTZ_V_REF_UNDEFINED;
		return r;
	}
	if (len == 0) {
		return r; /* produced == 0 maps to Z_BUF_ERROR */
	}
	memcpy (out, in + 5, len);
	r.ret = r.correct_ret = OTZ_Z_STREAM_END;
	r.total_in = in_len;
	r.total_out = len;
	return r;
}

/* ------------------------------------------------- private "brotli" framing */

otz_codec_result otz_brotli_decode(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap) {
	otz_codec_result r;
	memset (&r, 0, sizeof (r));
	uint64_t stored = 0;
	int ok = 0;
	if (in_len >= 17 && memcmp (in, "BROT", 4) == 0) {
		for (int i = 0; i < 8; i++) {
			stored |= (uint64_t)in[5 + i] << (8 * i);
		}
		if (stored <= out_cap && (uint64_t)in_len >= 17u + stored) {
			uint32_t want = (uint32_t)in[13] | ((uint32_t)in[14] << 8) | ((uint32_t)in[15] << 16) | ((uint32_t)in[16] << 24);
			if (otz_crc32 (0, in + 17, (size_t)stored) == want) {
				ok = 1;
			}
		}
	}
	if (ok && stored > 0) {
		memcpy (out, in + 17, (size_t)stored);
		r.ret = r.correct_ret = OTZ_Z_STREAM_END;
		r.total_in = in_len;
		r.total_out = (uint32_t)stored;
		return r;
	}
	/* ref :158-171: anything 13+ bytes long whose length field is zero ends
	 * the stream with no output, whatever the magic says */
	if (in_len >= 13) {
		uint64_t l = 0;
		for (int i = 0; i < 8; i++) {
			l |= (uint64_t)in[5 + i] << (8 * i);
		}
		if (l == 0) {
			r.ret = r.correct_ret = OTZ_Z_STREAM_END;
			r.total_in = in_len;
			return r;
		}
	}
	r.ret = r.correct_ret = OTZ_Z_DATA_ERROR;
	return r;
}

/* ---------------------------------------------- private "lzma" token format */

otz_codec_result otz_lzma_decode(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap) {
	otz_codec_result r;
	memset (&r, 0, sizeof (r));
	if (in_len < 13) {
		r.ret = r.correct_ret = OTZ_Z_BUF_ERROR; /* ref :222-224 */
		return r;
	}
	uint64_t hdr = 0;
	for (int i = 0; i < 8; i++) {
		hdr |= (uint64_t)in[5 + i] << (8 * i);
	}
	r.total_in = 13;
	uint32_t total = 0;
	if (in_len > 13) {
		/* ref simple_lzma_decompress, :102-155, into a 64 KiB staging buffer (:181, :193) */
		enum { CAP = 1 << 16 };
		uint8_t *stage = (uint8_t *)malloc (CAP);
		if (!stage) {
			r.ret = r.correct_ret = OTZ_Z_MEM_ERROR;
			return r;
		}
		const uint8_t *src = in + 13;
		size_t n = in_len - 13, sp = 0, dp = 0;
		int bad = 0;
		while (sp < n) {
			if (sp + 2 >= n) {
				bad = 1;
				break;
			}
			uint8_t marker = src[sp++];
			if (marker == 0x00) {
				uint8_t byte = src[sp++];
				uint8_t len = src[sp++];
				if (dp + len > CAP) {
					bad = 1;
					break;
				}
				memset (stage + dp, byte, len);
				dp += len;
			} else if (marker == 0x01) {
				uint8_t len = src[sp++];
				if (sp + len > n || dp + len > CAP) {
					bad = 1;
					break;
				}
				memcpy (stage + dp, src + sp, len);
				dp += len;
				sp += len;
			} else {
				bad = 1;
				break;
			}
		}
		if (bad || dp == 0) {
			free (stage);
			r.ret = r.correct_ret = OTZ_Z_DATA_ERROR; /* ref :271-273 */
			return r;
		}
		r.total_in = in_len;
		if (out_cap >= dp) {
			memcpy (out, stage, dp);
			total = (uint32_t)dp;
		} else {
			memcpy (out, stage, out_cap); /* ref :286-295 */
			free (stage);
			r.total_out = out_cap;
			r.ret = r.correct_ret = OTZ_Z_OK;
			return r;
		}
		free (stage);
	}
	r.total_out = total;
	/* ref :297-310 */
	if (hdr != 0xFFFFFFFFFFFFFFFFull) {
		r.ret = (uint64_t)total >= hdr ? OTZ_Z_STREAM_END : OTZ_Z_OK;
	} else {
		r.ret = OTZ_Z_STREAM_END;
	}
	r.correct_ret = r.ret;
	return r;
}

/* ------------------------------------------------------------ entry driver */

int otz_extract_entry(uint16_t method, const uint8_t *comp, uint32_t comp_size, uint32_t uncomp_size, uint32_t dir_crc,
	int strict_crc, int ignore_zipbomb, uint64_t max_ratio, uint64_t max_slack,
	uint8_t *out, uint32_t *out_crc, uint32_t *verdict, int *correct_rc) {
	uint32_t v = 0;
	int rc = 0, crc_rc = 0;
	if (out_crc) {
		*out_crc = 0;
	}
	/* ref src/lib/otezip.c:454-462 */
	if (!ignore_zipbomb && comp_size > 0) {
		uint64_t allowed = (uint64_t)comp_size * max_ratio + max_slack;
		if ((uint64_t)uncomp_size > allowed) {
			rc = crc_rc = -1;
			goto done;
		}
	}
	if (method == OTZ_M_STORE) {
		/* ref :481-487 */
		if (comp_size != uncomp_size) {
			rc = crc_rc = -1;
			goto done;
		}
		memcpy (out, comp, uncomp_size);
	} else {
		memset (out, 0, uncomp_size); /* ref :500, :541, :570, :611, :640 */
		otz_codec_result r;
		switch (method) {
		case OTZ_M_DEFLATE:
			r = otz_inflate_raw (comp, comp_size, out, uncomp_size);
			v |= r.verdict;
			/* ref :525: no total_out check on this branch */
			rc = r.ret == OTZ_Z_STREAM_END ? 0 : -1;
			crc_rc = r.correct_ret == OTZ_Z_STREAM_END ? 0 : -1;
			if (crc_rc == 0 && r.total_out < uncomp_size) {
				v |= OTZ_V_SHORT_STREAM;
			}
			break;
		case OTZ_M_ZSTD:
			r = otz_zstd_decode (comp, comp_size, out, uncomp_size);
			goto sized;
		case OTZ_M_LZFSE:
			r = otz_lzfse_decode (comp, comp_size, out, uncomp_size);
			goto sized;
		case OTZ_M_LZMA:
			r = otz_lzma_decode (comp, comp_size, out, uncomp_size);
			goto sized;
		case OTZ_M_BROTLI:
			r = otz_brotli_decode (comp, comp_size, out, uncomp_size);
		sized:
			v |= r.verdict;
			/* ref :555, :584, :625, :654 */
			rc = crc_rc = (r.ret == OTZ_Z_STREAM_END && r.total_out == uncomp_size) ? 0 : -1;
			break;
		default:
			rc = crc_rc = -1; /* ref :662-665 */
			break;
		}
	}
	if (crc_rc == 0) {
		/* ref :667-679 */
		uint32_t c = otz_crc32 (0, out, uncomp_size);
		if (out_crc) {
			*out_crc = c;
		}
		if (c != dir_crc) {
			v |= OTZ_V_CRC_MISMATCH;
			if (strict_crc) {
				rc = crc_rc = -1;
			}
		}
	}
done:
	if (verdict) {
		*verdict = v;
	}
	if (correct_rc) {
		*correct_rc = crc_rc;
	}
	return rc;
}

/* -------------------------------------------------------------- encode side */

uint64_t otz_compress_bound(uint16_t method, uint64_t n) {
	switch (method) {
	case OTZ_M_DEFLATE: return n + (n >> 3) + 64u + 11u; /* ref deflate.inc.c:72-77 */
	case OTZ_M_ZSTD: return n + 1024u;                   /* ref otezip.c:860 */
	case OTZ_M_LZFSE: return n + n / 10u + 100u;         /* :931 */
	case OTZ_M_LZMA: return n + 13u + n / 8u;            /* :975 */
	case OTZ_M_BROTLI: return 2u * n + 64u;              /* :1019 */
	case OTZ_M_STORE: return n;
	default: return 0;
	}
}

/* ref zstd.inc.c:254-414 driven by the loop at otezip.c:878-881 */
uint64_t otz_zstd_encode(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap) {
	uint64_t op = 0, ip = 0;
	if (cap < 5) {
		return 0;
	}
	static const uint8_t hdr[5] = { 0xFD, 0x2F, 0xB5, 0x28, 0x70 };
	memcpy (out, hdr, 5);
	op = 5;
	while (ip < n) {
		uint64_t bs = n - ip < 65535u ? n - ip : 65535u;
		if (cap - op < 3 || cap - op - 3 < bs) {
			return 0; /* Z_BUF_ERROR: the driver falls back to STORE */
		}
		out[op++] = 0x00;
		out[op++] = (uint8_t)bs;
		out[op++] = (uint8_t) (bs >> 8);
		memcpy (out + op, in + ip, bs);
		op += bs;
		ip += bs;
	}
	if (cap - op < 3) {
		return 0;
	}
	out[op++] = 0x01;
	out[op++] = 0;
	out[op++] = 0;
	return op;
}

/* ref lzfse.inc.c:179-219 */
uint64_t otz_lzfse_encode(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap) {
	if (n + 5 > cap) {
		return 0;
	}
	out[0] = 0x06;
	out[1] = (uint8_t)n;
	out[2] = (uint8_t) (n >> 8);
	out[3] = (uint8_t) (n >> 16);
	out[4] = (uint8_t) (n >> 24);
	memcpy (out + 5, in, n);
	return n + 5;
}

/* ref brotli.inc.c:41-55 */
uint64_t otz_brotli_encode(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap) {
	if (cap < 17 || n > cap - 17) {
		return 0;
	}
	memcpy (out, "BROT", 4);
	out[4] = 1;
	for (int i = 0; i < 8; i++) {
		out[5 + i] = (uint8_t) (n >> (8 * i));
	}
	uint32_t c = otz_crc32 (0, in, n);
	for (int i = 0; i < 4; i++) {
		out[13 + i] = (uint8_t) (c >> (8 * i));
	}
	memcpy (out + 17, in, n);
	return n + 17;
}

/* ref lzma-enc.inc.c:110-161 (token writer) inside lzmaCompress :220-300 (13-byte header, 64 KiB
 * chunks, the "chunk + 32 bytes must still fit" test before every chunk). Returns 0 where the
 * reference returns an error (the driver then stores the entry, otezip.c:992-1001). */
uint64_t otz_lzma_encode(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap) {
	if (cap < 13) {
		return 0;
	}
	static const uint8_t props[5] = { 0x5D, 0x00, 0x00, 0x00, 0x01 };
	memcpy (out, props, 5);
	for (int i = 0; i < 8; i++) {
		out[5 + i] = (uint8_t) (n >> (8 * i));
	}
	uint64_t op = 13, ip = 0;
	while (ip < n) {
		uint64_t chunk = n - ip < 65536u ? n - ip : 65536u;
		if (cap - op < chunk + 32u) {
			return 0; /* Z_BUF_ERROR, ref :277-280 */
		}
		/* tokens of this chunk: runs of one byte, at most 255 long, never across chunks */
		uint64_t sp = 0;
		uint64_t start = op;
		while (sp < chunk) {
			uint8_t b = in[ip + sp];
			uint64_t run = 1;
			while (sp + run < chunk && in[ip + sp + run] == b && run < 255) {
				run++;
			}
			uint64_t need = run >= 4 ? 3 : 2 + run;
			if (cap - op < need) {
				return 0; /* compressed_size > avail_out, ref :288-290 */
			}
			if (run >= 4) {
				out[op++] = 0x00;
				out[op++] = b;
				out[op++] = (uint8_t)run;
			} else {
				out[op++] = 0x01;
				out[op++] = (uint8_t)run;
				memset (out + op, b, run);
				op += run;
			}
			sp += run;
		}
		(void)start;
		ip += chunk;
	}
	return op;
}

/* otezip_compress_data for the byte-adding/private methods (ref otezip.c:855-1071): the
 * payload the reference writes and the method it ends up with.  Returns the payload size and
 * sets *method_out; DEFLATE is not covered (its reference bytes are invalid). */
uint64_t otz_compress_data_private(uint16_t method, const uint8_t *in, uint64_t n, uint8_t *out, uint16_t *method_out) {
	uint64_t sz = 0;
	*method_out = OTZ_M_STORE;
	if (n == 0) {
		return 0; /* ref :793-801 */
	}
	uint64_t cap = otz_compress_bound (method, n);
	switch (method) {
	case OTZ_M_ZSTD: sz = otz_zstd_encode (in, n, out, cap); break;
	case OTZ_M_LZFSE: sz = otz_lzfse_encode (in, n, out, cap); break;
	case OTZ_M_BROTLI: sz = otz_brotli_encode (in, n, out, cap); break;
	case OTZ_M_LZMA: sz = otz_lzma_encode (in, n, out, cap); break;
	default: sz = 0; break;
	}
	if (sz == 0 || sz >= n) {
		memcpy (out, in, n); /* "not smaller -> STORE", ref :894-899 and siblings */
		return n;
	}
	*method_out = method;
	return sz;
}
