/* otz_zstd_rfc.c -- TEST INFRASTRUCTURE: a CPU decoder for Zstandard frames written from RFC 8878,
 * the checker for the real-format decoder under ZIP method 93.
 *
 * The reference implements no Zstandard bitstream at all: its "zstd" is a private raw-block
 * container (src/lib/zstd.inc.c:172-182 compress_block returns 0; :689-692 rejects RLE and
 * compressed blocks), so there is no reference code to restate.  This file restates the published
 * algorithm (RFC 8878 sections 3.1.1 frames, 3.1.1.2 blocks, 3.1.1.3 compressed blocks, 4.1 FSE,
 * 4.2 Huffman, appendix A default distributions) and is PINNED against a third-party implementation:
 * libzstd 1.4.8 as shipped in the image (/usr/lib/x86_64-linux-gnu/libzstd.so.1, reached through
 * ctypes by oracle/gen_zstd_golden.py and tests/test_zstd_real.py), plus the XXH64 content checksum
 * inside every frame, which is a known answer neither side of the comparison wrote.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "otz_oracle.h"

#define ZE_OK 0
#define ZE_DATA (-3)   /* corrupt / unsupported (dictionary) */
#define ZE_BUF (-5)    /* output does not fit */

/* ---------------------------------------------------------------- XXH64 (for the content checksum) */
#define XP1 11400714785074694791ULL
#define XP2 14029467366897019727ULL
#define XP3 1609587929392839161ULL
#define XP4 9650029242287828579ULL
#define XP5 2870177450012600261ULL
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t rd64(const uint8_t *p) {
	uint64_t v = 0;
	for (int i = 7; i >= 0; i--) {
		v = (v << 8) | p[i];
	}
	return v;
}
static uint32_t rd32le(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t xround(uint64_t acc, uint64_t in) { return rotl64 (acc + in * XP2, 31) * XP1; }
static uint64_t xmerge(uint64_t h, uint64_t v) { return (h ^ xround (0, v)) * XP1 + XP4; }
uint64_t otz_xxh64(const uint8_t *p, size_t n, uint64_t seed) {
	const uint8_t *end = p + n;
	uint64_t h;
	if (n >= 32) {
		uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
		while (p + 32 <= end) {
			v1 = xround (v1, rd64 (p));
			v2 = xround (v2, rd64 (p + 8));
			v3 = xround (v3, rd64 (p + 16));
			v4 = xround (v4, rd64 (p + 24));
			p += 32;
		}
		h = rotl64 (v1, 1) + rotl64 (v2, 7) + rotl64 (v3, 12) + rotl64 (v4, 18);
		h = xmerge (h, v1);
		h = xmerge (h, v2);
		h = xmerge (h, v3);
		h = xmerge (h, v4);
	} else {
		h = seed + XP5;
	}
	h += (uint64_t)n;
	while (p + 8 <= end) {
		h ^= xround (0, rd64 (p));
		h = rotl64 (h, 27) * XP1 + XP4;
		p += 8;
	}
	if (p + 4 <= end) {
		h ^= (uint64_t)rd32le (p) * XP1;
		h = rotl64 (h, 23) * XP2 + XP3;
		p += 4;
	}
	while (p < end) {
		h ^= (uint64_t)*p * XP5;
		h = rotl64 (h, 11) * XP1;
		p++;
	}
	h ^= h >> 33;
	h *= XP2;
	h ^= h >> 29;
	h *= XP3;
	h ^= h >> 32;
	return h;
}

/* ---------------------------------------------------------------- backward bit reader (RFC 8878 4.1: the
 * stream is read from its last byte; the highest set bit of the last byte is a padding marker) */
typedef struct {
	const uint8_t *p;
	int64_t bits; /* number of unread bits; may go negative: reads past the start yield zeros and are an error at the end */
} zbr;
static int zbr_init(zbr *b, const uint8_t *p, size_t n) {
	if (n == 0 || p[n - 1] == 0) {
		return -1;
	}
	int hi = 7;
	while (!((p[n - 1] >> hi) & 1)) {
		hi--;
	}
	b->p = p;
	b->bits = (int64_t) (n - 1) * 8 + hi;
	return 0;
}
/* n <= 32 bits; most significant bit = the bit nearest the end of the stream */
static uint32_t zbr_read(zbr *b, unsigned n) {
	uint32_t v = 0;
	for (unsigned i = 0; i < n; i++) {
		b->bits--;
		uint32_t bit = b->bits >= 0 ? (b->p[b->bits >> 3] >> (b->bits & 7)) & 1u : 0u;
		v = (v << 1) | bit;
	}
	return v;
}

/* ---------------------------------------------------------------- FSE tables */
typedef struct {
	uint8_t sym, nbits;
	uint16_t base;
} fse_ent;
typedef struct {
	fse_ent e[512];
	int log;
} fse_tab;

/* forward (little-endian) bit reader for FSE table descriptions */
typedef struct {
	const uint8_t *p;
	size_t n, bit;
} fbr;
static uint32_t fbr_peek(const fbr *b, unsigned n) {
	uint32_t v = 0;
	for (unsigned i = 0; i < n; i++) {
		size_t k = b->bit + i;
		uint32_t bit = (k >> 3) < b->n ? (b->p[k >> 3] >> (k & 7)) & 1u : 0u;
		v |= bit << i;
	}
	return v;
}

static int highbit(uint32_t v) {
	int r = 0;
	while (v >>= 1) {
		r++;
	}
	return r;
}

static int fse_build(fse_tab *t, const int16_t *norm, int nsym, int log) {
	const int size = 1 << log;
	uint16_t next[256];
	uint8_t spread[512];
	int high = size - 1;
	t->log = log;
	for (int s = 0; s < nsym; s++) {
		if (norm[s] == -1) {
			spread[high--] = (uint8_t)s;
			next[s] = 1;
		} else {
			next[s] = (uint16_t)norm[s];
		}
	}
	const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
	int pos = 0;
	for (int s = 0; s < nsym; s++) {
		for (int i = 0; i < norm[s]; i++) {
			spread[pos] = (uint8_t)s;
			do {
				pos = (pos + step) & mask;
			} while (pos > high);
		}
	}
	if (pos != 0) {
		return -1;
	}
	for (int i = 0; i < size; i++) {
		const int s = spread[i];
		const uint16_t x = next[s]++;
		const int nb = log - highbit (x);
		t->e[i].sym = (uint8_t)s;
		t->e[i].nbits = (uint8_t)nb;
		t->e[i].base = (uint16_t) ((x << nb) - size);
	}
	return 0;
}

/* normalized counts (RFC 8878 4.1.1); returns bytes consumed or -1 */
static int fse_read_norm(const uint8_t *p, size_t n, int16_t *norm, int *nsym, int *log, int max_log, int max_sym) {
	fbr b = { p, n, 0 };
	if (n == 0) {
		return -1;
	}
	const int al = (int)fbr_peek (&b, 4) + 5;
	b.bit += 4;
	if (al > max_log) {
		return -1;
	}
	int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, s = 0, prev0 = 0;
	while (remaining > 1 && s <= max_sym) {
		if (prev0) {
			/* repeat flags: 2 bits at a time, 3 = three more zeros and another flag follows */
			for (;;) {
				const int r = (int)fbr_peek (&b, 2);
				b.bit += 2;
				for (int k = 0; k < r && s <= max_sym; k++) {
					norm[s++] = 0;
				}
				if (r != 3) {
					break;
				}
			}
			prev0 = 0;
			if (s > max_sym) {
				break;
			}
			continue;
		}
		const int max = (2 * threshold - 1) - remaining;
		int count;
		const uint32_t v = fbr_peek (&b, (unsigned)nbits);
		if ((int) (v & (uint32_t) (threshold - 1)) < max) {
			count = (int) (v & (uint32_t) (threshold - 1));
			b.bit += (size_t)nbits - 1;
		} else {
			count = (int) (v & (uint32_t) (2 * threshold - 1));
			if (count >= threshold) {
				count -= max;
			}
			b.bit += (size_t)nbits;
		}
		count--; /* -1: "less than one" */
		remaining -= count < 0 ? -count : count;
		norm[s++] = (int16_t)count;
		prev0 = count == 0;
		while (remaining < threshold) {
			nbits--;
			threshold >>= 1;
		}
	}
	if (remaining != 1 || s > max_sym + 1) {
		return -1;
	}
	if ((b.bit + 7) / 8 > n) {
		return -1;
	}
	*nsym = s;
	*log = al;
	return (int) ((b.bit + 7) / 8);
}

/* ---------------------------------------------------------------- default distributions and code tables (RFC 8878 3.1.1.3.2.2) */
static const int16_t LL_DEF[36] = { 4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1 };
static const int16_t ML_DEF[53] = { 1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
	-1, -1, -1, -1, -1, -1, -1 };
static const int16_t OF_DEF[29] = { 1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1 };
static const uint32_t LL_BASE[36] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192,
	16384, 32768, 65536 };
static const uint8_t LL_BITS[36] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };
static const uint32_t ML_BASE[53] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
	43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539 };
static const uint8_t ML_BITS[53] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
	12, 13, 14, 15, 16 };

/* ---------------------------------------------------------------- frame state */
typedef struct {
	/* Huffman table of the literals (kept for "treeless" blocks): 2^max_bits entries */
	uint8_t h_sym[2048], h_nb[2048];
	int h_bits; /* 0: none yet */
	fse_tab ll, of, ml;
	int have_ll, have_of, have_ml;
	uint32_t rep[3];
	uint8_t *lit; /* literal buffer of the current block */
} zframe;

/* Huffman tree description (RFC 8878 4.2.1); returns bytes consumed or -1 */
static int huf_read_tree(zframe *f, const uint8_t *p, size_t n) {
	uint8_t w[256];
	int nw = 0;
	size_t used;
	if (n == 0) {
		return -1;
	}
	const int hb = p[0];
	if (hb >= 128) {
		nw = hb - 127;
		used = 1 + (size_t) (nw + 1) / 2;
		if (used > n) {
			return -1;
		}
		for (int i = 0; i < nw; i++) {
			const uint8_t byte = p[1 + i / 2];
			w[i] = (i & 1) ? byte & 15 : byte >> 4;
		}
	} else {
		/* FSE-compressed weights: two interleaved states over a backward bitstream */
		used = 1 + (size_t)hb;
		if (hb == 0 || used > n) {
			return -1;
		}
		int16_t norm[16];
		int nsym = 0, log = 0;
		const int hl = fse_read_norm (p + 1, (size_t)hb, norm, &nsym, &log, 6, 11);
		if (hl < 0 || hl >= hb) {
			return -1;
		}
		fse_tab t;
		if (fse_build (&t, norm, nsym, log) != 0) {
			return -1;
		}
		zbr b;
		if (zbr_init (&b, p + 1 + hl, (size_t) (hb - hl)) != 0) {
			return -1;
		}
		uint32_t s1 = zbr_read (&b, (unsigned)log), s2 = zbr_read (&b, (unsigned)log);
		if (b.bits < 0) {
			return -1;
		}
		for (;;) {
			if (nw >= 254) {
				return -1;
			}
			w[nw++] = t.e[s1].sym;
			if (b.bits < (int64_t)t.e[s1].nbits) { /* not enough bits to update state 1: state 2 holds the last symbol */
				w[nw++] = t.e[s2].sym;
				break;
			}
			s1 = t.e[s1].base + zbr_read (&b, t.e[s1].nbits);
			if (nw >= 254) {
				return -1;
			}
			w[nw++] = t.e[s2].sym;
			if (b.bits < (int64_t)t.e[s2].nbits) {
				w[nw++] = t.e[s1].sym;
				break;
			}
			s2 = t.e[s2].base + zbr_read (&b, t.e[s2].nbits);
		}
	}
	/* the last weight is implied: the total must be a power of two */
	uint32_t total = 0;
	for (int i = 0; i < nw; i++) {
		if (w[i] > 11) {
			return -1;
		}
		total += w[i] ? 1u << (w[i] - 1) : 0u;
	}
	if (total == 0 || nw > 255) {
		return -1;
	}
	const int max_bits = highbit (total) + 1;
	if (max_bits > 11) {
		return -1;
	}
	const uint32_t left = (1u << max_bits) - total;
	if (left & (left - 1)) {
		return -1; /* not a power of two */
	}
	w[nw++] = (uint8_t) (highbit (left) + 1);
	/* table: symbols sorted by weight ascending (= code length descending), then by value */
	uint32_t rank_start[13] = { 0 };
	uint32_t cnt[13] = { 0 };
	for (int i = 0; i < nw; i++) {
		cnt[w[i]]++;
	}
	if (cnt[1] < 2 || (cnt[1] & 1)) {
		return -1; /* (a complete prefix code has an even number of longest codes) */
	}
	uint32_t at = 0;
	for (int wt = 1; wt <= max_bits; wt++) {
		rank_start[wt] = at;
		at += cnt[wt] << (wt - 1);
	}
	if (at != (1u << max_bits)) {
		return -1;
	}
	for (int s = 0; s < nw; s++) {
		if (!w[s]) {
			continue;
		}
		const uint32_t len = 1u << (w[s] - 1);
		for (uint32_t k = 0; k < len; k++) {
			f->h_sym[rank_start[w[s]] + k] = (uint8_t)s;
			f->h_nb[rank_start[w[s]] + k] = (uint8_t) (max_bits + 1 - w[s]);
		}
		rank_start[w[s]] += len;
	}
	f->h_bits = max_bits;
	return (int)used;
}

static int huf_decode_stream(const zframe *f, const uint8_t *p, size_t n, uint8_t *out, size_t nout) {
	zbr b;
	if (zbr_init (&b, p, n) != 0) {
		return -1;
	}
	const int hb = f->h_bits;
	for (size_t i = 0; i < nout; i++) {
		/* peek hb bits (zeros beyond the start) */
		zbr t = b;
		const uint32_t idx = zbr_read (&t, (unsigned)hb);
		out[i] = f->h_sym[idx];
		b.bits -= f->h_nb[idx];
		if (b.bits < 0) {
			return -1;
		}
	}
	return b.bits == 0 ? 0 : -1;
}

/* literals section; returns bytes consumed or -1; *nlit = regenerated size */
static int read_literals(zframe *f, const uint8_t *p, size_t n, size_t *nlit) {
	if (n < 1) {
		return -1;
	}
	const int type = p[0] & 3, sf = (p[0] >> 2) & 3;
	size_t regen, comp = 0, hdr;
	int streams = 1;
	if (type < 2) { /* raw, RLE */
		if (sf == 0 || sf == 2) {
			regen = p[0] >> 3;
			hdr = 1;
		} else if (sf == 1) {
			if (n < 2) {
				return -1;
			}
			regen = (p[0] >> 4) | ((size_t)p[1] << 4);
			hdr = 2;
		} else {
			if (n < 3) {
				return -1;
			}
			regen = (p[0] >> 4) | ((size_t)p[1] << 4) | ((size_t)p[2] << 12);
			hdr = 3;
		}
		if (regen > 128 * 1024) {
			return -1;
		}
		if (type == 0) {
			if (hdr + regen > n) {
				return -1;
			}
			memcpy (f->lit, p + hdr, regen);
			*nlit = regen;
			return (int) (hdr + regen);
		}
		if (hdr + 1 > n) {
			return -1;
		}
		memset (f->lit, p[hdr], regen);
		*nlit = regen;
		return (int) (hdr + 1);
	}
	/* compressed (2) or treeless (3) */
	if (sf == 0 || sf == 1) {
		if (n < 3) {
			return -1;
		}
		streams = sf == 0 ? 1 : 4;
		const uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
		regen = (v >> 4) & 0x3FF;
		comp = (v >> 14) & 0x3FF;
		hdr = 3;
	} else if (sf == 2) {
		if (n < 4) {
			return -1;
		}
		streams = 4;
		const uint32_t v = rd32le (p);
		regen = (v >> 4) & 0x3FFF;
		comp = v >> 18;
		hdr = 4;
	} else {
		if (n < 5) {
			return -1;
		}
		streams = 4;
		const uint64_t v = (uint64_t)rd32le (p) | ((uint64_t)p[4] << 32);
		regen = (size_t) ((v >> 4) & 0x3FFFF);
		comp = (size_t) (v >> 22);
		hdr = 5;
	}
	if (regen > 128 * 1024 || hdr + comp > n) {
		return -1;
	}
	const uint8_t *q = p + hdr;
	size_t left = comp;
	if (type == 2) {
		const int t = huf_read_tree (f, q, left);
		if (t < 0) {
			return -1;
		}
		q += t;
		left -= (size_t)t;
	} else if (f->h_bits == 0) {
		return -1; /* treeless without a previous table */
	}
	if (streams == 1) {
		if (huf_decode_stream (f, q, left, f->lit, regen) != 0) {
			return -1;
		}
	} else {
		if (left < 6) {
			return -1;
		}
		const size_t s1 = q[0] | ((size_t)q[1] << 8), s2 = q[2] | ((size_t)q[3] << 8), s3 = q[4] | ((size_t)q[5] << 8);
		if (6 + s1 + s2 + s3 > left) {
			return -1;
		}
		const size_t s4 = left - 6 - s1 - s2 - s3;
		const size_t per = (regen + 3) / 4;
		if (per * 3 > regen) {
			return -1;
		}
		const uint8_t *a = q + 6;
		if (huf_decode_stream (f, a, s1, f->lit, per) != 0 || huf_decode_stream (f, a + s1, s2, f->lit + per, per) != 0 ||
			huf_decode_stream (f, a + s1 + s2, s3, f->lit + 2 * per, per) != 0 ||
			huf_decode_stream (f, a + s1 + s2 + s3, s4, f->lit + 3 * per, regen - 3 * per) != 0) {
			return -1;
		}
	}
	*nlit = regen;
	return (int) (hdr + comp);
}

/* one of the three sequence tables; returns bytes consumed or -1 */
static int read_seq_table(fse_tab *t, int *have, int mode, const uint8_t *p, size_t n, const int16_t *def, int def_n, int def_log, int max_log, int max_sym) {
	if (mode == 0) {
		if (fse_build (t, def, def_n, def_log) != 0) {
			return -1;
		}
		*have = 1;
		return 0;
	}
	if (mode == 1) {
		if (n < 1 || p[0] > max_sym) {
			return -1;
		}
		t->log = 0;
		t->e[0].sym = p[0];
		t->e[0].nbits = 0;
		t->e[0].base = 0;
		*have = 1;
		return 1;
	}
	if (mode == 2) {
		int16_t norm[64];
		int nsym = 0, log = 0;
		const int used = fse_read_norm (p, n, norm, &nsym, &log, max_log, max_sym);
		if (used < 0 || fse_build (t, norm, nsym, log) != 0) {
			return -1;
		}
		*have = 1;
		return used;
	}
	return *have ? 0 : -1; /* repeat */
}

static int decode_block(zframe *f, const uint8_t *p, size_t n, uint8_t *dst, size_t cap, size_t *pos, uint64_t window) {
	size_t nlit = 0;
	const int lu = read_literals (f, p, n, &nlit);
	if (lu < 0) {
		return ZE_DATA;
	}
	p += lu;
	n -= (size_t)lu;
	if (n < 1) {
		return ZE_DATA;
	}
	size_t nseq = p[0], hdr = 1;
	if (nseq >= 128) {
		if (nseq == 255) {
			if (n < 3) {
				return ZE_DATA;
			}
			nseq = (size_t)p[1] + ((size_t)p[2] << 8) + 0x7F00;
			hdr = 3;
		} else {
			if (n < 2) {
				return ZE_DATA;
			}
			nseq = ((nseq - 128) << 8) + p[1];
			hdr = 2;
		}
	}
	p += hdr;
	n -= hdr;
	size_t lit_at = 0;
	const size_t block_start = *pos;
	if (nseq > 0) {
		if (n < 1) {
			return ZE_DATA;
		}
		const int modes = p[0];
		if (modes & 3) {
			return ZE_DATA; /* reserved bits */
		}
		p++;
		n--;
		int u = read_seq_table (&f->ll, &f->have_ll, (modes >> 6) & 3, p, n, LL_DEF, 36, 6, 9, 35);
		if (u < 0) {
			return ZE_DATA;
		}
		p += u;
		n -= (size_t)u;
		u = read_seq_table (&f->of, &f->have_of, (modes >> 4) & 3, p, n, OF_DEF, 29, 5, 8, 31);
		if (u < 0) {
			return ZE_DATA;
		}
		p += u;
		n -= (size_t)u;
		u = read_seq_table (&f->ml, &f->have_ml, (modes >> 2) & 3, p, n, ML_DEF, 53, 6, 9, 52);
		if (u < 0) {
			return ZE_DATA;
		}
		p += u;
		n -= (size_t)u;
		zbr b;
		if (zbr_init (&b, p, n) != 0) {
			return ZE_DATA;
		}
		uint32_t sl = zbr_read (&b, (unsigned)f->ll.log), so = zbr_read (&b, (unsigned)f->of.log), sm = zbr_read (&b, (unsigned)f->ml.log);
		if (b.bits < 0) {
			return ZE_DATA;
		}
		for (size_t i = 0; i < nseq; i++) {
			const int oc = f->of.e[so].sym, mc = f->ml.e[sm].sym, lc = f->ll.e[sl].sym;
			if (oc > 31 || mc > 52 || lc > 35) {
				return ZE_DATA;
			}
			const uint64_t ov = ((uint64_t)1 << oc) + zbr_read (&b, (unsigned)oc);
			const uint32_t mlen = ML_BASE[mc] + zbr_read (&b, ML_BITS[mc]);
			const uint32_t llen = LL_BASE[lc] + zbr_read (&b, LL_BITS[lc]);
			uint64_t off;
			if (ov > 3) {
				off = ov - 3;
				f->rep[2] = f->rep[1];
				f->rep[1] = f->rep[0];
				f->rep[0] = (uint32_t)off;
			} else {
				uint32_t idx = (uint32_t)ov - 1 + (llen == 0 ? 1u : 0u);
				if (idx == 0) {
					off = f->rep[0];
				} else {
					off = idx < 3 ? f->rep[idx] : f->rep[0] - 1;
					if (idx > 1) {
						f->rep[2] = f->rep[1];
					}
					f->rep[1] = f->rep[0];
					f->rep[0] = (uint32_t)off;
				}
			}
			if (i + 1 < nseq) {
				sl = f->ll.e[sl].base + zbr_read (&b, f->ll.e[sl].nbits);
				sm = f->ml.e[sm].base + zbr_read (&b, f->ml.e[sm].nbits);
				so = f->of.e[so].base + zbr_read (&b, f->of.e[so].nbits);
			}
			if (b.bits < 0) {
				return ZE_DATA;
			}
			if (llen > nlit - lit_at) {
				return ZE_DATA;
			}
			if ((uint64_t)llen + mlen > cap - *pos) {
				return ZE_BUF;
			}
			memcpy (dst + *pos, f->lit + lit_at, llen);
			lit_at += llen;
			*pos += llen;
			(void)window; /* (libzstd bounds an offset by the bytes produced so far, not by the window: same here) */
			if (off == 0 || off > *pos) {
				return ZE_DATA;
			}
			for (uint32_t k = 0; k < mlen; k++) {
				dst[*pos + k] = dst[*pos + k - off];
			}
			*pos += mlen;
		}
		if (b.bits != 0) {
			return ZE_DATA;
		}
	}
	const size_t rest = nlit - lit_at;
	if (rest > cap - *pos) {
		return ZE_BUF;
	}
	memcpy (dst + *pos, f->lit + lit_at, rest);
	*pos += rest;
	if (*pos - block_start > 128 * 1024) {
		return ZE_DATA;
	}
	return ZE_OK;
}

/* Decodes every frame of src (skippable frames are skipped).  Returns 0, ZE_DATA or ZE_BUF;
 * *out_len = bytes produced.  flags (optional): bit 0 = a frame carried a content checksum (verified). */
int otz_zstd_rfc_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len, uint32_t *flags) {
	size_t ip = 0, pos = 0;
	int frames = 0;
	if (flags) {
		*flags = 0;
	}
	*out_len = 0;
	zframe *f = (zframe *)calloc (1, sizeof (*f));
	uint8_t *lit = (uint8_t *)malloc (128 * 1024 + 32);
	if (!f || !lit) {
		free (f);
		free (lit);
		return -4;
	}
	int rc = ZE_OK;
	while (ip < n && rc == ZE_OK) {
		if (n - ip < 4) {
			rc = ZE_DATA;
			break;
		}
		const uint32_t magic = rd32le (src + ip);
		if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
			if (n - ip < 8) {
				rc = ZE_DATA;
				break;
			}
			const uint32_t sz = rd32le (src + ip + 4);
			if ((uint64_t)sz + 8 > n - ip) {
				rc = ZE_DATA;
				break;
			}
			ip += 8 + (size_t)sz;
			continue;
		}
		if (magic != 0xFD2FB528u || n - ip < 5) {
			rc = ZE_DATA;
			break;
		}
		const uint8_t fhd = src[ip + 4];
		size_t h = ip + 5;
		const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
		if (fhd & 0x08) {
			rc = ZE_DATA; /* reserved bit */
			break;
		}
		uint64_t window = 0;
		if (!single) {
			if (h >= n) {
				rc = ZE_DATA;
				break;
			}
			const uint8_t wd = src[h++];
			const uint64_t base = (uint64_t)1 << (10 + (wd >> 3));
			window = base + (base >> 3) * (wd & 7);
		}
		static const int did_len[4] = { 0, 1, 2, 4 };
		if (did) {
			uint32_t id = 0;
			if (h + (size_t)did_len[did] > n) {
				rc = ZE_DATA;
				break;
			}
			for (int k = 0; k < did_len[did]; k++) {
				id |= (uint32_t)src[h + (size_t)k] << (8 * k);
			}
			h += (size_t)did_len[did];
			if (id != 0) {
				rc = ZE_DATA; /* dictionaries are not supported */
				break;
			}
		}
		const int fcs_len = fcs_flag == 0 ? (single ? 1 : 0) : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
		uint64_t fcs = 0;
		if (h + (size_t)fcs_len > n) {
			rc = ZE_DATA;
			break;
		}
		for (int k = 0; k < fcs_len; k++) {
			fcs |= (uint64_t)src[h + (size_t)k] << (8 * k);
		}
		if (fcs_len == 2) {
			fcs += 256;
		}
		h += (size_t)fcs_len;
		if (single) {
			window = fcs;
		}
		const size_t frame_start = pos;
		memset (f, 0, sizeof (*f));
		f->lit = lit;
		f->rep[0] = 1;
		f->rep[1] = 4;
		f->rep[2] = 8;
		int last = 0;
		while (!last && rc == ZE_OK) {
			if (h + 3 > n) {
				rc = ZE_DATA;
				break;
			}
			const uint32_t bh = (uint32_t)src[h] | ((uint32_t)src[h + 1] << 8) | ((uint32_t)src[h + 2] << 16);
			h += 3;
			last = (int) (bh & 1);
			const int type = (int) ((bh >> 1) & 3);
			const size_t bs = bh >> 3;
			if (bs > 128 * 1024) {
				rc = ZE_DATA; /* Block_Maximum_Size */
				break;
			}
			if (type == 0) {
				if (bs > n - h) {
					rc = ZE_DATA;
				} else if (bs > cap - pos) {
					rc = ZE_BUF;
				} else {
					memcpy (dst + pos, src + h, bs);
					pos += bs;
					h += bs;
				}
			} else if (type == 1) {
				if (h >= n) {
					rc = ZE_DATA;
				} else if (bs > cap - pos) {
					rc = ZE_BUF;
				} else {
					memset (dst + pos, src[h], bs);
					pos += bs;
					h += 1;
				}
			} else if (type == 2) {
				if (bs > n - h || bs > 128 * 1024) {
					rc = ZE_DATA;
				} else {
					rc = decode_block (f, src + h, bs, dst, cap, &pos, window ? window : ~(uint64_t)0);
					h += bs;
				}
			} else {
				rc = ZE_DATA;
			}
		}
		if (rc != ZE_OK) {
			break;
		}
		if (fcs_len && pos - frame_start != fcs) {
			rc = ZE_DATA;
			break;
		}
		if (checksum) {
			if (h + 4 > n) {
				rc = ZE_DATA;
				break;
			}
			if ((uint32_t)otz_xxh64 (dst + frame_start, pos - frame_start, 0) != rd32le (src + h)) {
				rc = ZE_DATA;
				break;
			}
			h += 4;
			if (flags) {
				*flags |= 1u;
			}
		}
		ip = h;
		frames++;
	}
	if (rc == ZE_OK && frames == 0) {
		rc = ZE_DATA;
	}
	*out_len = pos;
	free (f);
	free (lit);
	return rc;
}
