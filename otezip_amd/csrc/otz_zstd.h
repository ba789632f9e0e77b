// otz_zstd.h -- Zstandard frames (RFC 8878) under ZIP method 93: ONE WAVEFRONT per entry.
//
// The reference has no Zstandard decoder: its "zstd" is a private raw-block container
// (ref src/lib/zstd.inc.c:172-182 compress_block returns 0, :188-201, :689-692 rejects RLE and
// compressed blocks).  That container stays with otz_copy_kernel (otz_containers.h); this file is
// the real format, which SURVEY.md section 8 lists as row f4 and BASELINE.json's north_star names
// ("ZSTD sequence decode + FSE/Huffman").  Checked against libzstd 1.4.8 of the image and an
// independent CPU decoder written from the RFC (oracle/otz_zstd_rfc.c).
//
// Which of the two a method-93 payload is: both start with the bytes 28 B5 2F FD.  The fifth byte
// is the Frame_Header_Descriptor of a real frame, and a byte the reference skips unread (:495-503).
// Bit 4 of the descriptor is the format's "unused" bit, which no Zstandard encoder sets and which
// the hybrid frames the reference accepts carry (0x70): set -> private framing.  Otherwise the
// payload is decoded as Zstandard here, and if that fails it is handed to the private-framing
// decoder after all (a private frame may carry any fifth byte): route[] below.
//
// Shape of the work.  A frame is a chain of blocks (<= 128 KiB); a compressed block is a
// literals section (Huffman, 1 or 4 backward bitstreams) and a sequences section (three
// interleaved FSE streams, one backward bitstream, strictly serial: every sequence's table
// states depend on the previous one's).  Per block:
//   * literals: lanes 0-3 decode the (up to) four streams side by side into the wavefront's
//     literal buffer in global memory (tables in LDS);
//   * sequences, 64 at a time: lane 0 decodes 64 (literal length, match length, offset) triples
//     (repeat-offset history resolved on the way) into LDS; then ALL lanes execute them: wave
//     prefix sums give every sequence its output position, literal bytes and match bytes are
//     copied "flattened" (lane l takes byte l, l + 64, ... of the group's concatenated runs; a
//     6-step search over the prefix array finds its sequence), so the byte streams are coalesced
//     whatever the run lengths are.  A match whose source overlaps a match of the same group that
//     is not finished yet waits for the next round (the first unfinished match is always ready).
// The window is the entry's own output in global memory (Zstandard windows are megabytes: LDS
// cannot hold them); it was just written by this wavefront, so it is served from L2.
#pragma once

#include "otz_simt.h"
#include "otz_crc.h"
#include "../../include/otezip_gpu.h"

#define OTZ_ZS_BLOCK_MAX (128u * 1024u)
#define OTZ_ZS_LIT_BYTES (OTZ_ZS_BLOCK_MAX + 64u) // literal buffer per wavefront (global memory)
#define OTZ_ZS_HUF_LOG_MAX 11u
#define OTZ_ZS_SBUF 1024u // staged bytes of the sequence bitstream: a group of 64 sequences reads at most 64 x 105 bits = 840 bytes
// route[] values (one byte per entry of the batch, zeroed per batch)
#define OTZ_ZS_ROUTE_NONE 0u
#define OTZ_ZS_ROUTE_PRIVATE 1u // not a Zstandard frame after all: otz_copy_kernel decodes the private framing
#define OTZ_ZS_ROUTE_DONE 2u    // decoded here; otz_zstd_crc_kernel adds the CRC-32

typedef int32_t i32;
typedef int64_t i64;

struct otz_zs_lds {
	u16 huf[1u << OTZ_ZS_HUF_LOG_MAX]; // literal code: symbol | bits << 8
	u32 fse_ll[512], fse_ml[512], fse_of[256]; // entries: OTZ_ZS_E_* below
	u32 ll_tab[36], ml_tab[53]; // baseline | extra bits << 24 of the literal-length / match-length codes
	union {
		struct { // a group of sequences and the prefix arrays of its flattened copies
			u32 g_ll[64], g_ml[64], g_off[64];
			u32 pre_a[65], pre_b[64];
		} g;
		struct { // table construction (lane 0)
			int16_t norm[64];
			u16 next[64];
			u8 spread[512];
			u8 weights[256];
			u32 rank_start[16];
			u32 wtab[64]; // FSE table of the Huffman weights (accuracy <= 6)
		} b;
	};
	u8 sbuf[OTZ_ZS_SBUF + 32u]; // the stretch of the sequence bitstream lane 0 is reading (staged by all lanes):
	                            // stream byte k sits at sbuf[16 + k - wlo]
	u32 huf_log;     // 0: no literal code yet (treeless blocks need one)
	u32 ll_log, ml_log, of_log;
	u32 have_ll, have_ml, have_of;
};

// ---------------------------------------------------------------- small helpers

OTZ_DEV u32 otz_zs_highbit(u32 v) { return 31u - (u32)__builtin_clz (v); }

// bytes [at, at + 8) of p as a little-endian word; bytes at or beyond `avail` read as zero
OTZ_DEV u64 otz_zs_ld64(const u8 *p, u32 at, u32 avail) {
	if (at + 8u <= avail) {
		return otz_ld64u (p + at);
	}
	u64 v = 0;
	for (u32 k = 0; k < 8u; k++) {
		if (at + k < avail) {
			v |= (u64)p[at + k] << (8u * k);
		}
	}
	return v;
}

// Backward bit reader of ONE lane (RFC 8878 4.1): the stream is read from its end; the highest set
// bit of its last byte is a marker.  `bit` counts the unread bits; a 64-bit container holds the
// stream bits [cbit, cbit + 64) and is reloaded (one global load) only when a read reaches below it.
struct otz_zs_bits {
	const u8 *p;
	u32 n;    // bytes
	i32 bit;  // unread bits; negative: the stream was over-read (corrupt input)
	u64 c;
	i32 cbit;
};
OTZ_DEV void otz_zs_reload(otz_zs_bits &b) {
	const i32 lo = b.bit > 64 ? (b.bit - 64 + 7) >> 3 : 0;
	b.cbit = lo * 8;
	b.c = otz_zs_ld64 (b.p, (u32)lo, b.n);
}

// The sequence bitstream's reader (lane 0): the same stream, read from the LDS window the wavefront
// stages, with libzstd's container discipline -- c holds the stream bytes [ptr, ptr + 8), `used` bits of
// it (from the top) are consumed; a read is two shifts, a reload (at fixed points of a sequence, so the
// loop is straight-line code) moves ptr down by the whole bytes consumed.
struct otz_zs_sbits {
	const u8 *w; // L.sbuf
	i32 wlo;     // stream byte at w[16]
	i32 ptr;
	u64 c;
	u32 used;
};
OTZ_DEV void otz_zs_sreload(otz_zs_sbits &b) {
	b.ptr -= (i32) (b.used >> 3);
	b.used &= 7u;
	i32 at = 16 + b.ptr - b.wlo;
	at = at < 0 ? 0 : at; // (only a corrupt stream gets here: its position goes negative and the group is rejected)
	b.c = otz_ld64u (b.w + at);
}
OTZ_DEV u32 otz_zs_sread(otz_zs_sbits &b, u32 nb) { // nb <= 32
	const u32 v = (u32) (((b.c << b.used) >> 1) >> (63u - nb));
	b.used += nb;
	return v;
}
OTZ_DEV i32 otz_zs_spos(const otz_zs_sbits &b) { return 8 * b.ptr + 64 - (i32)b.used; } // unread bits; negative: over-read
OTZ_DEV bool otz_zs_bits_init(otz_zs_bits &b, const u8 *p, u32 n) {
	b.p = p;
	b.n = n;
	b.bit = 0;
	b.c = 0;
	b.cbit = 0;
	if (n == 0 || p[n - 1u] == 0) {
		return false;
	}
	b.bit = (i32) ((n - 1u) * 8u + otz_zs_highbit (p[n - 1u]));
	otz_zs_reload (b);
	return true;
}
// the next nb (<= 32) bits, most significant first; zeros once the stream is over-read
OTZ_DEV u32 otz_zs_read(otz_zs_bits &b, u32 nb) {
	const i32 lo = b.bit - (i32)nb;
	b.bit = lo;
	if (lo < 0) {
		return 0; // (callers check b.bit < 0 at the end)
	}
	if (lo < b.cbit) {
		const i32 keep = b.bit;
		b.bit = lo + (i32)nb;
		otz_zs_reload (b);
		b.bit = keep;
	}
	const u64 v = b.c >> (u32) (lo - b.cbit);
	return nb ? (u32)v & (nb >= 32u ? 0xFFFFFFFFu : (1u << nb) - 1u) : 0u;
}
// peek nb (<= 16) bits without consuming; bits beyond the start read as zero
OTZ_DEV u32 otz_zs_peek(otz_zs_bits &b, u32 nb) {
	if (b.bit <= 0) {
		return 0;
	}
	i32 lo = b.bit - (i32)nb;
	u32 sh = 0;
	if (lo < 0) {
		sh = (u32) (-lo);
		lo = 0;
	}
	if (lo < b.cbit) {
		otz_zs_reload (b);
	}
	const u32 v = (u32) (b.c >> (u32) (lo - b.cbit)) & ((1u << (nb - sh)) - 1u);
	return v << sh;
}

// default distributions (RFC 8878 3.1.1.3.2.2.x)
OTZ_CONST_TABLE int8_t otz_zs_ll_def[36] = { 4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1 };
OTZ_CONST_TABLE int8_t otz_zs_ml_def[53] = { 1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
	1, 1, 1, -1, -1, -1, -1, -1, -1, -1 };
OTZ_CONST_TABLE int8_t otz_zs_of_def[29] = { 1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1 };
OTZ_CONST_TABLE u32 otz_zs_ll_base_c[36] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
	4096, 8192, 16384, 32768, 65536 };
OTZ_CONST_TABLE u8 otz_zs_ll_bits_c[36] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };
OTZ_CONST_TABLE u32 otz_zs_ml_base_c[53] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35,
	37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539 };
OTZ_CONST_TABLE u8 otz_zs_ml_bits_c[53] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5,
	7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };

// ---------------------------------------------------------------- FSE (lane 0)

// forward bit reader over a byte range for table descriptions
OTZ_DEV u32 otz_zs_fpeek(const u8 *p, u32 n, u32 bit, u32 nb) {
	const u32 at = bit >> 3;
	u64 v = 0;
	for (u32 k = 0; k < 5u; k++) {
		if (at + k < n) {
			v |= (u64)p[at + k] << (8u * k);
		}
	}
	return (u32) (v >> (bit & 7u)) & ((1u << nb) - 1u);
}

// normalized counts (RFC 8878 4.1.1) into L.b.norm; returns bytes used or 0 on error
OTZ_DEV_NOINLINE u32 otz_zs_read_norm(otz_zs_lds &L, const u8 *p, u32 n, u32 *nsym, u32 *log, u32 max_log, u32 max_sym) {
	if (n == 0) {
		return 0;
	}
	u32 bit = 4;
	const u32 al = otz_zs_fpeek (p, n, 0, 4) + 5u;
	if (al > max_log) {
		return 0;
	}
	i32 remaining = (i32) (1u << al) + 1, threshold = (i32) (1u << al);
	u32 nbits = al + 1u, s = 0;
	bool prev0 = false;
	while (remaining > 1 && s <= max_sym) {
		if (prev0) {
			for (;;) {
				const u32 r = otz_zs_fpeek (p, n, bit, 2);
				bit += 2;
				for (u32 k = 0; k < r && s <= max_sym; k++) {
					L.b.norm[s++] = 0;
				}
				if (r != 3u) {
					break;
				}
			}
			prev0 = false;
			if (s > max_sym) {
				break;
			}
			continue;
		}
		const i32 max = (2 * threshold - 1) - remaining;
		const u32 v = otz_zs_fpeek (p, n, bit, nbits);
		i32 count;
		if ((i32) (v & (u32) (threshold - 1)) < max) {
			count = (i32) (v & (u32) (threshold - 1));
			bit += nbits - 1u;
		} else {
			count = (i32) (v & (u32) (2 * threshold - 1));
			if (count >= threshold) {
				count -= max;
			}
			bit += nbits;
		}
		count--;
		remaining -= count < 0 ? -count : count;
		L.b.norm[s++] = (int16_t)count;
		prev0 = count == 0;
		while (remaining < threshold) {
			nbits--;
			threshold >>= 1;
		}
	}
	if (remaining != 1 || (bit + 7u) / 8u > n) {
		return 0;
	}
	*nsym = s;
	*log = al;
	return (bit + 7u) / 8u;
}

// Table entry: [5:0] symbol, [10:6] the symbol's extra-bit count (sequence tables; so that the bit
// positions of a sequence follow from its three entries alone, without waiting for the baseline
// lookups), [15:11] bits of the state update, [31:16] its baseline.
#define OTZ_ZS_E_SYM(e) ((e) & 63u)
#define OTZ_ZS_E_ADD(e) (((e) >> 6) & 31u)
#define OTZ_ZS_E_NB(e) (((e) >> 11) & 31u)
#define OTZ_ZS_E_BASE(e) ((e) >> 16)
#define OTZ_ZS_KIND_LL 0u
#define OTZ_ZS_KIND_ML 1u
#define OTZ_ZS_KIND_OF 2u
#define OTZ_ZS_KIND_WEIGHTS 3u
OTZ_DEV u32 otz_zs_extra_bits(const otz_zs_lds &L, u32 kind, u32 sym) {
	if (kind == OTZ_ZS_KIND_OF) {
		return sym;
	}
	if (kind == OTZ_ZS_KIND_LL) {
		return L.ll_tab[sym < 36u ? sym : 0u] >> 24; // (the LDS copies: a constant-memory load per table cell is slow)
	}
	if (kind == OTZ_ZS_KIND_ML) {
		return L.ml_tab[sym < 53u ? sym : 0u] >> 24;
	}
	return 0;
}

// decoding table from L.b.norm[0..nsym) (RFC 8878 4.1.1); false on an inconsistent distribution
OTZ_DEV_NOINLINE bool otz_zs_fse_build(otz_zs_lds &L, u32 *tab, u32 nsym, u32 log, u32 kind) {
	const u32 size = 1u << log;
	i32 high = (i32)size - 1;
	for (u32 s = 0; s < nsym; s++) {
		if (L.b.norm[s] == -1) {
			if (high < 0) {
				return false;
			}
			L.b.spread[high--] = (u8)s;
			L.b.next[s] = 1;
		} else {
			L.b.next[s] = (u16)L.b.norm[s];
		}
	}
	const u32 step = (size >> 1) + (size >> 3) + 3u, mask = size - 1u;
	u32 pos = 0;
	for (u32 s = 0; s < nsym; s++) {
		for (i32 i = 0; i < L.b.norm[s]; i++) {
			L.b.spread[pos] = (u8)s;
			do {
				pos = (pos + step) & mask;
			} while ((i32)pos > high);
		}
	}
	if (pos != 0) {
		return false;
	}
	for (u32 i = 0; i < size; i++) {
		const u32 s = L.b.spread[i];
		const u32 x = L.b.next[s]++;
		const u32 nb = log - otz_zs_highbit (x);
		tab[i] = s | (otz_zs_extra_bits (L, kind, s) << 6) | (nb << 11) | ((((x << nb) - size) & 0xFFFFu) << 16);
	}
	return true;
}

// one of the three sequence tables (lane 0); returns bytes used, or 0xFFFFFFFF on error
OTZ_DEV_NOINLINE u32 otz_zs_seq_table(otz_zs_lds &L, u32 *tab, u32 *log, u32 *have, u32 mode, const u8 *p, u32 n, const int8_t *def, u32 def_n, u32 def_log,
	u32 max_log, u32 max_sym, u32 kind) {
	if (mode == 0) {
		for (u32 s = 0; s < def_n; s++) {
			L.b.norm[s] = def[s];
		}
		if (!otz_zs_fse_build (L, tab, def_n, def_log, kind)) {
			return 0xFFFFFFFFu;
		}
		*log = def_log;
		*have = 1;
		return 0;
	}
	if (mode == 1) {
		if (n < 1u || p[0] > max_sym) {
			return 0xFFFFFFFFu;
		}
		tab[0] = p[0] | (otz_zs_extra_bits (L, kind, p[0]) << 6);
		*log = 0;
		*have = 1;
		return 1;
	}
	if (mode == 2) {
		u32 nsym = 0, al = 0;
		const u32 used = otz_zs_read_norm (L, p, n, &nsym, &al, max_log, max_sym);
		if (used == 0 || !otz_zs_fse_build (L, tab, nsym, al, kind)) {
			return 0xFFFFFFFFu;
		}
		*log = al;
		*have = 1;
		return used;
	}
	return *have ? 0u : 0xFFFFFFFFu; // repeat
}

// ---------------------------------------------------------------- Huffman literals

// Huffman tree description (RFC 8878 4.2.1) -> L.huf (lane 0); returns bytes used or 0 on error
OTZ_DEV_NOINLINE u32 otz_zs_read_tree(otz_zs_lds &L, const u8 *p, u32 n) {
	if (n == 0) {
		return 0;
	}
	u32 nw = 0, used;
	const u32 hb = p[0];
	if (hb >= 128u) {
		nw = hb - 127u;
		used = 1u + (nw + 1u) / 2u;
		if (used > n) {
			return 0;
		}
		for (u32 i = 0; i < nw; i++) {
			const u32 byte = p[1u + i / 2u];
			L.b.weights[i] = (u8) ((i & 1u) ? byte & 15u : byte >> 4);
		}
	} else {
		used = 1u + hb;
		if (hb == 0 || used > n) {
			return 0;
		}
		u32 nsym = 0, log = 0;
		const u32 hl = otz_zs_read_norm (L, p + 1, hb, &nsym, &log, 6, 11);
		if (hl == 0 || hl >= hb) {
			return 0;
		}
		u32 *tab = L.b.wtab;
		if (!otz_zs_fse_build (L, tab, nsym, log, OTZ_ZS_KIND_WEIGHTS)) {
			return 0;
		}
		otz_zs_bits b;
		if (!otz_zs_bits_init (b, p + 1u + hl, hb - hl)) {
			return 0;
		}
		u32 s1 = otz_zs_read (b, log), s2 = otz_zs_read (b, log);
		if (b.bit < 0) {
			return 0;
		}
		for (;;) {
			if (nw >= 254u) {
				return 0;
			}
			L.b.weights[nw++] = (u8)OTZ_ZS_E_SYM (tab[s1]);
			u32 nb = OTZ_ZS_E_NB (tab[s1]);
			if (b.bit < (i32)nb) {
				L.b.weights[nw++] = (u8)OTZ_ZS_E_SYM (tab[s2]);
				break;
			}
			s1 = OTZ_ZS_E_BASE (tab[s1]) + otz_zs_read (b, nb);
			if (nw >= 254u) {
				return 0;
			}
			L.b.weights[nw++] = (u8)OTZ_ZS_E_SYM (tab[s2]);
			nb = OTZ_ZS_E_NB (tab[s2]);
			if (b.bit < (i32)nb) {
				L.b.weights[nw++] = (u8)OTZ_ZS_E_SYM (tab[s1]);
				break;
			}
			s2 = OTZ_ZS_E_BASE (tab[s2]) + otz_zs_read (b, nb);
		}
	}
	u32 total = 0;
	for (u32 i = 0; i < nw; i++) {
		const u32 w = L.b.weights[i];
		if (w > OTZ_ZS_HUF_LOG_MAX) {
			return 0;
		}
		total += w ? 1u << (w - 1u) : 0u;
	}
	if (total == 0 || nw > 255u) {
		return 0;
	}
	const u32 max_bits = otz_zs_highbit (total) + 1u;
	if (max_bits > OTZ_ZS_HUF_LOG_MAX) {
		return 0;
	}
	const u32 left = (1u << max_bits) - total;
	if (left & (left - 1u)) {
		return 0;
	}
	L.b.weights[nw++] = (u8) (otz_zs_highbit (left) + 1u);
	u32 cnt[OTZ_ZS_HUF_LOG_MAX + 2u];
	for (u32 w = 0; w <= OTZ_ZS_HUF_LOG_MAX + 1u; w++) {
		cnt[w] = 0;
	}
	for (u32 i = 0; i < nw; i++) {
		cnt[L.b.weights[i]]++;
	}
	if (cnt[1] < 2u || (cnt[1] & 1u)) {
		return 0;
	}
	u32 at = 0;
	for (u32 w = 1; w <= max_bits; w++) {
		L.b.rank_start[w] = at;
		at += cnt[w] << (w - 1u);
	}
	if (at != (1u << max_bits)) {
		return 0;
	}
	for (u32 s = 0; s < nw; s++) {
		const u32 w = L.b.weights[s];
		if (!w) {
			continue;
		}
		const u32 len = 1u << (w - 1u), e = s | ((max_bits + 1u - w) << 8);
		const u32 r0 = L.b.rank_start[w];
		for (u32 k = 0; k < len; k++) {
			L.huf[r0 + k] = (u16)e;
		}
		L.b.rank_start[w] = r0 + len;
	}
	L.huf_log = max_bits;
	return used;
}

// one stream of nout literals (one lane); false on a corrupt stream
OTZ_DEV bool otz_zs_huf_stream(const otz_zs_lds &L, const u8 *p, u32 n, u8 *out, u32 nout) {
	otz_zs_bits b;
	if (!otz_zs_bits_init (b, p, n)) {
		return false;
	}
	const u32 hl = L.huf_log;
	u32 i = 0;
	for (; i + 8u <= nout; i += 8u) { // eight literals per (unaligned) 8-byte store
		u64 acc = 0;
#pragma unroll
		for (u32 k = 0; k < 8u; k++) {
			const u32 e = L.huf[otz_zs_peek (b, hl)];
			acc |= (u64) (e & 0xFFu) << (8u * k);
			b.bit -= (i32) (e >> 8);
		}
		if (b.bit < 0) {
			return false;
		}
		otz_st64u (out + i, acc);
	}
	for (; i < nout; i++) {
		const u32 e = L.huf[otz_zs_peek (b, hl)];
		out[i] = (u8)e;
		b.bit -= (i32) (e >> 8);
		if (b.bit < 0) {
			return false;
		}
	}
	return b.bit == 0;
}

// ---------------------------------------------------------------- execution of a group of sequences

// largest s in [0, 64) with pre[s] <= f  (pre is non-decreasing, pre[0] = 0)
OTZ_DEV u32 otz_zs_find(const u32 *pre, u32 f) {
	u32 s = 0;
#pragma unroll
	for (u32 step = 32; step >= 1u; step >>= 1) {
		const u32 t = s + step;
		s = pre[t] <= f ? t : s; // (t <= 63)
	}
	return s;
}

// One flattened copy of the matches of the lanes in `sel` (their lengths in L.g.g_ml, offsets in L.g.g_off).
OTZ_DEV void otz_zs_copy_matches(otz_zs_lds &L, bool sel, u32 ml, u32 mstart, u8 *out) {
	const u32 lane = otz_lane ();
	const u32 rml = sel ? ml : 0u;
	const u32 incl = otz_wave_scan_incl (rml);
	const u32 tot_r = otz_readlane (incl, 63u);
	otz_wave_sync_lds ();
	L.g.pre_a[lane] = incl - rml;
	L.g.pre_b[lane] = mstart;
	otz_wave_sync_lds ();
	for (u32 f = lane; f < tot_r; f += 64u) {
		const u32 s = otz_zs_find (L.g.pre_a, f);
		const u32 k = f - L.g.pre_a[s], o = L.g.g_off[s], ms = L.g.pre_b[s];
		const u32 src = o >= L.g.g_ml[s] ? ms + k - o : ms - o + k % o; // an overlapping match repeats its first `o` bytes
		out[ms + k] = out[src];
	}
}

// All 64 lanes.  ng (1..64) sequences in L.g.*; literals from lit[*lpos ...), output at out[*opos ...).
// frame_start: offsets may not reach in front of it.  *synced: every output byte below this position is
// known to be in memory (a fence has run since it was stored): a match whose source lies below it can be
// copied without one.  Returns 0, OZG_ST_DATA or OZG_ST_SIZE.
OTZ_DEV int otz_zs_execute(otz_zs_lds &L, u32 ng, u8 *out, u32 cap, u32 *opos, const u8 *lit, u32 nlit, u32 *lpos, u32 frame_start, u32 *synced) {
	const u32 lane = otz_lane ();
	const bool mine = lane < ng;
	const u32 ll = mine ? L.g.g_ll[lane] : 0u, ml = mine ? L.g.g_ml[lane] : 0u, off = mine ? L.g.g_off[lane] : 1u;
	const u32 incl_ll = otz_wave_scan_incl (ll);
	const u32 incl_tot = otz_wave_scan_incl (ll + ml);
	const u32 tot_ll = otz_readlane (incl_ll, 63u), tot = otz_readlane (incl_tot, 63u);
	if (tot_ll > nlit - *lpos) {
		return OZG_ST_DATA;
	}
	if (tot > cap - *opos || tot < tot_ll) {
		return OZG_ST_SIZE; // more than uncomp_size bytes (ref otezip.c:555: total_out must equal it)
	}
	const u32 start = *opos + incl_tot - (ll + ml); // this sequence's literals go here, its match behind them
	const u32 mstart = start + ll;
	if (otz_ballot (mine && (off == 0 || off > mstart - frame_start)) != 0) {
		return OZG_ST_DATA;
	}
	// ---- literals
	otz_wave_sync_lds ();
	L.g.pre_a[lane] = incl_ll - ll;
	L.g.pre_b[lane] = start;
	otz_wave_sync_lds ();
	for (u32 f = lane; f < tot_ll; f += 64u) {
		const u32 s = otz_zs_find (L.g.pre_a, f);
		out[L.g.pre_b[s] + (f - L.g.pre_a[s])] = lit[*lpos + f];
	}
	// ---- matches whose source is already in memory: no fence needed (most of them: offsets are long)
	bool done = !mine || ml == 0;
	const u32 span = ml < off ? ml : off;
	const u32 src_hi = mstart - off + span;
	{
		const bool far = !done && src_hi <= *synced;
		if (otz_ballot (far) != 0) {
			otz_zs_copy_matches (L, far, ml, mstart, out);
			done = done || far;
		}
	}
	// ---- the others in rounds: a match is ready once everything its source needs is final and in memory
	while (otz_ballot (!done) != 0) {
		otz_wave_sync (); // what has been stored so far is in memory now
		const u64 pend = otz_ballot (!done);
		const u32 F = otz_readlane (mstart, otz_ctz64 (pend)); // start of the first unfinished match
		*synced = F;
		const bool ready = !done && src_hi <= F;
		otz_zs_copy_matches (L, ready, ml, mstart, out);
		done = done || ready;
	}
	*opos += tot;
	*lpos += tot_ll;
	return 0;
}

// cooperative byte copy / fill (any alignment)
OTZ_DEV void otz_zs_copy(u8 *dst, const u8 *src, u32 n) {
	for (u32 i = otz_lane (); i < n; i += 64u) {
		dst[i] = src[i];
	}
}
OTZ_DEV void otz_zs_fill(u8 *dst, u32 v, u32 n) {
	for (u32 i = otz_lane (); i < n; i += 64u) {
		dst[i] = (u8)v;
	}
}

// ---------------------------------------------------------------- XXH64 (content checksum)

#define OTZ_XP1 11400714785074694791ull
#define OTZ_XP2 14029467366897019727ull
#define OTZ_XP3 1609587929392839161ull
#define OTZ_XP4 9650029242287828579ull
#define OTZ_XP5 2870177450012600261ull
OTZ_DEV u64 otz_rotl64(u64 x, u32 r) { return (x << r) | (x >> (64u - r)); }
OTZ_DEV u64 otz_xround(u64 acc, u64 in) { return otz_rotl64 (acc + in * OTZ_XP2, 31) * OTZ_XP1; }
OTZ_DEV u64 otz_xmerge(u64 h, u64 v) { return (h ^ otz_xround (0, v)) * OTZ_XP1 + OTZ_XP4; }
// all lanes call; lanes 0-3 run the four accumulators, every lane gets the digest.  The input streams
// through LDS a kilobyte at a time (every lane loads 16 bytes, the next kilobyte is already on its way
// while the four lanes work through the 32 stripes of this one): the accumulators' chain of 64-bit
// multiplications then never waits for global memory.
OTZ_DEV_NOINLINE u64 otz_zs_xxh64(otz_zs_lds &L, const u8 *p, u32 n) {
	const u32 lane = otz_lane ();
	const u32 stripes = n >> 5;
	u64 v = lane == 0 ? OTZ_XP1 + OTZ_XP2 : lane == 1u ? OTZ_XP2 : lane == 2u ? 0ull : 0ull - OTZ_XP1;
	const u32 chunks = (stripes + 31u) >> 5; // of 1 KiB
	otz_u4 nx = { 0, 0, 0, 0 };
	if (chunks && 16u * lane + 16u <= n) {
		nx = otz_ld16 (p + 16u * lane);
	}
	for (u32 c = 0; c < chunks; c++) {
		otz_wave_sync_lds ();
		otz_st64u (&L.sbuf[16u * lane], (u64)nx.x | ((u64)nx.y << 32));
		otz_st64u (&L.sbuf[16u * lane + 8u], (u64)nx.z | ((u64)nx.w << 32));
		otz_wave_sync_lds ();
		const u32 nat = 1024u * (c + 1u) + 16u * lane;
		if (c + 1u < chunks && nat + 16u <= n) {
			nx = otz_ld16 (p + nat);
		}
		if (lane < 4u) {
			const u32 ns = stripes - 32u * c < 32u ? stripes - 32u * c : 32u;
			for (u32 k = 0; k < ns; k++) {
				v = otz_xround (v, otz_ld64u (&L.sbuf[32u * k + 8u * lane]));
			}
		}
	}
	otz_wave_sync_lds ();
	const u32 v0l = otz_readlane ((u32)v, 0), v0h = otz_readlane ((u32) (v >> 32), 0);
	const u32 v1l = otz_readlane ((u32)v, 1), v1h = otz_readlane ((u32) (v >> 32), 1);
	const u32 v2l = otz_readlane ((u32)v, 2), v2h = otz_readlane ((u32) (v >> 32), 2);
	const u32 v3l = otz_readlane ((u32)v, 3), v3h = otz_readlane ((u32) (v >> 32), 3);
	const u64 v1 = ((u64)v0h << 32) | v0l, v2 = ((u64)v1h << 32) | v1l, v3 = ((u64)v2h << 32) | v2l, v4 = ((u64)v3h << 32) | v3l;
	u64 h;
	if (n >= 32u) {
		h = otz_rotl64 (v1, 1) + otz_rotl64 (v2, 7) + otz_rotl64 (v3, 12) + otz_rotl64 (v4, 18);
		h = otz_xmerge (h, v1);
		h = otz_xmerge (h, v2);
		h = otz_xmerge (h, v3);
		h = otz_xmerge (h, v4);
	} else {
		h = OTZ_XP5;
	}
	h += (u64)n;
	u32 at = stripes << 5;
	while (at + 8u <= n) {
		h ^= otz_xround (0, otz_ld64u (p + at));
		h = otz_rotl64 (h, 27) * OTZ_XP1 + OTZ_XP4;
		at += 8u;
	}
	if (at + 4u <= n) {
		h ^= (u64)otz_ld32u (p + at) * OTZ_XP1;
		h = otz_rotl64 (h, 23) * OTZ_XP2 + OTZ_XP3;
		at += 4u;
	}
	while (at < n) {
		h ^= (u64)p[at] * OTZ_XP5;
		h = otz_rotl64 (h, 11) * OTZ_XP1;
		at++;
	}
	h ^= h >> 33;
	h *= OTZ_XP2;
	h ^= h >> 29;
	h *= OTZ_XP3;
	h ^= h >> 32;
	return h;
}

// ---------------------------------------------------------------- blocks and frames

// value of lane 0's v in every lane (control flow below is kept wave-uniform this way)
OTZ_DEV u32 otz_zs_bcast(u32 v) { return otz_readlane (v, 0); }

// One compressed block.  All lanes.  p/n: the block's bytes.  Returns 0 or a status.
// pc (PROF builds): cycles of 0 literals, 1 sequence tables, 2 sequence decoding (lane 0), 3 sequence execution, 4 bitstream staging
#define OTZ_ZS_MARK(i) \
	do { \
		if (PROF) { \
			const u64 _t = otz_cycles (); \
			pc[i] += _t - plast; \
			plast = _t; \
		} \
	} while (0)
template <bool PROF>
OTZ_DEV int otz_zs_block(otz_zs_lds &L, const u8 *p, u32 n, u8 *out, u32 cap, u32 *opos, u8 *lit, u32 frame_start, u32 rep[3], u32 *synced, u64 *pc) {
	u64 plast = PROF ? otz_cycles () : 0;
	const u32 lane = otz_lane ();
	if (n < 1u) {
		return OZG_ST_DATA;
	}
	// ---- literals section header (every lane reads the same bytes)
	const u32 b0 = p[0];
	const u32 type = b0 & 3u, sf = (b0 >> 2) & 3u;
	u32 regen, comp = 0, hdr, streams = 1;
	if (type < 2u) {
		if (sf == 0 || sf == 2u) {
			regen = b0 >> 3;
			hdr = 1;
		} else if (sf == 1u) {
			if (n < 2u) {
				return OZG_ST_DATA;
			}
			regen = (b0 >> 4) | ((u32)p[1] << 4);
			hdr = 2;
		} else {
			if (n < 3u) {
				return OZG_ST_DATA;
			}
			regen = (b0 >> 4) | ((u32)p[1] << 4) | ((u32)p[2] << 12);
			hdr = 3;
		}
	} else if (sf < 2u) {
		if (n < 3u) {
			return OZG_ST_DATA;
		}
		streams = sf == 0 ? 1u : 4u;
		const u32 v = b0 | ((u32)p[1] << 8) | ((u32)p[2] << 16);
		regen = (v >> 4) & 0x3FFu;
		comp = (v >> 14) & 0x3FFu;
		hdr = 3;
	} else if (sf == 2u) {
		if (n < 4u) {
			return OZG_ST_DATA;
		}
		streams = 4;
		const u32 v = b0 | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
		regen = (v >> 4) & 0x3FFFu;
		comp = v >> 18;
		hdr = 4;
	} else {
		if (n < 5u) {
			return OZG_ST_DATA;
		}
		streams = 4;
		const u64 v = (u64)b0 | ((u64)p[1] << 8) | ((u64)p[2] << 16) | ((u64)p[3] << 24) | ((u64)p[4] << 32);
		regen = (u32) ((v >> 4) & 0x3FFFFu);
		comp = (u32) (v >> 22);
		hdr = 5;
	}
	if (regen > OTZ_ZS_BLOCK_MAX) {
		return OZG_ST_DATA;
	}
	u32 used; // bytes of the literals section
	const u8 *litp = lit; // where the block's literals are read from
	if (type == 0) {
		if (hdr + regen > n) {
			return OZG_ST_DATA;
		}
		litp = p + hdr; // raw literals are read in place
		used = hdr + regen;
	} else if (type == 1u) {
		if (hdr + 1u > n) {
			return OZG_ST_DATA;
		}
		otz_zs_fill (lit, p[hdr], regen);
		used = hdr + 1u;
	} else {
		if (hdr + comp > n) {
			return OZG_ST_DATA;
		}
		const u8 *q = p + hdr;
		u32 left = comp;
		u32 ok = 1;
		if (type == 2u) {
			u32 t = 0;
			otz_wave_sync_lds (); // (the group arrays share storage with the construction scratch)
			if (lane == 0) {
				t = otz_zs_read_tree (L, q, left);
			}
			otz_wave_sync_lds ();
			t = otz_zs_bcast (t);
			if (t == 0) {
				return OZG_ST_DATA;
			}
			q += t;
			left -= t;
		} else if (otz_zs_bcast (L.huf_log) == 0) {
			return OZG_ST_DATA; // treeless block without a previous code
		}
		if (streams == 1u) {
			if (lane == 0) {
				ok = otz_zs_huf_stream (L, q, left, lit, regen) ? 1u : 0u;
			}
		} else {
			if (left < 6u) {
				return OZG_ST_DATA;
			}
			const u32 s1 = q[0] | ((u32)q[1] << 8), s2 = q[2] | ((u32)q[3] << 8), s3 = q[4] | ((u32)q[5] << 8);
			if (6u + s1 + s2 + s3 > left) {
				return OZG_ST_DATA;
			}
			const u32 s4 = left - 6u - s1 - s2 - s3;
			const u32 per = (regen + 3u) / 4u;
			if (per * 3u > regen) {
				return OZG_ST_DATA;
			}
			if (lane < 4u) {
				const u32 so = lane == 0 ? 0u : lane == 1u ? s1 : lane == 2u ? s1 + s2 : s1 + s2 + s3;
				const u32 sn = lane == 0 ? s1 : lane == 1u ? s2 : lane == 2u ? s3 : s4;
				const u32 cnt = lane < 3u ? per : regen - 3u * per;
				ok = otz_zs_huf_stream (L, q + 6u + so, sn, lit + lane * per, cnt) ? 1u : 0u;
			}
		}
		if (otz_ballot (ok == 0) != 0) {
			return OZG_ST_DATA;
		}
		used = hdr + comp;
	}
	otz_wave_sync (); // the literal buffer is complete
	OTZ_ZS_MARK (0);
	p += used;
	n -= used;
	// ---- sequences section header
	if (n < 1u) {
		return OZG_ST_DATA;
	}
	u32 nseq = p[0], shdr = 1;
	if (nseq >= 128u) {
		if (nseq == 255u) {
			if (n < 3u) {
				return OZG_ST_DATA;
			}
			nseq = (u32)p[1] + ((u32)p[2] << 8) + 0x7F00u;
			shdr = 3;
		} else {
			if (n < 2u) {
				return OZG_ST_DATA;
			}
			nseq = ((nseq - 128u) << 8) + p[1];
			shdr = 2;
		}
	}
	p += shdr;
	n -= shdr;
	u32 lpos = 0;
	const u32 block_out0 = *opos;
	if (nseq > 0) {
		if (n < 1u) {
			return OZG_ST_DATA;
		}
		const u32 modes = p[0];
		if (modes & 3u) {
			return OZG_ST_DATA;
		}
		p++;
		n--;
		// ---- the three tables (lane 0), then the initial states
		u32 tused = 0;
		otz_wave_sync_lds ();
		if (lane == 0) {
			u32 at = 0;
			u32 u = otz_zs_seq_table (L, L.fse_ll, &L.ll_log, &L.have_ll, (modes >> 6) & 3u, p, n, otz_zs_ll_def, 36, 6, 9, 35, OTZ_ZS_KIND_LL);
			if (u != 0xFFFFFFFFu) {
				at = u;
				u = otz_zs_seq_table (L, L.fse_of, &L.of_log, &L.have_of, (modes >> 4) & 3u, p + at, n - at, otz_zs_of_def, 29, 5, 8, 31, OTZ_ZS_KIND_OF);
			}
			if (u != 0xFFFFFFFFu) {
				at += u;
				u = otz_zs_seq_table (L, L.fse_ml, &L.ml_log, &L.have_ml, (modes >> 2) & 3u, p + at, n - at, otz_zs_ml_def, 53, 6, 9, 52, OTZ_ZS_KIND_ML);
			}
			tused = u == 0xFFFFFFFFu ? u : at + u;
		}
		otz_wave_sync_lds ();
		tused = otz_zs_bcast (tused);
		if (tused == 0xFFFFFFFFu || tused > n) {
			return OZG_ST_DATA;
		}
		p += tused;
		n -= tused;
		// ---- the bitstream: lane 0 decodes groups of 64 sequences, all lanes execute them
		if (n == 0 || p[n - 1u] == 0) {
			return OZG_ST_DATA; // no end marker
		}
		otz_zs_sbits b;
		b.w = L.sbuf;
		b.wlo = 0;
		b.ptr = (i32)n - 8;
		b.c = 0;
		b.used = 8u - otz_zs_highbit (p[n - 1u]);
		u32 sl = 0, so = 0, sm = 0;
		u32 bad = 0;
		u32 r0 = rep[0], r1 = rep[1], r2 = rep[2];
		i32 wlo = -0x7FFFFFFF; // first stream byte of the staged window (none yet)
		OTZ_ZS_MARK (1);
		for (u32 base = 0; base < nseq; base += 64u) {
			const u32 ng = nseq - base < 64u ? nseq - base : 64u;
			// the stretch of the bitstream this group can read, into LDS (all lanes; lane 0 then reads it at
			// LDS latency instead of waiting on global memory): 64 sequences take at most 64 x 105 bits
			{
				const i32 ptr = (i32)otz_zs_bcast ((u32)b.ptr) - (i32) (otz_zs_bcast (b.used) >> 3); // lowest byte of the next container
				const i32 top = ptr + 8; // one past its highest byte
				if (wlo == -0x7FFFFFFF || ptr - 848 < wlo) {
					wlo = top > (i32)OTZ_ZS_SBUF ? top - (i32)OTZ_ZS_SBUF : 0;
					otz_wave_sync_lds ();
					for (u32 c = lane; c < (OTZ_ZS_SBUF + 32u) / 16u; c += 64u) {
						// sbuf[16 c ...] = stream bytes [wlo - 16 + 16 c, ...): zeros outside the stream
						const i32 at = wlo - 16 + 16 * (i32)c;
						u64 lo8 = 0, hi8 = 0;
						if (at >= 0) {
							lo8 = otz_zs_ld64 (p, (u32)at, n);
							hi8 = otz_zs_ld64 (p, (u32)at + 8u, n);
						} else if (at + 16 > 0) { // straddles the stream's first byte (wlo < 16)
							for (i32 k = 0; k < 16; k++) {
								const i32 q = at + k;
								const u64 byte = q >= 0 && (u32)q < n ? p[q] : 0u;
								if (k < 8) {
									lo8 |= byte << (8 * k);
								} else {
									hi8 |= byte << (8 * (k - 8));
								}
							}
						}
						otz_st64u (&L.sbuf[16u * c], lo8);
						otz_st64u (&L.sbuf[16u * c + 8u], hi8);
					}
					otz_wave_sync_lds ();
				}
				b.wlo = wlo;
			}
			OTZ_ZS_MARK (4);
			otz_wave_sync_lds ();
			if (lane == 0) {
				if (base == 0) {
					otz_zs_sreload (b);
					sl = otz_zs_sread (b, L.ll_log);
					so = otz_zs_sread (b, L.of_log);
					sm = otz_zs_sread (b, L.ml_log);
				}
				for (u32 i = 0; i < ng; i++) {
					// The chain that limits this loop is state -> table entry -> next state.  The entries carry their
					// symbols' extra-bit counts, so every bit position of the sequence follows from them at once;
					// the baseline lookups (ll_tab / ml_tab) are off that chain.
					otz_zs_sreload (b); // >= 57 bits in hand: an offset's extra bits (<= 31) fit
					const u32 el = L.fse_ll[sl], eo = L.fse_of[so], em = L.fse_ml[sm];
					const u32 oc = OTZ_ZS_E_ADD (eo), mb = OTZ_ZS_E_ADD (em), lb = OTZ_ZS_E_ADD (el);
					const u32 mt = L.ml_tab[OTZ_ZS_E_SYM (em)], lt = L.ll_tab[OTZ_ZS_E_SYM (el)];
					const u32 ox = otz_zs_sread (b, oc);
					otz_zs_sreload (b); // the two lengths' extra bits (<= 16 + 16)
					const u32 mx = otz_zs_sread (b, mb);
					const u32 lx = otz_zs_sread (b, lb);
					if (base + i + 1u < nseq) {
						otz_zs_sreload (b); // three state updates (<= 9 + 9 + 8 bits)
						sl = OTZ_ZS_E_BASE (el) + otz_zs_sread (b, OTZ_ZS_E_NB (el));
						sm = OTZ_ZS_E_BASE (em) + otz_zs_sread (b, OTZ_ZS_E_NB (em));
						so = OTZ_ZS_E_BASE (eo) + otz_zs_sread (b, OTZ_ZS_E_NB (eo));
					}
					const u32 ov = (1u << oc) + ox;
					const u32 mlen = (mt & 0xFFFFFFu) + mx;
					const u32 llen = (lt & 0xFFFFFFu) + lx;
					u32 off;
					if (ov > 3u) {
						off = ov - 3u;
						r2 = r1;
						r1 = r0;
						r0 = off;
					} else {
						const u32 idx = ov - 1u + (llen == 0 ? 1u : 0u);
						if (idx == 0) {
							off = r0;
						} else {
							off = idx == 1u ? r1 : idx == 2u ? r2 : r0 - 1u;
							if (idx > 1u) {
								r2 = r1;
							}
							r1 = r0;
							r0 = off;
						}
					}
					L.g.g_ll[i] = llen;
					L.g.g_ml[i] = mlen;
					L.g.g_off[i] = off;
				}
				const i32 left = otz_zs_spos (b);
				if (left < 0 || (base + ng == nseq && left != 0)) {
					bad = 1; // over-read, or the bitstream was not consumed exactly
				}
			}
			otz_wave_sync_lds ();
			if (otz_zs_bcast (bad) != 0) {
				return OZG_ST_DATA;
			}
			OTZ_ZS_MARK (2);
			const int st = otz_zs_execute (L, ng, out, cap, opos, litp, regen, &lpos, frame_start, synced);
			if (st != 0) {
				return st;
			}
			OTZ_ZS_MARK (3);
		}
		rep[0] = otz_zs_bcast (r0);
		rep[1] = otz_zs_bcast (r1);
		rep[2] = otz_zs_bcast (r2);
	}
	// ---- the literals behind the last sequence
	const u32 rest = regen - lpos;
	if (rest > cap - *opos) {
		return OZG_ST_SIZE;
	}
	otz_zs_copy (out + *opos, litp + lpos, rest);
	*opos += rest;
	otz_wave_sync (); // (the literal buffer is rewritten by the next block)
	*synced = *opos;
	if (*opos - block_out0 > OTZ_ZS_BLOCK_MAX) {
		return OZG_ST_DATA;
	}
	return 0;
}

// Every frame of in[0, n_in) (skippable frames skipped).  All 64 lanes.  lit: the wavefront's
// OTZ_ZS_LIT_BYTES of global scratch.  Returns OZG_ST_OK (with *total) or the failure.
template <bool PROF>
OTZ_DEV int otz_zs_entry(otz_zs_lds &L, const u8 *in, u32 n_in, u8 *out, u32 cap, u8 *lit, u32 *total, u64 *pc) {
	const u32 lane = otz_lane ();
	for (u32 i = lane; i < 36u; i += 64u) {
		L.ll_tab[i] = otz_zs_ll_base_c[i] | ((u32)otz_zs_ll_bits_c[i] << 24);
	}
	for (u32 i = lane; i < 53u; i += 64u) {
		L.ml_tab[i] = otz_zs_ml_base_c[i] | ((u32)otz_zs_ml_bits_c[i] << 24);
	}
	otz_wave_sync_lds ();
	u32 ip = 0, opos = 0, frames = 0;
	while (ip < n_in) {
		if (n_in - ip < 4u) {
			return OZG_ST_DATA;
		}
		const u32 magic = (u32)in[ip] | ((u32)in[ip + 1u] << 8) | ((u32)in[ip + 2u] << 16) | ((u32)in[ip + 3u] << 24);
		if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
			if (n_in - ip < 8u) {
				return OZG_ST_DATA;
			}
			const u32 sz = (u32)in[ip + 4u] | ((u32)in[ip + 5u] << 8) | ((u32)in[ip + 6u] << 16) | ((u32)in[ip + 7u] << 24);
			if ((u64)sz + 8u > (u64) (n_in - ip)) {
				return OZG_ST_DATA;
			}
			ip += 8u + sz;
			continue;
		}
		if (magic != 0xFD2FB528u || n_in - ip < 5u) {
			return OZG_ST_DATA;
		}
		const u32 fhd = in[ip + 4u];
		u32 h = ip + 5u;
		const u32 fcs_flag = fhd >> 6, single = (fhd >> 5) & 1u, checksum = (fhd >> 2) & 1u, did = fhd & 3u;
		if (fhd & 0x08u) {
			return OZG_ST_DATA;
		}
		if (!single) {
			if (h >= n_in) {
				return OZG_ST_DATA;
			}
			h++; // Window_Descriptor: the window is the output buffer itself
		}
		const u32 did_len = did == 3u ? 4u : did;
		if (h + did_len > n_in) {
			return OZG_ST_DATA;
		}
		for (u32 k = 0; k < did_len; k++) {
			if (in[h + k] != 0) {
				return OZG_ST_DATA; // a dictionary: not supported
			}
		}
		h += did_len;
		const u32 fcs_len = fcs_flag == 0 ? single : fcs_flag == 1u ? 2u : fcs_flag == 2u ? 4u : 8u;
		if (h + fcs_len > n_in) {
			return OZG_ST_DATA;
		}
		u64 fcs = 0;
		for (u32 k = 0; k < fcs_len; k++) {
			fcs |= (u64)in[h + k] << (8u * k);
		}
		if (fcs_len == 2u) {
			fcs += 256u;
		}
		h += fcs_len;
		const u32 frame_start = opos;
		u32 synced = opos; // (frames before this one ended with a fence)
		u32 rep[3] = { 1u, 4u, 8u };
		if (lane == 0) {
			L.huf_log = 0;
			L.have_ll = L.have_ml = L.have_of = 0;
		}
		otz_wave_sync_lds ();
		bool last = false;
		while (!last) {
			if (h + 3u > n_in) {
				return OZG_ST_DATA;
			}
			const u32 bh = (u32)in[h] | ((u32)in[h + 1u] << 8) | ((u32)in[h + 2u] << 16);
			h += 3u;
			last = (bh & 1u) != 0;
			const u32 type = (bh >> 1) & 3u, bs = bh >> 3;
			if (bs > OTZ_ZS_BLOCK_MAX || type == 3u) {
				return OZG_ST_DATA;
			}
			if (type == 0) {
				if (bs > n_in - h) {
					return OZG_ST_DATA;
				}
				if (bs > cap - opos) {
					return OZG_ST_SIZE;
				}
				otz_zs_copy (out + opos, in + h, bs);
				opos += bs;
				h += bs;
			} else if (type == 1u) {
				if (h >= n_in) {
					return OZG_ST_DATA;
				}
				if (bs > cap - opos) {
					return OZG_ST_SIZE;
				}
				otz_zs_fill (out + opos, in[h], bs);
				opos += bs;
				h += 1u;
			} else {
				if (bs > n_in - h) {
					return OZG_ST_DATA;
				}
				const int st = otz_zs_block<PROF> (L, in + h, bs, out, cap, &opos, lit, frame_start, rep, &synced, pc);
				if (st != 0) {
					return st;
				}
				h += bs;
			}
		}
		otz_wave_sync (); // the frame's bytes are in memory (checksum below, next frame, CRC kernel)
		if (fcs_len && (u64) (opos - frame_start) != fcs) {
			return OZG_ST_DATA;
		}
		if (checksum) {
			if (h + 4u > n_in) {
				return OZG_ST_DATA;
			}
			const u32 want = (u32)in[h] | ((u32)in[h + 1u] << 8) | ((u32)in[h + 2u] << 16) | ((u32)in[h + 3u] << 24);
			const u64 t0 = PROF ? otz_cycles () : 0;
			if ((u32)otz_zs_xxh64 (L, out + frame_start, opos - frame_start) != want) {
				return OZG_ST_DATA;
			}
			if (PROF) {
				pc[5] += otz_cycles () - t0;
			}
			h += 4u;
		}
		ip = h;
		frames++;
	}
	if (frames == 0) {
		return OZG_ST_DATA;
	}
	*total = opos;
	return OZG_ST_OK;
}

// does a method-93 payload go to this decoder first?  (see the head of this file)
OTZ_DEV bool otz_zs_is_candidate(const u8 *in, u32 n_in) {
	if (n_in < 5u) {
		return false;
	}
	const u32 magic = (u32)in[0] | ((u32)in[1] << 8) | ((u32)in[2] << 16) | ((u32)in[3] << 24);
	if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
		return true; // a skippable frame in front
	}
	return magic == 0xFD2FB528u && (in[4] & 0x10u) == 0;
}

// One method-93 entry.  All 64 lanes.  Writes *res and route (lane 0).
template <bool PROF = false>
OTZ_DEV void otz_zs_run(otz_zs_lds &L, const u8 *in, u32 n_in, u8 *out, u32 cap, u8 *lit, ozg_result *res, u8 *route, u64 *prof = nullptr) {
	u32 total = 0;
	u64 pc[6] = { 0, 0, 0, 0, 0, 0 };
	const u64 t0 = PROF ? otz_cycles () : 0;
	const int st = otz_zs_entry<PROF> (L, in, n_in, out, cap, lit, &total, pc);
	if (PROF && prof && otz_lane () == 0) {
		for (u32 i = 0; i < 6u; i++) {
			prof[i] = pc[i];
		}
		prof[15] = otz_cycles () - t0;
	}
	if (otz_lane () == 0) {
		if (st == OZG_ST_OK && total == cap) {
			res->status = OZG_ST_OK;
			res->total_out = total;
			res->crc32 = 0;
			res->flags = 0;
			*route = OTZ_ZS_ROUTE_DONE;
		} else {
			// Not a Zstandard frame of uncomp_size bytes.  The payload may still be the reference's private
			// framing with an arbitrary fifth byte (ref zstd.inc.c:495-503 skips it unread): that decoder
			// has the last word, and its verdict is the entry's (a broken Zstandard frame fails there too).
			*route = OTZ_ZS_ROUTE_PRIVATE;
		}
	}
}
