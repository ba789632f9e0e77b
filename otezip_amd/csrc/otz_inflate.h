// otz_inflate.h -- raw DEFLATE (RFC 1951) decode of ONE archive entry by ONE
// 512-lane workgroup, self-contained (no index).  Replaces
// inflateInit2(-15)/inflate(Z_FINISH)/inflateEnd as otezip_extract_entry drives
// them (ref src/lib/otezip.c:490-532, src/lib/deflate-dec.inc.c:452-518,
// :547-831) plus the CRC-32 of :668.
//
// Since round 2 this is the FALLBACK: entries normally go through the index
// kernel (otz_huff_index.h) and the apply kernel (otz_lz_apply.h); only entries
// whose index does not fit their arena region (OTZ_IX_OVERFLOW) are decoded here,
// by otz_inflate_kernel over the overflow list.  The ring, bitmaps, table
// builders, flush and CRC in this file are shared with the apply kernel.
//
//   * the entry's output passes through a 56 KiB LDS ring (32 KiB LZ77 window +
//     one round of output; the window never touches HBM); it is flushed with
//     coalesced 16-byte stores and the CRC is taken from LDS on the way out;
//   * Huffman decode is parallel INSIDE the stream: each round stages 8 KiB of
//     compressed bytes into LDS (coalesced 16-byte loads), every lane decodes
//     the 128-bit subsequence that starts at its own speculative bit offset
//     (pass A), then re-decodes from where its left neighbour really ended
//     until it meets its own pass-A token boundaries (pass B; prefix codes
//     self-synchronise within a few tokens).  A fixpoint over the per-lane
//     exit positions yields the true token chain; a workgroup prefix sum of
//     per-lane output bytes places every lane's tokens;
//   * literals and LZ77 copies are written by all lanes at once (pass C); a
//     copy whose source bytes are not final yet waits on a per-byte "final"
//     bitmap in LDS instead of on program order.
//
// Results are bit-identical to a correct inflate; where the reference itself
// misbehaves the rules of SURVEY.md section 8 apply (F2 repaired and flagged,
// F12 zero-fill mirrored, malformed streams fail cleanly).
//
// Deliberate difference on MALFORMED input only: an over-subscribed Huffman
// code set is rejected (Z_DATA_ERROR) instead of being decoded by the
// reference's first-match search (ref :671-691); parity rule 4.
#pragma once

#include "otz_simt.h"
#include "otz_crc.h"
#include "../../include/otezip_gpu.h"

#ifndef OTZ_NT
#define OTZ_NT 512u            // work-items per workgroup (8 wavefronts)
#endif
#define OTZ_RING 57344u          // 56 KiB: 32 KiB LZ77 window + one round of output (OTZ_RMAX); not a power of two
#define OTZ_RIDX(p) ((p) % OTZ_RING) // absolute output position -> ring index (constant modulo: mul_hi + sub)
// ring index + small offset -> ring index, for i < 2 * OTZ_RING: the hot loops take one
// OTZ_RIDX per token/match and step with this (compare + subtract) from there
#define OTZ_RWRAP(i) ((i) >= OTZ_RING ? (i) - OTZ_RING : (i))
#ifndef OTZ_S_BITS
#define OTZ_S_BITS 128u         // speculative subsequence per lane per round (at most 256)
#endif
#define OTZ_RMAX 24576u         // output bytes per round (size of the bitmaps); OTZ_RING - OTZ_RMAX = the 32 KiB window
#define OTZ_STAGE_BYTES 8704u   // OTZ_NT * OTZ_S_BITS / 8 = 8192 B + skew + lookahead; also holds the CRC tables
#define OTZ_STAGE_WORDS (OTZ_STAGE_BYTES / 4)
#define OTZ_LIT_ROOT 9          // two-level tables: root + sub-tables for the longer codes
#define OTZ_DIST_ROOT 8
#define OTZ_LIT_LUT 1024u       // 512 root entries + room for the sub-tables (zlib's ENOUGH_LENS is 852)
#define OTZ_DIST_LUT 512u       // 256 root entries + sub-tables (<= 146 for 32 symbols)
#define OTZ_SPIN_LIMIT 1000000u
#define OTZ_ST_NEED_OUT_INT (-100) // OZG_ST_NEED_OUT is 0, which L.status uses for "still decoding"

// exit word of a lane's path: [23:0] stage-relative bit position, [26:24] kind
#define OTZ_X_NORMAL 0u  // first token start at or beyond the lane's range
#define OTZ_X_EOB 1u     // position just after an end-of-block code
#define OTZ_X_INVALID 3u // no code matches / reserved symbol
#define OTZ_X_TRUNC 4u   // ran past the end of the input
#define OTZ_X_POS(x) ((x) & 0xFFFFFFu)
#define OTZ_X_KIND(x) ((x) >> 24)
#define OTZ_X_MAKE(kind, pos) (((kind) << 24) | (pos))

// LUT entry: [3:0] code length, [5:4] kind, [11:8] extra-bit count,
// [31:16] literal / base length / base distance.
// Code length 0 marks a root slot without a code of its own: kind OTZ_K_LEN there
// means "link": [11:8] = index bits of the sub-table, [31:16] = its first slot.
#define OTZ_K_LIT 0u
#define OTZ_K_LEN 1u
#define OTZ_K_EOB 2u
#define OTZ_K_BAD 3u

struct otz_canon {
	u16 first[16]; // first code value of each length
	u16 count[16];
	u16 off[16];   // index of the first symbol of each length in sorted[]
};

struct otz_inflate_lds {
	u8 ring[OTZ_RING];
	u32 lit_lut[OTZ_LIT_LUT];
	u32 dist_lut[OTZ_DIST_LUT];
	union {
		u32 stage_w[OTZ_STAGE_WORDS];
		u8 stage[OTZ_STAGE_BYTES];
		u32 crc_t4[1024];
	};
	union {
		u32 fin[OTZ_RMAX / 32]; // pass C: byte is final
		u8 ll[320];             // table build: code lengths
	};
	union {
		u32 mstart[OTZ_RMAX / 32]; // pass C: an unresolved match starts at this byte
		u32 exits[OTZ_NT];         // pass A/B: per-lane exit word
		struct {                   // header parse + table build only
			u32 wcnt[6][16]; // table build: symbols of each code length per wavefront
			otz_canon lit_c, dist_c;
			u32 cl_lut[128];
			u32 hist[16 * 2]; // code length histograms (lit, dist)
		};
	};
	u32 scan[8];      // per-wave partials
	// workgroup-uniform scalars
	u32 status;       // 0 while decoding, else (u32) final status code (nonzero ends the entry)
	u32 flag;
	u32 first_term;
	u32 first_rej;
	u32 next_start;
	u32 hdr_bits;     // header: bits consumed
	u32 eob_start_lo, eob_start_hi;
	u32 rej_bytes_incl;
	u32 acc_bytes;
	u32 dbg[4];      // failure diagnostics: reason, lane, position, aux
	u32 bmin[2][2];  // pass B: per-iteration {first lane that changed, first terminated lane}
	u64 prof_last;
	u64 prof[16];    // diagnostic cycle counters per phase (only with a profile buffer)
};

struct otz_job {
	const u8 *in;
	u32 n_in;
	u8 *out;
	u32 out_cap;
	u64 *prof; // optional: 16 x u64 phase cycle counters for this entry (diagnostics), or null
};

// phase ids of the diagnostic counters
#define OTZ_P_HEADER 0
#define OTZ_P_LENGTHS 1
#define OTZ_P_TABLES 2
#define OTZ_P_STAGE 3
#define OTZ_P_PASS_A 4
#define OTZ_P_PASS_B 5
#define OTZ_P_SCAN 6
#define OTZ_P_EMIT 7
#define OTZ_P_FLUSH 8
#define OTZ_P_STORED 9
#define OTZ_P_RESOLVE 10
#define OTZ_P_ROUNDS 12
#define OTZ_P_B_ITERS 13
#define OTZ_P_BLOCKS 14
#define OTZ_P_TOTAL 15

// ------------------------------------------------------------------ helpers

// work-item 0 charges the cycles since the previous mark to `phase`
#define OTZ_MARK(L, J, phase) \
	do { \
		if ((J).prof && otz_tid () == 0) { \
			u64 _t = otz_cycles (); \
			(L).prof[phase] += _t - (L).prof_last; \
			(L).prof_last = _t; \
		} \
	} while (0)
#define OTZ_COUNT(L, J, phase, n) \
	do { \
		if ((J).prof && otz_tid () == 0) { \
			(L).prof[phase] += (n); \
		} \
	} while (0)

template <class LDS>
OTZ_DEV void otz_set_status(LDS &L, int st) { otz_lds_store (&L.status, (u32)st); }
template <class LDS>
OTZ_DEV int otz_get_status(const LDS &L) { return (int)otz_lds_load (&L.status); }

OTZ_DEV u32 otz_mk_lit_entry(u32 sym, u32 len) {
	if (sym < 256u) {
		return (sym << 16) | (OTZ_K_LIT << 4) | len;
	}
	if (sym == 256u) {
		return (OTZ_K_EOB << 4) | len;
	}
	if (sym > 285u) {
		return (OTZ_K_BAD << 4) | len; // ref deflate-dec.inc.c:794-797
	}
	u32 k = sym - 257u, base, xb;
	if (k < 8u) {
		base = 3u + k;
		xb = 0;
	} else if (k == 28u) {
		base = 258u;
		xb = 0;
	} else {
		xb = (k >> 2) - 1u;
		base = 3u + ((4u + (k & 3u)) << xb);
	}
	return (base << 16) | (xb << 8) | (OTZ_K_LEN << 4) | len;
}

OTZ_DEV u32 otz_mk_dist_entry(u32 sym, u32 len) {
	if (sym >= 30u) {
		return (OTZ_K_BAD << 4) | len; // reference indexes past dist_base[] here (:766-774)
	}
	u32 base, xb;
	if (sym < 4u) {
		base = 1u + sym;
		xb = 0;
	} else {
		xb = (sym >> 1) - 1u;
		base = 1u + ((2u + (sym & 1u)) << xb);
	}
	return (base << 16) | (xb << 8) | (OTZ_K_LEN << 4) | len;
}

struct otz_tok {
	u32 kind;  // OTZ_K_*
	u32 nbits; // stream bits of the whole token
	u32 val;   // literal byte or match length
	u32 dist;
};

// Table entries: code length in bits 0-3, OTZ_K_* in bits 4-5, extra-bit count in bits
// 8-11, value (literal, length base, distance base) in bits 16+.  A slot no code maps to is
// OTZ_LUT_NONE (kind BAD, length 0); a root slot whose codes are longer than the root is a
// link (OTZ_LUT_LINK, the sign bit).
#define OTZ_LUT_NONE (OTZ_K_BAD << 4)
#define OTZ_LUT_LINK 0x80000000u
// root lookup, then at most one sub-table lookup
template <u32 ROOT>
OTZ_DEV u32 otz_lut2(const u32 *lut, u32 bits) {
	u32 e = lut[bits & ((1u << ROOT) - 1u)];
	if ((int)e < 0) { // OTZ_LUT_LINK: sub-table base in bits 16-30, index width in bits 8-11
		e = lut[((e >> 16) & 0x7FFFu) + otz_bfe (bits, ROOT, (e >> 8) & 15u)];
	}
	return e;
}

// Decode the token (literal / end-of-block / length+distance pair) whose bits start
// at bit 0 of (hi:lo): 64 stream bits always cover a whole token (at most
// 15+5+15+13 = 48 bits).  (Fetching them with one unaligned 8-byte LDS read instead
// of aligned words measured slower.)
template <class LDS>
OTZ_DEV otz_tok otz_decode_bits(const LDS &L, u32 lo, u32 hi) {
	otz_tok t;
	u32 e = otz_lut2<OTZ_LIT_ROOT> (L.lit_lut, lo);
	u32 len = e & 15u;
	t.kind = (e >> 4) & 3u;
	t.val = e >> 16;
	t.dist = 0;
	t.nbits = len;
	if (t.kind != OTZ_K_LEN) {
		return t;
	}
	u32 xb = (e >> 8) & 15u;
	t.val += otz_bfe (lo, len, xb);
	u32 used = len + xb; // <= 20
	u32 db = otz_alignbit (hi, lo, used);
	u32 d = otz_lut2<OTZ_DIST_ROOT> (L.dist_lut, db);
	u32 dl = d & 15u, dxb = (d >> 8) & 15u; // dl + dxb <= 28
	if (((d >> 4) & 3u) != OTZ_K_LEN) {
		t.kind = OTZ_K_BAD;
		t.nbits = 0;
		return t;
	}
	t.dist = (d >> 16) + otz_bfe (db, dl, dxb);
	t.nbits = used + dl + dxb;
	return t;
}

// A token walk keeps three stage words (96 stream bits from the word holding the
// current position) in registers.  Each step asks for the next two words BEFORE it
// decodes: they arrive while the table lookups are in flight, and the window then
// slides by 0-2 words to the token's end -- so the stream bits of the next token are
// already there and a step waits for the table lookups only (one LDS round trip for
// a literal, two for a match, instead of two and three).
struct otz_win {
	u32 w0, w1, w2;
	u32 a; // stage word index of w0; the current position lies in that word
};
OTZ_DEV void otz_win_init(otz_win &W, const otz_inflate_lds &L, u32 pos) {
	W.a = pos >> 5;
	W.w0 = L.stage_w[W.a];
	W.w1 = L.stage_w[W.a + 1u];
	W.w2 = L.stage_w[W.a + 2u];
}
OTZ_DEV otz_tok otz_decode_adv(const otz_inflate_lds &L, otz_win &W, u32 pos) {
	u32 p0 = L.stage_w[W.a + 3u], p1 = L.stage_w[W.a + 4u];
	u32 sh = pos & 31u;
	otz_tok t = otz_decode_bits (L, otz_alignbit (W.w1, W.w0, sh), otz_alignbit (W.w2, W.w1, sh));
	u32 na = (pos + t.nbits) >> 5; // a token is at most 48 bits: 0-2 words further
	u32 d = na - W.a;
	u32 n0 = d == 0 ? W.w0 : d == 1u ? W.w1 : W.w2;
	u32 n1 = d == 0 ? W.w1 : d == 1u ? W.w2 : p0;
	u32 n2 = d == 0 ? W.w2 : d == 1u ? p0 : p1;
	W.w0 = n0;
	W.w1 = n1;
	W.w2 = n2;
	W.a = na;
	return t;
}

// Inclusive prefix sum over the workgroup; *total gets the sum. Uniform control flow.
OTZ_DEV u32 otz_block_scan_incl(u32 v, u32 *scratch, u32 *total) {
	u32 lane = otz_lane ();
	v = otz_wave_scan_incl (v);
	otz_sync ();
	if (lane == 63u) {
		scratch[otz_wave ()] = v;
	}
	otz_sync ();
	u32 add = 0, tot = 0;
	for (u32 w = 0; w < OTZ_NT / 64; w++) {
		u32 s = scratch[w];
		if (w < otz_wave ()) {
			add += s;
		}
		tot += s;
	}
	*total = tot;
	return v + add;
}

// ------------------------------------------------------- staging compressed bytes

// Stage OTZ_STAGE_BYTES of input starting at the 16-byte-aligned global address at
// or below in+byte0; zero beyond the entry. Returns the skew (in+byte0 - aligned).
template <class LDS>
OTZ_DEV u32 otz_stage_load(LDS &L, const otz_job &J, u32 byte0) {
	const u8 *p = J.in + byte0;
	u32 skew = (u32) ((uintptr_t)p & 15u);
	const u8 *ga = p - skew;
	const u8 *lo = J.in, *hi = J.in + J.n_in;
	for (u32 c = otz_tid (); c < OTZ_STAGE_BYTES / 16; c += OTZ_NT) {
		const u8 *a = ga + 16u * c;
		otz_u4 v;
		if (a >= lo && a + 16 <= hi) {
			v = otz_ld16_aligned (a);
		} else {
			u32 w[4] = { 0, 0, 0, 0 };
			if (a + 16 > lo && a < hi) {
				for (u32 k = 0; k < 16; k++) {
					const u8 *q = a + k;
					if (q >= lo && q < hi) {
						w[k >> 2] |= (u32)*q << (8u * (k & 3u));
					}
				}
			}
			v.x = w[0];
			v.y = w[1];
			v.z = w[2];
			v.w = w[3];
		}
		u32 *d = &L.stage_w[4u * c];
		d[0] = v.x;
		d[1] = v.y;
		d[2] = v.z;
		d[3] = v.w;
	}
	return skew;
}

// ------------------------------------------------------------- flush + CRC

// Write ring[flushed, upto) to global memory (coalesced 16-byte stores) and fold
// it into the running CRC register.  `flushed` is a multiple of 16; `upto` too
// unless this is the last flush.  Uses the stage area for the CRC tables.
OTZ_DEV void otz_flush(otz_inflate_lds &L, const otz_job &J, u32 &flushed, u32 upto, u32 &crc_reg) {
	u32 n = upto - flushed;
	if (n == 0) {
		return;
	}
	otz_sync ();
	OTZ_MARK (L, J, OTZ_P_SCAN);
	otz_crc_load_slice4 (L.crc_t4);
	u32 nch = n >> 4, tail = n & 15u;
	u32 tid = otz_tid ();
	for (u32 c = tid; c < nch; c += OTZ_NT) {
		u32 a = OTZ_RIDX (flushed + 16u * c);
		otz_u4 v = *reinterpret_cast<const otz_u4 *> (&L.ring[a]);
		otz_st16_aligned (J.out + flushed + 16u * c, v);
	}
	if (tid < tail) {
		J.out[flushed + 16u * nch + tid] = L.ring[OTZ_RIDX (flushed + 16u * nch + tid)];
	}
	otz_sync ();
	// contiguous run of chunks per lane; an odd run length keeps the 16-byte LDS
	// reads of a wave on distinct bank quads
	u32 cpt = ((nch + OTZ_NT - 1u) / OTZ_NT) | 1u;
	u32 c0 = tid * cpt, c1 = c0 + cpt;
	if (c1 > nch) {
		c1 = nch;
	}
	u32 r = 0;
	for (u32 c = c0; c < c1; c++) {
		u32 a = OTZ_RIDX (flushed + 16u * c);
		otz_u4 v = *reinterpret_cast<const otz_u4 *> (&L.ring[a]);
		r = otz_crc_chunk16 (r, v, L.crc_t4);
	}
	if (c0 < c1) {
		u32 after = n - 16u * c1;
		if (after) {
			r = otz_gf_mul (r, otz_xpow_bytes (after));
		}
	}
	if (tail && tid == OTZ_NT - 1u) {
		u32 rt = 0;
		for (u32 k = 0; k < tail; k++) {
			rt = otz_crc_byte (rt, L.ring[OTZ_RIDX (flushed + 16u * nch + k)], L.crc_t4);
		}
		r ^= rt;
	}
	u32 R = otz_block_xor (r, L.scan);
	crc_reg = otz_gf_mul (crc_reg, otz_xpow_bytes (n)) ^ R;
	flushed = upto;
	otz_sync ();
	OTZ_MARK (L, J, OTZ_P_FLUSH);
}

// Make room for `need` more bytes at opos: everything older than 64 KiB - need
// must already be in global memory.
OTZ_DEV void otz_make_room(otz_inflate_lds &L, const otz_job &J, u32 opos, u32 need, u32 &flushed, u32 &crc_reg) {
	if (opos + need - flushed > OTZ_RING) {
		otz_flush (L, J, flushed, opos & ~15u, crc_reg);
	}
}

// -------------------------------------------------------------- table build

// Histogram + canonical bookkeeping for n code lengths at ll[0..n). Returns
// false when the set is over-subscribed.  Called by one work-item.
OTZ_DEV bool otz_canon_build(otz_canon &c, const u32 *hist) {
	u32 code = 0, off = 0;
	int left = 1;
	c.first[0] = 0;
	c.count[0] = 0;
	c.off[0] = 0;
	u32 prev = 0;
	for (u32 l = 1; l <= 15; l++) {
		u32 cnt = hist[l];
		code = (code + prev) << 1;
		c.first[l] = (u16)code;
		c.count[l] = (u16)cnt;
		c.off[l] = (u16)off;
		off += cnt;
		prev = cnt;
		left = (left << 1) - (int)cnt;
		if (left < 0) {
			return false;
		}
	}
	return true;
}

// Build both two-level decode tables from L.ll[0..hlit+hdist).  All work-items;
// returns false (uniformly) on an over-subscribed set, or if the sub-tables do not
// fit (which no code set a compressor emits can cause).
//
// One work-item per SYMBOL (wavefronts 0-4: the 288 literal/length symbols, wavefront
// 5: the 32 distance symbols).  A symbol's canonical code is first[len] + its rank
// among the symbols of the same length, and the rank comes from wavefront ballots --
// no sorted symbol list is ever materialised.  The symbol then writes its own root
// slots (code no longer than the root), or claims a sub-table under its root slot.
#ifdef OTZ_TB_PROF
#define OTZ_TBMARK(slot) do { if (otz_tid () == 0) { u64 _t = otz_cycles (); L.prof[slot] += _t - L.prof_last; L.prof_last = _t; } } while (0)
#else
#define OTZ_TBMARK(slot) do { } while (0)
#endif
template <class LDS>
OTZ_DEV bool otz_build_tables(LDS &L, u32 hlit, u32 hdist) {
	static_assert (OTZ_NT >= 384 && OTZ_NT >= (1u << OTZ_LIT_ROOT), "one work-item per symbol and per literal root slot");
	constexpr u32 LRSZ = 1u << OTZ_LIT_ROOT, DRSZ = 1u << OTZ_DIST_ROOT;
	// (opaque: nothing derived from them is computed once before the caller's block loop and kept -- spilled -- across it)
	const u32 tid = otz_opaque (otz_tid ()), lane = otz_opaque (otz_lane ()), wave = otz_opaque (otz_wave ());
	const bool is_dist = tid >= 320u;
	const u32 sym = is_dist ? tid - 320u : tid;
	u32 l = 0;
	if (!is_dist && sym < hlit) {
		l = L.ll[sym];
	} else if (is_dist && sym < hdist) {
		l = L.ll[hlit + sym];
	}
	for (u32 i = tid; i < OTZ_LIT_LUT; i += OTZ_NT) {
		L.lit_lut[i] = OTZ_LUT_NONE;
	}
	for (u32 i = tid; i < OTZ_DIST_LUT; i += OTZ_NT) {
		L.dist_lut[i] = OTZ_LUT_NONE;
	}
	u32 rank = 0; // symbols of the same length before this one in its wavefront
	if (wave < 6u) {
		// one ballot per code length that occurs in this wavefront (typically 4-8 of the 15)
		const u64 below = ((u64)1 << lane) - 1u;
		if (lane < 16u) {
			L.wcnt[wave][lane] = 0; // (the counts below come from the same wavefront: LDS keeps its order)
		}
		u64 rem = otz_ballot (l != 0);
		while (rem != 0) {
			const u32 k = otz_readlane (l, otz_ctz64 (rem));
			const u64 m = otz_ballot (l == k);
			if (l == k) {
				rank = otz_popc64 (m & below);
			}
			if (lane == 0) {
				L.wcnt[wave][k] = otz_popc64 (m);
			}
			rem &= ~m;
		}
	}
	otz_sync ();
	OTZ_TBMARK (1);
	// Canonical bookkeeping, one lane per code length (wavefront 0: literal/length code,
	// wavefront 1: distance code): first[l] = sum over j < l of count[j] << (l - j),
	// off[l] = sum over j < l of count[j].  A set is over-subscribed (ref: "left < 0")
	// exactly when first[15] + count[15] exceeds 2^15.
	if (wave < 2u) {
		u32 cnt = 0;
		if (lane >= 1u && lane < 16u) {
			const u32 *wc = &L.wcnt[0][0] + lane; // one address, constant offsets
			cnt = wave == 1u ? wc[80] : wc[0] + wc[16] + wc[32] + wc[48] + wc[64];
		}
		u32 code = 0, off = 0;
		for (u32 j = 1; j < 15u; j++) {
			u32 cj = otz_shfl (cnt, j);
			if (lane > j && lane < 16u) {
				code += cj << (lane - j);
				off += cj;
			}
		}
		if (lane < 16u) {
			otz_canon &c = wave == 1u ? L.dist_c : L.lit_c;
			c.first[lane] = (u16)code;
			c.count[lane] = (u16)cnt;
			c.off[lane] = (u16)off;
		}
		if (lane == 15u) {
			u32 ok = code + cnt <= (1u << 15) ? 1u : 0u;
			if (wave == 0) {
				L.flag = ok;
			} else {
				L.first_rej = ok;
			}
		}
	}
	otz_sync ();
	OTZ_TBMARK (6);
	if (L.flag == 0u || L.first_rej == 0u) {
		return false;
	}
	// the symbol's code, bit-reversed (DEFLATE sends codes most significant bit first)
	const u32 root = is_dist ? OTZ_DIST_ROOT : OTZ_LIT_ROOT, rsz = is_dist ? DRSZ : LRSZ;
	u32 *const lut = is_dist ? L.dist_lut : L.lit_lut;
	u32 rr = 0, e = 0;
	if (l) {
		const otz_canon &c = is_dist ? L.dist_c : L.lit_c;
		u32 before = rank;
		for (u32 w = 0; w < wave && !is_dist; w++) {
			before += L.wcnt[w][l];
		}
		rr = otz_brev ((u32)c.first[l] + before) >> (32u - l);
		e = is_dist ? otz_mk_dist_entry (sym, l) : otz_mk_lit_entry (sym, l);
		if (l > root) {
			otz_lds_max (&lut[rr & (rsz - 1u)], (l - root) << 8); // widest code under this root slot
		} else if (l + 5u > root) {
			for (u32 idx = rr; idx < rsz; idx += 1u << l) { // at most sixteen root slots
				lut[idx] = e;
			}
		}
	}
	if (wave < 6u) {
		// A very short code owns many root slots (a 4-bit literal code 32 of the 512): the wavefront fills those
		// together, a symbol at a time, instead of waiting for the lane that owns the most.  (lut, root and rsz
		// are the same for all lanes of a wavefront: the distance symbols have wavefront 5 to themselves.)
		u64 heavy = otz_ballot (l != 0 && l + 5u <= root);
		while (heavy != 0) {
			const u32 s = otz_ctz64 (heavy);
			heavy &= heavy - 1u;
			const u32 ls = otz_readlane (l, s), rs = otz_readlane (rr, s), es = otz_readlane (e, s);
			for (u32 idx = rs + (lane << ls); idx < rsz; idx += 64u << ls) {
				lut[idx] = es;
			}
		}
	}
	otz_sync ();
	OTZ_TBMARK (9);
	// sub-table placement: one prefix sum over the root slots of both tables (literal
	// sizes in the low half of the scanned word, distance sizes in the high half)
	u32 el = L.lit_lut[tid < LRSZ ? tid : 0u], ed = L.dist_lut[tid < DRSZ ? tid : 0u];
	// a root slot claimed by longer codes holds (widest code - root) << 8 at this point
	u32 ml = tid < LRSZ && el && (el & 0xFFu) == 0 ? 1u << (el >> 8) : 0u;
	u32 md = tid < DRSZ && ed && (ed & 0xFFu) == 0 ? 1u << (ed >> 8) : 0u;
	u32 total;
	u32 incl = otz_block_scan_incl (ml | (md << 16), L.scan, &total);
	const bool fits = (total & 0xFFFFu) <= OTZ_LIT_LUT - LRSZ && (total >> 16) <= OTZ_DIST_LUT - DRSZ;
	if (fits) {
		if (ml) {
			L.lit_lut[tid] = OTZ_LUT_LINK | ((LRSZ + (incl & 0xFFFFu) - ml) << 16) | ((el >> 8) << 8) | OTZ_LUT_NONE;
		}
		if (md) {
			L.dist_lut[tid] = OTZ_LUT_LINK | ((DRSZ + (incl >> 16) - md) << 16) | ((ed >> 8) << 8) | OTZ_LUT_NONE;
		}
	}
	otz_sync ();
	OTZ_TBMARK (5);
	if (!fits) {
		return false;
	}
	if (l > root) {
		u32 link = lut[rr & (rsz - 1u)];
		u32 sb = (link >> 8) & 15u, base = (link >> 16) & 0x7FFFu;
		for (u32 idx = rr >> root; idx < (1u << sb); idx += 1u << (l - root)) {
			lut[base + idx] = e;
		}
	}
	otz_sync ();
	return true;
}

// ------------------------------------------------------------ block headers

// Block header, every work-item for itself: BFINAL/BTYPE and, for a dynamic block,
// HLIT/HDIST/HCLEN (ref deflate-dec.inc.c:613-627, :122-156) are read straight from the
// staged bits, so the verdict and the fields are uniform without a broadcast.
// Wavefront 0 also builds the 7-bit lookup table of the code-length code: lane s takes
// symbol s (0..18), its 3-bit length sits at slot inv_order[s] of the header; ranks and
// per-length counts come from ballots, canonical first codes from a 7-step recurrence
// in registers; L.flag = 0 reports an over-subscribed set.
// p = stage-relative bit position, avail = input bits left from p.
struct otz_hdr {
	int st; // 0, or the status that ends the entry
	u32 bfinal, btype, hlit, hdist, bits;
};
OTZ_DEV otz_hdr otz_parse_header_par(otz_inflate_lds &L, u32 p, u64 avail) {
	otz_hdr H;
	H.st = 0;
	H.bfinal = H.btype = H.hlit = H.hdist = 0;
	H.bits = 3;
	const u32 w = p >> 5, sh = p & 31u;
	const u32 d0 = L.stage_w[w], d1 = L.stage_w[w + 1u];
	const u32 lo = otz_alignbit (d1, d0, sh);
	if (avail < 3u) {
		H.st = OZG_ST_BUF;
		return H;
	}
	H.bfinal = lo & 1u;
	H.btype = (lo >> 1) & 3u;
	if (H.btype == 3u) {
		H.st = OZG_ST_DATA;
		return H;
	}
	if (H.btype != 2u) {
		return H;
	}
	H.hlit = ((lo >> 3) & 31u) + 257u;
	H.hdist = ((lo >> 8) & 31u) + 1u;
	const u32 hclen = ((lo >> 13) & 15u) + 4u;
	H.bits = 17u + 3u * hclen;
	// reference limit: LL_LENGTHS_MAX = 318 (:165-168); beyond it the reference reads
	// outside ll_lengths[] -- fail cleanly instead
	if (avail < H.bits || H.hlit + H.hdist > 318u) {
		H.st = OZG_ST_DATA;
		return H;
	}
	if (otz_wave () == 0) {
		const u32 lane = otz_lane ();
		L.cl_lut[lane] = 0;
		L.cl_lut[lane + 64u] = 0;
		otz_wave_sync ();
		// slot of symbol s in the transmitted order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 (5 bits each)
		const u64 inv_lo = 3ull | (17ull << 5) | (15ull << 10) | (13ull << 15) | (11ull << 20) | (9ull << 25) | (7ull << 30) | (5ull << 35) |
			(4ull << 40) | (6ull << 45) | (8ull << 50) | (10ull << 55);                                     // symbols 0..11
		const u64 inv_hi = 12ull | (14ull << 5) | (16ull << 10) | (18ull << 15) | (0ull << 20) | (1ull << 25) | (2ull << 30); // symbols 12..18
		u32 l = 0;
		if (lane < 19u) {
			u32 slot = (u32) ((lane < 12u ? inv_lo >> (5u * lane) : inv_hi >> (5u * (lane - 12u))) & 31u);
			if (slot < hclen) {
				u32 at = p + 17u + 3u * slot;
				l = otz_alignbit (L.stage_w[(at >> 5) + 1u], L.stage_w[at >> 5], at & 31u) & 7u;
			}
		}
		const u64 below = ((u64)1 << lane) - 1u;
		u32 code = 0, kraft = 0, my_first = 0, rank = 0;
		for (u32 k = 1; k <= 7u; k++) {
			u64 m = otz_ballot (l == k);
			u32 cnt = otz_popc64 (m);
			if (l == k) {
				my_first = code;
				rank = otz_popc64 (m & below);
			}
			code = (code + cnt) << 1;
			kraft += cnt << (7u - k);
		}
		if (l) {
			u32 rr = otz_brev (my_first + rank) >> (32u - l);
			for (u32 idx = rr; idx < 128u; idx += 1u << l) {
				L.cl_lut[idx] = (lane << 8) | l;
			}
		}
		if (lane == 0) {
			L.flag = kraft <= 128u ? 1u : 0u;
		}
	}
	return H;
}

// *p = min (*p, lowest work-item id of this wavefront for which pred holds): one ballot and
// at most one LDS atomic per wavefront.  (A per-lane atomic min is serialised by the
// compiler into a loop over the active lanes.)  Wavefront-uniform control flow only.
OTZ_DEV void otz_wave_min_tid(u32 *p, bool pred) {
	const u64 m = otz_ballot (pred);
	const u32 lo = (u32)m, hi = (u32) (m >> 32);
	const u32 first = lo ? otz_ctz (lo) : 32u + otz_ctz (hi | 0x80000000u);
	if (m != 0 && otz_lane () == first) {
		otz_lds_min (p, otz_tid ());
	}
}

// Every work-item calls this with the same st: the barriers keep the write to
// L.status away from work-items that are still reading it.
template <class LDS>
OTZ_DEV void otz_fail_all(LDS &L, int st) {
	otz_sync ();
	if (otz_tid () == 0) {
		otz_set_status (L, st);
	}
	otz_sync ();
}

// All work-items: decode the hlit+hdist code lengths of a dynamic block in parallel
// (ref deflate-dec.inc.c:164-253: symbols 0-15 are lengths, 16 repeats the previous
// length 3-6 times, 17/18 write 3-10 / 11-138 zeros; a bad code, a repeat with no
// previous length, a run past hlit+hdist or running out of input is a data error).
//   1. every bit position i of the length section decodes the one code-length token
//      that would start there (7-bit LUT + extra bits);
//   2. every position chains 8 of those tokens: jump[i] = (position after them, lengths
//      they produce);
//   3. work-item 0 follows the jumps from position 0 -- at most 41 dependent steps
//      instead of one per token -- and notes where each group of 8 tokens starts;
//   4. one work-item per group expands its tokens into ll[]; "repeat previous" runs
//      (symbol 16) are written as a marker and filled in afterwards.
// Scratch (token and jump tables) lives in the ring slots of the not-yet-written
// output region starting at ring index `scr` (4-byte aligned, OTZ_RMAX bytes free).
#define OTZ_HP_NPOS 2560u // >= 7 bits x 318 lengths + one token
#define OTZ_HP_GROUP 8u
#define OTZ_HP_HEADS 64u
// scr < OTZ_RING and 4-byte aligned; the tables may wrap around the end of the ring once
OTZ_DEV u32 &otz_scr32(otz_inflate_lds &L, u32 scr, u32 i) {
	u32 at = scr + 4u * i;
	at -= at >= OTZ_RING ? OTZ_RING : 0u;
	return *reinterpret_cast<u32 *> (&L.ring[at]);
}

OTZ_DEV void otz_parse_lengths_par(otz_inflate_lds &L, u32 p, u64 avail, u32 scr, u32 n) {
	static_assert (OTZ_NT >= 320 && OTZ_NT >= OTZ_HP_HEADS * OTZ_HP_GROUP, "one work-item per code length / per token");
	static_assert ((2u * OTZ_HP_NPOS + OTZ_HP_HEADS) * 4u <= OTZ_RMAX - 4u, "scratch fits the free ring region");
	static_assert (OTZ_HP_NPOS % OTZ_NT == 0, "whole chunks");
	const u32 tid = otz_tid ();
	// a chain that is still valid has spent at most 7 bits per length
	const u32 npos = 7u * n + 16u < OTZ_HP_NPOS ? 7u * n + 16u : OTZ_HP_NPOS;
	for (u32 i = tid; i < npos; i += OTZ_NT) {
		u32 pos = p + i;
		u32 w = pos >> 5, sh = pos & 31u;
		u32 bits = otz_alignbit (L.stage_w[w + 1], L.stage_w[w], sh);
		u32 e = L.cl_lut[bits & 127u];
		u32 l = e & 15u, sym = e >> 8;
		u32 info = 0;
		if (l) {
			u32 xb = 0, cnt = 1, val = sym;
			if (sym >= 16u) {
				xb = sym == 16u ? 2u : sym == 17u ? 3u : 7u;
				cnt = ((bits >> l) & ((1u << xb) - 1u)) + (sym == 18u ? 11u : 3u);
				val = sym == 16u ? 0xFFu : 0u;
			}
			u32 nb = l + xb;
			if ((u64)i + nb <= avail) {
				info = nb | (cnt << 4) | (val << 12) | (1u << 20);
			}
		}
		otz_scr32 (L, scr, i) = info;
	}
	otz_sync ();
	// jumps are built one chunk of OTZ_NT positions at a time, only as far as the chain
	// from position 0 actually reaches (typical headers end inside the first chunk)
	u32 wpos = 0, widx = 0, wg = 0; // work-item 0: the chain walk
	u32 groups = 0;
	for (u32 base = 0; base < npos; base += OTZ_NT) {
		{
			u32 i = base + tid, j = i, c = 0;
			for (u32 t = 0; t < OTZ_HP_GROUP && j < npos; t++) {
				u32 info = otz_scr32 (L, scr, j);
				if (!(info >> 20)) {
					break;
				}
				c += (info >> 4) & 0xFFu;
				j += info & 15u;
			}
			otz_scr32 (L, scr, OTZ_HP_NPOS + i) = j | (c << 16);
		}
		otz_sync ();
		if (tid == 0) {
			u32 verdict = 0; // 1: chain complete, 2: bad
			while (widx < n && wpos < base + OTZ_NT) {
				u32 jv = otz_scr32 (L, scr, OTZ_HP_NPOS + wpos);
				u32 j = jv & 0xFFFFu;
				if (j == wpos || wg == OTZ_HP_HEADS) { // no valid token here: bad code or out of input
					verdict = 2;
					break;
				}
				otz_scr32 (L, scr, 2u * OTZ_HP_NPOS + wg) = wpos | (widx << 16);
				wg++;
				wpos = j;
				widx += jv >> 16;
			}
			if (widx >= n) {
				verdict = 1;
			} else if (wpos >= npos) {
				verdict = 2;
			}
			// verdicts travel in their own words: L.status is only written between
			// barriers that no work-item can be ahead of (otz_fail_all)
			L.flag = wg | (verdict << 16);
			L.first_rej = 0;
		}
		otz_sync ();
		groups = L.flag;
		if (groups >> 16) {
			break;
		}
	}
	if ((groups >> 16) != 1u) {
		otz_fail_all (L, OZG_ST_DATA);
		return;
	}
	groups &= 0xFFFFu;
	// one work-item per token: skip the tokens before it in its group, then expand it
	if (tid < groups * OTZ_HP_GROUP) {
		u32 hv = otz_scr32 (L, scr, 2u * OTZ_HP_NPOS + tid / OTZ_HP_GROUP);
		u32 pos = hv & 0xFFFFu, idx = hv >> 16;
		u32 info = otz_scr32 (L, scr, pos);
		for (u32 t = tid % OTZ_HP_GROUP; t > 0 && (info >> 20) && idx < n; t--) {
			idx += (info >> 4) & 0xFFu;
			pos += info & 15u;
			info = pos < npos ? otz_scr32 (L, scr, pos) : 0u;
		}
		if ((info >> 20) && idx < n) {
			u32 cnt = (info >> 4) & 0xFFu, val = (info >> 12) & 0xFFu;
			if ((val == 0xFFu && idx == 0) || idx + cnt > n) {
				otz_lds_or (&L.first_rej, 1u);
			} else {
				for (u32 k = 0; k < cnt; k++) {
					L.ll[idx + k] = (u8)val;
				}
				if (idx + cnt == n) {
					L.hdr_bits = pos + (info & 15u);
				}
			}
		}
	}
	otz_sync ();
	if (L.first_rej != 0) {
		otz_fail_all (L, OZG_ST_DATA);
		return;
	}
	// "repeat previous" markers take the nearest earlier real length
	u32 v = tid < n ? L.ll[tid] : 0u;
	const bool marker = v == 0xFFu;
	if (marker) {
		u32 j = tid;
		do {
			j--;
			v = L.ll[j];
		} while (v == 0xFFu);
	}
	otz_sync ();
	if (marker) {
		L.ll[tid] = (u8)v;
	}
	otz_sync ();
}

// ------------------------------------------------------------------- passes

// The visited-token-start bitmap of a lane's pass-A path lives in registers, as two
// 64-bit halves (a runtime-indexed array would be demoted to scratch memory, i.e.
// global-memory round trips inside the hottest loop).
static_assert (OTZ_S_BITS == 128u, "otz_vis covers a 128-bit range");
struct otz_vis {
	u64 lo, hi;
};
OTZ_DEV void otz_vis_set(otz_vis &v, u32 rel) {
	u64 bit = (u64)1 << (rel & 63u);
	v.lo |= rel < 64u ? bit : 0u;
	v.hi |= rel < 64u ? 0u : bit;
}
OTZ_DEV bool otz_vis_test(const otz_vis &v, u32 rel) {
	return (((rel < 64u ? v.lo : v.hi) >> (rel & 63u)) & 1u) != 0;
}
// visited positions below rel
OTZ_DEV u32 otz_vis_rank(const otz_vis &v, u32 rel) {
	u64 below = ((u64)1 << (rel & 63u)) - 1u;
	return rel < 64u ? otz_popc64 (v.lo & below) : otz_popc64 (v.lo) + otz_popc64 (v.hi & below);
}

struct otz_path {
	u32 exit;  // exit word
	u32 bytes; // output bytes of the tokens walked
	u32 syncq; // MODE_B: position where the walk met pass A's path, or ~0u
};

#define OTZ_MODE_A 0      // from the speculative start; records visited token starts
#define OTZ_MODE_B 1      // from the true start; stops on a visited position
#define OTZ_MODE_PREFIX 2 // from the speculative start up to `stop`

// Pass A also notes, per visited token start, the output bytes its path produced before
// that token (u16, first OTZ_CUM_CAP tokens).  The notes live in the ring slots of the
// round's own not-yet-written output region, which is free until placement.  `notes`
// points at the lane's record; note_cap is 0 for the one lane whose record would
// straddle the end of the ring (it falls back to re-walking its prefix).
#define OTZ_WARM_BITS 64u // pass A's head start (must stay below OTZ_S_BITS: lane 1 starts inside the stage)
#define OTZ_CUM_CAP 23u
#define OTZ_CUM_STRIDE (2u * OTZ_CUM_CAP)

// A range of OTZ_S_BITS bits holds at most OTZ_S_BITS tokens, of which at most half are
// matches (a match is at least two bits): a lane's path yields < 64 * 258 + 64 bytes,
// which always fits a round (OTZ_RMAX) and a u16 note.
template <int MODE>
OTZ_DEV otz_path otz_walk(otz_inflate_lds &L, u32 start, u32 s, u32 e, u32 limit_bits, u32 stop, otz_vis &vis, u16 *notes = nullptr, u32 note_cap = 0) {
	otz_path P;
	u32 pos = start, ntok = 0, bytes = 0;
	const u32 end = MODE == OTZ_MODE_PREFIX ? stop : e;
	// The loop body is straight-line: ONE condition keeps a lane in the loop, everything
	// else is a select.  (Several `break`s cost more scalar mask bookkeeping per step than
	// the decode itself.)  A lane that stops this step still decodes its token -- it leaves
	// the loop right after, so that costs the wavefront nothing.
	bool stopped = false, hit = false;
	u32 last_kind = OTZ_K_LIT, last_end = 0;
	otz_win W;
	otz_win_init (W, L, pos);
	while (pos < end && !stopped) {
		const u32 rel = pos - s; // MODE_B/PREFIX: pos >= s always (a left neighbour exits at or after s)
		// MODE_A may start before s (warm-up, see pass A): those tokens are neither
		// recorded nor counted
		const bool counted = MODE != OTZ_MODE_A || pos >= s;
		if (MODE == OTZ_MODE_A && counted) {
			otz_vis_set (vis, rel);
			if (ntok < note_cap) {
				notes[ntok] = (u16)bytes;
			}
		}
		if (MODE == OTZ_MODE_B) {
			hit = otz_vis_test (vis, rel);
		}
		const otz_tok t = otz_decode_adv (L, W, pos);
		last_kind = t.kind;
		last_end = pos + t.nbits;
		stopped = hit || t.kind >= OTZ_K_EOB || last_end > limit_bits;
		pos = stopped ? pos : last_end;
		const bool adv = counted && !stopped;
		ntok += adv ? 1u : 0u;
		bytes += adv ? (t.kind == OTZ_K_LIT ? 1u : t.val) : 0u;
	}
	// how the walk ended; pos is the start of the token it stopped at, or the first token start
	// at or beyond `end`
	u32 xk = OTZ_X_NORMAL;
	P.syncq = hit ? pos : ~0u;
	if (stopped && !hit) {
		if (last_kind == OTZ_K_BAD) {
			xk = pos >= limit_bits ? OTZ_X_TRUNC : OTZ_X_INVALID;
		} else if (last_end > limit_bits) {
			xk = OTZ_X_TRUNC;
		} else {
			xk = OTZ_X_EOB; // position just after the end-of-block code
			pos = last_end;
		}
	}
	P.exit = OTZ_X_MAKE (xk, pos);
	P.bytes = bytes;
	return P;
}

OTZ_DEV void otz_mark_final(otz_inflate_lds &L, u32 a, u32 b) {
	// bits [a, b) of the round-relative final bitmap
	if (a >= b) {
		return;
	}
	otz_fence_release ();
	u32 w0 = a >> 5, w1 = (b - 1u) >> 5;
	u32 mlo = 0xFFFFFFFFu << (a & 31u), mhi = 0xFFFFFFFFu >> (31u - ((b - 1u) & 31u));
	if (w0 == w1) {
		otz_lds_or (&L.fin[w0], mlo & mhi);
		return;
	}
	otz_lds_or (&L.fin[w0], mlo);
	for (u32 w = w0 + 1u; w < w1; w++) {
		otz_lds_or (&L.fin[w], 0xFFFFFFFFu);
	}
	otz_lds_or (&L.fin[w1], mhi);
}

OTZ_DEV bool otz_all_final(const otz_inflate_lds &L, u32 a, u32 b) {
	if (a >= b) {
		return true;
	}
	u32 w0 = a >> 5, w1 = (b - 1u) >> 5;
	u32 mlo = 0xFFFFFFFFu << (a & 31u), mhi = 0xFFFFFFFFu >> (31u - ((b - 1u) & 31u));
	bool ok;
	if (w0 == w1) {
		u32 m = mlo & mhi;
		ok = (otz_lds_load (&L.fin[w0]) & m) == m;
	} else {
		ok = (otz_lds_load (&L.fin[w0]) & mlo) == mlo && (otz_lds_load (&L.fin[w1]) & mhi) == mhi;
		for (u32 w = w0 + 1u; ok && w < w1; w++) {
			ok = otz_lds_load (&L.fin[w]) == 0xFFFFFFFFu;
		}
	}
	otz_fence_acquire ();
	return ok;
}

// Pass C1 ("place"): every lane decodes its tokens again from its true start.
// Literals go straight to their final bytes.  A match is NOT copied here: its 23
// bits (distance-1 : 15, length-3 : 8) are parked in the first three bytes of its
// own destination (a match is at least three bytes long, so the space is always
// there), and the destination's bit is set in the match-start bitmap.  No lane
// waits for another lane in this pass.
OTZ_DEV void otz_place(otz_inflate_lds &L, bool active, u32 start, u32 exit_word, u32 o, u32 round_base, u64 stage_abs_bit) {
	if (!active) {
		return;
	}
	const u32 exit_pos = OTZ_X_POS (exit_word);
	u32 pos = start;
	u32 rel = o - round_base; // round-relative output position
	u32 ri = OTZ_RIDX (o);    // its ring index
	// bits of the bitmap word being filled: literal bytes (final) and match starts.  They
	// go to LDS when the output moves on to the next word; nobody reads the bitmaps
	// before the barrier after this pass, so no ordering is needed.
	u32 curw = rel >> 5, finm = 0, stm = 0;
	u32 kind = OTZ_K_LIT, tpos = pos, tbits = 0;
	bool stopped = false, too_far = false;
	otz_win W;
	otz_win_init (W, L, pos);
	// straight-line body, one loop condition (see otz_walk)
	while (pos < exit_pos && !stopped) {
		const otz_tok t = otz_decode_adv (L, W, pos);
		kind = t.kind;
		tpos = pos;
		tbits = t.nbits;
		const bool lit = kind == OTZ_K_LIT;
		// ref reads window bytes nobody wrote (:524): no parity, fail cleanly
		too_far = kind == OTZ_K_LEN && t.dist > round_base + rel;
		stopped = kind >= OTZ_K_EOB || too_far;
		const u32 w = rel >> 5;
		if (w != curw) {
			otz_lds_or (&L.fin[curw], finm); // (an OR of zero is cheaper than a branch around it)
			otz_lds_or (&L.mstart[curw], stm);
			curw = w;
			finm = stm = 0;
		}
		if (!stopped) {
			const u32 bit = 1u << (rel & 31u);
			const u32 d = (t.dist - 1u) | ((t.val - 3u) << 15); // the parked match
			L.ring[ri] = (u8) (lit ? t.val : d);
			if (!lit) {
				L.ring[OTZ_RWRAP (ri + 1u)] = (u8) (d >> 8);
				L.ring[OTZ_RWRAP (ri + 2u)] = (u8) (d >> 16);
			}
			finm |= lit ? bit : 0u;
			stm |= lit ? 0u : bit;
			const u32 adv = lit ? 1u : t.val;
			rel += adv;
			ri = OTZ_RWRAP (ri + adv);
			pos += t.nbits;
		}
	}
	if (too_far) {
		otz_set_status (L, OZG_ST_DATA);
	}
	if (finm) {
		otz_lds_or (&L.fin[curw], finm);
	}
	if (stm) {
		otz_lds_or (&L.mstart[curw], stm);
	}
	if (kind == OTZ_K_EOB) {
		u64 a = stage_abs_bit + tpos;
		L.eob_start_lo = (u32)a;
		L.eob_start_hi = (u32) (a >> 32);
	} else if (kind == OTZ_K_BAD) {
		L.dbg[0] = 2; // validated path cannot hold a bad token
		L.dbg[1] = otz_tid ();
		L.dbg[2] = tpos;
		L.dbg[3] = kind | (tbits << 8);
		otz_set_status (L, OZG_ST_INTERNAL);
	}
}

// Pass C2 ("resolve"): the LZ77 copies, all lanes at once.  Lane j owns the
// matches that START in the 8-byte granules j, j + OTZ_NT, ... of the round's output
// (strided, so that every lane works on the same stretch of the round at the same
// time and the matches it depends on lie mostly in stretches that are already done);
// it keeps trying the pending matches of a granule round-robin and copies one as soon
// as every source byte is final (literal, older round, or an already resolved
// match).  The earliest unresolved match of the round is always ready, so the
// loop cannot stall.  One loop per wavefront, closed by a ballot: a convergent
// operation keeps the compiler from peeling the "not ready yet" path into an
// inner loop, which would park the lanes being waited for (SIMT deadlock).
#define OTZ_RES_GRAN_LOG 3u // a lane owns the matches that start in 8-byte granules g = lane, lane + OTZ_NT, ...
#define OTZ_RES_CHUNK 16u   // bytes copied per step (two 8-byte LDS accesses each way)
OTZ_DEV void otz_resolve(otz_inflate_lds &L, u32 round_base, u32 round_bytes) {
	constexpr u32 GB = 1u << OTZ_RES_GRAN_LOG, GMASK = (1u << GB) - 1u;
	constexpr u32 FIN_WORDS = OTZ_RMAX / 32u;
	const u32 ngran = (round_bytes + GB - 1u) >> OTZ_RES_GRAN_LOG;
	const u32 rb = OTZ_RIDX (round_base); // ring index of the round's first byte
	u32 g = otz_tid ();
	u32 pending = g < ngran ? (L.mstart[g >> (5u - OTZ_RES_GRAN_LOG)] >> (GB * (g & (32u / GB - 1u)))) & GMASK : 0u;
	u32 cand = pending;
	// the candidate being polled: cb = its bit in the granule (32: none), cd = its parked descriptor
	u32 cb = 32u, cd = 0;
	// the match being copied: o = round-relative destination, ri/si = ring indices of
	// destination and source, pend = bytes left
	u32 o = 0, ri = 0, si = 0, pend = 0, pdist = 1, spins = 0;
	bool done = false;
	for (;;) {
		// The body is written as selects wherever a branch would only guard a few register
		// moves: every extra branch level costs the wavefront half a dozen scalar mask
		// instructions per iteration (cf. otz_walk).
		// ---- 1. who needs a candidate: next granule, or next pending match of this one
		const bool idle = !done && pend == 0;
		const bool adv = idle && pending == 0;
		if (adv) {
			g += OTZ_NT;
			const bool has = g < ngran;
			const u32 gi = has ? g : 0u;
			const u32 bits = (L.mstart[gi >> (5u - OTZ_RES_GRAN_LOG)] >> (GB * (gi & (32u / GB - 1u)))) & GMASK;
			pending = has ? bits : 0u;
			cand = pending;
			cb = 32u;
			done = !has;
		}
		bool progressed = adv && !done;
		const bool poll = idle && !adv;
		const u32 pool = cand ? cand : pending; // round-robin over the granule's pending matches
		const u32 b = poll ? otz_ctz (pool | 0x80000000u) : 0u;
		cand = poll ? pool & (pool - 1u) : cand;
		const u32 at = GB * g + b; // round-relative start of the candidate
		const u32 ai = OTZ_RWRAP (rb + at);
		if (poll && b != cb) { // a retry of the same candidate keeps its descriptor
			cd = otz_ld32u (&L.ring[ai]) & 0xFFFFFFu; // (may read past the ring's end: still inside the LDS block)
			if (ai + 3u > OTZ_RING) {
				cd = (u32)L.ring[ai] | ((u32)L.ring[OTZ_RWRAP (ai + 1u)] << 8) | ((u32)L.ring[OTZ_RWRAP (ai + 2u)] << 16);
			}
			cb = b;
		}
		const u32 dist = (cd & 0x7FFFu) + 1u, len = (cd >> 15) + 3u;
		// source bytes [at - dist, at - dist + min(len, dist)) must be final; everything
		// before the round already is
		const u32 span = len < dist ? len : dist;
		const u32 fa = at > dist ? at - dist : 0u, fb = at + span - dist; // round-relative [fa, fb) when fb > 0
		const u32 csi = ai >= dist ? ai - dist : ai + OTZ_RING - dist;
		// ---- 2. one LDS round trip: the final bits of the candidate's source AND the
		// next OTZ_RES_CHUNK source bytes (of the candidate, or of the match in flight)
		const u32 rs = pend != 0 ? si : csi;
		const u32 w0 = fa >> 5;
		u32 f0 = otz_lds_load (&L.fin[w0 < FIN_WORDS ? w0 : 0u]);
		u32 f1 = otz_lds_load (&L.fin[w0 + 1u < FIN_WORDS ? w0 + 1u : 0u]);
		otz_order_loads ();
		// reads past the bytes used (and past the ring's end) land in the LDS block and are ignored
		u64 lo = otz_ld64u (&L.ring[rs]), hi = otz_ld64u (&L.ring[rs + 8u]);
		const bool before = dist >= at + span; // source entirely before the round
		const u32 flen = fb - fa;             // meaningful when !before
		const u64 need = ((((u64)1 << (flen & 63u)) - 1u) << (fa & 31u));
		const u64 have = (u64)f0 | ((u64)f1 << 32);
		bool ready = poll && (before || (flen <= 32u && (have & need) == need));
		if (poll && !before && flen > 32u) { // long source span: the general check
			ready = otz_all_final (L, fa, fb);
			if (ready) { // the bytes read above may predate the final ones
				lo = otz_ld64u (&L.ring[csi]);
				hi = otz_ld64u (&L.ring[csi + 8u]);
			}
		}
		o = ready ? at : o;
		ri = ready ? ai : ri;
		si = ready ? csi : si;
		pend = ready ? len : pend;
		pdist = ready ? dist : pdist;
		pending &= ~(ready ? 1u << b : 0u);
		cb = ready ? 32u : cb;
		if (poll && !ready && (++spins & 63u) == 0) {
			if (otz_get_status (L) != 0) {
				done = true;
			} else if (spins > OTZ_SPIN_LIMIT) {
				L.dbg[0] = 3; // waited too long for source bytes
				L.dbg[1] = otz_tid ();
				L.dbg[2] = round_base + at;
				L.dbg[3] = dist | (len << 16);
				otz_set_status (L, OZG_ST_INTERNAL);
				done = true;
			}
		}
		// ---- 3. copy one step
		if (!done && pend != 0) {
			progressed = true;
			// Up to OTZ_RES_CHUNK bytes that neither overlap their own source nor cross
			// the end of the ring, so every byte is ring[si + k] -> ring[ri + k] with
			// constant k.  An overlapping match (distance < chunk) copies one period,
			// then two, four, ... from the same source.
			u32 n = pend < OTZ_RES_CHUNK ? pend : OTZ_RES_CHUNK;
			n = n < pdist ? n : pdist;
			n = n < OTZ_RING - ri ? n : OTZ_RING - ri;
			n = n < OTZ_RING - si ? n : OTZ_RING - si;
			u8 *dst = &L.ring[ri];
			if (n == OTZ_RES_CHUNK) {
				otz_st64u (dst, lo);
				otz_st64u (dst + 8, hi);
			} else {
				u64 v = lo;
				if (n & 8u) {
					otz_st64u (dst, lo);
					v = hi;
					dst += 8;
				}
				if (n & 4u) {
					otz_st32u (dst, (u32)v);
					v >>= 32;
					dst += 4;
				}
				if (n & 2u) {
					otz_st16u (dst, (u16)v);
					v >>= 16;
					dst += 2;
				}
				if (n & 1u) {
					*dst = (u8)v;
				}
			}
			{
				// bytes [o, o + n) are final: at most two words of the bitmap
				otz_fence_release ();
				u64 mask = (((u64)1 << n) - 1u) << (o & 31u);
				otz_lds_or (&L.fin[o >> 5], (u32)mask);
				if ((u32) (mask >> 32)) {
					otz_lds_or (&L.fin[(o >> 5) + 1u], (u32) (mask >> 32));
				}
			}
			o += n;
			ri = OTZ_RWRAP (ri + n);
			pend -= n;
			// a whole period written: same source, twice the reach; otherwise the source
			// slides along (destination - source stays a multiple of the distance)
			const bool dbl = n == pdist && pdist < OTZ_RES_CHUNK;
			si = dbl ? si : OTZ_RWRAP (si + n);
			pdist += dbl ? n : 0u;
		}
		u64 alive = otz_ballot (!done);
		u64 moved = otz_ballot (progressed);
		if (alive == 0) {
			break;
		}
		if (moved == 0) {
			otz_backoff (); // every live lane of this wavefront waits on another wavefront
		}
	}
}

// Cooperative copy of n input bytes (entry offset src_byte) to the ring at opos.
OTZ_DEV void otz_copy_in(otz_inflate_lds &L, const otz_job &J, u32 src_byte, u32 opos, u32 n) {
	for (u32 i = otz_tid (); i < n; i += OTZ_NT) {
		L.ring[OTZ_RIDX (opos + i)] = J.in[src_byte + i];
	}
}
OTZ_DEV void otz_fill_zero(otz_inflate_lds &L, u32 opos, u32 n) {
	for (u32 i = otz_tid (); i < n; i += OTZ_NT) {
		L.ring[OTZ_RIDX (opos + i)] = 0;
	}
}

// ----------------------------------------------------------------- the entry

// Decode one entry. All OTZ_NT work-items of the workgroup call this together.
// Work-item 0 writes *res.
OTZ_DEV void otz_inflate_entry(otz_inflate_lds &L, const otz_job &J, u32 expect_crc, ozg_result *res) {
	const u32 tid = otz_tid ();
	u64 abs_bit = 0;               // next input bit
	const u64 total_bits = (u64)J.n_in * 8u;
	u32 opos = 0, flushed = 0;     // bytes produced / already in global memory
	u32 crc_reg = 0xFFFFFFFFu;
	u32 flags = 0;
	bool last_block = false, ended = false;
	u32 last_btype = 0; // type of the block decoded last

	{
		// uniform start-up verdict; nobody reads L.status before the barrier
		int st0 = 0;
		if (((uintptr_t)J.out & 15u) != 0) {
			st0 = OZG_ST_INTERNAL;
		} else if (J.n_in == 0) {
			// ref: the loop at :610 never runs; Z_OK when there is no output space, else Z_BUF_ERROR
			st0 = J.out_cap == 0 ? OTZ_ST_NEED_OUT_INT : OZG_ST_BUF;
		}
		if (tid == 0) {
			otz_set_status (L, st0);
			L.dbg[0] = st0 == OZG_ST_INTERNAL ? 1u : 0u;
			L.dbg[1] = L.dbg[2] = L.dbg[3] = 0;
			for (u32 i = 0; i < 16; i++) {
				L.prof[i] = 0;
			}
			L.prof_last = otz_cycles ();
			L.prof[OTZ_P_TOTAL] = 0 - L.prof_last;
			L.eob_start_lo = L.eob_start_hi = 0;
		}
		otz_sync ();
	}

	while (otz_get_status (L) == 0 && !ended) {
		// ---- block header
		u32 byte0 = (u32) (abs_bit >> 3);
		// the header decode uses the free output region of the ring as scratch; flushing
		// must come before the stage load (the flush borrows the stage buffer)
		otz_make_room (L, J, opos, OTZ_RMAX, flushed, crc_reg);
		otz_sync ();
		u32 skew = otz_stage_load (L, J, byte0);
		otz_sync ();
		u32 p0 = skew * 8u + (u32) (abs_bit & 7u);
		u64 stage_abs = abs_bit - p0; // absolute bit position of stage bit 0 (may wrap below zero; only sums are used)
		OTZ_MARK (L, J, OTZ_P_STAGE);
		const otz_hdr H = otz_parse_header_par (L, p0, total_bits - abs_bit);
		otz_sync ();
		OTZ_MARK (L, J, OTZ_P_HEADER);
		OTZ_COUNT (L, J, OTZ_P_BLOCKS, 1);
		if (H.st != 0 || (H.btype == 2u && L.flag == 0u)) {
			otz_fail_all (L, H.st != 0 ? H.st : OZG_ST_DATA);
			break;
		}
		last_block = H.bfinal != 0;
		const u32 btype = H.btype;
		last_btype = btype;
		abs_bit += H.bits;
		if (btype == 0) {
			// ---- stored block (ref read_uncompressed_block :269-319)
			abs_bit = (abs_bit + 7u) & ~(u64)7u;
			u32 b = (u32) (abs_bit >> 3);
			if (J.n_in - b < 4u || b > J.n_in) {
				otz_fail_all (L, OZG_ST_DATA);
				break;
			}
			u32 len = (u32)J.in[b] | ((u32)J.in[b + 1] << 8);
			u32 nlen = (u32)J.in[b + 2] | ((u32)J.in[b + 3] << 8);
			b += 4;
			int st = 0;
			if (len != ((~nlen) & 0xFFFFu)) {
				st = OZG_ST_DATA;
			} else if (J.n_in - b < len) {
				st = OZG_ST_DATA;
			} else if (J.out_cap - opos < len) {
				st = OZG_ST_BUF; // ref :299-301
			}
			if (st != 0) {
				otz_fail_all (L, st);
				break;
			}
			u32 done = 0;
			while (done < len) {
				u32 n = len - done;
				if (n > OTZ_RMAX) {
					n = OTZ_RMAX;
				}
				otz_make_room (L, J, opos, n, flushed, crc_reg);
				otz_copy_in (L, J, b + done, opos, n);
				opos += n;
				done += n;
				otz_sync ();
			}
			abs_bit = ((u64)b + len) * 8u;
			OTZ_MARK (L, J, OTZ_P_STORED);
			if (last_block) {
				ended = true;
			}
			continue;
		}
		// ---- Huffman block: code lengths
		u32 hlit = 288, hdist = 32;
		if (btype == 1) {
			for (u32 i = tid; i < 320u; i += OTZ_NT) {
				L.ll[i] = (u8) (i <= 143u ? 8 : i <= 255u ? 9 : i <= 279u ? 7 : i <= 287u ? 8 : 5);
			}
			otz_sync ();
		} else {
			// the run-length coded lengths, through the 7-bit LUT of the code-length code
			hlit = H.hlit;
			hdist = H.hdist;
			otz_parse_lengths_par (L, p0 + H.bits, total_bits - abs_bit, OTZ_RIDX ((opos + 3u) & ~3u), hlit + hdist);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_LENGTHS);
			if (otz_get_status (L) != 0) {
				break;
			}
			abs_bit += L.hdr_bits;
		}
		if (!otz_build_tables (L, hlit, hdist)) {
			otz_fail_all (L, OZG_ST_DATA);
			break;
		}

		OTZ_MARK (L, J, OTZ_P_TABLES);
		// ---- rounds over the block's symbols
		bool block_done = false;
		// The bytes staged for the block header are still there (nothing has flushed since:
		// the room was made before the header): the block's first round reuses them when
		// a whole round still fits behind the header.
		bool staged = true;
		while (!block_done) {
			u32 room = J.out_cap - opos;
			u32 limit = room < OTZ_RMAX ? room : OTZ_RMAX;
			// a full OTZ_RMAX is freed even near the end of the output: passes A/B use
			// the round's output region as scratch
			otz_make_room (L, J, opos, OTZ_RMAX, flushed, crc_reg);
			if (staged && abs_bit - stage_abs + OTZ_NT * OTZ_S_BITS + 8u * 128u <= 8u * OTZ_STAGE_BYTES) {
				p0 = (u32) (abs_bit - stage_abs);
			} else {
				byte0 = (u32) (abs_bit >> 3);
				otz_sync ();
				skew = otz_stage_load (L, J, byte0);
				p0 = skew * 8u + (u32) (abs_bit & 7u);
				stage_abs = abs_bit - p0;
			}
			staged = false;
			u64 lb = total_bits - stage_abs; // input ends at this stage-relative bit
			u32 limit_bits = lb > 0xFFFFFFu ? 0xFFFFFFu : (u32)lb;
			if (tid == 0) {
				L.flag = 0;
				L.first_term = OTZ_NT;
				L.first_rej = OTZ_NT;
				L.next_start = 0;
			}
			otz_sync ();

			OTZ_MARK (L, J, OTZ_P_STAGE);
			OTZ_COUNT (L, J, OTZ_P_ROUNDS, 1);
			// pass A
			const u32 s = p0 + tid * OTZ_S_BITS, e = s + OTZ_S_BITS;
			otz_vis vis = { 0, 0 };
			const u32 note_at = OTZ_RIDX (((opos + 1u) & ~1u) + tid * OTZ_CUM_STRIDE);
			u16 *const notes = reinterpret_cast<u16 *> (&L.ring[note_at]);
			const u32 note_cap = note_at + OTZ_CUM_STRIDE <= OTZ_RING ? OTZ_CUM_CAP : 0u;
			// Lanes other than 0 start OTZ_WARM_BITS early: a Huffman stream decoded from a
			// wrong bit position falls into step with the real token chain after a few
			// tokens, so by the time the path reaches s it is usually already the true
			// one, and pass B's walk meets it at its first token.
			const u32 a_start = tid ? s - OTZ_WARM_BITS : s;
			otz_path A = otz_walk<OTZ_MODE_A> (L, a_start, s, e, limit_bits, 0, vis, notes, note_cap);
			L.exits[tid] = A.exit;
			otz_sync ();

			OTZ_MARK (L, J, OTZ_P_PASS_A);
			// pass B: fixpoint over true starts.  A lane whose left neighbour's path
			// terminated (EOB, cut, bad code, end of input) has no true start and is
			// ignored below: only lanes up to the FIRST terminator are on the chain, so
			// the iteration stops as soon as nothing at or left of it changed (lanes
			// beyond it decode the next block's header as symbols and need not settle).
			u32 t = tid == 0 ? s : ~0u;
			u32 q = tid == 0 ? s : ~0u; // where the true path meets pass A's path (~0u: never)
			u32 wbytes = 0;
			u32 my_exit = A.exit;
			if (tid == 0) {
				L.bmin[0][0] = L.bmin[0][1] = OTZ_NT;
			}
			otz_set_priority_high ();
			for (u32 it = 0; it <= OTZ_NT; it++) {
				u32 slot = it & 1u;
				u32 prev = tid ? L.exits[tid - 1] : 0u;
				otz_sync ();
				if (tid == 0) {
					L.bmin[slot ^ 1u][0] = L.bmin[slot ^ 1u][1] = OTZ_NT;
				}
				bool changed = false;
				if (tid && OTZ_X_KIND (prev) == OTZ_X_NORMAL && OTZ_X_POS (prev) != t) {
					t = OTZ_X_POS (prev);
					otz_path B = otz_walk<OTZ_MODE_B> (L, t, s, e, limit_bits, 0, vis);
					wbytes = B.bytes;
					q = B.syncq;
					u32 ne = q != ~0u ? A.exit : B.exit;
					if (ne != my_exit) {
						my_exit = ne;
						L.exits[tid] = ne;
						changed = true;
					}
				}
				otz_wave_min_tid (&L.bmin[slot][0], changed);
				otz_wave_min_tid (&L.bmin[slot][1], OTZ_X_KIND (my_exit) != OTZ_X_NORMAL);
				otz_sync ();
				OTZ_COUNT (L, J, OTZ_P_B_ITERS, 1);
				u32 first_changed = L.bmin[slot][0], first_term = L.bmin[slot][1];
				if (first_changed == OTZ_NT || first_changed > first_term) {
					break;
				}
			}
			otz_set_priority_normal ();
			OTZ_MARK (L, J, OTZ_P_PASS_B);
			// output bytes of this lane's true path
			u32 bytes = 0;
			if (t != ~0u) {
				if (q == ~0u) {
					bytes = wbytes;
				} else {
					u32 pre = 0;
					if (q != s) {
						// ordinal of q among pass A's token starts -> bytes before it
						u32 ord = otz_vis_rank (vis, q - s);
						if (ord < note_cap) {
							pre = notes[ord];
						} else {
							// re-walk pass A's recorded path from its first token
							otz_vis dummy = { 0, 0 };
							const u32 first = s + (vis.lo ? otz_ctz64 (vis.lo) : 64u + otz_ctz64 (vis.hi));
							pre = otz_walk<OTZ_MODE_PREFIX> (L, first, s, e, limit_bits, q, dummy).bytes;
						}
					}
					bytes = wbytes + A.bytes - pre;
				}
			}
			otz_wave_min_tid (&L.first_term, OTZ_X_KIND (my_exit) != OTZ_X_NORMAL);
			u32 total;
			u32 incl = otz_block_scan_incl (bytes, L.scan, &total);
			const u32 F = L.first_term; // first lane whose path terminates; lanes <= F are the true chain
			bool onchain = tid <= F;
			bool acc = onchain && incl <= limit;
			otz_wave_min_tid (&L.first_rej, !acc);
			otz_sync ();
			const u32 K = L.first_rej; // lanes [0, K) are materialised this round
			if (tid == K && onchain) {
				L.next_start = t; // true start of the first lane that did not fit
				L.rej_bytes_incl = incl;
			}
			if (K > 0 && tid == K - 1u) {
				L.acc_bytes = incl;
			}
			u32 term_exit = F < OTZ_NT ? L.exits[F] : 0u;
			u32 last_exit = L.exits[OTZ_NT - 1u];
			otz_sync (); // exits[] is about to be reused as the final bitmap
			int st = 0;
			u32 tk = OTZ_X_KIND (term_exit);
			if (F < K && (tk == OTZ_X_INVALID || tk == OTZ_X_TRUNC)) {
				st = OZG_ST_DATA; // ref: Z_DATA_ERROR from get_bit / no matching code
			} else if (K <= F && K < OTZ_NT && L.rej_bytes_incl > room) {
				st = OTZ_ST_NEED_OUT_INT; // stream wants more than uncomp_size bytes (ref pending copy/literal, Z_OK)
			}
			if (st != 0) {
				otz_fail_all (L, st);
				break;
			}
			u32 produced = K > 0 ? L.acc_bytes : 0u;
			u32 next_rel;
			if (K > 0 && K - 1u == F) {
				next_rel = OTZ_X_POS (term_exit);
				block_done = tk == OTZ_X_EOB;
			} else if (K == OTZ_NT) {
				next_rel = OTZ_X_POS (last_exit);
			} else {
				next_rel = L.next_start;
			}
			OTZ_MARK (L, J, OTZ_P_SCAN);
			// pass C
			for (u32 i = tid; i < OTZ_RMAX / 32; i += OTZ_NT) {
				L.fin[i] = 0;
				L.mstart[i] = 0;
			}
			otz_sync ();
			otz_place (L, tid < K, tid == 0 ? s : t, my_exit, opos + incl - bytes, opos, stage_abs);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_EMIT);
			if (otz_get_status (L) != 0) {
				break;
			}
			otz_resolve (L, opos, produced);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_RESOLVE);
			if (otz_get_status (L) != 0) {
				break;
			}
			opos += produced;
			abs_bit = stage_abs + next_rel;
		}
		if (otz_get_status (L) != 0) {
			break;
		}
		if (last_block) {
			ended = true;
		}
	}

	int status = otz_get_status (L);
	if (J.prof && tid == 0) {
		L.prof[OTZ_P_TOTAL] += otz_cycles ();
	}
	if (status == 0) {
		// ---- reached the end of the final block
		status = OZG_ST_OK;
		if (last_btype != 0) {
			// the reference returns Z_BUF_ERROR when every input byte had
			// already been loaded before the step that decodes the final EOB
			u64 eob = ((u64)L.eob_start_hi << 32) | L.eob_start_lo;
			if (((eob + 7u) >> 3) >= (u64)J.n_in) {
				flags |= OZG_F_REF_FALSE_REJECT;
			}
		}
		u32 produced = opos;
		if (opos < J.out_cap) {
			// short stream, zero-filled tail
			flags |= OZG_F_SHORT_STREAM;
			while (opos < J.out_cap) {
				u32 n = J.out_cap - opos;
				if (n > OTZ_RMAX) {
					n = OTZ_RMAX;
				}
				otz_make_room (L, J, opos, n, flushed, crc_reg);
				otz_sync ();
				otz_fill_zero (L, opos, n);
				opos += n;
				otz_sync ();
			}
		}
		otz_flush (L, J, flushed, opos, crc_reg);
		u32 crc = ~crc_reg;
		if (crc != expect_crc) {
			flags |= OZG_F_CRC_MISMATCH;
		}
		if (tid == 0) {
			res->status = OZG_ST_OK;
			res->total_out = produced;
			res->crc32 = crc;
			res->flags = flags;
		}
		if (J.prof && tid == 0) {
			L.prof[OTZ_P_FLUSH] += otz_cycles () - L.prof_last;
			for (u32 i = 0; i < 16; i++) {
				J.prof[i] = L.prof[i];
			}
		}
	} else if (tid == 0) {
		res->status = status == OTZ_ST_NEED_OUT_INT ? OZG_ST_NEED_OUT : status;
		res->total_out = status == OZG_ST_INTERNAL ? L.dbg[1] : opos;
		res->crc32 = status == OZG_ST_INTERNAL ? L.dbg[2] : 0;
		res->flags = status == OZG_ST_INTERNAL ? (L.dbg[0] << 28) | (L.dbg[3] & 0x0FFFFFFF) : 0;
	}
}
