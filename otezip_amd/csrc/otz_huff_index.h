// otz_huff_index.h -- first half of the DEFLATE decode: ONE WAVEFRONT finds the token chain
// of one entry and writes the index of otz_index.h.  (ref: the symbol loop of inflate(),
// src/lib/deflate-dec.inc.c:610-817, block headers :122-253, stored blocks :269-319.)
//
// A Huffman stream can only be cut where a token starts, and that is only known once
// everything before it has been decoded.  The wavefront breaks the chain by speculation, and
// pays for the speculation ONCE PER LANE, not once per stretch of input:
//
//   * a span covers 64 segments of S bits -- up to OTZ_K1_SMAX bits each, i.e. a whole block
//     of a typical stream -- and lane l owns the segment [s_l, s_l + S) from end to end;
//   * a lane's bits pass through LDS a PIECE (OTZ_K1_P bits) at a time: every lane stages the
//     next piece of its own segment (plus a token's reach behind it) into its column of a
//     [row][lane] window, so the lanes of a wavefront read 64 different banks wherever each of
//     them stands, and carries its position, byte and token counts from piece to piece;
//   * walk A: every lane but lane 0 starts OTZ_K1_WARM bits BEFORE its segment at a
//     guessed position.  A prefix code decoded from a wrong position falls into step with
//     the real chain after a few tokens, so by s_l the lane is usually on it.  From s_l on
//     the lane notes a checkpoint (bit position, output bytes so far) every OTZ_IX_G
//     tokens, in a row of its region's temp area;
//   * fixpoint: lane l's true start is where lane l-1 really ended (one v_mov_dpp).  If
//     that is the first token walk A saw in the segment, the lane is confirmed; otherwise
//     walk B re-decodes from the true start until it lands on one of walk A's checkpoints
//     (or the segment ends) -- usually within the segment's first piece.  A lane whose end
//     moved makes its right neighbour repeat; that almost never cascades;
//   * the first lane whose path ends the block, breaks, or runs out of input (or that
//     exceeds its token budget) closes the span; the lanes before it are accepted: prefix
//     sums give every item its output position and its slot in the block's item list.  The
//     lanes behind a block's end walked the next block with the wrong code: a span is cut to
//     what the previous block took, so that they stay few.
//
// Nothing here waits on another wavefront: the kernel is launched with one 64-lane
// workgroup per entry and the chip keeps twenty of them per CU in flight.
#pragma once

#include "otz_inflate.h"
#include "otz_index.h"

#ifndef OTZ_K1_P
#define OTZ_K1_P 512u     // bits of its segment a lane stages per piece
#endif
#ifndef OTZ_K1_WARM
#define OTZ_K1_WARM 256u  // head start of walk A (bits; a multiple of 8).  Measured, index kernel at full occupancy: 64 bits 0.82 ms, 96 0.90,
                          // 128 0.89, 192 0.80, 256 0.79-0.80, 320 0.80, 384 0.78-0.79, 440 0.79 (re-walk passes per entry 2.8 ... 1.7)
#endif
#define OTZ_K1_PW (OTZ_K1_P / 32u)    // words of a piece
#define OTZ_K1_WROWS (OTZ_K1_PW + 2u) // window rows (words per lane): the piece and the reach of a token that starts on its
                                      // last bit (two words more).  With 512-bit pieces the wavefront's LDS is 7680 bytes: the
                                      // CU hands LDS out in 1280-byte granules (measured: 7808 bytes admit 18 wavefronts, not 20),
                                      // six of them is the most that leaves room for twenty wavefronts
#define OTZ_K1_HDR_WORDS 264u // staged bytes for a block header: 17 + 57 + 318 x 14 bits < 1056 bytes

static_assert (OTZ_K1_P % 128u == 0 && OTZ_K1_P >= 256u && OTZ_K1_WARM % 8u == 0 && OTZ_K1_WARM + 7u + 64u <= OTZ_K1_P, "a piece is a whole number of 16-byte loads; the warm-up and the early token starts lie in the first piece");
static_assert (OTZ_K1_SMIN % 32u == 0 && OTZ_K1_SMAX % 32u == 0 && OTZ_K1_WARM + 7u + OTZ_K1_SMAX + 48u < (1u << 24), "segments are whole words; positions fit the exit word");
static_assert (OTZ_IX_G >= 2u && OTZ_IX_G <= 8u, "an item is two to eight steps");

#define OTZ_K1_NONE 0xFFFFFFFFu
#define OTZ_X_CAP 2u // exit kind: token budget of the round used up (not an error; the block goes on)
#define OTZ_X_TRUNC_END 5u // the input ends exactly where the next token would start: the reference's loop ends
                           // with Z_BUF_ERROR there (:610, :821-830), while running dry INSIDE a token is Z_DATA_ERROR

// Decode tables of the index kernel: 16-bit entries, 10-bit root for the literal/length code.
// This kernel never needs a match's distance -- only how many bits a token takes and how many
// bytes it yields -- so an entry is small, a long root fits, and the second-level lookup (codes
// longer than the root) becomes rare.  The layout is chosen for the walk's inner loop: every
// field it needs is one VALU instruction away, and "this lane cannot take a token here" is ONE
// test (extra-bit field == 7).
//   lit[]:  [3:0] code length, [6:4] X, [15:7] B
//           X < 7: a token.  literal: X = 0, B = 1.  length code: X = extra-bit count (0..5),
//                  B = base length (3..258): the token yields B + (X extra bits) bytes, and
//                  B > 2 says that a distance code follows
//           X = 7: special.  B = 0: no code / reserved symbol 286, 287 (ref deflate-dec.inc.c:794-797);
//                  B = 1: end of block ([3:0] its code length); B >= 3: link, [3:0] index width,
//                  B - 3 = (sub-table start - root size) / 2
//   dist[]: code length + extra-bit count (1..28); OTZ_K1_DBAD: reserved symbol (30, 31) or no code
//           (large: a token ending there lies beyond every limit, so the walk needs no test of its own);
//           bit 15: link ([3:0] index width, [14:4] sub-table start - root size)
#define OTZ_K1_SPECIAL 0x70u
#define OTZ_K1_DBAD 0x3FFFu
#ifndef OTZ_K1_LROOT
#define OTZ_K1_LROOT 10u
#endif
#define OTZ_K1_DROOT 8u
// Sub-table room: zlib's worst cases (ENOUGH: 1332 / 592 entries) need adversarial code sets; streams that
// compressors write stay far below these sizes, and a block that does not fit sends its entry to the
// self-contained kernel (OTZ_IX_OVERFLOW), so the sizes are a speed choice, never a correctness one.
#ifndef OTZ_K1_LLUT
#define OTZ_K1_LLUT 1152u // 1024 + 128
#endif
#ifndef OTZ_K1_DLUT
#define OTZ_K1_DLUT 384u  // 256 + 128
#endif

struct otz_k1_lds {
	union {
		u32 win[OTZ_K1_WROWS * 64u]; // [row][lane]: the piece every lane has staged
		struct { // block prologue only
			u32 hdr_w[OTZ_K1_HDR_WORDS + 4u];
			u32 cl_lut[128];
			u8 ll[OTZ_IX_LENS];
			u16 rank[OTZ_IX_LENS];
			u8 subw[1u << OTZ_K1_LROOT]; // per root slot: index width of its sub-table (0: none)
		} h;
	};
	u16 lit[OTZ_K1_LLUT + OTZ_K1_DLUT]; // the literal/length table, then the distance table (OTZ_K1_DIST): one array,
	                                    // so that a step's SECOND lookup is one load whichever table it goes to
};
#define OTZ_K1_DIST(L) ((L).lit + OTZ_K1_LLUT)
static_assert (OTZ_K1_P != 512u || sizeof (otz_k1_lds) <= 6u * 1280u, "twenty wavefronts per CU: six 1280-byte granules of LDS each");

// ------------------------------------------------------------------ wave helpers

// 16 bytes of the entry at byte offset off (may be negative or run past the end: zeros there)
OTZ_DEV otz_u4 otz_k1_ld16(const u8 *in, u32 n_in, int64_t off) {
	if (off >= 0 && off + 16 <= (int64_t)n_in) {
		return otz_ld16 (in + off);
	}
	u32 w[4] = { 0, 0, 0, 0 };
	if (off + 16 > 0 && off < (int64_t)n_in) {
		for (u32 k = 0; k < 16; k++) {
			int64_t q = off + k;
			if (q >= 0 && q < (int64_t)n_in) {
				w[k >> 2] |= (u32)in[q] << (8u * (k & 3u));
			}
		}
	}
	otz_u4 v;
	v.x = w[0];
	v.y = w[1];
	v.z = w[2];
	v.w = w[3];
	return v;
}

// ------------------------------------------------------------------ block header

OTZ_DEV u32 otz_k1_hbits(const otz_k1_lds &L, u32 p) { // 32 staged bits from bit p
	const u32 w = p >> 5;
	return otz_alignbit (L.h.hdr_w[w + 1u], L.h.hdr_w[w], p & 31u);
}

// (The block prologue below synchronises LDS traffic only -- otz_wave_sync_lds: the LDS executes a wavefront's
// instructions in order -- so that it never waits for the stores of the index that are still on their way.)
// BFINAL/BTYPE and HLIT/HDIST/HCLEN of a dynamic block plus the 7-bit table of the
// code-length code (ref deflate-dec.inc.c:613-627, :122-156); same construction as
// otz_parse_header_par, one wavefront.  *cl_ok = 0 for an over-subscribed code-length code.
OTZ_DEV otz_hdr otz_k1_parse_header(otz_k1_lds &L, u32 p, u64 avail, u32 *cl_ok) {
	otz_hdr H;
	H.st = 0;
	H.bfinal = H.btype = H.hlit = H.hdist = 0;
	H.bits = 3;
	*cl_ok = 1;
	const u32 lo = otz_k1_hbits (L, p);
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
	if (avail < H.bits || H.hlit + H.hdist > 318u) { // ref LL_LENGTHS_MAX (:165-168)
		H.st = OZG_ST_DATA;
		return H;
	}
	const u32 lane = otz_opaque (otz_lane ()); // (opaque: nothing derived from it is computed once at the top of the kernel and kept, spilled, across everything else)
	L.h.cl_lut[lane] = 0;
	L.h.cl_lut[lane + 64u] = 0;
	otz_wave_sync_lds ();
	const u64 inv_lo = 3ull | (17ull << 5) | (15ull << 10) | (13ull << 15) | (11ull << 20) | (9ull << 25) | (7ull << 30) | (5ull << 35) |
		(4ull << 40) | (6ull << 45) | (8ull << 50) | (10ull << 55);
	const u64 inv_hi = 12ull | (14ull << 5) | (16ull << 10) | (18ull << 15) | (0ull << 20) | (1ull << 25) | (2ull << 30);
	u32 l = 0;
	if (lane < 19u) {
		u32 slot = (u32) ((lane < 12u ? inv_lo >> (5u * lane) : inv_hi >> (5u * (lane - 12u))) & 31u);
		if (slot < hclen) {
			l = otz_k1_hbits (L, p + 17u + 3u * slot) & 7u;
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
			L.h.cl_lut[idx] = (lane << 8) | l;
		}
	}
	*cl_ok = kraft <= 128u ? 1u : 0u;
	otz_wave_sync_lds ();
	return H;
}

// The hlit + hdist run-length coded code lengths (ref deflate-dec.inc.c:164-253), 64 bit
// positions per step: every lane decodes the code-length token that WOULD start at its
// bit, a scalar walk (v_readlane) follows the real chain through the 64 positions, a
// prefix sum places the tokens.  Tokens write only the START of their run ("this length
// from here on"); a fill-forward pass expands the runs, which also resolves symbol 16
// ("repeat the previous length") without a serial pass.
// Returns 0 and the bits consumed, or OZG_ST_DATA.
OTZ_DEV int otz_k1_parse_lengths(otz_k1_lds &L, u32 p, u64 avail, u32 n, u32 *bits_used) {
	const u32 lane = otz_opaque (otz_lane ());
	for (u32 i = lane; i < OTZ_IX_LENS; i += 64u) {
		L.h.ll[i] = 0;
	}
	otz_wave_sync_lds ();
	u32 idx = 0, pos = 0;
	while (idx < n) {
		const u32 at = pos + lane;
		const u32 bits = otz_k1_hbits (L, p + at);
		const u32 e = L.h.cl_lut[bits & 127u];
		const u32 l = e & 15u, sym = e >> 8;
		const u32 xb = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u;
		const u32 cnt = sym < 16u ? 1u : ((bits >> l) & ((1u << xb) - 1u)) + (sym == 18u ? 11u : 3u);
		u32 nb = l + xb;
		if (l == 0 || (u64)at + nb > avail) {
			nb = 0; // no token can start here: unused code or out of input
		}
		u64 starts = 0;
		u32 cur = 0, got = idx;
		bool fail = false;
		const u32 tokw = nb | (cnt << 8); // (one cross-lane read per token of the chain: bits taken | lengths made)
		while (cur < 64u && got < n) {
			const u32 t = otz_readlane (tokw, cur);
			const u32 step = t & 0xFFu;
			if (step == 0) {
				fail = true;
				break;
			}
			starts |= (u64)1 << cur;
			got += t >> 8;
			cur += step;
		}
		if (fail || got > n) { // bad code, out of input, or a run past hlit + hdist
			return OZG_ST_DATA;
		}
		const bool is_start = ((starts >> lane) & 1u) != 0;
		const u32 mine = is_start ? cnt : 0u;
		const u32 my_idx = idx + otz_wave_scan_incl (mine) - mine;
		const bool rep0 = is_start && sym == 16u && my_idx == 0; // nothing to repeat
		if (otz_ballot (rep0) != 0) {
			return OZG_ST_DATA;
		}
		if (is_start && sym != 16u) {
			L.h.ll[my_idx] = (u8) (0x80u | (sym < 16u ? sym : 0u));
		}
		idx = got;
		pos += cur;
	}
	otz_wave_sync_lds ();
	u32 carry = 0xFFu;
	for (u32 base = 0; base < n; base += 64u) {
		const u32 i = base + lane;
		const u32 v = i < n ? L.h.ll[i] : 0u;
		const u64 m = otz_ballot ((v & 0x80u) != 0);
		const u64 mm = m & (~(u64)0 >> (63u - lane));
		const u32 srcl = mm ? 63u - otz_clz64 (mm) : 0u;
		u32 val = otz_shfl (v & 0x1Fu, srcl);
		val = mm ? val : carry;
		otz_wave_sync_lds ();
		if (i < n) {
			L.h.ll[i] = (u8)val;
		}
		carry = otz_readlane (val, 63u);
	}
	otz_wave_sync_lds ();
	*bits_used = pos;
	return 0;
}

OTZ_DEV u32 otz_k1_mk_lit(u32 sym, u32 len) {
	if (sym < 256u) {
		return len | (1u << 7);
	}
	if (sym == 256u) {
		return len | OTZ_K1_SPECIAL | (1u << 7);
	}
	if (sym > 285u) {
		return len | OTZ_K1_SPECIAL; // ref deflate-dec.inc.c:794-797
	}
	const u32 k = sym - 257u;
	u32 base, xb;
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
	return len | (xb << 4) | (base << 7);
}

OTZ_DEV u32 otz_k1_mk_dist(u32 sym, u32 len) {
	if (sym >= 30u) {
		return OTZ_K1_DBAD; // the reference indexes past dist_base[] here (:766-774)
	}
	return len + (sym < 4u ? 0u : (sym >> 1) - 1u);
}

// Both decode tables from L.h.ll[0..hlit+hdist): canonical code = first[len] + rank among the
// symbols of that length (ranks from ballots, no sorted list), root slots written by the
// symbols themselves, sub-tables placed by a running sum over the root slots -- the
// construction of otz_build_tables with one wavefront looping over chunks of 64.
// Returns 0, OZG_ST_DATA for an over-subscribed set, or OTZ_IX_OVERFLOW when the sub-tables
// do not fit these (smaller) tables.
OTZ_DEV int otz_k1_build_tables(otz_k1_lds &L, u32 hlit, u32 hdist) {
	const u32 lane = otz_opaque (otz_lane ());
	const u64 below = ((u64)1 << lane) - 1u;
	for (u32 i = lane; i < OTZ_K1_LLUT / 2u; i += 64u) {
		reinterpret_cast<u32 *> (L.lit)[i] = OTZ_K1_SPECIAL * 0x10001u; // no code
	}
	for (u32 i = lane; i < OTZ_K1_DLUT / 2u; i += 64u) {
		reinterpret_cast<u32 *> (OTZ_K1_DIST (L))[i] = OTZ_K1_DBAD * 0x10001u;
	}
	int st = 0;
	for (u32 tbl = 0; tbl < 2u; tbl++) {
		const u32 nsym = tbl ? hdist : hlit, at0 = tbl ? hlit : 0u;
		const u32 root = tbl ? OTZ_K1_DROOT : OTZ_K1_LROOT, rsz = 1u << root, cap = tbl ? OTZ_K1_DLUT : OTZ_K1_LLUT;
		u16 *const lut = tbl ? OTZ_K1_DIST (L) : L.lit;
		for (u32 i = lane; i < rsz / 4u; i += 64u) {
			reinterpret_cast<u32 *> (L.h.subw)[i] = 0;
		}
		// lane k (1..15) counts the symbols of length k; every symbol notes its rank
		u32 cnt = 0;
		for (u32 base = 0; base < nsym; base += 64u) {
			const u32 sym = base + lane;
			const u32 l = sym < nsym ? L.h.ll[at0 + sym] : 0u;
			const u32 before = otz_shfl (cnt, l);
			u32 rank = 0;
			u64 rem = otz_ballot (l != 0); // one ballot per code length that occurs in this chunk
			while (rem != 0) {
				const u32 k = otz_readlane (l, otz_ctz64 (rem));
				const u64 m = otz_ballot (l == k);
				if (l == k) {
					rank = otz_popc64 (m & below);
				}
				if (lane == k) {
					cnt += otz_popc64 (m);
				}
				rem &= ~m;
			}
			if (sym < nsym) {
				L.h.rank[at0 + sym] = (u16) (before + rank);
			}
		}
		// lane l: first code of length l = sum over j < l of count[j] << (l - j)
		u32 first = 0;
		for (u32 j = 1; j < 15u; j++) {
			const u32 cj = otz_readlane (cnt, j);
			if (lane > j && lane < 16u) {
				first += cj << (lane - j);
			}
		}
		// over-subscribed (ref: "left < 0") exactly when first[15] + count[15] > 2^15
		if (otz_readlane (first + cnt, 15u) > (1u << 15)) {
			st = OZG_ST_DATA;
		}
		otz_wave_sync_lds ();
		// short codes: the symbol writes every root slot that starts with its code
		for (u32 base = 0; base < nsym; base += 64u) {
			const u32 sym = base + lane;
			const u32 l = sym < nsym ? L.h.ll[at0 + sym] : 0u;
			const u32 f = otz_shfl (first, l);
			u32 rr = 0, e = 0;
			if (l && l <= root) {
				rr = otz_brev (f + L.h.rank[at0 + sym]) >> (32u - l);
				e = tbl ? otz_k1_mk_dist (sym, l) : otz_k1_mk_lit (sym, l);
				if (l + 5u > root) {
					for (u32 idx = rr; idx < rsz; idx += 1u << l) { // at most sixteen root slots
						lut[idx] = (u16)e;
					}
				}
			}
			// a very short code owns 32 root slots or more: the wavefront fills those together, a symbol at a time
			u64 heavy = otz_ballot (l != 0 && l + 5u <= root);
			while (heavy != 0) {
				const u32 hs = otz_ctz64 (heavy);
				heavy &= heavy - 1u;
				const u32 ls = otz_readlane (l, hs), rs = otz_readlane (rr, hs), es = otz_readlane (e, hs);
				for (u32 idx = rs + (lane << ls); idx < rsz; idx += 64u << ls) {
					lut[idx] = (u16)es;
				}
			}
		}
		// long codes: their root slot needs a sub-table as wide as the longest of them.  Lengths
		// in ascending order, plain stores: the LDS executes a wave's stores in order, the last wins.
		for (u32 ll = root + 1u; ll <= 15u; ll++) {
			if (otz_readlane (cnt, ll) == 0) {
				continue;
			}
			for (u32 base = 0; base < nsym; base += 64u) {
				const u32 sym = base + lane;
				const u32 l = sym < nsym ? L.h.ll[at0 + sym] : 0u;
				const u32 f = otz_shfl (first, l);
				if (l == ll) {
					const u32 rr = otz_brev (f + L.h.rank[at0 + sym]) >> (32u - l);
					L.h.subw[rr & (rsz - 1u)] = (u8) (ll - root);
				}
			}
		}
		otz_wave_sync_lds ();
		// sub-table placement: running sum over the root slots
		u32 total = 0;
		for (u32 base = 0; base < rsz; base += 64u) {
			const u32 slot = base + lane;
			const u32 wd = L.h.subw[slot];
			const u32 sz = wd ? 1u << wd : 0u;
			const u32 incl = otz_wave_scan_incl (sz);
			const u32 tot = otz_readlane (incl, 63u);
			const u32 off = total + incl - sz; // relative to the end of the root table; even (every size is)
			if (sz && total + tot <= cap - rsz) {
				lut[slot] = (u16) (tbl ? 0x8000u | (off << 4) | wd : wd | OTZ_K1_SPECIAL | ((3u + (off >> 1)) << 7));
			}
			total += tot;
		}
		if (total > cap - rsz) {
			st = st ? st : OTZ_IX_OVERFLOW;
		}
		otz_wave_sync_lds ();
		if (st == 0) {
			for (u32 base = 0; base < nsym; base += 64u) {
				const u32 sym = base + lane;
				const u32 l = sym < nsym ? L.h.ll[at0 + sym] : 0u;
				const u32 f = otz_shfl (first, l);
				if (l > root) {
					const u32 rr = otz_brev (f + L.h.rank[at0 + sym]) >> (32u - l);
					const u16 e = (u16) (tbl ? otz_k1_mk_dist (sym, l) : otz_k1_mk_lit (sym, l));
					const u32 link = lut[rr & (rsz - 1u)];
					const u32 sb = link & 15u, sbase = rsz + (tbl ? (link >> 4) & 0x7FFu : 2u * ((link >> 7) - 3u));
					for (u32 idx = rr >> root; idx < (1u << sb); idx += 1u << (l - root)) {
						lut[sbase + idx] = e;
					}
				}
			}
		}
		otz_wave_sync_lds ();
	}
	return st;
}

// ------------------------------------------------------------------ the walks

struct otz_k1_tok {
	u32 nbits;  // stream bits of the whole token
	u32 obytes; // output bytes: 1 for a literal, the match length
	u32 kind;   // 0: literal or match, 1: end of block, 2: invalid
};

// One token whose bits start at bit 0 of (hi:lo), in full: for everything outside the walks' loops
// (classification of a stop, the recount of the fixpoint, the serial finisher).
OTZ_DEV otz_k1_tok otz_k1_decode_bits(const otz_k1_lds &L, u32 lo, u32 hi) {
	u32 e = L.lit[lo & ((1u << OTZ_K1_LROOT) - 1u)];
	if ((e & OTZ_K1_SPECIAL) == OTZ_K1_SPECIAL && (e >> 7) >= 3u) {
		e = L.lit[(1u << OTZ_K1_LROOT) + 2u * ((e >> 7) - 3u) + otz_bfe (lo, OTZ_K1_LROOT, e & 15u)];
	}
	const u32 len = e & 15u, xb = otz_bfe (e, 4, 3), B = e >> 7;
	otz_k1_tok t;
	if (xb == 7u) {
		t.nbits = len;
		t.obytes = 0;
		t.kind = B == 1u ? 1u : 2u;
		return t;
	}
	const u32 used = len + xb;
	const u32 db = otz_alignbit (hi, lo, used);
	u32 d = OTZ_K1_DIST (L)[db & ((1u << OTZ_K1_DROOT) - 1u)];
	if (d & 0x8000u) {
		d = OTZ_K1_DIST (L)[(1u << OTZ_K1_DROOT) + ((d >> 4) & 0x7FFu) + otz_bfe (db, OTZ_K1_DROOT, d & 15u)];
	}
	const bool is_len = B > 2u;
	t.nbits = is_len ? used + d : len;
	t.obytes = B + otz_bfe (lo, len, xb);
	t.kind = is_len && d >= OTZ_K1_DBAD ? 2u : 0u;
	return t;
}

// The three window words from the one that holds slot-relative bit u.  The window is [row][lane]: word r
// of the piece a lane has staged lives at win[64 r + lane], so the lanes of a wavefront hit 64 different
// banks wherever each of them stands, and the three words are one address and two constant offsets.
// lb = 4 * lane - 256 * (first word of the piece), modulo 2^32.
OTZ_DEV void otz_k1_words(const otz_k1_lds &L, u32 lb, u32 u, u32 &w0, u32 &w1, u32 &w2) {
	const u32 a = (otz_opaque (u >> 5) << 8) + lb;
	const u32 *p = reinterpret_cast<const u32 *> (reinterpret_cast<const u8 *> (L.win) + a);
	w0 = p[0];
	w1 = p[64];
	w2 = p[128];
}
OTZ_DEV otz_k1_tok otz_k1_decode(const otz_k1_lds &L, u32 lb, u32 u) {
	u32 w0, w1, w2;
	otz_k1_words (L, lb, u, w0, w1, w2);
	return otz_k1_decode_bits (L, otz_alignbit (w1, w0, u), otz_alignbit (w2, w1, u));
}

// The step of the walks' loops: window words (2 VALU for the address), 2 funnel shifts, literal/length
// lookup, field extractions, funnel shift + SECOND lookup, the step's bits and bytes.  Straight-line but
// for the rare second-level lookups.  The second lookup is fetched for every lane and means what the first
// one makes of it: behind a length code it is the distance entry (a branch around it would cost more than the
// lookup and every step has both kinds); behind a literal it is the literal/length entry of the NEXT token, and
// if that is a literal too the step takes both -- TWO LITERALS PER STEP.  That is not about the instruction
// count: segments are equal in bits, a segment of literals holds three times the tokens of a segment of long
// matches, and a walk lasts as long as its busiest lane -- the lanes that gain are the ones everybody waits
// for (bench corpus, first block: busiest lane 433-458 steps -> 262-266, mean 256 -> 204).
// allow: this step may take two (the caller keeps a pair from straddling a multiple of OTZ_IX_G tokens); bound: the
// second literal starts in front of it (otz_k1_pair_bound of the end of the piece or segment, or of the start of the
// segment in the warm-up: where a walk starts and where it leaves its segment are token starts its neighbours must
// find it on; and a pair never ends beyond the limit).  A step that may not take two takes the first literal, as before.
// stop: no token can be taken here (end of block, reserved symbol, no such code).
#ifndef OTZ_K1_PAIRS
#define OTZ_K1_PAIRS 1
#endif
struct otz_k1_step {
	u32 nb;   // bits of the step; beyond every limit when its distance code is reserved
	u32 ob;   // bytes it yields
	u32 l1;   // two literals: bits of the first
	bool stop;
	bool pair; // two literals
};
OTZ_DEV otz_k1_step otz_k1_hot_w(const otz_k1_lds &L, u32 w0, u32 w1, u32 w2, u32 u, bool allow, u32 bound) {
	const u32 lo = otz_alignbit (w1, w0, u), hi = otz_alignbit (w2, w1, u);
	u32 e = L.lit[lo & ((1u << OTZ_K1_LROOT) - 1u)];
	u32 xb = otz_bfe (e, 4, 3);
	if (xb == 7u && (e >> 7) >= 3u) { // rare: a code longer than the root
		e = L.lit[(1u << OTZ_K1_LROOT) + 2u * ((e >> 7) - 3u) + otz_bfe (lo, OTZ_K1_LROOT, e & 15u)];
		xb = otz_bfe (e, 4, 3);
	}
	otz_k1_step r;
	r.stop = xb == 7u;
	const u32 len = e & 15u, used = len + xb;
	const u32 db = otz_alignbit (hi, lo, used);
	const u32 B = e >> 7;
	const bool is_len = B > 2u;
#if OTZ_K1_PAIRS
	u32 d = L.lit[is_len ? OTZ_K1_LLUT + (db & ((1u << OTZ_K1_DROOT) - 1u)) : db & ((1u << OTZ_K1_LROOT) - 1u)];
#else
	u32 d = OTZ_K1_DIST (L)[db & ((1u << OTZ_K1_DROOT) - 1u)];
#endif
	otz_touch32 (d); // (fetched here, for every lane)
	if (is_len && (d & 0x8000u) != 0) { // rare: a distance code longer than the root
		d = OTZ_K1_DIST (L)[(1u << OTZ_K1_DROOT) + ((d >> 4) & 0x7FFu) + otz_bfe (db, OTZ_K1_DROOT, d & 15u)];
	}
#if OTZ_K1_PAIRS
	// (a literal is X = 0, B = 1: the entry is 0x80 | code length -- for the first and for the second)
	// (bitwise, not &&: the conditions are all computed, nothing branches)
	r.l1 = len;
	const u32 l2 = d & 15u;
	const bool two = ((e | (d << 16)) & 0xFFF0FFF0u) == 0x00800080u;
	r.pair = (bool) ((int)allow & (int)two & (int) (u + len < bound));
	r.nb = used + (is_len ? d : r.pair ? l2 : 0u);
	r.ob = B + otz_bfe (lo, len, xb) + (r.pair ? 1u : 0u);
#else
	r.l1 = len;
	r.pair = false;
	r.nb = used + (is_len ? d : 0u);
	r.ob = B + otz_bfe (lo, len, xb);
#endif
	return r;
}
OTZ_DEV otz_k1_step otz_k1_hot(const otz_k1_lds &L, u32 lb, u32 u, bool allow, u32 bound) {
	u32 w0, w1, w2;
	otz_k1_words (L, lb, u, w0, w1, w2);
	return otz_k1_hot_w (L, w0, w1, w2, u, allow, bound);
}
// Where a step's second literal has to start in front of: `bound`, and far enough from the limit for any literal code
// (15 bits) to end inside it -- one comparison in the step instead of two.
OTZ_DEV u32 otz_k1_pair_bound(u32 bound, u32 lim2) {
	const u32 safe = lim2 >= 15u ? lim2 - 15u : 0u;
	return bound < safe ? bound : safe;
}

// How a walk that stopped at slot-relative bit u (in front of e, inside the staged piece) ends: the exit word.
OTZ_DEV u32 otz_k1_classify(const otz_k1_lds &L, u32 lb, u32 u, u32 limit) {
	const otz_k1_tok t = otz_k1_decode (L, lb, u); // the token it stopped at
	if (u >= limit) {
		return OTZ_X_MAKE (OTZ_X_TRUNC_END, u);
	}
	if (t.kind == 2u) {
		return OTZ_X_MAKE (OTZ_X_INVALID, u);
	}
	if (u + t.nbits > limit) {
		return OTZ_X_MAKE (OTZ_X_TRUNC, u);
	}
	return OTZ_X_MAKE (OTZ_X_EOB, u + t.nbits); // (kind 1: a token that neither breaks nor ends the input is taken)
}

// Checkpoint (row r, this lane) of a walk's rows, [row][lane], 512 bytes a row.  The address is the rows' base (the
// same in every lane: scalar registers) plus a 32-bit byte offset -- no 64-bit per-lane index arithmetic.
// OTZ_K1_ROWS_LM: the rows LANE-major instead -- a lane's checkpoints one after the other (8 bytes each, `cap` of
// them per lane), so that the eight checkpoints a lane notes in a row of time fill one 64-byte line instead of
// touching eight.
#ifndef OTZ_K1_ROWS_LM
#define OTZ_K1_ROWS_LM 0
#endif
#if OTZ_K1_ROWS_LM
#define OTZ_K1_RSHIFT 3u
OTZ_DEV u32 otz_k1_lane_off(u32 cap) { return 8u * cap * otz_opaque (otz_lane ()); }
#else
#define OTZ_K1_RSHIFT 9u
OTZ_DEV u32 otz_k1_lane_off(u32 cap) { return 8u * otz_opaque (otz_lane ()); }
#endif
OTZ_DEV const otz_ix_item *otz_k1_row(const otz_ix_item *rows, u32 r, u32 lane8) {
	return reinterpret_cast<const otz_ix_item *> (reinterpret_cast<const u8 *> (rows) + ((r << OTZ_K1_RSHIFT) + lane8));
}

#ifdef OTZ_K1_EXP
OTZ_DEV u32 otz_wave_max_u32(u32 v) { // (diagnostics only)
	for (u32 d = 1; d < 64u; d <<= 1) {
		const u32 x = otz_shfl_xor (v, d);
		v = x > v ? x : v;
	}
	return v;
}
#endif
// State of a walk (registers), carried from piece to piece of the lane's segment.
struct otz_k1_ws {
	u32 u;       // slot-relative bit position: start of the next token
	u32 bytes;   // output bytes of the tokens taken
	u32 cnt;     // tokens taken (each produced output)
	bool alive;  // still walking
	bool halted; // stopped in front of a token it cannot take (classified while the piece is staged)
	bool merged; // walk B: landed on walk A's path
	bool capped; // token budget used up
	u64 bm;      // walk A: bit (u - s) set for every token start u < s + 64
	u32 first;   // walk A: where its chain starts (the first token start at or behind s, or where it started over)
	u32 c, cpos; // walk B: walk A's next checkpoint at or behind u
	u32 exit;    // exit word once the walk has ended
#ifdef OTZ_K1_EXP
	u32 steps;   // (diagnostics: trips of the walk loops)
	u32 loop_cyc; // (diagnostics: cycles inside the walk loops)
	u32 stage_cyc; // (diagnostics: cycles staging pieces)
#endif
};

// One piece of a walk: the lanes with W.alive take tokens while they stand in front of `pend`.
//   SPEC (walk A): notes a checkpoint (position, bytes so far) before every OTZ_IX_G-th token of the lane, in its
//     column of `rows` ([row][lane]), and -- EARLY -- the token starts of the segment's first 64 bits in a bitmap.
//   !SPEC (walk B): the same checkpoints in its own rows; stops as soon as it stands on walk A's path: on one of
//     the early token starts (bmA, EARLY) or on a checkpoint (rowsA / nckA).
// ncap: token budget (a multiple of OTZ_IX_G).
// The loop has ONE exit test and no branch in its body but for the rare paths: a lane that has to stop does so by
// raising a bit above every position in its own u (a break out of the middle of the body costs a dozen scalar
// instructions of execution-mask bookkeeping per way out, every iteration), and a step that must not count is
// computed and then not added.
#define OTZ_K1_HALT 0x80000000u
#define OTZ_K1_MERGE 0x40000000u
#ifndef OTZ_K1_STAGE_ALL
#define OTZ_K1_STAGE_ALL 0 // 1: every lane stages its piece, not only the lanes that walk in it (a re-walk pass is a few lanes: -3.5 % of the kernel)
#endif
#ifndef OTZ_K1_ACC_BATCH
#define OTZ_K1_ACC_BATCH 16u // rows of walk A whose loads are in flight together at acceptance (even)
#endif
#ifndef OTZ_K1_YOUNG
#define OTZ_K1_YOUNG 768u // bits: a guessed chain that stops in the first so many bits of its segment starts over
#endif
template <bool SPEC, bool EARLY, bool YOUNG>
OTZ_DEV void otz_k1_walk_loop(const otz_k1_lds &L, u32 lb, otz_k1_ws &W, u32 pend, u32 s, u32 lim2, u32 ncap, otz_ix_item *rows,
	const otz_ix_item *rowsA, u64 bmA, u32 nckA, u32 capA) {
	const u32 lane8 = otz_k1_lane_off (ncap / OTZ_IX_G);       // this walk's rows
	const u32 lane8A = SPEC ? lane8 : otz_k1_lane_off (capA);  // walk A's rows (walk B reads them)
	u32 u = W.alive ? W.u : OTZ_K1_HALT;
	u32 bytes = W.bytes, cnt = W.cnt, first = W.first;
	u64 bm = W.bm;
	// Walk A is a guess until it has met the real chain, and a guessed chain runs into an end-of-block code (or
	// a reserved one) once in a few thousand tokens -- in one span out of four some lane does.  With segments
	// this long that lane must not give up (its whole segment would be walked again, alone): it starts over
	// one bit further on, as a new chain -- first token, counts, checkpoints and all.  Only early on (YOUNG: the
	// first OTZ_K1_YOUNG bits of the segment, a loop of their own so that the rest of the walk does not carry the
	// bookkeeping): a chain that has come that far has met the real one, and what stops it then stops the block.
	// (Whatever walk A does is a hint: the true path is walk B's, up to where it lands on walk A's chain.)
	const bool may_restart = SPEC && YOUNG && otz_lane () != 0;
#ifdef OTZ_K1_EXP
	const u64 exp_t0 = otz_cycles ();
#endif
	const u32 pbound = otz_k1_pair_bound (pend, lim2);
	while (u < pend && cnt < ncap) {
#ifdef OTZ_K1_EXP
		W.steps++;
#endif
		bool hit = false;
		if (!SPEC) {
			while (W.cpos < u) { // (rare: passed a checkpoint of walk A without landing on it)
				W.c++;
				W.cpos = W.c < nckA ? otz_k1_row (rowsA, W.c, lane8A)->bit : OTZ_K1_NONE;
			}
			hit = W.cpos == u;
			if (EARLY) { // the early token starts of walk A
				const u32 rel = u - s;
				hit = hit || (rel < 64u && ((bmA >> rel) & 1u) != 0);
			}
		}
		if (SPEC && EARLY) {
			const u32 rel = u - s;
			bm |= rel < 64u ? (u64)1 << rel : 0u;
		}
		// A checkpoint in front of every OTZ_IX_G-th STEP of the lane (cnt counts steps, a step is one token or two
		// literals): the lanes of a walk count in lockstep (but for the few that started over), so a row is stored
		// by all of them in the same trip -- one coalesced 512-byte store -- and an item is at most OTZ_IX_G steps
		// of the apply kernel's placement, which pairs literals at least as eagerly (otz_lz_apply.h).
		if (cnt % OTZ_IX_G == 0) {
			otz_ix_item k;
			k.bit = u;
			k.out = bytes;
			*reinterpret_cast<otz_ix_item *> (reinterpret_cast<u8 *> (rows) + (((cnt / OTZ_IX_G) << OTZ_K1_RSHIFT) + lane8)) = k;
		}
		otz_k1_step h = otz_k1_hot (L, lb, u, true, pbound);
#if OTZ_K1_PAIRS
		if (!SPEC) {
			// walk B does not step over a position of walk A it could land on: the second literal's start
			const u32 mid = u + h.l1;
			bool on_a = mid == W.cpos;
			if (EARLY) {
				const u32 relm = mid - s;
				on_a = on_a || (relm < 64u && ((bmA >> relm) & 1u) != 0);
			}
			if (h.pair && on_a) {
				h.pair = false;
				h.nb = h.l1;
				h.ob = 1u;
			}
		}
#endif
		const u32 tend = u + h.nb;
		const bool bad = h.stop || tend > lim2;
		const bool adv = !bad && !hit;
		if (SPEC && YOUNG) {
			const bool again = may_restart && h.stop && tend <= lim2;
			bytes = again ? 0u : bytes + (adv ? h.ob : 0u);
			cnt = again ? 0u : cnt + (adv ? 1u : 0u);
			first = again ? u + 1u : first;
			if (EARLY) {
				bm = again ? 0u : bm;
			}
			u = again ? u + 1u : bad ? u | OTZ_K1_HALT : tend;
		} else {
			bytes += adv ? h.ob : 0u;
			cnt += adv ? 1u : 0u;
			u = hit ? u | OTZ_K1_MERGE : bad ? u | OTZ_K1_HALT : tend;
		}
	}
#ifdef OTZ_K1_EXP
	W.loop_cyc += (u32) (otz_cycles () - exp_t0);
#endif
	if (W.alive) {
		W.bytes = bytes;
		W.cnt = cnt;
		W.bm = bm;
		W.first = first;
		W.u = u & ~(OTZ_K1_HALT | OTZ_K1_MERGE);
		if (u & OTZ_K1_MERGE) {
			W.merged = true;
			W.alive = false;
		} else if (u & OTZ_K1_HALT) {
			W.halted = true;
			W.alive = false;
		} else if (u < pend) { // (cnt == ncap)
			W.capped = true;
			W.alive = false;
		}
	}
}

// Entry byte of byte 0 of this lane's slot: the span starts in byte rb, a segment is sb bytes, the slot starts
// OTZ_K1_WARM bits ahead of the segment.  (Computed where it is needed, from an opaque lane number: a per-lane
// 64-bit value kept across the walks is two registers the kernel does not have.)
OTZ_DEV int64_t otz_k1_slot0(u32 rb, u32 sb) {
	return (int64_t)rb + (int64_t)sb * (int64_t)otz_opaque (otz_lane ()) - (int64_t) (OTZ_K1_WARM / 8u);
}

// The input ends at this slot-relative bit of the lane's slot (0: the slot lies behind the input).
OTZ_DEV u32 otz_k1_limit(u32 n_in, u32 rb, u32 sb) {
	const int64_t lb = 8 * ((int64_t)n_in - otz_k1_slot0 (rb, sb));
	return lb <= 0 ? 0u : lb > 0xFFFFFF ? 0xFFFFFFu : (u32)lb;
}

// Stage piece k of every lane's slot: OTZ_K1_WROWS words from entry byte slot0 + k * P / 8 (zeros outside the
// entry), each lane its own column of the window.  ALL the 16-byte loads are issued before any of them is waited
// for: written the obvious way -- load a chunk, store it to LDS, next chunk -- every chunk is a global-memory round
// trip of its own (each sits behind a bounds test), five in a row per piece, and under load that was 14 % of the
// kernel's time.  The chunks that straddle an end of the entry (one lane, in few pieces) are patched afterwards.
OTZ_DEV void otz_k1_stage_piece(otz_k1_lds &L, const u8 *in, u32 n_in, u32 rb, u32 sb, u32 k, bool need) {
	const u32 lane = otz_lane ();
	const int64_t b = otz_k1_slot0 (rb, sb) + (int64_t) (k * (OTZ_K1_P / 8u));
	static_assert (OTZ_K1_WROWS % 4u == 2u, "whole 16-byte loads and half of one");
	constexpr u32 NQ = (OTZ_K1_WROWS + 3u) / 4u;
	otz_u4 v[NQ];
	bool edge = false;
	// (a chunk that does not lie inside the entry loads the entry's first bytes instead and is zeroed afterwards:
	// a conditional load would have to wait for the ones in flight before its registers could be cleared)
	// need: this lane walks the piece (a re-walk pass is a few lanes: their loads touch a few lines, not 64 x 2)
	const bool big = n_in >= 16u;
	if (big) {
		if (need) {
#pragma unroll
			for (u32 q = 0; q < NQ; q++) {
				const int64_t off = b + 16 * (int64_t)q;
				const bool inside = off >= 0 && off + 16 <= (int64_t)n_in;
				v[q] = otz_ld16 (in + (inside ? off : 0));
			}
		}
	} else {
#pragma unroll
		for (u32 q = 0; q < NQ; q++) {
			const otz_u4 zero = { 0, 0, 0, 0 };
			v[q] = zero;
		}
	}
#pragma unroll
	for (u32 q = 0; q < NQ; q++) {
		const int64_t off = b + 16 * (int64_t)q;
		const bool inside = big && off >= 0 && off + 16 <= (int64_t)n_in;
		edge = edge || (need && !inside && off + 16 > 0 && off < (int64_t)n_in);
	}
#pragma unroll
	for (u32 q = 0; q < NQ; q++) {
		const int64_t off = b + 16 * (int64_t)q;
		const bool inside = big && off >= 0 && off + 16 <= (int64_t)n_in;
		v[q].x = inside ? v[q].x : 0u;
		v[q].y = inside ? v[q].y : 0u;
		v[q].z = inside ? v[q].z : 0u;
		v[q].w = inside ? v[q].w : 0u;
	}
	if (otz_ballot (edge) != 0) { // rare: some chunk holds an end of the entry
#pragma unroll
		for (u32 q = 0; q < NQ; q++) {
			const int64_t off = b + 16 * (int64_t)q;
			if (need && !(big && off >= 0 && off + 16 <= (int64_t)n_in) && off + 16 > 0 && off < (int64_t)n_in) {
				v[q] = otz_k1_ld16 (in, n_in, off);
			}
		}
	}
	if (need) {
#pragma unroll
	for (u32 q = 0; q < NQ; q++) {
		u32 *d = &L.win[(4u * q) * 64u + lane];
		d[0] = v[q].x;
		d[64] = v[q].y;
		if (4u * q + 2u < OTZ_K1_WROWS) {
			d[128] = v[q].z;
			d[192] = v[q].w;
		}
	}
	}
}

// A whole walk over the lanes' segments [s, e) of the span, piece by piece.  All 64 lanes call.
//   SPEC (walk A): every lane; lane 0 from s (the span's true start), the others from OTZ_K1_WARM bits ahead of
//     their segment at a guessed position: a warm-up (tokens neither counted nor noted) up to the first token
//     start at or behind s -- *first (OTZ_K1_NONE: none) -- then the walk proper.
//   !SPEC (walk B): the lanes with `run`, from their true start `start` (inside the first piece); a lane that
//     lands on one of walk A's early step starts also recounts walk A's steps up to there (*jm steps,
//     *offA bytes; firstA: walk A's first token) while that piece is still staged.
// Ends with every lane's W.exit set.
template <bool SPEC>
OTZ_DEV void otz_k1_walk(otz_k1_lds &L, const u8 *in, u32 n_in, u32 rb, u32 sb, otz_k1_ws &W, bool run, u32 start, u32 s, u32 e,
	u32 ncap, otz_ix_item *rows, const otz_ix_item *rowsA, u64 bmA, u32 nckA, u32 capA, u32 firstA, u32 *first, u32 *jm, u32 *offA) {
	const u32 lane = otz_lane ();
	const u32 limit = otz_k1_limit (n_in, rb, sb);
	// a token that starts in front of e ends in front of e + 48: the clamp changes nothing for
	// tokens that exist and puts a reserved distance code (OTZ_K1_DBAD bits) beyond the limit
	const u32 lim2 = limit < e + 48u ? limit : e + 48u;
	W.u = run ? start : s;
	W.bytes = W.cnt = 0;
	W.alive = run;
	W.halted = W.merged = W.capped = false;
	W.bm = 0;
	W.first = OTZ_K1_NONE;
	W.c = 1; // walk B: walk A's next checkpoint at or behind u (row 0 is its first token: bmA has it)
	W.cpos = OTZ_K1_NONE;
	W.exit = OTZ_X_MAKE (OTZ_X_INVALID, W.u);
#ifdef OTZ_K1_EXP
	W.steps = 0;
	W.loop_cyc = 0;
	W.stage_cyc = 0;
#endif
	if (!SPEC) {
		W.cpos = run && nckA > 1u ? otz_k1_row (rowsA, 1u, otz_k1_lane_off (capA))->bit : OTZ_K1_NONE;
	}
	const u32 npieces = (e + OTZ_K1_P - 1u) / OTZ_K1_P;
	for (u32 k = 0; k < npieces; k++) {
		const u32 pend = (k + 1u) * OTZ_K1_P < e ? (k + 1u) * OTZ_K1_P : e;
		if (!SPEC) {
			if (otz_ballot (W.alive) == 0) {
				break;
			}
			if (otz_ballot (W.alive && W.u < pend) == 0) {
				continue; // (no lane stands in this piece)
			}
		}
		const u32 lb = 4u * lane - 256u * OTZ_K1_PW * k;
		// (LDS only, and the LDS executes a wavefront's instructions in order: nothing has to drain -- in particular
		// not the checkpoint stores, which stay in flight across pieces)
#ifdef OTZ_K1_EXP
#if OTZ_K1_EXP & 4
		otz_wave_sync (); // (timing experiment: the checkpoint stores drain before the clock starts)
#endif
		const u64 exp_s0 = otz_cycles ();
#endif
		otz_wave_sync_lds (); // the previous piece (or the block prologue's scratch) is dead
		otz_k1_stage_piece (L, in, n_in, rb, sb, k, OTZ_K1_STAGE_ALL || (W.alive && W.u < pend + 64u));
		otz_wave_sync_lds ();
#ifdef OTZ_K1_EXP
		W.stage_cyc += (u32) (otz_cycles () - exp_s0);
#endif
		if (SPEC && k == 0) {
			// the warm-up: nothing is counted or noted, and a guess that runs into a code it cannot take moves on a bit
			u32 u = W.u;
			const u32 wbound = otz_k1_pair_bound (s, lim2);
			while (u < s) {
#ifdef OTZ_K1_EXP
				W.steps++;
#endif
				const otz_k1_step h = otz_k1_hot (L, lb, u, true, wbound);
				const u32 tend = u + h.nb;
				u = tend > lim2 ? u | OTZ_K1_HALT : h.stop ? u + 1u : tend;
			}
			W.alive = (u & OTZ_K1_HALT) == 0;
			W.u = u & ~OTZ_K1_HALT;
			W.first = W.alive ? W.u : OTZ_K1_NONE;
			W.exit = OTZ_X_MAKE (OTZ_X_INVALID, W.u);
		}
		if (k == 0) {
			const u32 pe = s + 64u < pend ? s + 64u : pend;
			otz_k1_walk_loop<SPEC, true, SPEC> (L, lb, W, pe, s, lim2, ncap, rows, rowsA, bmA, nckA, capA);
		}
		if (SPEC && k * OTZ_K1_P < s + OTZ_K1_YOUNG) {
			const u32 py = s + OTZ_K1_YOUNG < pend ? s + OTZ_K1_YOUNG : pend;
			otz_k1_walk_loop<SPEC, false, true> (L, lb, W, py, s, lim2, ncap, rows, rowsA, bmA, nckA, capA);
		}
		otz_k1_walk_loop<SPEC, false, false> (L, lb, W, pend, s, lim2, ncap, rows, rowsA, bmA, nckA, capA);
		// the lanes that stopped in this piece: how their walk ends
		if (otz_ballot (W.halted) != 0) {
			if (W.halted) {
				W.exit = otz_k1_classify (L, lb, W.u, limit);
				W.halted = false;
			}
		}
		if (!SPEC && k == 0) {
			// landed on one of walk A's early token starts: count walk A's tokens up to there
			const bool early = W.merged && W.cpos != W.u;
			if (otz_ballot (early) != 0) {
				// (walk A's STEPS, as its early loop took them: the same lookups, the same bound for a pair)
				const u32 pe = s + 64u < pend ? s + 64u : pend;
				const u32 pb = otz_k1_pair_bound (pe, lim2);
				u32 x = firstA, j = 0, ob = 0;
				while (early && x < W.u) {
					const otz_k1_step h = otz_k1_hot (L, lb, x, true, pb);
					x += h.nb;
					ob += h.ob;
					j++;
				}
				if (early) {
					*jm = j;
					*offA = ob;
				}
			}
		}
	}
	if (SPEC) {
		*first = W.first;
		if (W.first >= s + 64u) {
			W.bm = 0; // (a chain that started over behind the early stretch: the bits are its predecessor's)
		}
	}
	if (W.alive) {
		W.exit = OTZ_X_MAKE (OTZ_X_NORMAL, W.u); // ran through its segment: the next token starts at u >= e
	} else if (W.capped) {
		W.exit = OTZ_X_MAKE (OTZ_X_CAP, W.u); // token budget used up (not an error; the block goes on)
	}
	otz_wave_sync (); // the checkpoint rows are read by other lanes from here on
}

// ------------------------------------------------------------------ the entry

#define OTZ_K1_ST_NEED_OUT OTZ_IX_NEED_OUT

// All 64 lanes.  in/n_in: the compressed payload; out_cap: uncomp_size; out_aligned: the
// entry's output address is 16-byte aligned; region/region_bytes: the entry's arena region
// (region_bytes = 0: none could be carved).  head: the entry's index head.
// Returns true when the index did not fit (head->status = OTZ_IX_OVERFLOW).
// prof: optional 16 x u64 diagnostic cycle counters (0 header+lengths, 1 tables, 3 walk A,
// 4 fixpoint, 5 accept+items, 6 spans, 7 fixpoint iterations, 8 re-walks, 9 blocks, 15 total)
// (PROF is a compile-time switch: the counters would otherwise occupy two dozen scalar registers
// of the product kernel)
#define OTZ_K1_MARK(i) \
	do { \
		if (PROF) { \
			const u64 _t = otz_cycles (); \
			pc[i] += _t - plast; \
			plast = _t; \
		} \
	} while (0)
#define OTZ_K1_COUNT(i) \
	do { \
		if (PROF) { \
			pc[i] += 1; \
		} \
	} while (0)
template <bool PROF>
OTZ_DEV bool otz_index_entry(otz_k1_lds &L, const u8 *in, u32 n_in, u32 out_cap, bool out_aligned, u8 *region, u32 region_bytes, otz_ix_head *head, u64 *prof = nullptr) {
	const u32 lane = otz_lane ();
	u64 pc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#ifdef OTZ_K1_EXP
	u32 exp_steps[2] = { 0, 0 };
	u32 exp_pro[3] = { 0, 0, 0 };
#endif
	u64 plast = PROF ? otz_cycles () : 0;
	const u64 pstart = plast;
	const u64 total_bits = (u64)n_in * 8u;
	u64 abs_bit = 0;
	u32 opos = 0;
	int status = 0;
	u32 flags = 0, n_blocks = 0;
	u32 rec = 32u; // region offset of the next block record
	u32 last_btype = 0;
	u64 eob_abs = 0, clip_abs = 0;
	bool last_block = false, ended = false;
	u32 prev_next_at = 0; // region offset of the previous block's `next` field (0: none)
	// how much input a span covers at most: what the previous block took (compressors cut blocks of similar
	// size), so that the lanes behind the block's end -- which walk the next block with the wrong code -- stay few
	u64 span_cap = 64ull * OTZ_K1_SMAX;

	const u32 rowsA_cap = otz_k1_rows_a (n_in), rowsB_cap = otz_k1_rows_b (n_in);
	const u32 temp_bytes = (rowsA_cap + rowsB_cap) * 512u;
	if (!out_aligned) {
		status = OZG_ST_INTERNAL;
	} else if (n_in == 0) {
		// ref: the loop at :610 never runs; Z_OK when there is no output space, else Z_BUF_ERROR
		status = out_cap == 0 ? OTZ_K1_ST_NEED_OUT : OZG_ST_BUF;
	} else if (region_bytes < temp_bytes + 1024u) {
		status = OTZ_IX_OVERFLOW;
	}
	const u32 rec_limit = region_bytes >= temp_bytes + 1024u ? region_bytes - temp_bytes : 0u;
	// checkpoint rows of the two walks, [row][lane]
	otz_ix_item *const rowsA = reinterpret_cast<otz_ix_item *> (region + rec_limit); // (the same in every lane)
	otz_ix_item *const rowsB = rowsA + rowsA_cap * 64u;

	while (status == 0 && !ended) {
#ifdef OTZ_K1_EXP
		u64 exp_t = otz_cycles ();
#define OTZ_K1_EXP_LAP(i) do { const u64 _n = otz_cycles (); exp_pro[i] += (u32) (_n - exp_t); exp_t = _n; } while (0)
#else
#define OTZ_K1_EXP_LAP(i) do { } while (0)
#endif
		// ---- block header: stage the bytes behind the current position
		const u32 hb = (u32) (abs_bit >> 3);
		{
			const otz_u4 v = otz_k1_ld16 (in, n_in, (int64_t)hb + 16 * (int64_t)lane);
			otz_u4 t = { 0, 0, 0, 0 };
			if (lane < 2u) { // (asked for together with the first: one trip to memory, not two)
				t = otz_k1_ld16 (in, n_in, (int64_t)hb + 1024 + 16 * (int64_t)lane);
			}
			otz_wave_sync (); // the window of the previous span is dead from here
			u32 *d = &L.h.hdr_w[4u * lane];
			d[0] = v.x;
			d[1] = v.y;
			d[2] = v.z;
			d[3] = v.w;
			if (lane < 2u) {
				u32 *d2 = &L.h.hdr_w[256u + 4u * lane];
				d2[0] = t.x;
				d2[1] = t.y;
				d2[2] = t.z;
				d2[3] = t.w;
			}
			otz_wave_sync ();
		}
		OTZ_K1_EXP_LAP (0); // header bytes staged
		const u32 hp = (u32) (abs_bit & 7u);
		u32 cl_ok = 1;
		const otz_hdr H = otz_k1_parse_header (L, hp, total_bits - abs_bit, &cl_ok);
		OTZ_K1_EXP_LAP (1); // header fields, code-length code
		if (H.st != 0 || (H.btype == 2u && !cl_ok)) {
			status = H.st != 0 ? H.st : OZG_ST_DATA;
			break;
		}
		last_block = H.bfinal != 0;
		last_btype = H.btype;
		abs_bit += H.bits;
		OTZ_K1_COUNT (9);
		if (rec + 32u > rec_limit) {
			status = OTZ_IX_OVERFLOW;
			break;
		}
		otz_ix_block *const blk = reinterpret_cast<otz_ix_block *> (region + rec);
		if (H.btype == 0) {
			// ---- stored block (ref read_uncompressed_block :269-319)
			abs_bit = (abs_bit + 7u) & ~(u64)7u;
			u32 b = (u32) (abs_bit >> 3);
			if (b > n_in || n_in - b < 4u) {
				status = OZG_ST_DATA;
				break;
			}
			const u32 len = (u32)in[b] | ((u32)in[b + 1] << 8);
			const u32 nlen = (u32)in[b + 2] | ((u32)in[b + 3] << 8);
			b += 4;
			if (len != ((~nlen) & 0xFFFFu) || n_in - b < len) {
				status = OZG_ST_DATA;
				break;
			}
			if (out_cap - opos < len) {
				status = OZG_ST_BUF; // ref :299-301
				break;
			}
			if (lane == 0) {
				blk->kind = OTZ_IX_KIND_STORED;
				blk->n_items = 0;
				blk->bit_lo = b;
				blk->bit_hi = 0;
				blk->out_base = opos;
				blk->out_len = len;
				blk->hlit_hdist = 0;
				blk->next = 0;
				if (prev_next_at) {
					*reinterpret_cast<u32 *> (region + prev_next_at) = rec;
				}
			}
			prev_next_at = rec + 28u;
			rec += 32u;
			n_blocks++;
			opos += len;
			abs_bit = ((u64)b + len) * 8u;
			ended = last_block;
			continue;
		}
		// ---- Huffman block: code lengths, tables
		u32 hlit = 288, hdist = 32;
		if (H.btype == 1) {
			for (u32 i = otz_opaque (lane); i < OTZ_IX_LENS; i += 64u) { // (opaque: see otz_k1_parse_header)
				L.h.ll[i] = (u8) (i <= 143u ? 8 : i <= 255u ? 9 : i <= 279u ? 7 : i <= 287u ? 8 : 5);
			}
			otz_wave_sync ();
		} else {
			hlit = H.hlit;
			hdist = H.hdist;
			u32 used = 0;
			const int st = otz_k1_parse_lengths (L, hp + H.bits, total_bits - abs_bit, hlit + hdist, &used);
			if (st != 0) {
				status = st;
				break;
			}
			abs_bit += used;
		}
		OTZ_K1_EXP_LAP (2); // code lengths
		if (rec + OTZ_IX_ITEMS_OFF + 8u > rec_limit) {
			status = OTZ_IX_OVERFLOW;
			break;
		}
		for (u32 i = otz_opaque (lane); i < OTZ_IX_LENS / 4u; i += 64u) { // the apply kernel rebuilds the tables from these
			u32 w = 0;
			for (u32 k = 0; k < 4u; k++) {
				const u32 j = 4u * i + k;
				w |= (j < hlit + hdist ? (u32)L.h.ll[j] : 0u) << (8u * k);
			}
			reinterpret_cast<u32 *> (region + rec + 32u)[i] = w;
		}
		OTZ_K1_MARK (0);
		{
			const int st = otz_k1_build_tables (L, hlit, hdist);
			if (st != 0) {
				status = st;
				break;
			}
		}
		OTZ_K1_MARK (1);
		otz_ix_item *const items = reinterpret_cast<otz_ix_item *> (region + rec + OTZ_IX_ITEMS_OFF);
		const u32 items_cap = (rec_limit - rec - OTZ_IX_ITEMS_OFF) / 8u; // >= 1
		const u64 block_bit = abs_bit;
		const u32 block_out = opos;
		u32 n_items = 0;
		u32 end_bit = 0; // block-relative position that closes the item list
		bool block_done = false, serial_finish = false;

		// ---- spans: 64 segments of S bits, one per lane
		while (!block_done && status == 0) {
			const u32 rb = (u32) (abs_bit >> 3), phi = (u32) (abs_bit & 7u);
			const u64 left = total_bits - 8ull * rb;
			const u32 S = otz_k1_seg_bits (left < span_cap ? left : span_cap);
			// slot-relative bit coordinates: the lane's slot starts OTZ_K1_WARM bits ahead of its segment [s, e)
			const u32 s = OTZ_K1_WARM + phi, e = s + S;
			OTZ_K1_COUNT (6);
			// ---- walk A
			otz_k1_ws A;
			u32 firstA = OTZ_K1_NONE, dummy0 = 0, dummy1 = 0;
			otz_k1_walk<true> (L, in, n_in, rb, S / 8u, A, true, lane ? s - OTZ_K1_WARM : s, s, e, rowsA_cap * OTZ_IX_G, rowsA, rowsA, 0, 0, rowsA_cap, 0, &firstA, &dummy0, &dummy1);
			const u32 nckA = (A.cnt + OTZ_IX_G - 1u) / OTZ_IX_G; // checkpoint rows in front of tokens walk A took
#ifdef OTZ_K1_EXP
			exp_steps[0] += otz_wave_max_u32 (A.steps);
			pc[2] += otz_wave_max_u32 (A.loop_cyc); // (reported as "slots": cycles inside walk A's loops)
			exp_steps[1] += otz_wave_max_u32 (A.stage_cyc); // (reported as stepsB#: cycles staging walk A's pieces)
#endif
			OTZ_K1_MARK (3);
			// ---- fixpoint over the true starts.  A lane's true path: nB tokens of its own walk B
			// (none when it starts on walk A's first token), then -- if it lands on walk A's path at
			// mp -- walk A's tokens from its token number jm on (offA bytes precede that token there).
			u32 t = lane == 0 ? s : OTZ_K1_NONE;
			u32 exit_cur = A.exit;
			bool onA = true;
			u32 mp = firstA, jm = 0, offA = 0; // (lane 0: walk A is the true path)
			u32 nB = 0, bytesB = 0, tstartB = 0;
#pragma unroll 1
			for (u32 it = 0; it < 64u; it++) {
				const u32 prev = otz_lane_up1 (exit_cur, exit_cur);
				// the left neighbour's exit is slot-relative to ITS slot: one segment further left
				const u32 nt = lane && OTZ_X_KIND (prev) == OTZ_X_NORMAL ? OTZ_X_POS (prev) - S : OTZ_K1_NONE;
				const bool moved = lane != 0 && nt != t && nt != OTZ_K1_NONE;
				t = lane ? nt : t;
				const bool rewalk = moved && nt != firstA;
				const u32 old_exit = exit_cur;
				OTZ_K1_COUNT (7);
				if (otz_ballot (rewalk) != 0) {
					OTZ_K1_COUNT (8);
					otz_k1_ws W;
					u32 unused = 0, j = 0, ob = 0;
					otz_k1_walk<false> (L, in, n_in, rb, S / 8u, W, rewalk, nt, s, e, rowsB_cap * OTZ_IX_G, rowsB, rowsA, A.bm, nckA, rowsA_cap, firstA, &unused, &j, &ob);
					if (rewalk) {
						const bool early = W.merged && W.cpos != W.u;
						nB = W.cnt;
						bytesB = W.bytes;
						tstartB = W.u;
						onA = W.merged;
						mp = W.merged ? W.u : OTZ_K1_NONE;
						jm = early ? j : W.c * OTZ_IX_G;
						offA = early ? ob : onA ? otz_k1_row (rowsA, W.c, otz_k1_lane_off (rowsA_cap))->out : 0u;
						exit_cur = onA ? A.exit : W.exit;
					}
				}
				if (moved && !rewalk) { // lands on walk A's first token: confirmed as it stands
					nB = bytesB = 0;
					onA = true;
					mp = firstA;
					jm = offA = 0;
					exit_cur = A.exit;
				}
				if (otz_ballot (moved && exit_cur != old_exit) == 0) {
					break;
				}
			}
			OTZ_K1_MARK (4);
			// ---- acceptance: lanes up to the first one whose path does not simply run on
			const u64 termm = otz_ballot (OTZ_X_KIND (exit_cur) != OTZ_X_NORMAL);
			u32 F = termm ? otz_ctz64 (termm) : 63u; // last accepted lane
			bool acc = lane <= F;
			// items: walk B's checkpoint rows, then (on walk A's path, with tokens left there) one
			// item at the landing point and walk A's rows behind it
			const u32 nB_items = (nB + OTZ_IX_G - 1u) / OTZ_IX_G;
			const bool tailA = onA && jm < A.cnt;
			const u32 rA0 = jm / OTZ_IX_G + 1u; // first row of walk A behind the landing point
			const u32 nA_rows = tailA && nckA > rA0 ? nckA - rA0 : 0u;
			const u32 cnt_all = nB_items + (tailA ? 1u + nA_rows : 0u);
			const u32 bytes_all = bytesB + (onA ? A.bytes - offA : 0u);
			u32 cnt = acc ? cnt_all : 0u, bytes = acc ? bytes_all : 0u;
			u32 bincl = otz_wave_scan_incl (bytes);
			u32 cincl = otz_wave_scan_incl (cnt);
			// the stream wants more than uncomp_size bytes (ref: pending copy/literal, Z_OK) --
			// unless it breaks before that point
			const u32 room = out_cap - opos;
			const u64 overm = otz_ballot (acc && bincl > room);
			if (overm != 0) {
				// A lane's path crosses the end of the output.  The reference fills the buffer, keeps the
				// rest of the token pending and returns Z_OK (:693-760, :805-830).  The lanes in front of
				// that one are accepted as they are; its own segment is indexed again, token by token,
				// below (cold path).
				const u32 Fo = otz_ctz64 (overm);
				serial_finish = true;
				acc = lane < Fo;
				cnt = acc ? cnt_all : 0u;
				bytes = acc ? bytes_all : 0u;
				bincl = otz_wave_scan_incl (bytes);
				cincl = otz_wave_scan_incl (cnt);
				F = Fo; // (the lane the stream goes on in)
			}
			const u32 produced = otz_readlane (bincl, 63u), n_new = otz_readlane (cincl, 63u);
			const u32 fkind = OTZ_X_KIND (otz_readlane (exit_cur, F));
			if (!serial_finish) {
				if (fkind == OTZ_X_INVALID || fkind == OTZ_X_TRUNC) {
					status = OZG_ST_DATA; // ref: Z_DATA_ERROR from get_bit / no matching code
					break;
				}
				if (fkind == OTZ_X_TRUNC_END) {
					status = OZG_ST_BUF;
					break;
				}
			}
			if (n_items + n_new + 1u > items_cap) {
				status = OTZ_IX_OVERFLOW;
				break;
			}
			// ---- the accepted lanes' items, in stream order.  Row by row: every load is
			// independent of the others (and a row is one coalesced access).
			{
				// block-relative bit position of this lane's slot bit 0 (modulo 2^32: lane 0's may lie in front of the block)
				const u32 bit0 = 8u * rb - OTZ_K1_WARM - (u32)block_bit + S * otz_opaque (lane);
				const u32 obase = opos - block_out + bincl - bytes;
				otz_ix_item *dst = items + n_items + (cincl - cnt);
				// (rows several at a time: the loads of a batch are in flight together -- one trip to memory per
				// batch, not per row; every lane loads: a row it does not need is row 0, which exists)
				if (otz_ballot (acc && nB_items != 0) != 0) {
					for (u32 r0 = 0; r0 < rowsB_cap; r0 += 4u) {
						if (otz_ballot (acc && r0 < nB_items) == 0) {
							break; // (no lane has that many rows)
						}
						otz_ix_item t[4];
#pragma unroll
						for (u32 k = 0; k < 4u; k++) {
							t[k] = *otz_k1_row (rowsB, acc && r0 + k < nB_items ? r0 + k : 0u, otz_k1_lane_off (rowsB_cap));
						}
#pragma unroll
						for (u32 k = 0; k < 4u; k++) {
							if (acc && r0 + k < nB_items) {
								otz_ix_item it = t[k];
								it.bit += bit0;
								it.out += obase;
								dst[r0 + k] = it;
							}
						}
					}
				}
				if (acc && tailA) {
					otz_ix_item it;
					it.bit = mp + bit0;
					it.out = obase + bytesB;
					dst[nB_items] = it;
				}
				for (u32 r0 = 1; r0 < rowsA_cap; r0 += OTZ_K1_ACC_BATCH) {
					if (otz_ballot (acc && tailA && r0 < nckA) == 0) {
						break; // (walk A's paths are shorter than that)
					}
					otz_ix_item t[OTZ_K1_ACC_BATCH];
#pragma unroll
					for (u32 k = 0; k < OTZ_K1_ACC_BATCH; k++) {
						const u32 r = r0 + k;
						t[k] = *otz_k1_row (rowsA, acc && tailA && r >= rA0 && r < nckA ? r : 0u, otz_k1_lane_off (rowsA_cap));
					}
					// (two items per store where a lane has both: its items lie one behind the other, and what a store
					// costs in the CU's memory path is the lines it touches -- 64 lanes, 64 places -- not its width)
#pragma unroll
					for (u32 k = 0; k < OTZ_K1_ACC_BATCH; k += 2u) {
						const u32 r = r0 + k;
						const bool v0 = acc && tailA && r >= rA0 && r < nckA, v1 = acc && tailA && r + 1u >= rA0 && r + 1u < nckA;
						otz_ix_item i0 = t[k], i1 = t[k + 1u];
						i0.bit += bit0;
						i0.out += obase + bytesB - offA;
						i1.bit += bit0;
						i1.out += obase + bytesB - offA;
						if (v0 && v1) {
							otz_u4 w;
							w.x = i0.bit;
							w.y = i0.out;
							w.z = i1.bit;
							w.w = i1.out;
							otz_st16 (&dst[nB_items + 1u + r - rA0], w);
						} else if (v0) {
							dst[nB_items + 1u + r - rA0] = i0;
						} else if (v1) {
							dst[nB_items + 1u + (r + 1u) - rA0] = i1;
						}
					}
				}
			}
			OTZ_K1_MARK (5);
			n_items += n_new;
			opos += produced;
			const u64 fslot_bit = (u64) (8 * ((int64_t)rb + (int64_t) (S / 8u) * (int64_t)F - (int64_t) (OTZ_K1_WARM / 8u)));
			if (serial_finish) {
				abs_bit = fslot_bit + otz_readlane (t, F); // lane F's true start
				break;
			}
			// where the stream goes on (lane F's exit), and how the block ends
			const u32 fexit = otz_readlane (exit_cur, F);
			const u32 ftstart = otz_readlane (onA ? A.u : tstartB, F);
			abs_bit = fslot_bit + OTZ_X_POS (fexit);
			if (fkind == OTZ_X_EOB) {
				block_done = true;
				eob_abs = fslot_bit + ftstart;
				end_bit = (u32) (eob_abs - block_bit);
				// the next block's first span: what this block took, and a little more
				const u64 took = abs_bit - block_bit;
				span_cap = took + (took >> 4) + 2048u;
				span_cap = span_cap < 64ull * OTZ_K1_SMIN ? 64ull * OTZ_K1_SMIN : span_cap;
			} else {
				end_bit = (u32) (abs_bit - block_bit);
			}
		}
		if (serial_finish) {
			// From the start of the round that overflowed (abs_bit, opos): every lane decodes the same
			// tokens (uniform control flow, LDS broadcasts) from 1 KB stretches staged like a block
			// header; lane 0 notes one item per OTZ_IX_G tokens until a token does not fit.
			const u32 room = out_cap - opos;
			u64 pos = abs_bit;
			u32 got = 0, ntok = 0;
			bool found = false;
			while (!found && status == 0) {
				const u32 hb = (u32) (pos >> 3);
				{
					const otz_u4 v = otz_k1_ld16 (in, n_in, (int64_t)hb + 16 * (int64_t)lane);
					otz_wave_sync ();
					u32 *d = &L.h.hdr_w[4u * lane];
					d[0] = v.x;
					d[1] = v.y;
					d[2] = v.z;
					d[3] = v.w;
					otz_wave_sync ();
				}
				u32 rel = (u32) (pos & 7u);
				while (rel + 128u <= 8u * 1024u) {
					const otz_k1_tok tk = otz_k1_decode_bits (L, otz_k1_hbits (L, rel), otz_k1_hbits (L, rel + 32u));
					if (tk.kind != 0 || got + tk.obytes > room) {
						if (tk.kind == 2u || pos + tk.nbits > total_bits) {
							status = OZG_ST_DATA; // (cannot happen on a chain the walks validated)
						}
						clip_abs = pos;
						abs_bit = pos + tk.nbits;
						found = true;
						break;
					}
					if (ntok % OTZ_IX_G == 0) {
						if (n_items + 2u > items_cap) {
							status = OTZ_IX_OVERFLOW;
							break;
						}
						if (lane == 0) {
							otz_ix_item it;
							it.bit = (u32) (pos - block_bit);
							it.out = opos - block_out + got;
							items[n_items] = it;
						}
						n_items++;
					}
					got += tk.obytes;
					pos += tk.nbits;
					rel += tk.nbits;
					ntok++;
				}
			}
			if (status == 0) {
				opos += got;
				end_bit = (u32) (clip_abs - block_bit);
				status = OTZ_K1_ST_NEED_OUT;
			}
		}
		if (status != 0 && status != OTZ_K1_ST_NEED_OUT) {
			break;
		}
		if (lane == 0) {
			otz_ix_item sent;
			sent.bit = end_bit;
			sent.out = opos - block_out;
			items[n_items] = sent;
			blk->kind = OTZ_IX_KIND_HUFF;
			blk->n_items = n_items;
			blk->bit_lo = (u32)block_bit;
			blk->bit_hi = (u32) (block_bit >> 32);
			blk->out_base = block_out;
			blk->out_len = opos - block_out;
			blk->hlit_hdist = hlit | (hdist << 16);
			blk->next = 0;
			if (prev_next_at) {
				*reinterpret_cast<u32 *> (region + prev_next_at) = rec;
			}
		}
		prev_next_at = rec + 28u;
		rec += (OTZ_IX_ITEMS_OFF + 8u * (n_items + 1u) + 31u) & ~31u;
		n_blocks++;
		if (status != 0) {
			break;
		}
		ended = last_block;
	}

	if (status == 0 && last_btype != 0) {
		// the reference returns Z_BUF_ERROR when every input byte had already
		// been loaded before the step that decodes the final EOB
		if (((eob_abs + 7u) >> 3) >= (u64)n_in) {
			flags |= OZG_F_REF_FALSE_REJECT;
		}
	}
	if (lane == 0) {
		head->status = status;
		head->total_out = opos;
		head->flags = flags;
		head->n_blocks = n_blocks;
		head->end_lo = (u32)abs_bit;
		head->end_hi = (u32) (abs_bit >> 32);
		head->clip_lo = (u32)clip_abs;
		head->clip_hi = (u32) (clip_abs >> 32);
	}
	if (PROF && prof && lane == 0) {
		for (u32 i = 0; i < 10; i++) {
			prof[i] = pc[i];
		}
		prof[10] = pstart; // (timeline diagnostics: start, end, place)
		prof[11] = otz_cycles ();
		prof[12] = otz_where ();
		prof[15] = prof[11] - pstart;
#ifdef OTZ_K1_EXP
		prof[13] = exp_steps[0];
		prof[14] = exp_steps[1];
		prof[10] = exp_pro[0]; // (instead of the timeline: cycles of the block prologues' parts)
		prof[11] = exp_pro[1];
		prof[12] = exp_pro[2];
#endif
	}
	return status == OTZ_IX_OVERFLOW;
}
