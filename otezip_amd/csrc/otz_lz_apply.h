// otz_lz_apply.h -- second half of the DEFLATE decode: one workgroup of OTZ_A_NT work-items (256: four
// wavefronts, eight such workgroups per CU) turns the index of otz_index.h into the entry's bytes.  (ref: the output side of inflate(),
// src/lib/deflate-dec.inc.c:693-812 -- literal store, do_copy_from_window :521-545.)
//
// Per Huffman block: rebuild the decode tables from the stored code lengths (all lanes),
// then take the block's items OTZ_A_NT at a time (fewer when their output would not fit one
// round, or their bits the stage buffer): lane j decodes item j -- at most OTZ_IX_G tokens
// between two known token boundaries, at a known output position.  Literals go to their
// final bytes of the round buffer in LDS, matches are parked at their destination and
// resolved once their source bytes are final; the round is then written out with coalesced
// 16-byte stores.
//
// The kernel is bound by the latency of dependent LDS round trips, not by issue slots: what
// buys time is wavefronts per CU.  So LDS holds only the round being decoded and the last
// OTZ_A_WIN bytes before it; the rest of the 32 KiB LZ77 window is the entry's own output in
// global memory, written at the end of every round -- a match that reaches further back reads
// it from there (L2 hits: the bytes were stored microseconds ago by this CU).  The CRC-32 is
// left to a streaming pass over the output (otz_out_crc_kernel).
#pragma once

#include "otz_inflate.h"
#include "otz_index.h"

#ifndef OTZ_A_NT
#define OTZ_A_NT 256u     // work-items of the apply kernel's workgroup: 256 or 512
#endif
// The sizes below put the workgroup's LDS at 18 allocation granules of 1280 bytes: seven workgroups share a CU (28
// wavefronts).  Measured, ms per step of the benchmark on one box: 512 work-items x 4 workgroups (4 KiB window,
// 16 KiB rounds) 2.51-2.53; 256 x 7 (2 KiB window, 8 KiB rounds) 2.40-2.43; 256 x 8 (1 KiB window, 7 KiB rounds: 16
// granules) 2.395, with 128 resolving work-items 2.34 -- the choice while an item was eight tokens (256 items are
// 7.4 KB of text).  Since an item is eight STEPS (up to 16 tokens, 36 bytes on average) 256 items are 9.2 KB and the
// 7 KiB rounds left a fifth of the lanes without an item: 256 x 7 with 8.75 KiB rounds 1.923-1.932 against 256 x 8
// 1.964-1.978 (the window 0.5 ... 1 KiB and the stage buffer 17 ... 18 bytes per work-item make no difference).
#ifndef OTZ_A_RMAX
#define OTZ_A_RMAX (OTZ_A_NT * 35u) // output bytes per round, at most
#endif
#ifndef OTZ_A_WIN
#define OTZ_A_WIN (OTZ_A_NT * 4u) // bytes before the round kept in LDS (a multiple of 16)
#endif
#ifndef OTZ_A_STAGE_PER
#define OTZ_A_STAGE_PER 18u
#endif
#define OTZ_A_STAGE_BYTES (OTZ_A_NT * OTZ_A_STAGE_PER) // the round's compressed bytes (16 bits a token on average; more: a shorter round), then its match lists
#define OTZ_A_STAGE_WORDS (OTZ_A_STAGE_BYTES / 4u)
static_assert (OTZ_A_NT == 512u || OTZ_A_NT == 256u, "eight or four wavefronts");
// The resolution loop is a chain of dependent iterations (as many as the round's longest chain of matches copying
// from matches: 5-25), so what counts is how long ONE iteration takes, and that is set by how many wavefronts of
// the CU are in it: half of the workgroup's wavefronts resolve, at raised priority, the others wait at the barrier
// and leave their issue slots to the bulk phases of the other workgroups.  Measured (ms per step of the
// benchmark), 512-lane workgroups: all 512 resolving 2.68, with priority 2.68; 256 with priority 2.57; 128 2.61;
// 192 / 320 2.59.  256-lane workgroups, 7 per CU: 256 resolving 2.43, 192 2.42, 128 2.38-2.41, 64 2.43; without
// the raised priority 2.43.
#ifndef OTZ_A_RES_LANES
#define OTZ_A_RES_LANES (OTZ_A_NT / 2u) // work-items that resolve the round's matches (whole wavefronts; the others wait at the barrier)
#endif
static_assert (OTZ_A_RES_LANES <= OTZ_A_NT && OTZ_A_RES_LANES % 64u == 0, "whole wavefronts of the workgroup");
#ifndef OTZ_A_RES_PRIO
#define OTZ_A_RES_PRIO 1
#endif
#ifndef OTZ_A_FAR_WIDE
#define OTZ_A_FAR_WIDE 1 // a copy from the entry's output in global memory reads 16 bytes in one load
#endif
#ifndef OTZ_A_DECIDE_AHEAD
#define OTZ_A_DECIDE_AHEAD 1 // a round's size is decided a round ahead (two barriers fewer per round)
#endif
#ifndef OTZ_A_ITEMS_AHEAD
#define OTZ_A_ITEMS_AHEAD 1
#endif
#define OTZ_A_RBUF (OTZ_A_WIN + 16u + OTZ_A_RMAX) // + the last, partial 16-byte chunk before the window
#define OTZ_A_FIN_WORDS (OTZ_A_RBUF / 32u + 2u)

// rbuf[i] is output byte p0 + i of the entry (p0 a multiple of 16, everything before it is in
// global memory); fin bit i: that byte is final.
struct otz_apply_lds {
	u8 rbuf[OTZ_A_RBUF + 32u]; // (+ what the wide reads of otz_resolve_list may touch past the end)
	u32 lit_lut[OTZ_LIT_LUT];
	u32 dist_lut[OTZ_DIST_LUT];
	union {
		u32 stage_w[OTZ_A_STAGE_WORDS];
		u8 stage[OTZ_A_STAGE_BYTES];
	};
	union {
		u32 fin[OTZ_A_FIN_WORDS];
		u8 ll[320]; // table build: code lengths
	};
	u32 wcnt[6][16]; // table build: symbols of each code length per wavefront
	otz_canon lit_c, dist_c;
	u32 scan[8];
	u32 status; // 0 while decoding, else the final status code
	u32 flag;
	u32 first_rej;
	u32 last_out;
	u32 dec[2][2]; // the round-size decision a round ahead, two slots in turn: [0] first item that does not fit, [1] bits of the round
	u32 dbg[4];
	u64 prof_last;
	u64 prof[16];
};
static_assert (OTZ_A_WIN % 16u == 0 && OTZ_A_RMAX % 32u == 0, "chunked flush and bitmap words");
static_assert (OTZ_A_RBUF <= 65535u, "buffer positions fit 16 bits");

// where the entry stands: opos = bytes decoded, flushed = bytes in global memory (a multiple of
// 16 until the end), p0 = output position of rbuf[0]
struct otz_a_pos {
	u32 opos, flushed, p0;
};

// ---------------------------------------------------------------- workgroup helpers
//
// Inclusive prefix sum over the workgroup's OTZ_A_NT work-items; *total gets the sum.  Uniform control flow.
OTZ_DEV u32 otz_a_block_scan_incl(u32 v, u32 *scratch, u32 *total) {
	const u32 lane = otz_lane ();
	v = otz_wave_scan_incl (v);
	otz_sync ();
	if (lane == 63u) {
		scratch[otz_wave ()] = v;
	}
	otz_sync ();
	u32 add = 0, tot = 0;
	for (u32 w = 0; w < OTZ_A_NT / 64u; w++) {
		const u32 x = scratch[w];
		if (w < otz_wave ()) {
			add += x;
		}
		tot += x;
	}
	*total = tot;
	return v + add;
}

// Both two-level decode tables from L.ll[0..hlit+hdist): the construction of otz_build_tables (otz_inflate.h: a
// symbol's canonical code is first[len] + its rank among the symbols of the same length, ranks from wavefront
// ballots, the symbol writes its own root slots or claims a sub-table under its root slot) for a workgroup of
// OTZ_A_NT work-items: the symbols are taken in chunks of 64 -- chunks 0..4 the 288 literal/length symbols
// (320 slots), chunk 5 the 32 distance symbols -- wavefront w takes chunks w, w + OTZ_A_NT / 64, ...; the root
// slots OTZ_A_NT at a time.  Returns false (uniformly) on an over-subscribed set, or if the sub-tables do not fit.
OTZ_DEV bool otz_a_build_tables(otz_apply_lds &L, u32 hlit, u32 hdist) {
	constexpr u32 LRSZ = 1u << OTZ_LIT_ROOT, DRSZ = 1u << OTZ_DIST_ROOT;
	constexpr u32 NW = OTZ_A_NT / 64u, PASSES = (6u + NW - 1u) / NW;
	static_assert (OTZ_A_NT >= 128u, "wavefronts 0 and 1 do the canonical bookkeeping");
	// (opaque: nothing derived from them is computed once before the caller's block loop and kept -- spilled -- across it)
	const u32 tid = otz_opaque (otz_tid ()), lane = otz_opaque (otz_lane ()), wave = otz_opaque (otz_wave ());
	const u64 below = ((u64)1 << lane) - 1u;
	for (u32 i = tid; i < OTZ_LIT_LUT; i += OTZ_A_NT) {
		L.lit_lut[i] = OTZ_LUT_NONE;
	}
	for (u32 i = tid; i < OTZ_DIST_LUT; i += OTZ_A_NT) {
		L.dist_lut[i] = OTZ_LUT_NONE;
	}
	// this work-item's symbols: one per pass
	u32 ls[PASSES], ranks[PASSES];
#pragma unroll
	for (u32 p = 0; p < PASSES; p++) {
		const u32 chunk = wave + p * NW;
		const u32 slot = 64u * chunk + lane; // 0..287: literal/length symbols, 320..351: distance symbols
		u32 l = 0;
		if (chunk < 5u && slot < hlit) {
			l = L.ll[slot];
		} else if (chunk == 5u && slot - 320u < hdist) {
			l = L.ll[hlit + slot - 320u];
		}
		u32 rank = 0; // symbols of the same length before this one in its chunk
		if (chunk < 6u) {
			if (lane < 16u) {
				L.wcnt[chunk][lane] = 0; // (the counts below come from the same wavefront: LDS keeps its order)
			}
			u64 rem = otz_ballot (l != 0); // one ballot per code length that occurs in this chunk
			while (rem != 0) {
				const u32 k = otz_readlane (l, otz_ctz64 (rem));
				const u64 m = otz_ballot (l == k);
				if (l == k) {
					rank = otz_popc64 (m & below);
				}
				if (lane == 0) {
					L.wcnt[chunk][k] = otz_popc64 (m);
				}
				rem &= ~m;
			}
		}
		ls[p] = l;
		ranks[p] = rank;
	}
	otz_sync ();
	// Canonical bookkeeping, one lane per code length (wavefront 0: literal/length code, wavefront 1: distance
	// code): first[l] = sum over j < l of count[j] << (l - j), off[l] = sum over j < l of count[j].  A set is
	// over-subscribed (ref: "left < 0") exactly when first[15] + count[15] exceeds 2^15.
	if (wave < 2u) {
		u32 cnt = 0;
		if (lane >= 1u && lane < 16u) {
			const u32 *wc = &L.wcnt[0][0] + lane; // one address, constant offsets
			cnt = wave == 1u ? wc[80] : wc[0] + wc[16] + wc[32] + wc[48] + wc[64];
		}
		u32 code = 0, off = 0;
		for (u32 j = 1; j < 15u; j++) {
			const u32 cj = otz_shfl (cnt, j);
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
			const u32 ok = code + cnt <= (1u << 15) ? 1u : 0u;
			if (wave == 0) {
				L.flag = ok;
			} else {
				L.first_rej = ok;
			}
		}
	}
	otz_sync ();
	if (L.flag == 0u || L.first_rej == 0u) {
		return false;
	}
	// the symbols' codes, bit-reversed (DEFLATE sends codes most significant bit first)
	u32 rrs[PASSES], es[PASSES];
#pragma unroll
	for (u32 p = 0; p < PASSES; p++) {
		const u32 chunk = wave + p * NW;
		const bool is_dist = chunk == 5u;
		const u32 sym = is_dist ? lane : 64u * chunk + lane;
		const u32 root = is_dist ? OTZ_DIST_ROOT : OTZ_LIT_ROOT, rsz = is_dist ? DRSZ : LRSZ;
		u32 *const lut = is_dist ? L.dist_lut : L.lit_lut;
		const u32 l = ls[p];
		u32 rr = 0, e = 0;
		if (l) {
			const otz_canon &c = is_dist ? L.dist_c : L.lit_c;
			u32 before = ranks[p];
			for (u32 w = 0; w < chunk && !is_dist; w++) {
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
		if (chunk < 6u) {
			// A very short code owns many root slots (a 4-bit literal code 32 of the 512): the wavefront fills those
			// together, a symbol at a time, instead of waiting for the lane that owns the most.
			u64 heavy = otz_ballot (l != 0 && l + 5u <= root);
			while (heavy != 0) {
				const u32 hs = otz_ctz64 (heavy);
				heavy &= heavy - 1u;
				const u32 lh = otz_readlane (l, hs), rh = otz_readlane (rr, hs), eh = otz_readlane (e, hs);
				for (u32 idx = rh + (lane << lh); idx < rsz; idx += 64u << lh) {
					lut[idx] = eh;
				}
			}
		}
		rrs[p] = rr;
		es[p] = e;
	}
	otz_sync ();
	// sub-table placement: one prefix sum over the root slots of both tables (literal sizes in the low half of the
	// scanned word, distance sizes in the high half), OTZ_A_NT root slots at a time
	u32 carry = 0;
	bool fits = true;
	for (u32 base = 0; base < LRSZ; base += OTZ_A_NT) {
		const u32 slot = base + tid;
		const u32 el = L.lit_lut[slot < LRSZ ? slot : 0u], ed = L.dist_lut[slot < DRSZ ? slot : 0u];
		// a root slot claimed by longer codes holds (widest code - root) << 8 at this point
		const u32 ml = slot < LRSZ && el && (el & 0xFFu) == 0 ? 1u << (el >> 8) : 0u;
		const u32 md = slot < DRSZ && ed && (ed & 0xFFu) == 0 ? 1u << (ed >> 8) : 0u;
		u32 total;
		const u32 incl = carry + otz_a_block_scan_incl (ml | (md << 16), L.scan, &total);
		carry += total;
		fits = fits && (carry & 0xFFFFu) <= OTZ_LIT_LUT - LRSZ && (carry >> 16) <= OTZ_DIST_LUT - DRSZ;
		if (fits) {
			if (ml) {
				L.lit_lut[slot] = OTZ_LUT_LINK | ((LRSZ + (incl & 0xFFFFu) - ml) << 16) | ((el >> 8) << 8) | OTZ_LUT_NONE;
			}
			if (md) {
				L.dist_lut[slot] = OTZ_LUT_LINK | ((DRSZ + (incl >> 16) - md) << 16) | ((ed >> 8) << 8) | OTZ_LUT_NONE;
			}
		}
		otz_sync ();
	}
	if (!fits) {
		return false;
	}
#pragma unroll
	for (u32 p = 0; p < PASSES; p++) {
		const u32 chunk = wave + p * NW;
		const bool is_dist = chunk == 5u;
		const u32 root = is_dist ? OTZ_DIST_ROOT : OTZ_LIT_ROOT, rsz = is_dist ? DRSZ : LRSZ;
		u32 *const lut = is_dist ? L.dist_lut : L.lit_lut;
		const u32 l = ls[p];
		if (l > root) {
			const u32 link = lut[rrs[p] & (rsz - 1u)];
			const u32 sb = (link >> 8) & 15u, sbase = (link >> 16) & 0x7FFFu;
			for (u32 idx = rrs[p] >> root; idx < (1u << sb); idx += 1u << (l - root)) {
				lut[sbase + idx] = es[p];
			}
		}
	}
	otz_sync ();
	return true;
}

// ---------------------------------------------------------------- placement
//
// Lane j decodes item j: at most OTZ_IX_G STEPS (a step is one token, or two literals) from stage bit s_bit
// up to e_bit, first output byte at buffer position ri.  Literals go to their final bytes (and are marked
// final); a match is parked in the first three bytes of its own destination (23 bits:
// distance-1 : 15, length-3 : 8 -- a match is at least three bytes long) and its destination
// is noted.  The loop has one exit and no branch in its body (what a lane does not need is computed and
// not used: the kernel's time follows its instruction count, and the execution-mask bookkeeping of a loop
// that lanes leave from the middle costs more), and a step reloads its three stage words instead of
// sliding a window.
// Afterwards the destinations of all matches of the round are written, in stream order, to
// `list` (buffer positions, u16): a prefix sum over the lanes' match counts gives every lane
// its slice.  Returns the number of matches of the round.  All OTZ_A_NT work-items.
static_assert (OTZ_IX_G <= 8u, "up to eight 16-bit destinations are packed into four 32-bit registers");
#define OTZ_A_LIST_CAP (OTZ_A_NT * OTZ_IX_G) // matches of a round, at most
static_assert (OTZ_A_LIST_CAP * 2u <= OTZ_A_STAGE_BYTES, "the match lists of a round fit the stage buffer");
OTZ_DEV u32 otz_place_items(otz_apply_lds &L, bool active, u32 s_bit, u32 e_bit, u32 ri, u32 p0, u16 *list, u32 *n_far) {
	u32 nm = 0;
	u32 k0 = 0, k1 = 0, k2 = 0, k3 = 0; // the matches' destinations, newest in the low half of k0
	u32 farm = 0;                       // bit k: the source of the k-th newest match starts before rbuf[0]
	// One exit test and no way out of the middle of the body: a lane that has to stop raises a bit above every
	// position in its own `pos` (1: the index and the stream disagree, 2: distance beyond the output start).  The
	// distance entry is looked up for every lane -- a literal reads some entry and ignores it: a branch around the
	// lookup would cost more than the lookup, and every step has both kinds.
	u32 pos = active ? s_bit : 0xC0000000u;
	for (u32 i = 0; i < OTZ_IX_G && pos < e_bit; i++) {
		const u32 *wp = &L.stage_w[otz_opaque (pos >> 5)];
		const u32 w0 = wp[0], w1 = wp[1], w2 = wp[2];
		const u32 lo = otz_alignbit (w1, w0, pos), hi = otz_alignbit (w2, w1, pos);
		const u32 e = otz_lut2<OTZ_LIT_ROOT> (L.lit_lut, lo);
		const u32 len = e & 15u, kind = (e >> 4) & 3u;
		const bool is_lit = kind == OTZ_K_LIT, is_len = kind == OTZ_K_LEN;
		const u32 xb = (e >> 8) & 15u;
		const u32 val = (e >> 16) + otz_bfe (lo, len, xb); // the literal, or the match length
		const u32 used = len + xb; // <= 20
		const u32 db = otz_alignbit (hi, lo, used);
		// The second lookup means what the first one makes of it (as in the index kernel's walk): behind a length code
		// the distance entry, behind a literal the entry of the NEXT token -- and if that is a literal too, and lies
		// inside the item, the step places both.  An item is at most OTZ_IX_G steps of the index kernel's walk, which
		// pairs literals no more eagerly than this (otz_huff_index.h), so OTZ_IX_G steps here always reach its end.
		const u32 *lut2 = is_len ? L.dist_lut : L.lit_lut;
		const u32 rbits = is_len ? OTZ_DIST_ROOT : OTZ_LIT_ROOT;
		u32 d = lut2[db & ((1u << rbits) - 1u)];
		if ((int)d < 0) { // (a link: see otz_lut2)
			d = lut2[((d >> 16) & 0x7FFFu) + otz_bfe (db, rbits, (d >> 8) & 15u)];
		}
		otz_touch32 (d);
		const u32 dl = d & 15u, dxb = (d >> 8) & 15u; // dl + dxb <= 28
		const u32 dist = (d >> 16) + otz_bfe (db, dl, dxb);
		const bool pair = is_lit && ((d >> 4) & 3u) == OTZ_K_LIT && dl != 0 && pos + len + dl <= e_bit;
		// (a distance beyond the output start: ref reads window bytes nobody wrote (:524): no parity, fail cleanly;
		// anything else that is neither a literal nor a length with a distance: the index kernel validated the
		// chain, so this is an internal error)
		const bool ok_len = is_len && ((d >> 4) & 3u) == OTZ_K_LEN;
		const bool beyond = ok_len && dist > p0 + ri;
		const bool take = ok_len && !beyond;
		if (is_lit) {
			L.rbuf[ri] = (u8)val;
			const u64 fm = (u64) (pair ? 3u : 1u) << (ri & 31u);
			otz_lds_or (&L.fin[ri >> 5], (u32)fm);
			if ((u32) (fm >> 32)) { // (the second literal's bit lies in the next word)
				otz_lds_or (&L.fin[(ri >> 5) + 1u], 1u);
			}
		}
		if (pair) {
			L.rbuf[ri + 1u] = (u8) (d >> 16);
		}
		if (take) {
			const u32 cd = (dist - 1u) | ((val - 3u) << 15); // the parked match
			L.rbuf[ri] = (u8)cd;
			L.rbuf[ri + 1u] = (u8) (cd >> 8);
			L.rbuf[ri + 2u] = (u8) (cd >> 16);
			k3 = otz_alignbit (k3, k2, 16);
			k2 = otz_alignbit (k2, k1, 16);
			k1 = otz_alignbit (k1, k0, 16);
			k0 = (k0 << 16) | ri;
			farm = (farm << 1) | (dist > ri ? 1u : 0u);
			nm += 1u;
		}
		ri += is_lit ? (pair ? 2u : 1u) : take ? val : 0u;
		pos = is_lit ? pos + len + (pair ? dl : 0u) : take ? pos + used + dl + dxb : beyond ? 0x80000002u : 0x80000001u;
	}
	u32 err = 0;
	if (active) {
		err = (pos & 0x80000000u) ? pos & 3u : pos != e_bit ? 1u : 0u;
		pos &= 0x7FFFFFFFu;
	}
	if (err == 2u) {
		otz_set_status (L, OZG_ST_DATA);
	} else if (err != 0) {
		L.dbg[0] = 2; // the index and the stream disagree
		L.dbg[1] = otz_tid ();
		L.dbg[2] = pos;
		L.dbg[3] = e_bit;
		otz_set_status (L, OZG_ST_INTERNAL);
	}
	// two lists: the matches whose source lies in the buffer in stream order from list[0] on, the ones whose source
	// starts in the entry's output in global memory from the end of the list's room downwards (in any order)
	const u32 nf = otz_popc (farm), nn = nm - nf;
	u32 total;
	const u32 incl = otz_a_block_scan_incl (nn | (nf << 16), L.scan, &total); // (two barriers: every stage read is done)
	const u32 at_n = (incl & 0xFFFFu) - nn, at_f = (incl >> 16) - nf;
	// match number q (in stream order) is the (nm - 1 - q)-th newest
#pragma unroll
	for (u32 k = 0; k < OTZ_IX_G; k++) {
		if (k < nm) {
			const u32 w = k < 2u ? k0 : k < 4u ? k1 : k < 6u ? k2 : k3;
			const u32 older_far = otz_popc (farm >> (k + 1u));
			const bool isfar = ((farm >> k) & 1u) != 0;
			const u32 idx = isfar ? OTZ_A_LIST_CAP - 1u - (at_f + otz_popc (farm & ((1u << k) - 1u))) : at_n + (nm - 1u - k) - older_far;
			list[idx] = (u16) (w >> (16u * (k & 1u)));
		}
	}
	*n_far = total >> 16;
	return total & 0xFFFFu;
}

// One step of a copy: n <= 16 bytes (lo: the first eight, hi: the rest) to buffer position o, then their final bits.
// (An unaligned 2-byte store is cheap, an unaligned 4-byte store costs three of them.)
OTZ_DEV void otz_a_put(otz_apply_lds &L, u32 o, u64 lo, u64 hi, u32 n) {
	u8 *dst = &L.rbuf[o];
	u64 v = lo;
	if (n & 16u) {
		otz_st64u (dst, lo);
		otz_st64u (dst + 8, hi);
	} else {
		if (n & 8u) {
			otz_st64u (dst, lo);
			v = hi;
			dst += 8;
		}
		if (n & 4u) {
			otz_st16u (dst, (u16)v);
			otz_st16u (dst + 2, (u16) (v >> 16));
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
	otz_order_stores ();
	const u64 mask = (((u64)1 << n) - 1u) << (o & 31u);
	otz_lds_or (&L.fin[o >> 5], (u32)mask);
	if ((u32) (mask >> 32)) {
		otz_lds_or (&L.fin[(o >> 5) + 1u], (u32) (mask >> 32));
	}
}

// The matches whose source STARTS before rbuf[0], in the entry's output in global memory.  Such a source ends
// before the round does (the buffer keeps OTZ_A_WIN >= 258 bytes in front of the round, or the entry's whole
// output so far), so every byte of it was final before the round began: these copies depend on nothing and are
// made first, all of a lane's loads in flight together -- one trip to the L2 per round instead of one inside
// every iteration of the resolution loop below.  (The part of a source that lies in the buffer is read there.)
OTZ_DEV void otz_far_copies(otz_apply_lds &L, const otz_job &J, u32 p0, const u16 *list, u32 n_far) {
	for (u32 m = otz_tid (); m < n_far; m += OTZ_A_NT) {
		u32 o = list[OTZ_A_LIST_CAP - 1u - m];
		const u64 dw = otz_ld64u (&L.rbuf[o & ~3u]);
		const u32 cd = otz_alignbyte ((u32) (dw >> 32), (u32)dw, o) & 0xFFFFFFu;
		u32 pend = (cd >> 15) + 3u;
		int si = (int)o - (int) ((cd & 0x7FFFu) + 1u); // (p0 + si >= 0: the distance was checked against the position)
		while (pend != 0) {
			u32 n = pend < 16u ? pend : 16u;
			const bool far = si < 0;
			if (far) {
				n = n < (u32)-si ? n : (u32)-si; // (a step does not straddle rbuf[0])
			}
			u64 lo, hi;
			const u32 gpos = p0 + (u32)si; // (far: where the source starts in the entry's output)
			if (far && OTZ_A_FAR_WIDE && gpos + 16u <= J.out_cap) {
				// one 16-byte load at the source's own byte address (the buffer reaches that far; what lies behind the
				// n bytes is not used): a wavefront's copies come from 64 different places, and what a load costs in the
				// CU's memory path is the lines it touches -- one or two here, five with a word at a time
				otz_ldg16u (J.out + gpos, &lo, &hi);
			} else {
				u32 s0, s1, s2, s3, s4;
				u32 q;
				if (far) {
					const u8 *g = J.out + gpos;
					const u32 *gp = reinterpret_cast<const u32 *> ((uintptr_t)g & ~(uintptr_t)3);
					q = (u32) (uintptr_t)g & 3u;
					const u32 span = q + n; // 1..19: only the words that hold bytes of the source (they lie inside the entry's output)
					s0 = otz_ldg32 (gp);
					s1 = span > 4u ? otz_ldg32 (gp + 1) : 0u;
					s2 = span > 8u ? otz_ldg32 (gp + 2) : 0u;
					s3 = span > 12u ? otz_ldg32 (gp + 3) : 0u;
					s4 = span > 16u ? otz_ldg32 (gp + 4) : 0u;
				} else {
					const u32 *sp = reinterpret_cast<const u32 *> (&L.rbuf[(u32)si & ~3u]);
					q = (u32)si & 3u;
					s0 = sp[0];
					s1 = sp[1];
					s2 = sp[2];
					s3 = sp[3];
					s4 = sp[4];
				}
				lo = otz_alignbyte (s1, s0, q) | ((u64)otz_alignbyte (s2, s1, q) << 32);
				hi = otz_alignbyte (s3, s2, q) | ((u64)otz_alignbyte (s4, s3, q) << 32);
			}
			otz_a_put (L, o, lo, hi, n);
			o += n;
			si += (int)n;
			pend -= n;
		}
	}
}

// ---------------------------------------------------------------- LZ77 resolution
//
// The copies whose source lies in the buffer, in stream order: lane j takes matches j, j + OTZ_A_RES_LANES, ... of
// the list, one after the other, and takes the next as soon as one is done (no lane waits for the slowest of a
// batch).  Consecutive matches span a few KB of output, so most sources lie in stretches that are already final;
// the others wait for exactly the bytes they need: each step copies up to 16 bytes once the per-byte final bits
// of THOSE source bytes are set (two bitmap words and the 16 source bytes in one LDS round trip), then sets the
// bits of what it wrote.  The earliest unresolved match of the round is some lane's current one and never waits,
// so the loop cannot stall; it is closed by ballots (a convergent exit keeps the waiting lanes and the lanes they
// wait for in the same loop -- see otz_resolve in otz_inflate.h).
// The loop runs as many iterations as the round's longest chain of matches copying from matches, so an iteration
// is kept short: a lane's NEXT match -- its place in the list two ahead, its parked descriptor one ahead -- is
// fetched while the current one is copied, so that a step waits for one LDS round trip, not three.
// An overlapping match (distance < 16) copies one period, then two, four, ...
#define OTZ_A_NOMATCH 0u // (no match is written to buffer position 0: a match needs bytes in front of it)
OTZ_DEV void otz_resolve_list(otz_apply_lds &L, const otz_job &J, const u16 *list, u32 n_matches) {
	u32 m = otz_tid (); // list index of the match after next
	u32 on1 = OTZ_A_NOMATCH, on2 = OTZ_A_NOMATCH; // destinations of the next match and of the one after
	if (m < n_matches) {
		on1 = list[m];
		m += OTZ_A_RES_LANES;
	}
	if (m < n_matches) {
		on2 = list[m];
	}
	u64 dn1 = otz_ld64u (&L.rbuf[on1 & ~3u]); // the words that hold the next match's parked descriptor
	u32 pend = 0, pdist = 1, o = 0, si = 0;
	u32 spins = 0;
	for (;;) {
		OTZ_COUNT (L, J, OTZ_P_B_ITERS, 1);
		if (pend == 0 && on1 != OTZ_A_NOMATCH) {
			// ---- next match of this lane: it is in registers
			const u32 cd = otz_alignbyte ((u32) (dn1 >> 32), (u32)dn1, on1) & 0xFFFFFFu;
			o = on1;
			pdist = (cd & 0x7FFFu) + 1u;
			pend = (cd >> 15) + 3u;
			si = o - pdist; // (>= 0: the matches that reach in front of the buffer are otz_far_copies')
			on1 = on2;
			m += OTZ_A_RES_LANES;
		}
		// ---- ask for what the next take needs, every iteration and whether or not it changed: loads that are
		// always issued here are in flight while this step's reads are, and their registers are the loop's own
		// (a load inside the branch above is waited for on the spot: its value has to be merged with the old one)
		const u32 on2_next = list[m < n_matches ? m : 0u];
		const u64 dn1_next = otz_ld64u (&L.rbuf[on1 & ~3u]);
		// ---- one step: up to 16 bytes that do not overlap their own source
		u32 n = pend < 16u ? pend : 16u;
		n = n < pdist ? n : pdist;
		const u64 have = otz_ld64u (&L.fin[si >> 5]); // (n <= 16 bits: words si/32 and si/32 + 1 hold them)
		otz_order_loads (); // "flag, then data": the source bytes are read after the bits that declare them final
		const u64 need = (((u64)1 << n) - 1u) << (si & 31u);
		const bool go = pend != 0 && (have & need) == need;
		if (go) {
			// the source bytes: dword-aligned reads, then a byte funnel shift (LDS reads past the bytes
			// used land in the buffer's slack and are ignored)
			const u32 *sp = reinterpret_cast<const u32 *> (&L.rbuf[si & ~3u]);
			const u64 s01 = otz_ld64u (sp);
			const u32 s0 = (u32)s01, s1 = (u32) (s01 >> 32), s2 = sp[2];
			u32 s3 = 0, s4 = 0;
			if (n > 8u) {
				const u64 s34 = otz_ld64u (sp + 3);
				s3 = (u32)s34;
				s4 = (u32) (s34 >> 32);
			}
			const u32 q = si & 3u;
			const u64 lo = otz_alignbyte (s1, s0, q) | ((u64)otz_alignbyte (s2, s1, q) << 32);
			const u64 hi = otz_alignbyte (s3, s2, q) | ((u64)otz_alignbyte (s4, s3, q) << 32);
			otz_a_put (L, o, lo, hi, n);
			o += n;
			pend -= n;
			// a whole period written: same source, twice the reach; otherwise the source slides along
			const bool dbl = n == pdist && pdist < 16u;
			si += dbl ? 0u : n;
			pdist += dbl ? n : 0u;
		}
		on2 = m < n_matches ? on2_next : OTZ_A_NOMATCH;
		dn1 = dn1_next;
#ifdef OTZ_A_RES_STATS
		if (J.prof) { // diagnostics: lanes that copied / lanes that had a match in hand, this iteration (wavefront 0)
			const u32 ng = otz_popc64 (otz_ballot (go)), np = otz_popc64 (otz_ballot (go || pend != 0));
			OTZ_COUNT (L, J, 11, ng);
			OTZ_COUNT (L, J, 0, np);
		}
#endif
		if (otz_ballot (pend != 0 || on1 != OTZ_A_NOMATCH) == 0) {
			break;
		}
		if (otz_ballot (go) == 0) { // every live lane of this wavefront waits on another wavefront
#ifndef OTZ_A_RES_NOSLEEP
			otz_backoff ();
#endif
			if ((++spins & 63u) == 0) {
				if (otz_get_status (L) != 0) {
					break;
				}
				if (spins > OTZ_SPIN_LIMIT) {
					L.dbg[0] = 3; // waited too long for source bytes
					L.dbg[1] = otz_tid ();
					L.dbg[2] = o;
					L.dbg[3] = pdist | (pend << 16);
					otz_set_status (L, OZG_ST_INTERNAL);
					break;
				}
			}
		}
	}
}

// ---------------------------------------------------------------- rounds
//
// Stage the round's compressed bytes: n_bytes from in + byte0, from the 16-byte-aligned address
// at or below it (otz_stage_load, but only the chunks the round reads: one load per lane, not two).
OTZ_DEV void otz_a_stage(otz_apply_lds &L, const otz_job &J, u32 byte0, u32 n_bytes) {
	const u8 *p = J.in + byte0;
	const u32 skew = (u32) ((uintptr_t)p & 15u);
	const u8 *ga = p - skew;
	const u8 *lo = J.in, *hi = J.in + J.n_in;
	u32 chunks = (skew + n_bytes + 15u) >> 4;
	chunks = chunks < OTZ_A_STAGE_BYTES / 16u ? chunks : OTZ_A_STAGE_BYTES / 16u;
	for (u32 c = otz_opaque (otz_tid ()); c < chunks; c += OTZ_A_NT) { // (opaque: no per-lane address kept across the rounds)
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
}

//
// Start of a round: nothing after the decoded bytes is final.  (ll shares the bitmap's storage.)
OTZ_DEV void otz_a_begin(otz_apply_lds &L, const otz_a_pos &P) {
	const u32 h = P.opos - P.p0;
	for (u32 i = otz_tid (); i < OTZ_A_FIN_WORDS; i += OTZ_A_NT) {
		L.fin[i] = i < h >> 5 ? 0xFFFFFFFFu : i == h >> 5 ? (1u << (h & 31u)) - 1u : 0u;
	}
}

// End of a round that decoded `produced` bytes (all final, a barrier passed): store the whole
// 16-byte chunks not yet in global memory, then slide the buffer so that it starts OTZ_A_WIN
// bytes (rounded down to a chunk) before the end.  All work-items; ends with a barrier.
OTZ_DEV void otz_a_advance(otz_apply_lds &L, const otz_job &J, otz_a_pos &P, u32 produced) {
	const u32 tid = otz_tid ();
	const u32 end = P.opos + produced;
	const u32 up = end & ~15u;
	for (u32 c = tid; c < (up - P.flushed) >> 4; c += OTZ_A_NT) {
		otz_st16_aligned (J.out + P.flushed + 16u * c, otz_ld16_aligned (&L.rbuf[P.flushed - P.p0 + 16u * c]));
	}
	P.flushed = up;
	P.opos = end;
	const u32 keep = up > OTZ_A_WIN ? up - OTZ_A_WIN : 0u;
	if (keep > P.p0) {
		constexpr u32 CH = (OTZ_A_WIN / 16u + 1u + OTZ_A_NT - 1u) / OTZ_A_NT;
		const u32 d = keep - P.p0, chunks = (end - keep + 15u) >> 4;
		otz_u4 t[CH];
#pragma unroll
		for (u32 k = 0; k < CH; k++) {
			const u32 c = tid + k * OTZ_A_NT;
			if (c < chunks) {
				t[k] = otz_ld16_aligned (&L.rbuf[d + 16u * c]);
			}
		}
		otz_sync ();
#pragma unroll
		for (u32 k = 0; k < CH; k++) {
			const u32 c = tid + k * OTZ_A_NT;
			if (c < chunks) {
				otz_st16_aligned (&L.rbuf[16u * c], t[k]);
			}
		}
		P.p0 = keep;
	}
	otz_sync ();
}

// The last, partial chunk.  All work-items.
OTZ_DEV void otz_a_finish(otz_apply_lds &L, const otz_job &J, otz_a_pos &P) {
	const u32 tid = otz_tid ();
	if (tid < P.opos - P.flushed) {
		J.out[P.flushed + tid] = L.rbuf[P.flushed - P.p0 + tid];
	}
	P.flushed = P.opos;
}

// Output byte at position p, wherever it lives.
OTZ_DEV u8 otz_a_byte(const otz_apply_lds &L, const otz_job &J, const otz_a_pos &P, u32 p) {
	return p < P.p0 ? J.out[p] : L.rbuf[p - P.p0];
}

// The verdict on the items base, base + 1, ... of a block as one round (lane j: item base + j, a / z its start and
// end, first the round's first item): dec[0] = the first lane whose item no longer fits (the round's output within
// OTZ_A_RMAX, its bits within the stage buffer), dec[1] = the bits of the items in front of it.  All work-items;
// dec was reset a barrier ago and is read a barrier later.
OTZ_DEV void otz_a_decide(otz_apply_lds &L, const otz_job &J, u32 *dec, u64 bit_base, u32 base, u32 n_items, otz_ix_item a, otz_ix_item z, otz_ix_item first) {
	const u64 abs0 = bit_base + first.bit;
	const u32 byte0 = (u32) (abs0 >> 3);
	const u32 skew = (u32) ((uintptr_t) (J.in + byte0) & 15u);
	const u32 p0 = skew * 8u + (u32) (abs0 & 7u);
	// a walk reads up to five words past the one holding its position
	const u32 bit_room = 8u * OTZ_A_STAGE_BYTES - 192u - p0;
	const bool fits = base + otz_tid () < n_items && z.out - first.out <= OTZ_A_RMAX && z.bit - first.bit <= bit_room && z.out >= a.out && z.bit >= a.bit;
	otz_wave_min_tid (&dec[0], !fits);
	const u64 fm = otz_ballot (fits);
	if (fm != 0 && otz_lane () == 63u - otz_clz64 (fm)) { // (the last fitting lane of each wavefront knows how far the bits reach)
		otz_lds_max (&dec[1], z.bit - first.bit);
	}
}

// All OTZ_A_NT work-items.  Work-item 0 writes *res (crc32 and the CRC flag are left to the CRC pass).
// end_bit (optional): where the stream ended, for the z_stream seam's total_in.
OTZ_DEV void otz_apply_entry(otz_apply_lds &L, const otz_job &J, const otz_ix_head *head, const u8 *region, ozg_result *res, u64 *end_bit) {
	const u32 tid = otz_tid ();
	const int hst = head->status;
	if (hst == OTZ_IX_OVERFLOW) {
		return; // the index did not fit its region: otz_inflate_kernel decodes the entry on its own
	}
	if (end_bit && tid == 0) {
		*end_bit = hst == 0 || hst == OTZ_IX_NEED_OUT ? ((u64)head->end_hi << 32) | head->end_lo : 0u;
	}
	const bool need_out = hst == OTZ_IX_NEED_OUT && head->n_blocks != 0;
	if (hst != 0 && !need_out) {
		if (tid == 0) {
			res->status = hst == OTZ_IX_NEED_OUT ? OZG_ST_NEED_OUT : hst;
			res->total_out = head->total_out;
			res->crc32 = 0;
			res->flags = 0;
		}
		return;
	}
	otz_a_pos P = { 0, 0, 0 };
	u32 flags = head->flags;
	const u32 n_blocks = head->n_blocks;
	if (tid == 0) {
		otz_set_status (L, 0);
		L.dbg[0] = L.dbg[1] = L.dbg[2] = L.dbg[3] = 0;
		for (u32 i = 0; i < 16; i++) {
			L.prof[i] = 0;
		}
		L.prof_last = otz_cycles ();
		L.prof[OTZ_P_TOTAL] = 0 - L.prof_last;
	}
	otz_sync ();

	u32 boff = OTZ_IX_FIRST_BLOCK;
	for (u32 b = 0; b < n_blocks && otz_get_status (L) == 0; b++) {
		const otz_ix_block *blk = reinterpret_cast<const otz_ix_block *> (region + boff);
		const u32 kind = blk->kind, n_items = blk->n_items, out_len = blk->out_len;
		const u64 bit_base = ((u64)blk->bit_hi << 32) | blk->bit_lo;
		const u32 hlit = blk->hlit_hdist & 0xFFFFu, hdist = blk->hlit_hdist >> 16;
		const u32 next = blk->next;
		OTZ_COUNT (L, J, OTZ_P_BLOCKS, 1);
		if (blk->out_base != P.opos) {
			otz_fail_all (L, OZG_ST_INTERNAL);
			break;
		}
		if (kind == OTZ_IX_KIND_STORED) {
			const u32 src_byte = blk->bit_lo;
			u32 done = 0;
			while (done < out_len) {
				const u32 n = out_len - done < OTZ_A_RMAX ? out_len - done : OTZ_A_RMAX;
				const u32 h = P.opos - P.p0;
				for (u32 i = tid; i < n; i += OTZ_A_NT) {
					L.rbuf[h + i] = J.in[src_byte + done + i];
				}
				otz_sync ();
				otz_a_advance (L, J, P, n);
				done += n;
			}
			OTZ_MARK (L, J, OTZ_P_STORED);
			boff = next;
			continue;
		}
		// ---- Huffman block: tables from the stored code lengths
		for (u32 i = tid; i < OTZ_IX_LENS; i += OTZ_A_NT) {
			L.ll[i] = region[boff + 32u + i];
		}
		otz_sync ();
		if (!otz_a_build_tables (L, hlit, hdist)) {
			otz_fail_all (L, OZG_ST_INTERNAL);
			break;
		}
		OTZ_MARK (L, J, OTZ_P_TABLES);
		const otz_ix_item *items = reinterpret_cast<const otz_ix_item *> (region + boff + OTZ_IX_ITEMS_OFF);
#if OTZ_A_DECIDE_AHEAD
		// How many items a round takes is decided A ROUND AHEAD: the next round's items are asked for at the top of a
		// round, their verdict (the first item that does not fit the round buffer or the stage buffer) goes into one of
		// two LDS slots in front of the resolution loop and is read behind the barrier that closes it -- a barrier the
		// round has anyway.  Decided at the top of its own round it cost two barriers and an exposed trip to global
		// memory per round, with 32 wavefronts per CU waiting for issue slots at every barrier.
		u32 base = 0, par = 0;
		otz_ix_item a, z, first;
		{
			const u32 ia = tid < n_items ? tid : n_items;
			a = items[ia];
			z = items[ia < n_items ? ia + 1u : n_items];
			first = items[0];
		}
		if (tid == 0) {
			L.dec[0][0] = OTZ_A_NT;
			L.dec[0][1] = 0;
		}
		otz_sync ();
		otz_a_decide (L, J, L.dec[0], bit_base, base, n_items, a, z, first);
		otz_sync ();
		u32 n = otz_uniform (L.dec[0][0]), span_bits = L.dec[0][1];
		while (base < n_items) {
			const u64 abs0 = bit_base + first.bit;
			const u32 byte0 = (u32) (abs0 >> 3);
			const u32 skew = (u32) ((uintptr_t) (J.in + byte0) & 15u);
			const u32 p0 = skew * 8u + (u32) (abs0 & 7u);
			if (n == 0 || blk->out_base + first.out != P.opos) {
				otz_fail_all (L, OZG_ST_INTERNAL);
				break;
			}
			// ---- the next round's items
			const u32 nb = base + n;
			otz_ix_item a_n, z_n, first_n;
			{
				const u32 ia = nb + tid < n_items ? nb + tid : n_items;
				a_n = items[ia];
				z_n = items[ia < n_items ? ia + 1u : n_items];
				first_n = items[nb < n_items ? nb : n_items];
			}
			if (tid + 1u == n) {
				L.last_out = z.out; // (read after the barriers of otz_place_items)
			}
			if (tid == 0) { // the slot of the next decision (its last readers passed the barrier that ended the previous round)
				L.dec[par ^ 1u][0] = OTZ_A_NT;
				L.dec[par ^ 1u][1] = 0;
			}
#ifdef OTZ_ST_PROF
			OTZ_MARK (L, J, 1);
#endif
			otz_a_stage (L, J, byte0, (((u32)abs0 & 7u) + span_bits + 7u) / 8u + 24u);
#ifdef OTZ_ST_PROF
			OTZ_MARK (L, J, 6);
#endif
			otz_a_begin (L, P);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_STAGE);
			OTZ_COUNT (L, J, OTZ_P_ROUNDS, 1);
			const u32 s_bit = p0 + (a.bit - first.bit), e_bit = p0 + (z.bit - first.bit);
			// the match list takes over the stage buffer once every lane has read its tokens
			u16 *const list = reinterpret_cast<u16 *> (L.stage);
			u32 n_far = 0;
			const u32 n_matches = otz_place_items (L, tid < n, s_bit, e_bit, P.opos - P.p0 + (a.out - first.out), P.p0, list, &n_far);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_EMIT);
			if (otz_get_status (L) != 0) {
				break;
			}
			otz_far_copies (L, J, P.p0, list, n_far);
			otz_a_decide (L, J, L.dec[par ^ 1u], bit_base, nb, n_items, a_n, z_n, first_n);
#if OTZ_A_RES_PRIO
			otz_set_priority_high (); // (the resolution loop goes first, the bulk phases of the other workgroups fill the gaps)
#endif
			if (OTZ_A_RES_LANES == OTZ_A_NT || tid < OTZ_A_RES_LANES) {
				otz_resolve_list (L, J, list, n_matches);
			}
#if OTZ_A_RES_PRIO
			otz_set_priority_normal ();
#endif
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_RESOLVE);
			if (otz_get_status (L) != 0) {
				break;
			}
			const u32 n_next = otz_uniform (L.dec[par ^ 1u][0]), span_next = L.dec[par ^ 1u][1];
			otz_a_advance (L, J, P, L.last_out - first.out);
			OTZ_MARK (L, J, OTZ_P_FLUSH);
			base = nb;
			a = a_n;
			z = z_n;
			first = first_n;
			n = n_next;
			span_bits = span_next;
			par ^= 1u;
		}
#else
		u32 base = 0;
#if OTZ_A_ITEMS_AHEAD
		// the round's items are asked for a round ahead (behind the previous round's count, in front of its staging,
		// placement and resolution): a round then waits for ONE trip to global memory, its compressed bytes
		otz_ix_item a_n, z_n, first_n;
		{
			const u32 ia = tid < n_items ? tid : n_items;
			a_n = items[ia];
			z_n = items[ia < n_items ? ia + 1u : n_items];
			first_n = items[0];
		}
#endif
		while (base < n_items) {
			// ---- how many items this round takes
#if OTZ_A_ITEMS_AHEAD
			const otz_ix_item a = a_n, z = z_n, first = first_n;
#else
			const u32 ia = base + tid < n_items ? base + tid : n_items;
			const u32 ib = ia < n_items ? ia + 1u : n_items;
			const otz_ix_item a = items[ia], z = items[ib];
			const otz_ix_item first = items[base];
#endif
			const u64 abs0 = bit_base + first.bit;
			const u32 byte0 = (u32) (abs0 >> 3);
			const u32 skew = (u32) ((uintptr_t) (J.in + byte0) & 15u);
			const u32 p0 = skew * 8u + (u32) (abs0 & 7u);
			// a walk reads up to five words past the one holding its position
			const u32 bit_room = 8u * OTZ_A_STAGE_BYTES - 192u - p0;
			const bool fits = base + tid < n_items && z.out - first.out <= OTZ_A_RMAX && z.bit - first.bit <= bit_room && z.out >= a.out && z.bit >= a.bit;
			if (tid == 0) {
				L.first_rej = OTZ_A_NT;
				L.flag = 0;
			}
			otz_sync ();
			otz_wave_min_tid (&L.first_rej, !fits);
			{ // how far the round's bits reach (bounds the staging): the last fitting lane of each wavefront knows
				const u64 fm = otz_ballot (fits);
				if (fm != 0 && otz_lane () == 63u - otz_clz64 (fm)) {
					otz_lds_max (&L.flag, z.bit - first.bit);
				}
			}
			otz_sync ();
			const u32 n = otz_uniform (L.first_rej);
			const u32 span_bits = L.flag;
#if OTZ_A_ITEMS_AHEAD
			{
				const u32 nb = base + n;
				const u32 ia = nb + tid < n_items ? nb + tid : n_items;
				a_n = items[ia];
				z_n = items[ia < n_items ? ia + 1u : n_items];
				first_n = items[nb < n_items ? nb : n_items];
			}
#endif
#ifdef OTZ_ST_PROF
			OTZ_MARK (L, J, 1);
#endif
			if (n == 0 || blk->out_base + first.out != P.opos) {
				otz_fail_all (L, OZG_ST_INTERNAL);
				break;
			}
			if (tid + 1u == n) {
				L.last_out = z.out; // (read after the barriers of otz_place_items)
			}
			otz_a_stage (L, J, byte0, (((u32)abs0 & 7u) + span_bits + 7u) / 8u + 24u);
#ifdef OTZ_ST_PROF
			OTZ_MARK (L, J, 6);
#endif
			otz_a_begin (L, P);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_STAGE);
			OTZ_COUNT (L, J, OTZ_P_ROUNDS, 1);
			const u32 s_bit = p0 + (a.bit - first.bit), e_bit = p0 + (z.bit - first.bit);
			// the match list takes over the stage buffer once every lane has read its tokens
			u16 *const list = reinterpret_cast<u16 *> (L.stage);
			u32 n_far = 0;
			const u32 n_matches = otz_place_items (L, tid < n, s_bit, e_bit, P.opos - P.p0 + (a.out - first.out), P.p0, list, &n_far);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_EMIT);
			if (otz_get_status (L) != 0) {
				break;
			}
			otz_far_copies (L, J, P.p0, list, n_far);
#if OTZ_A_RES_PRIO
			otz_set_priority_high (); // (the resolution loop goes first, the bulk phases of the other workgroups fill the gaps)
#endif
			if (OTZ_A_RES_LANES == OTZ_A_NT || tid < OTZ_A_RES_LANES) {
				otz_resolve_list (L, J, list, n_matches);
			}
#if OTZ_A_RES_PRIO
			otz_set_priority_normal ();
#endif
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_RESOLVE);
			if (otz_get_status (L) != 0) {
				break;
			}
			otz_a_advance (L, J, P, L.last_out - first.out);
			OTZ_MARK (L, J, OTZ_P_FLUSH);
			base += n;
		}
#endif
		boff = next;
	}

	int status = otz_get_status (L);
	if (J.prof && tid == 0) {
		L.prof[OTZ_P_TOTAL] += otz_cycles ();
	}
	if (status == 0 && P.opos != head->total_out) {
		status = OZG_ST_INTERNAL;
	}
	if (status == 0 && need_out) {
		// The token at head->clip does not fit: the reference reads it, writes what fits of a match
		// and returns Z_OK with the buffer full (ref :521-545, :805-830).  Same here: the bytes are
		// written, the entry reports OZG_ST_NEED_OUT.  (The tables of the last block are still up.)
		const u64 clip = ((u64)head->clip_hi << 32) | head->clip_lo;
		const u32 byte0 = (u32) (clip >> 3);
		const u32 skew = (u32) ((uintptr_t) (J.in + byte0) & 15u);
		otz_a_stage (L, J, byte0, 32u);
		otz_sync ();
		const u32 p = skew * 8u + (u32) (clip & 7u);
		const u32 w = p >> 5;
		const otz_tok t = otz_decode_bits (L, otz_alignbit (L.stage_w[w + 1u], L.stage_w[w], p), otz_alignbit (L.stage_w[w + 2u], L.stage_w[w + 1u], p));
		const u32 room = J.out_cap - P.opos;
		u32 n = 0;
		if (t.kind == OTZ_K_LEN) {
			if (t.dist > P.opos) {
				status = OZG_ST_DATA;
			} else {
				n = t.val < room ? t.val : room;
				if (tid == 0) {
					for (u32 i = 0; i < n; i++) { // (byte by byte: the copy may overlap itself)
						L.rbuf[P.opos - P.p0 + i] = otz_a_byte (L, J, P, P.opos + i - t.dist);
					}
				}
			}
		}
		otz_sync ();
		if (status == 0) {
			otz_a_advance (L, J, P, n);
			otz_a_finish (L, J, P);
			if (tid == 0) {
				res->status = OZG_ST_NEED_OUT;
				res->total_out = P.opos;
				res->crc32 = 0;
				res->flags = 0;
			}
			return;
		}
	}
	if (status == 0) {
		const u32 produced = P.opos;
		otz_a_finish (L, J, P);
		if (produced < J.out_cap) {
			// short stream, zero-filled tail
			flags |= OZG_F_SHORT_STREAM;
			const u32 a16 = (produced + 15u) & ~15u;
			const u32 head_end = a16 < J.out_cap ? a16 : J.out_cap;
			if (produced + tid < head_end) {
				J.out[produced + tid] = 0;
			}
			const otz_u4 zero = { 0, 0, 0, 0 };
			u32 c = a16 + 16u * tid;
			for (; c + 16u <= J.out_cap; c += 16u * OTZ_A_NT) {
				otz_st16_aligned (J.out + c, zero);
			}
			if (c < J.out_cap) { // (one work-item: the partial chunk at the very end)
				for (; c < J.out_cap; c++) {
					J.out[c] = 0;
				}
			}
		}
		if (tid == 0) {
			res->status = OZG_ST_OK;
			res->total_out = produced;
			res->crc32 = 0;
			res->flags = flags;
		}
		if (J.prof && tid == 0) {
			for (u32 i = 0; i < 16; i++) {
				J.prof[i] = L.prof[i];
			}
		}
	} else if (tid == 0) {
		res->status = status;
		res->total_out = status == OZG_ST_INTERNAL ? L.dbg[1] : P.opos;
		res->crc32 = status == OZG_ST_INTERNAL ? L.dbg[2] : 0;
		res->flags = status == OZG_ST_INTERNAL ? (L.dbg[0] << 28) | (L.dbg[3] & 0x0FFFFFFF) : 0;
	}
}
