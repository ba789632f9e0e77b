// otz_lz_apply.h -- second half of the DEFLATE decode: one 512-lane workgroup turns the
// index of otz_index.h into the entry's bytes.  (ref: the output side of inflate(),
// src/lib/deflate-dec.inc.c:693-812 -- literal store, do_copy_from_window :521-545 -- and
// the CRC-32 of otezip_extract_entry, src/lib/otezip.c:667-679.)
//
// Per Huffman block: rebuild the decode tables from the stored code lengths (all lanes),
// then take the block's items 512 at a time (fewer when their output would not fit one
// round of the ring, or their bits the stage buffer): lane j decodes item j -- at most
// OTZ_IX_G tokens between two known token boundaries, at a known output position.
// Literals go to their final bytes of the LDS ring, matches are parked at their
// destination and resolved in LDS once their source bytes are final (otz_inflate.h); the
// ring is flushed with coalesced 16-byte stores and CRC'd on the way out.
#pragma once

#include "otz_inflate.h"
#include "otz_index.h"


// ---------------------------------------------------------------- placement
//
// Lane j decodes item j: at most OTZ_IX_G tokens from stage bit s_bit up to e_bit, first
// output byte at absolute position o_abs.  Literals go to their final ring bytes (and are
// marked final); a match is parked in the first three bytes of its own destination (23 bits:
// distance-1 : 15, length-3 : 8 -- a match is at least three bytes long) and its destination
// is noted.  The kernel is bound by VALU issue, so the loop is written to the instruction: lanes
// leave it, and take the literal or the match branch, through the execution mask (scalar
// instructions) -- nothing is computed for a lane and then selected away -- and a step reloads
// its three stage words instead of sliding a window (~55 VALU instructions per token; the
// unrolled select-based form took 129).
// Afterwards the destinations of all matches of the round are written, in stream order, to
// `list` (round-relative, u16): a prefix sum over the lanes' match counts gives every lane its
// slice.  Returns the number of matches of the round.  All OTZ_NT work-items.
static_assert (OTZ_IX_G == 8u, "eight 16-bit destinations are packed into four 32-bit registers");
static_assert (OTZ_NT * OTZ_IX_G * 2u <= OTZ_STAGE_BYTES, "the match list of a round fits the stage buffer");
static_assert (OTZ_RMAX <= 65536u, "round-relative positions fit 16 bits");
OTZ_DEV u32 otz_place_items(otz_inflate_lds &L, bool active, u32 s_bit, u32 e_bit, u32 o_abs, u32 round_base, u16 *list) {
	u32 nm = 0;
	u32 k0 = 0, k1 = 0, k2 = 0, k3 = 0; // the matches' destinations, newest in the low half of k0
	u32 err = 0;                        // 1: the index and the stream disagree, 2: distance beyond the output start
	u32 pos = s_bit;
	if (active) {
		u32 rel = o_abs - round_base; // round-relative output position
		u32 ri = OTZ_RIDX (o_abs);
		for (u32 i = 0; i < OTZ_IX_G; i++) {
			if (pos >= e_bit) {
				break;
			}
			const u32 *wp = &L.stage_w[otz_opaque (pos >> 5)];
			const u32 w0 = wp[0], w1 = wp[1], w2 = wp[2];
			const u32 lo = otz_alignbit (w1, w0, pos), hi = otz_alignbit (w2, w1, pos);
			const u32 e = otz_lut2<OTZ_LIT_ROOT> (L.lit_lut, lo);
			const u32 len = e & 15u, kind = (e >> 4) & 3u;
			if (kind == OTZ_K_LIT) {
				L.ring[ri] = (u8) (e >> 16);
				otz_lds_or (&L.fin[rel >> 5], 1u << (rel & 31u));
				rel += 1u;
				ri = ri + 1u == OTZ_RING ? 0u : ri + 1u;
				pos += len;
			} else if (kind == OTZ_K_LEN) {
				const u32 xb = (e >> 8) & 15u;
				const u32 val = (e >> 16) + otz_bfe (lo, len, xb);
				const u32 used = len + xb; // <= 20
				const u32 db = otz_alignbit (hi, lo, used);
				const u32 d = otz_lut2<OTZ_DIST_ROOT> (L.dist_lut, db);
				const u32 dl = d & 15u, dxb = (d >> 8) & 15u; // dl + dxb <= 28
				if (((d >> 4) & 3u) != OTZ_K_LEN) {
					err = 1; // (the index kernel validated the chain: an internal error)
					break;
				}
				const u32 dist = (d >> 16) + otz_bfe (db, dl, dxb);
				if (dist > round_base + rel) {
					err = 2; // ref reads window bytes nobody wrote (:524): no parity, fail cleanly
					break;
				}
				const u32 cd = (dist - 1u) | ((val - 3u) << 15); // the parked match
				if (ri + 2u < OTZ_RING) {
					L.ring[ri] = (u8)cd;
					L.ring[ri + 1u] = (u8) (cd >> 8);
					L.ring[ri + 2u] = (u8) (cd >> 16);
				} else {
					L.ring[ri] = (u8)cd;
					L.ring[OTZ_RWRAP (ri + 1u)] = (u8) (cd >> 8);
					L.ring[OTZ_RWRAP (ri + 2u)] = (u8) (cd >> 16);
				}
				k3 = otz_alignbit (k3, k2, 16);
				k2 = otz_alignbit (k2, k1, 16);
				k1 = otz_alignbit (k1, k0, 16);
				k0 = (k0 << 16) | rel;
				nm += 1u;
				rel += val;
				ri = OTZ_RWRAP (ri + val);
				pos += used + dl + dxb;
			} else {
				err = 1;
				break;
			}
		}
		if (err == 0 && pos != e_bit) {
			err = 1;
		}
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
	u32 total;
	const u32 incl = otz_block_scan_incl (nm, L.scan, &total); // (two barriers: every stage read is done)
	const u32 at = incl - nm;
	// match number q (in stream order) is the (nm - 1 - q)-th newest
#pragma unroll
	for (u32 k = 0; k < OTZ_IX_G; k++) {
		if (k < nm) {
			const u32 w = k < 2u ? k0 : k < 4u ? k1 : k < 6u ? k2 : k3;
			list[at + nm - 1u - k] = (u16) (w >> (16u * (k & 1u)));
		}
	}
	return total;
}

// ---------------------------------------------------------------- LZ77 resolution
//
// The copies, in stream order: lane j takes matches j, j + OTZ_NT, ... of the list, one after
// the other, and fetches the next as soon as one is done (no lane waits for the slowest of a
// batch).  OTZ_NT consecutive matches span a few KB of output, so most sources lie in
// stretches that are already final; the others wait for exactly the bytes they need: each
// step copies up to 16 bytes once the per-byte final bits of THOSE source bytes are set (two
// bitmap words and the 16 source bytes in one LDS round trip), then sets the bits of what it
// wrote.  The earliest unresolved match of the round is some lane's current one and never
// waits, so the loop cannot stall; it is closed by ballots (a convergent exit keeps the waiting lanes and the
// lanes they wait for in the same loop -- see otz_resolve in otz_inflate.h).
// An overlapping match (distance < 16) copies one period, then two, four, ...
OTZ_DEV void otz_resolve_list(otz_inflate_lds &L, const otz_job &J, u32 round_base, const u16 *list, u32 n_matches) {
	constexpr u32 FIN_WORDS = OTZ_RMAX / 32u;
	const u32 rb = OTZ_RIDX (round_base);
	u32 m = otz_tid (); // this lane's next match
	u32 pend = 0, pdist = 1, o = 0, ri = 0, si = 0;
	u32 spins = 0;
	for (;;) {
		OTZ_COUNT (L, J, OTZ_P_B_ITERS, 1);
		if (pend == 0 && m < n_matches) {
			// ---- next match of this lane: its parked descriptor
			const u32 at = list[m];
			const u32 ai = OTZ_RWRAP (rb + at);
			u32 cd;
			if (ai + 3u > OTZ_RING) {
				cd = (u32)L.ring[ai] | ((u32)L.ring[OTZ_RWRAP (ai + 1u)] << 8) | ((u32)L.ring[OTZ_RWRAP (ai + 2u)] << 16);
			} else {
				const u64 dw = otz_ld64u (&L.ring[ai & ~3u]); // (may read past the ring's end: still inside the LDS block)
				cd = otz_alignbyte ((u32) (dw >> 32), (u32)dw, ai) & 0xFFFFFFu;
			}
			pdist = (cd & 0x7FFFu) + 1u;
			pend = (cd >> 15) + 3u;
			o = at;
			ri = ai;
			si = ai >= pdist ? ai - pdist : ai + OTZ_RING - pdist;
			m += OTZ_NT;
		}
		// ---- one step: up to 16 bytes that neither overlap their own source nor cross the ring's end
		u32 n = pend < 16u ? pend : 16u;
		n = n < pdist ? n : pdist;
		n = n < OTZ_RING - ri ? n : OTZ_RING - ri;
		n = n < OTZ_RING - si ? n : OTZ_RING - si;
		// their source, round-relative [a, b); what lies before the round is final
		const int so = (int)o - (int)pdist;
		const u32 a = so < 0 ? 0u : (u32)so;
		const int be = so + (int)n;
		const u32 b = be < 0 ? 0u : (u32)be;
		const u32 w0 = a >> 5 < FIN_WORDS - 1u ? a >> 5 : FIN_WORDS - 2u; // (b - a <= 16 bits: words w0, w0 + 1 hold them)
		const u64 have = otz_ld64u (&L.fin[w0]);
		otz_order_loads (); // "flag, then data": the source bytes are read after the bits that declare them final
		const u64 need = b > a ? (((u64)1 << (b - a)) - 1u) << (a - 32u * w0) : 0u;
		const bool go = pend != 0 && (have & need) == need;
		if (go) {
			// the source bytes: dword-aligned reads, then a byte funnel shift (reads past the bytes
			// used, and past the ring's end, land in the LDS block and are ignored)
			const u32 *sp = reinterpret_cast<const u32 *> (&L.ring[si & ~3u]);
			const u64 s01 = otz_ld64u (sp);
			const u32 s2 = sp[2];
			const u64 lo = otz_alignbyte ((u32) (s01 >> 32), (u32)s01, si) | ((u64)otz_alignbyte (s2, (u32) (s01 >> 32), si) << 32);
			u64 hi = 0;
			if (n > 8u) {
				const u64 s34 = otz_ld64u (sp + 3);
				hi = otz_alignbyte ((u32)s34, s2, si) | ((u64)otz_alignbyte ((u32) (s34 >> 32), (u32)s34, si) << 32);
			}
			// an unaligned 2-byte store is cheap, an unaligned 4-byte store costs three of them
			u8 *dst = &L.ring[ri];
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
			o += n;
			ri = OTZ_RWRAP (ri + n);
			pend -= n;
			// a whole period written: same source, twice the reach; otherwise the source slides along
			const bool dbl = n == pdist && pdist < 16u;
			si = dbl ? si : OTZ_RWRAP (si + n);
			pdist += dbl ? n : 0u;
		}
		if (J.prof) { // diagnostics: lanes that copied / lanes that had a match in hand, this iteration (wavefront 0)
			const u32 ng = otz_popc64 (otz_ballot (go)), np = otz_popc64 (otz_ballot (go || pend != 0));
			OTZ_COUNT (L, J, 11, ng);
			OTZ_COUNT (L, J, 0, np);
		}
		if (otz_ballot (pend != 0 || m < n_matches) == 0) {
			break;
		}
		if (otz_ballot (go) == 0) { // every live lane of this wavefront waits on another wavefront
			otz_backoff ();
			if ((++spins & 63u) == 0) {
				if (otz_get_status (L) != 0) {
					break;
				}
				if (spins > OTZ_SPIN_LIMIT) {
					L.dbg[0] = 3; // waited too long for source bytes
					L.dbg[1] = otz_tid ();
					L.dbg[2] = round_base + o;
					L.dbg[3] = pdist | (pend << 16);
					otz_set_status (L, OZG_ST_INTERNAL);
					break;
				}
			}
		}
	}
}

// All OTZ_NT work-items.  Work-item 0 writes *res.
// end_bit (optional): where the stream ended, for the z_stream seam's total_in.
OTZ_DEV void otz_apply_entry(otz_inflate_lds &L, const otz_job &J, u32 expect_crc, const otz_ix_head *head, const u8 *region, ozg_result *res, u64 *end_bit) {
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
	u32 opos = 0, flushed = 0;
	u32 crc_reg = 0xFFFFFFFFu;
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
		if (blk->out_base != opos) {
			otz_fail_all (L, OZG_ST_INTERNAL);
			break;
		}
		if (kind == OTZ_IX_KIND_STORED) {
			const u32 src_byte = blk->bit_lo;
			u32 done = 0;
			while (done < out_len) {
				u32 n = out_len - done;
				if (n > OTZ_RMAX) {
					n = OTZ_RMAX;
				}
				otz_make_room (L, J, opos, n, flushed, crc_reg);
				otz_copy_in (L, J, src_byte + done, opos, n);
				opos += n;
				done += n;
				otz_sync ();
			}
			OTZ_MARK (L, J, OTZ_P_STORED);
			boff = next;
			continue;
		}
		// ---- Huffman block: tables from the stored code lengths
		otz_sync (); // (ll shares its storage with the final bitmap of the previous block's last round)
		for (u32 i = tid; i < OTZ_IX_LENS; i += OTZ_NT) {
			L.ll[i] = region[boff + 32u + i];
		}
		otz_sync ();
		if (!otz_build_tables (L, hlit, hdist)) {
			otz_fail_all (L, OZG_ST_INTERNAL);
			break;
		}
		OTZ_MARK (L, J, OTZ_P_TABLES);
		const otz_ix_item *items = reinterpret_cast<const otz_ix_item *> (region + boff + OTZ_IX_ITEMS_OFF);
		u32 base = 0;
		while (base < n_items) {
			otz_make_room (L, J, opos, OTZ_RMAX, flushed, crc_reg);
			// ---- how many items this round takes
			const u32 ia = base + tid < n_items ? base + tid : n_items;
			const u32 ib = ia < n_items ? ia + 1u : n_items;
			const otz_ix_item a = items[ia], z = items[ib];
			const otz_ix_item first = items[base];
			const u64 abs0 = bit_base + first.bit;
			const u32 byte0 = (u32) (abs0 >> 3);
			const u32 skew = (u32) ((uintptr_t) (J.in + byte0) & 15u);
			const u32 p0 = skew * 8u + (u32) (abs0 & 7u);
			// a walk reads up to five words past the one holding its position
			const u32 bit_room = 8u * OTZ_STAGE_BYTES - 192u - p0;
			const bool fits = base + tid < n_items && z.out - first.out <= OTZ_RMAX && z.bit - first.bit <= bit_room && z.out >= a.out && z.bit >= a.bit;
			if (tid == 0) {
				L.first_rej = OTZ_NT;
			}
			otz_sync ();
			otz_wave_min_tid (&L.first_rej, !fits);
			otz_sync ();
			const u32 n = L.first_rej;
			if (n == 0 || blk->out_base + first.out != opos) {
				otz_fail_all (L, OZG_ST_INTERNAL);
				break;
			}
			(void)otz_stage_load (L, J, byte0);
			for (u32 i = tid; i < OTZ_RMAX / 32; i += OTZ_NT) {
				L.fin[i] = 0;
			}
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_STAGE);
			OTZ_COUNT (L, J, OTZ_P_ROUNDS, 1);
			const u32 last_out = items[base + n].out;
			const u32 produced = last_out - first.out;
			const u32 s_bit = p0 + (a.bit - first.bit), e_bit = p0 + (z.bit - first.bit);
			// the match list takes over the stage buffer once every lane has read its tokens
			u16 *const list = reinterpret_cast<u16 *> (L.stage);
			const u32 n_matches = otz_place_items (L, tid < n, s_bit, e_bit, opos + (a.out - first.out), opos, list);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_EMIT);
			if (otz_get_status (L) != 0) {
				break;
			}
			otz_resolve_list (L, J, opos, list, n_matches);
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_RESOLVE);
			if (otz_get_status (L) != 0) {
				break;
			}
			opos += produced;
			base += n;
		}
		boff = next;
	}

	int status = otz_get_status (L);
	if (J.prof && tid == 0) {
		L.prof[OTZ_P_TOTAL] += otz_cycles ();
	}
	if (status == 0 && opos != head->total_out) {
		status = OZG_ST_INTERNAL;
	}
	if (status == 0 && need_out) {
		// The token at head->clip does not fit: the reference reads it, writes what fits of a match
		// and returns Z_OK with the buffer full (ref :521-545, :805-830).  Same here: the bytes are
		// written, the entry reports OZG_ST_NEED_OUT.  (The tables of the last block are still up.)
		const u64 clip = ((u64)head->clip_hi << 32) | head->clip_lo;
		const u32 byte0 = (u32) (clip >> 3);
		otz_make_room (L, J, opos, 512u, flushed, crc_reg);
		otz_sync ();
		const u32 skew = otz_stage_load (L, J, byte0);
		otz_sync ();
		const u32 p = skew * 8u + (u32) (clip & 7u);
		const u32 w = p >> 5;
		const otz_tok t = otz_decode_bits (L, otz_alignbit (L.stage_w[w + 1u], L.stage_w[w], p), otz_alignbit (L.stage_w[w + 2u], L.stage_w[w + 1u], p));
		const u32 room = J.out_cap - opos;
		u32 n = 0;
		if (t.kind == OTZ_K_LEN) {
			if (t.dist > opos) {
				status = OZG_ST_DATA;
			} else {
				n = t.val < room ? t.val : room;
				if (tid == 0) {
					for (u32 i = 0; i < n; i++) { // (byte by byte: the copy may overlap itself)
						L.ring[OTZ_RIDX (opos + i)] = L.ring[OTZ_RIDX (opos + i - t.dist)];
					}
				}
			}
		}
		otz_sync ();
		if (status == 0) {
			opos += n;
			otz_flush (L, J, flushed, opos, crc_reg);
			if (tid == 0) {
				res->status = OZG_ST_NEED_OUT;
				res->total_out = opos;
				res->crc32 = 0;
				res->flags = 0;
			}
			return;
		}
	}
	if (status == 0) {
		const u32 produced = opos;
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
		const u32 crc = ~crc_reg;
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
		res->status = status;
		res->total_out = status == OZG_ST_INTERNAL ? L.dbg[1] : opos;
		res->crc32 = status == OZG_ST_INTERNAL ? L.dbg[2] : 0;
		res->flags = status == OZG_ST_INTERNAL ? (L.dbg[0] << 28) | (L.dbg[3] & 0x0FFFFFFF) : 0;
	}
}
