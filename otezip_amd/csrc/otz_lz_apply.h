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

#define OTZ_K1_ST_NEED_OUT_HEAD (-100)

// All OTZ_NT work-items.  Work-item 0 writes *res.
OTZ_DEV void otz_apply_entry(otz_inflate_lds &L, const otz_job &J, u32 expect_crc, const otz_ix_head *head, const u8 *region, ozg_result *res) {
	const u32 tid = otz_tid ();
	const int hst = head->status;
	if (hst == OTZ_IX_OVERFLOW) {
		return; // the index did not fit its region: otz_inflate_kernel decodes the entry on its own
	}
	if (hst != 0) {
		if (tid == 0) {
			res->status = hst == OTZ_K1_ST_NEED_OUT_HEAD ? OZG_ST_NEED_OUT : hst;
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

	u32 boff = head->first_block;
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
				L.mstart[i] = 0;
			}
			otz_sync ();
			OTZ_MARK (L, J, OTZ_P_STAGE);
			OTZ_COUNT (L, J, OTZ_P_ROUNDS, 1);
			const u32 last_out = items[base + n].out;
			const u32 produced = last_out - first.out;
			const u32 s_bit = p0 + (a.bit - first.bit), e_bit = p0 + (z.bit - first.bit);
			otz_place (L, tid < n, s_bit, OTZ_X_MAKE (OTZ_X_NORMAL, e_bit), opos + (a.out - first.out), opos, 0);
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
