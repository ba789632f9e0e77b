// otz_lzma.h -- LZMA1 streams under ZIP method 14 as APPNOTE 5.8.8 defines them: ONE WORK-ITEM per entry.
//
// The reference's method 14 is a private run-length token format behind an LZMA-alone-shaped
// header (ref src/lib/lzma-dec.inc.c:102-155, :216-232: thirteen bytes are skipped unread, then
// "00 byte count" / "01 n bytes..." tokens; no range coder anywhere).  That format stays with
// otz_copy_kernel.  This file is the real one -- SURVEY.md section 8, row f4 -- which other ZIP
// writers (7-Zip, xz-based tools) put under the same method id:
//     version (2 bytes) | properties size = 5 (LE16) | lc/lp/pb byte | dictionary size (LE32) | LZMA1 stream
// Sniffing: bytes 2-3 are 05 00 here and 00 00 in the header the reference writes (5D 00 00 00 01 ...);
// a payload that looks like this but does not decode is handed to the private decoder after all
// (the reference reads its header bytes without looking at them): route[], as for Zstandard.
// Checked against liblzma (Python's lzma module: FORMAT_RAW, FILTER_LZMA1).
//
// Shape of the work.  An LZMA stream is one adaptive binary range coder: every decoded bit updates
// a probability that the next bit's decoding reads, so a stream has no parallelism at all.  The
// batch has: one stream per entry.  So each work-item runs the textbook decoder for its own entry --
// 64 independent streams per wavefront, their probability models (1846 + 768 << (lc + lp) 16-bit
// counters, 16-28 KB) in global memory (L2-resident), the entry's own output as the dictionary.
// Lanes of a wavefront diverge (one is inside a literal, its neighbour inside a match); the bit
// decoder they all spend their time in is the same code.  This is the slowest codec of the
// library per stream by far and is reported as such.
#pragma once

#include "otz_simt.h"
#include "../../include/otezip_gpu.h"

#define OTZ_LZMA_PROBS_MAX (1847u + (0x300u << 4)) // lc + lp <= 4
#define OTZ_LZMA_PROB_BYTES (OTZ_LZMA_PROBS_MAX * 2u)

struct otz_lzma_rc {
	const u8 *p;
	u32 n, pos;
	u32 range, code;
	u32 over; // bytes requested beyond the input (corrupt / truncated)
};

OTZ_DEV u32 otz_lzma_byte(otz_lzma_rc &rc) {
	if (rc.pos < rc.n) {
		return rc.p[rc.pos++];
	}
	rc.over++;
	return 0;
}

OTZ_DEV void otz_lzma_norm(otz_lzma_rc &rc) {
	if (rc.range < (1u << 24)) {
		rc.range <<= 8;
		rc.code = (rc.code << 8) | otz_lzma_byte (rc);
	}
}

OTZ_DEV u32 otz_lzma_bit(otz_lzma_rc &rc, u16 *prob) {
	otz_lzma_norm (rc);
	const u32 v = *prob;
	const u32 bound = (rc.range >> 11) * v;
	if (rc.code < bound) {
		rc.range = bound;
		*prob = (u16) (v + ((2048u - v) >> 5));
		return 0;
	}
	rc.range -= bound;
	rc.code -= bound;
	*prob = (u16) (v - (v >> 5));
	return 1;
}

OTZ_DEV u32 otz_lzma_tree(otz_lzma_rc &rc, u16 *probs, u32 nbits) {
	u32 m = 1;
	for (u32 i = 0; i < nbits; i++) {
		m = (m << 1) | otz_lzma_bit (rc, &probs[m]);
	}
	return m - (1u << nbits);
}

OTZ_DEV u32 otz_lzma_rtree(otz_lzma_rc &rc, u16 *probs, u32 nbits) {
	u32 m = 1, sym = 0;
	for (u32 i = 0; i < nbits; i++) {
		const u32 b = otz_lzma_bit (rc, &probs[m]);
		m = (m << 1) | b;
		sym |= b << i;
	}
	return sym;
}

OTZ_DEV u32 otz_lzma_direct(otz_lzma_rc &rc, u32 nbits) {
	u32 r = 0;
	for (u32 i = 0; i < nbits; i++) {
		otz_lzma_norm (rc);
		rc.range >>= 1;
		const u32 t = (rc.code - rc.range) >> 31; // 1 when code < range
		rc.code -= rc.range & (t - 1u);
		r = (r << 1) | (1u - t);
	}
	return r;
}

// probability model layout (the LZMA specification's reference decoder)
#define OTZ_LZ_ISMATCH 0u
#define OTZ_LZ_ISREP 192u
#define OTZ_LZ_ISREPG0 204u
#define OTZ_LZ_ISREPG1 216u
#define OTZ_LZ_ISREPG2 228u
#define OTZ_LZ_ISREP0LONG 240u
#define OTZ_LZ_POSSLOT 432u  // 4 x 64
#define OTZ_LZ_POSDEC 688u   // 115
#define OTZ_LZ_ALIGN 803u    // 16 (802: unused)
#define OTZ_LZ_LEN 819u      // 514: choice, choice2, low[16][8], mid[16][8], high[256]
#define OTZ_LZ_REPLEN 1333u  // 514
#define OTZ_LZ_LITERAL 1847u // (one past the specification's 1846: the alignment slot above)

OTZ_DEV u32 otz_lzma_len(otz_lzma_rc &rc, u16 *lp, u32 pos_state) {
	if (otz_lzma_bit (rc, &lp[0]) == 0) {
		return otz_lzma_tree (rc, &lp[2u + (pos_state << 3)], 3);
	}
	if (otz_lzma_bit (rc, &lp[1]) == 0) {
		return 8u + otz_lzma_tree (rc, &lp[130u + (pos_state << 3)], 3);
	}
	return 16u + otz_lzma_tree (rc, &lp[258u], 8);
}

// One stream.  in: the raw LZMA1 stream (behind the APPNOTE header); lc/lp/pb from the properties byte.
// Decodes until `cap` bytes exist or the end marker appears.  Returns OZG_ST_OK (*total set) or OZG_ST_DATA.
OTZ_DEV int otz_lzma_stream(const u8 *in, u32 n_in, u8 *out, u32 cap, u32 lc, u32 lpb, u32 pb, u16 *probs, u32 *total, bool *marked) {
	const u32 nprobs = OTZ_LZ_LITERAL + (0x300u << (lc + lpb));
	for (u32 i = 0; i < nprobs; i++) {
		probs[i] = 1024;
	}
	otz_lzma_rc rc;
	rc.p = in;
	rc.n = n_in;
	rc.pos = 0;
	rc.over = 0;
	rc.range = 0xFFFFFFFFu;
	rc.code = 0;
	if (n_in < 5u || in[0] != 0) {
		return OZG_ST_DATA;
	}
	rc.pos = 1;
	for (u32 i = 0; i < 4u; i++) {
		rc.code = (rc.code << 8) | otz_lzma_byte (rc);
	}
	u32 state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
	u32 pos = 0;
	const u32 pb_mask = (1u << pb) - 1u, lp_mask = (1u << lpb) - 1u;
	bool eos = false;
	for (;;) {
		if (pos >= cap && rc.pos + 4u >= rc.n) {
			break; // uncomp_size bytes exist and the input is used up (the coder may leave the last flush bytes
			       // unread; an end marker alone is longer than that): a stream without an end marker
		}
		// (with input left at pos == cap the only thing that may follow is the end marker)
		const u32 pos_state = pos & pb_mask;
		if (otz_lzma_bit (rc, &probs[OTZ_LZ_ISMATCH + (state << 4) + pos_state]) == 0) {
			// ---- literal
			if (pos >= cap) {
				return OZG_ST_SIZE; // the stream holds more than uncomp_size bytes
			}
			const u32 prev = pos ? out[pos - 1u] : 0u;
			u16 *lit = &probs[OTZ_LZ_LITERAL + 0x300u * (((pos & lp_mask) << lc) + (prev >> (8u - lc)))];
			u32 sym = 1;
			if (state >= 7u) {
				u32 mb = out[pos - rep0 - 1u]; // (rep0 < pos: every match checked its distance)
				do {
					const u32 mbit = (mb >> 7) & 1u;
					mb <<= 1;
					const u32 b = otz_lzma_bit (rc, &lit[((1u + mbit) << 8) + sym]);
					sym = (sym << 1) | b;
					if (mbit != b) {
						break;
					}
				} while (sym < 0x100u);
			}
			while (sym < 0x100u) {
				sym = (sym << 1) | otz_lzma_bit (rc, &lit[sym]);
			}
			out[pos++] = (u8)sym;
			state = state < 4u ? 0u : state < 10u ? state - 3u : state - 6u;
			if (rc.over > 4u) {
				return OZG_ST_DATA;
			}
			continue;
		}
		u32 len;
		if (otz_lzma_bit (rc, &probs[OTZ_LZ_ISREP + state]) != 0) {
			// ---- repeated match
			if (pos == 0) {
				return OZG_ST_DATA;
			}
			if (otz_lzma_bit (rc, &probs[OTZ_LZ_ISREPG0 + state]) == 0) {
				if (otz_lzma_bit (rc, &probs[OTZ_LZ_ISREP0LONG + (state << 4) + pos_state]) == 0) {
					// short rep: one byte
					state = state < 7u ? 9u : 11u;
					if (rep0 >= pos) {
						return OZG_ST_DATA;
					}
					if (pos >= cap) {
						return OZG_ST_SIZE;
					}
					out[pos] = out[pos - rep0 - 1u];
					pos++;
					continue;
				}
			} else {
				u32 dist;
				if (otz_lzma_bit (rc, &probs[OTZ_LZ_ISREPG1 + state]) == 0) {
					dist = rep1;
				} else {
					if (otz_lzma_bit (rc, &probs[OTZ_LZ_ISREPG2 + state]) == 0) {
						dist = rep2;
					} else {
						dist = rep3;
						rep3 = rep2;
					}
					rep2 = rep1;
				}
				rep1 = rep0;
				rep0 = dist;
			}
			len = otz_lzma_len (rc, &probs[OTZ_LZ_REPLEN], pos_state);
			state = state < 7u ? 8u : 11u;
		} else {
			// ---- match with a new distance
			rep3 = rep2;
			rep2 = rep1;
			rep1 = rep0;
			len = otz_lzma_len (rc, &probs[OTZ_LZ_LEN], pos_state);
			state = state < 7u ? 7u : 10u;
			const u32 len_state = len < 4u ? len : 3u;
			const u32 slot = otz_lzma_tree (rc, &probs[OTZ_LZ_POSSLOT + (len_state << 6)], 6);
			if (slot < 4u) {
				rep0 = slot;
			} else {
				const u32 nd = (slot >> 1) - 1u;
				u32 dist = (2u | (slot & 1u)) << nd;
				if (slot < 14u) {
					dist += otz_lzma_rtree (rc, &probs[OTZ_LZ_POSDEC + dist - slot], nd);
				} else {
					dist += otz_lzma_direct (rc, nd - 4u) << 4;
					dist += otz_lzma_rtree (rc, &probs[OTZ_LZ_ALIGN], 4);
				}
				rep0 = dist;
			}
			if (rep0 == 0xFFFFFFFFu) {
				eos = true; // end marker
				break;
			}
		}
		len += 2u;
		if (rep0 >= pos || rc.over > 4u) {
			return OZG_ST_DATA;
		}
		if (len > cap - pos) {
			return OZG_ST_SIZE; // the stream holds more than uncomp_size bytes
		}
		for (u32 k = 0; k < len; k++) { // (byte by byte: the copy may overlap itself)
			out[pos + k] = out[pos + k - rep0 - 1u];
		}
		pos += len;
	}
	*marked = eos;
	if (rc.over > 4u) { // (the coder reads up to four bytes ahead of the last bit it needs)
		return OZG_ST_DATA;
	}
	*total = pos;
	return OZG_ST_OK;
}

// does a method-14 payload go to this decoder first?  (see the head of this file)
OTZ_DEV bool otz_lzma_is_candidate(const u8 *in, u32 n_in) {
	return n_in >= 9u + 5u && in[2] == 5u && in[3] == 0 && in[4] < 225u;
}

// One method-14 entry (one work-item).  probs: OTZ_LZMA_PROBS_MAX counters of scratch.
OTZ_DEV void otz_lzma_run(const u8 *in, u32 n_in, u8 *out, u32 cap, u32 eflags, u16 *probs, ozg_result *res, u8 *route) {
	u32 d = in[4];
	const u32 lc = d % 9u;
	d /= 9u;
	const u32 lp = d % 5u, pb = d / 5u;
	u32 total = 0;
	bool marked = false;
	int st = OZG_ST_DATA;
	if (lc + lp <= 4u) { // (LZMA2's limit, and what every ZIP writer uses; more would not fit the scratch)
		st = otz_lzma_stream (in + 9u, n_in - 9u, out, cap, lc, lp, pb, probs, &total, &marked);
	}
	// (with OZG_EF_SIZE_IS_CAP only the end marker can say where the stream ends)
	if (st == OZG_ST_OK && (total == cap || ((eflags & OZG_EF_SIZE_IS_CAP) && marked))) {
		res->status = OZG_ST_OK;
		res->total_out = total;
		res->crc32 = 0;
		res->flags = 0;
		*route = 2u; // OTZ_ZS_ROUTE_DONE: the CRC pass follows
	} else if (st == OZG_ST_SIZE && (eflags & OZG_EF_SIZE_IS_CAP)) {
		res->status = OZG_ST_SIZE; // the codec seam: valid so far, out of room (see otz_brotli_run)
		res->total_out = 0;
		res->crc32 = 0;
		res->flags = 0;
		*route = 2u;
	} else {
		*route = 1u; // OTZ_ZS_ROUTE_PRIVATE: the reference's token decoder has the last word
	}
}
