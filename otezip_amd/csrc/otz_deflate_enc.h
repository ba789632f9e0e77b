// otz_deflate_enc.h -- raw DEFLATE encoder, one wavefront per buffer.  Replaces
// deflateInit2(level 6, -15)/deflate(Z_FINISH)/deflateEnd as otezip_compress_data
// drives them (ref src/lib/otezip.c:817-852, src/lib/deflate-enc.inc.c:199-525).
//
// This is a re-specification, not a mirror: the reference's encoder writes its
// Huffman codes without the bit reversal RFC 1951 section 3.1.1 requires and
// wraps its window position at 32 KiB, so nothing can decode its
// output -- only its SIZE is a meaningful observable (parity rule 5).  What is
// kept is its contract (one call, whole buffer, "level 6" class ratio); the
// stream ends with an EMPTY FINAL STORED BLOCK so that the reference's own
// inflate, which rejects a final end-of-block code that starts at bit 1 of the
// last byte, accepts every stream produced here.
//
// Per buffer, per block of OTZ_ENC_BLOCK input bytes, two passes:
//
//   PARSE  64 input positions per step, one per lane.  Every lane hashes the four bytes
//          at its position, reads its bucket (the two most recent earlier positions with
//          that hash), publishes its own position, and measures both candidates eight
//          bytes at a time (dword-aligned LDS reads + v_alignbyte; see the access-cost
//          table in otz_simt.h).  Matches are at least four bytes long (with a dynamic
//          code three-byte matches cost more than their literals on text).  One step of
//          lazy evaluation: a position whose right neighbour has a longer match becomes a
//          literal.  The token chain through the 64 positions -- the only serial part of
//          LZ77 parsing -- is found by POINTER DOUBLING: every lane knows where the chain
//          goes from its own position (position + advance), six rounds of ds_bpermute
//          give every lane the set of positions the chain visits from it and where it
//          leaves the step; the lane the previous step ended on holds the answer.  Chosen
//          lanes append their token to the block's token list (global scratch, 4 bytes
//          per token) and count their symbols in LDS histograms.
//   CODE   Huffman codes for the block from the histograms: symbols ranked by weight (all
//          lanes), the two-queue merge (one lane, 285 steps), leaf depths by walking up
//          (all lanes), depth limit 15 by the usual "move a leaf down" repair on the
//          length counts, lengths re-dealt in rank order, canonical codes from ballots.
//          The code lengths are run-length coded (symbols 16/17/18) and get their own
//          7-bit-limited code.  The block is written with the fixed code instead when
//          that is smaller (short inputs).
//   EMIT   64 tokens per step: table lookups give every token its bits (at most 48), a wave
//          prefix sum gives its bit offset, the tokens are OR-ed into a 2 KiB LDS bit
//          buffer that is drained with coalesced 16-byte stores.
//
// LDS: 8 KiB input ring (matches reach 7 168 bytes back), 2048 two-way buckets, the bit
// buffer, histograms, code tables and the scratch of the code construction: 26 KB per
// wavefront, six buffers in flight per CU.
#pragma once

#include "otz_simt.h"
#include "../../include/otezip_gpu.h"

#define OTZ_ENC_NT 64u
#define OTZ_ENC_WIN 8192u
#define OTZ_ENC_WMASK (OTZ_ENC_WIN - 1u)
#define OTZ_ENC_LOAD 512u                 // bytes per refill of the ring
#define OTZ_ENC_AHEAD (64u + 258u + 8u)   // a step looks this far beyond its first position
#define OTZ_ENC_MAXDIST 7168u             // ring - AHEAD - LOAD - slack
#ifndef OTZ_ENC_HBITS
#define OTZ_ENC_HBITS 11u
#endif
#define OTZ_ENC_HASH (1u << OTZ_ENC_HBITS)
#define OTZ_ENC_OUT 2048u                 // bit buffer: drained by halves, a step adds at most 64 * 48 bits
#define OTZ_ENC_BLOCK 65536u              // input bytes per DEFLATE block
#define OTZ_ENC_MINLEN 4u
#define OTZ_ENC_LAZY_MAX 32u              // a match at least this long is taken without looking at the next position
#define OTZ_ENC_TOK_WORDS OTZ_ENC_BLOCK   // token scratch per wavefront: one u32 per input byte of a block (all literals)
#define OTZ_ENC_DSYM 288u                 // distance symbols live at [288, 320) of hist / code / lens

#define OTZ_TOK_MATCH 0x80000000u         // token: literal = byte; match = this | (len - 3) << 16 | (dist - 1)

struct otz_enc_lds {
	u32 win_w[OTZ_ENC_WIN / 4];
	u32 head[OTZ_ENC_HASH];  // bucket: [15:0] most recent position with this hash (mod 64 Ki), [31:16] the one before
	u32 out_w[OTZ_ENC_OUT / 4]; // bit buffer: tokens are OR-ed in at their bit offsets
	u32 hist[320];           // literal/length symbols at 0..287, distance symbols at 288..319
	u32 code[320];           // same layout: bit-reversed code | length << 16
	u8 lens[320];
	// code construction scratch
	u32 sw[288];             // leaf weights, ascending
	u32 iw[288];             // internal-node weights, in creation order
	u16 ssym[288];           // symbol of the leaf with rank r
	u16 lpar[288];           // parent (internal node index) of leaf r
	u16 ipar[288];           // parent of internal node k
	u32 blc[16];             // leaves per code length
	u32 base[16];            // next canonical code per length
	// the code-length code
	u32 clhist[32];
	u32 clcode[32];
	u8 cllens[32];
	u16 cltok[320];          // run-length tokens: symbol | extra << 5
	u32 n_cltok;
};

struct otz_enc_state {
	u32 bitpos;  // bits pending in L.out_w
	u32 out_pos; // bytes already in global memory (a multiple of OTZ_ENC_OUT / 2)
	u32 cap;
	u8 *dst;
	bool overflow;
};

// OR nbits (<= 32) of v into the bit buffer at bit offset `off`
OTZ_DEV void otz_enc_or_bits(otz_enc_lds &L, u32 off, u32 v, u32 nbits) {
	u32 w = off >> 5, sh = off & 31u;
	otz_lds_or (&L.out_w[w], v << sh);
	if (sh + nbits > 32u) {
		otz_lds_or (&L.out_w[w + 1u], v >> (32u - sh));
	}
}

// Move the first half of the bit buffer (OTZ_ENC_OUT / 2 bytes) to global memory once it is full.
OTZ_DEV void otz_enc_drain(otz_enc_lds &L, otz_enc_state &S) {
	const u32 HALF_W = OTZ_ENC_OUT / 8, HALF_B = OTZ_ENC_OUT / 2;
	while (S.bitpos >= HALF_B * 8u) {
		otz_wave_sync ();
		if (S.out_pos + HALF_B > S.cap) {
			S.overflow = true;
		} else {
			for (u32 c = otz_tid (); c < HALF_B / 16u; c += OTZ_ENC_NT) {
				otz_st16_aligned (S.dst + S.out_pos + 16u * c, *reinterpret_cast<const otz_u4 *> (&L.out_w[4u * c]));
			}
		}
		otz_wave_sync ();
		for (u32 i = otz_tid (); i < HALF_W; i += OTZ_ENC_NT) {
			L.out_w[i] = L.out_w[i + HALF_W];
			L.out_w[i + HALF_W] = 0;
		}
		otz_wave_sync ();
		S.out_pos += HALF_B;
		S.bitpos -= HALF_B * 8u;
	}
}

// Every lane contributes nb (0..48) bits: a wave prefix sum places them behind S.bitpos, in lane order.
OTZ_DEV void otz_enc_put(otz_enc_lds &L, otz_enc_state &S, u64 bits, u32 nb) {
	const u32 lane = otz_tid ();
	u32 incl = nb;
#pragma unroll
	for (u32 d = 1; d < 64; d <<= 1) {
		u32 o = otz_shfl_up (incl, d);
		if (lane >= d) {
			incl += o;
		}
	}
	const u32 total = otz_shfl (incl, 63);
	if (nb) {
		const u32 off = S.bitpos + incl - nb, w = off >> 5, sh = off & 31u;
		const u64 lo = bits << sh;
		otz_lds_or (&L.out_w[w], (u32)lo);
		if (sh + nb > 32u) {
			otz_lds_or (&L.out_w[w + 1u], (u32) (lo >> 32));
		}
		if (sh + nb > 64u) {
			otz_lds_or (&L.out_w[w + 2u], (u32) (bits >> (64u - sh)));
		}
	}
	S.bitpos += total;
	otz_enc_drain (L, S);
}

// fixed literal/length code of RFC 1951 3.2.6, bit-reversed for LSB-first packing:
// returns the bits, *n their count (7..9)
OTZ_DEV u32 otz_enc_litlen_bits(u32 sym, u32 *n) {
	u32 code, k;
	if (sym <= 143u) {
		code = 0x30u + sym;
		k = 8;
	} else if (sym <= 255u) {
		code = 0x190u + (sym - 144u);
		k = 9;
	} else if (sym <= 279u) {
		code = sym - 256u;
		k = 7;
	} else {
		code = 0xC0u + (sym - 280u);
		k = 8;
	}
	*n = k;
	return otz_brev (code) >> (32u - k);
}

// length 3..258 -> literal/length symbol, number of extra bits, their value (RFC 1951 3.2.5)
OTZ_DEV u32 otz_enc_len_sym(u32 len, u32 *xb, u32 *xv) {
	u32 l = len - 3u;
	if (len == 258u) {
		*xb = 0;
		*xv = 0;
		return 285u;
	}
	if (l < 8u) {
		*xb = 0;
		*xv = 0;
		return 257u + l;
	}
	u32 hb = 31u - (u32)__builtin_clz (l), b = hb - 2u;
	*xb = b;
	*xv = l & ((1u << b) - 1u);
	return 257u + 4u * hb - 4u + ((l >> b) & 3u);
}

// distance 1..32768 -> distance symbol, extra bits, value
OTZ_DEV u32 otz_enc_dist_sym(u32 dist, u32 *xb, u32 *xv) {
	u32 d = dist - 1u;
	if (d < 4u) {
		*xb = 0;
		*xv = 0;
		return d;
	}
	u32 hb = 31u - (u32)__builtin_clz (d), b = hb - 1u;
	*xb = b;
	*xv = d & ((1u << b) - 1u);
	return 2u * hb + ((d >> b) & 1u);
}

// eight bytes of the ring starting at byte position q: three dword-aligned reads
OTZ_DEV u64 otz_enc_ld8(const otz_enc_lds &L, u32 q) {
	const u32 i = q >> 2, s = q & 3u, M = OTZ_ENC_WIN / 4u - 1u;
	const u32 w0 = L.win_w[i & M], w1 = L.win_w[(i + 1u) & M], w2 = L.win_w[(i + 2u) & M];
	return (u64)otz_alignbyte (w1, w0, s) | (u64)otz_alignbyte (w2, w1, s) << 32;
}

// length of the match between position pos and pos - dist (0 if shorter than the minimum), at most maxlen
OTZ_DEV u32 otz_enc_measure(const otz_enc_lds &L, u32 pos, u32 dist, u64 own, u32 maxlen) {
	u64 x = own ^ otz_enc_ld8 (L, pos - dist);
	if ((u32)x != 0u) {
		return 0; // fewer than four bytes
	}
	u32 k = 8;
	if (x != 0ull) {
		k = otz_ctz64 (x) >> 3;
	} else {
		while (k < maxlen) {
			x = otz_enc_ld8 (L, pos + k) ^ otz_enc_ld8 (L, pos - dist + k);
			if (x != 0ull) {
				k += otz_ctz64 (x) >> 3;
				break;
			}
			k += 8u;
		}
	}
	k = k < maxlen ? k : maxlen;
	return k >= OTZ_ENC_MINLEN ? k : 0u;
}

OTZ_DEV u32 otz_enc_wave_sum(u32 v) {
#pragma unroll
	for (u32 d = 1; d < 64; d <<= 1) {
		v += otz_shfl_xor (v, d);
	}
	return v;
}

OTZ_CONST_TABLE u8 otz_cl_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

// ---------------------------------------------------------------- Huffman code construction
// Code lengths (at most `limit`) and canonical codes for freq[0..nsym): all 64 lanes call.
// NCH = ceil(nsym / 64).  At least two symbols get a code (a zero-weight symbol among the
// first three is drafted when fewer are in use), as every inflate expects.
template <u32 NCH>
OTZ_DEV void otz_enc_build_code(otz_enc_lds &L, u32 *freq, u32 nsym, u32 limit, u8 *lens, u32 *code) {
	const u32 lane = otz_tid ();
	u32 m = 0;
#pragma unroll
	for (u32 k = 0; k < NCH; k++) {
		const u32 i = lane + 64u * k;
		m += otz_popc64 (otz_ballot (i < nsym && freq[i] != 0u));
	}
	if (m < 2u) {
		otz_wave_sync ();
		if (lane == 0) {
			u32 mm = m;
			for (u32 s = 0; s < 3u && mm < 2u; s++) {
				if (freq[s] == 0u) {
					freq[s] = 1;
					mm++;
				}
			}
		}
		m = 2;
		otz_wave_sync ();
	}
	// rank by (weight, symbol)
	u32 key[NCH], rk[NCH];
#pragma unroll
	for (u32 k = 0; k < NCH; k++) {
		const u32 i = lane + 64u * k;
		const u32 f = i < nsym ? freq[i] : 0u;
		key[k] = f ? (f << 9 | i) : 0u;
		rk[k] = 0;
		if (i < nsym) {
			lens[i] = 0;
		}
	}
	for (u32 j = 0; j < nsym; j++) {
		const u32 fj = freq[j];
		const u32 kj = fj ? (fj << 9 | j) : 0xFFFFFFFFu;
#pragma unroll
		for (u32 k = 0; k < NCH; k++) {
			rk[k] += kj < key[k] ? 1u : 0u;
		}
	}
#pragma unroll
	for (u32 k = 0; k < NCH; k++) {
		if (key[k]) {
			L.sw[rk[k]] = key[k] >> 9;
			L.ssym[rk[k]] = (u16) (key[k] & 511u);
		}
	}
	if (lane < 16u) {
		L.blc[lane] = 0;
	}
	otz_wave_sync ();
	// two-queue merge: leaves in ascending order, internal nodes in creation order (also ascending)
	if (lane == 0) {
		u32 li = 0, ih = 0;
		u32 lw = L.sw[0], inw = 0xFFFFFFFFu;
		for (u32 k = 0; k + 1u < m; k++) {
			u32 sum = 0;
			for (u32 t = 0; t < 2u; t++) {
				if (li < m && lw <= inw) {
					sum += lw;
					L.lpar[li] = (u16)k;
					li++;
					lw = li < m ? L.sw[li] : 0xFFFFFFFFu;
				} else {
					sum += inw;
					L.ipar[ih] = (u16)k;
					ih++;
					inw = ih < k ? L.iw[ih] : 0xFFFFFFFFu;
				}
			}
			L.iw[k] = sum;
			if (ih == k) {
				inw = sum; // the queue of internal nodes was empty: the new node is its head
			}
		}
	}
	otz_wave_sync ();
	// leaf depths, clamped; K = Kraft sum in units of 2^-limit
	u32 K = 0;
#pragma unroll
	for (u32 k = 0; k < NCH; k++) {
		const u32 r = lane + 64u * k;
		if (r < m) {
			u32 d = 1, q = L.lpar[r];
			while (q != m - 2u) {
				q = L.ipar[q];
				d++;
			}
			d = d < limit ? d : limit;
			otz_lds_add (&L.blc[d], 1u);
			K += 1u << (limit - d);
		}
	}
	K = otz_enc_wave_sum (K);
	otz_wave_sync ();
	if (K > (1u << limit) && lane == 0) {
		// too deep a tree was cut off at `limit`: every round turns a leaf above the limit into
		// a node with two leaves below it, one of them taken from the bottom row: Kraft sum - 1
		for (u32 over = K - (1u << limit); over > 0; over--) {
			u32 b = limit - 1u;
			while (L.blc[b] == 0u) {
				b--;
			}
			L.blc[b]--;
			L.blc[b + 1u] += 2u;
			L.blc[limit]--;
		}
	}
	otz_wave_sync ();
	// deal the lengths out again in rank order (lightest leaf = longest code), first canonical codes
	{
		u32 c = 0, mine = 0;
		for (u32 b = 1; b <= limit; b++) {
			c = (c + L.blc[b - 1u]) << 1;
			if (lane == b) {
				mine = c;
			}
		}
		if (lane >= 1u && lane < 16u) {
			L.base[lane] = mine;
		}
	}
#pragma unroll
	for (u32 k = 0; k < NCH; k++) {
		const u32 r = lane + 64u * k;
		if (r < m) {
			u32 cum = 0, len = 0;
			for (u32 b = limit; b >= 1u; b--) {
				const u32 c = L.blc[b];
				if (r >= cum && r < cum + c) {
					len = b;
				}
				cum += c;
			}
			lens[L.ssym[r]] = (u8)len;
		}
	}
	otz_wave_sync ();
	// canonical codes: symbols of one length in symbol order
#pragma unroll
	for (u32 k = 0; k < NCH; k++) {
		const u32 i = lane + 64u * k;
		const u32 len = i < nsym ? lens[i] : 0u;
		u64 same = 0;
		u32 add = 0;
		for (u32 b = 1; b <= limit; b++) {
			const u64 bal = otz_ballot (len == b);
			if (len == b) {
				same = bal;
			}
			if (lane == b) {
				add = otz_popc64 (bal);
			}
		}
		if (i < nsym) {
			u32 e = 0;
			if (len) {
				const u32 c = L.base[len] + otz_popc64 (same & ((1ull << lane) - 1ull));
				e = (otz_brev (c) >> (32u - len)) | len << 16;
			}
			code[i] = e;
		}
		otz_wave_sync ();
		if (lane >= 1u && lane < 16u) {
			L.base[lane] += add;
		}
		otz_wave_sync ();
	}
}

// Run-length tokens (code-length alphabet, RFC 1951 3.2.7) for lens[0..n), appended to L.cltok; one lane.
OTZ_DEV u32 otz_enc_rle_lengths(otz_enc_lds &L, const u8 *lens, u32 n, u32 nt) {
	u32 i = 0;
	while (i < n) {
		const u32 v = lens[i];
		u32 r = 1;
		while (i + r < n && lens[i + r] == v) {
			r++;
		}
		i += r;
		if (v == 0u) {
			while (r >= 11u) {
				const u32 t = r < 138u ? r : 138u;
				L.cltok[nt++] = (u16) (18u | (t - 11u) << 5);
				r -= t;
			}
			if (r >= 3u) {
				L.cltok[nt++] = (u16) (17u | (r - 3u) << 5);
				r = 0;
			}
		} else {
			L.cltok[nt++] = (u16)v;
			r--;
			while (r >= 3u) {
				const u32 t = r < 6u ? r : 6u;
				L.cltok[nt++] = (u16) (16u | (t - 3u) << 5);
				r -= t;
			}
		}
		while (r > 0u) {
			L.cltok[nt++] = (u16)v;
			r--;
		}
	}
	return nt;
}

// Encode src[0..n) into dst (capacity cap); tok = OTZ_ENC_TOK_WORDS words of scratch.  All 64
// work-items call together and all get the compressed size, or ~0u when the output does not
// fit (the caller then stores the entry, ref otezip.c:846-850).
OTZ_DEV u32 otz_deflate_entry(otz_enc_lds &L, const u8 *src, u32 n, u8 *dst, u32 cap, u32 *tok) {
	const u32 lane = otz_tid ();
	const u64 lt_mask = (1ull << lane) - 1ull;
	otz_enc_state S;
	S.bitpos = 0;
	S.out_pos = 0;
	S.cap = cap;
	S.dst = dst;
	S.overflow = false;
	for (u32 i = lane; i < OTZ_ENC_HASH; i += OTZ_ENC_NT) {
		L.head[i] = 0;
	}
	for (u32 i = lane; i < OTZ_ENC_OUT / 4; i += OTZ_ENC_NT) {
		L.out_w[i] = 0;
	}
	u32 loaded = 0, skip = 0;
	otz_wave_sync ();
	for (u32 bstart = 0; bstart < n && !S.overflow; bstart += OTZ_ENC_BLOCK) {
		const u32 bend = n - bstart < OTZ_ENC_BLOCK ? n : bstart + OTZ_ENC_BLOCK;
		for (u32 i = lane; i < 320u; i += OTZ_ENC_NT) {
			L.hist[i] = 0;
		}
		u32 ntok = 0;
		otz_wave_sync ();
		// ------------------------------------------------------------------ PARSE
		for (u32 p = bstart; p < bend; p += 64u) {
			// keep [p - MAXDIST, p + AHEAD) resident in the ring
			while (loaded < n && p + OTZ_ENC_AHEAD > loaded) {
				const u32 q = loaded + 16u * lane;
				if (lane < OTZ_ENC_LOAD / 16u) {
					if (q + 16u <= n) {
						*reinterpret_cast<otz_u4 *> (&L.win_w[(q & OTZ_ENC_WMASK) >> 2]) = otz_ld16 (src + q);
					} else {
						u8 *wb = reinterpret_cast<u8 *> (L.win_w);
						for (u32 t = q; t < n; t++) {
							wb[t & OTZ_ENC_WMASK] = src[t];
						}
					}
				}
				loaded += OTZ_ENC_LOAD;
			}
			otz_wave_sync ();
			const u32 pos = p + lane;
			const bool can = pos + OTZ_ENC_MINLEN <= n;
			const u64 own = otz_enc_ld8 (L, pos);
			const u32 h = ((u32)own * 0x9E3779B1u) >> (32u - OTZ_ENC_HBITS);
			const u32 c = can ? L.head[h] : 0u;
			otz_wave_sync (); // every lane has read the buckets of the earlier steps
			if (can) {
				L.head[h] = c << 16 | (pos & 0xFFFFu);
			}
			u32 len = 0, dist = 0;
			if (can && lane >= skip) {
				const u32 maxlen = n - pos < 258u ? n - pos : 258u;
				const u32 d0 = (pos - c) & 0xFFFFu, d1 = (pos - (c >> 16)) & 0xFFFFu;
				if (d0 >= 1u && d0 <= OTZ_ENC_MAXDIST && d0 <= pos) {
					len = otz_enc_measure (L, pos, d0, own, maxlen);
					dist = d0;
				}
				if (d1 >= 1u && d1 <= OTZ_ENC_MAXDIST && d1 <= pos && d1 != d0 && len < maxlen) {
					const u32 l1 = otz_enc_measure (L, pos, d1, own, maxlen);
					if (l1 > len) {
						len = l1;
						dist = d1;
					}
				}
			}
			if (skip >= 64u) {
				skip -= 64u; // the whole step lies inside the previous match
				continue;
			}
			// lazy evaluation, then where the chain goes from every position
			const u32 lnext = otz_shfl (len, (lane + 1u) & 63u);
			const bool take = len != 0u && (len >= OTZ_ENC_LAZY_MAX || lane == 63u || lnext <= len);
			u32 J = lane + (take ? len : 1u);
			u64 R = 1ull << lane;
#pragma unroll
			for (u32 k = 0; k < 6u; k++) {
				const u32 from = J & 63u;
				const u32 Jn = otz_shfl (J, from);
				const u32 Rlo = otz_shfl ((u32)R, from), Rhi = otz_shfl ((u32) (R >> 32), from);
				if (J < 64u) {
					R |= (u64)Rhi << 32 | Rlo;
					J = Jn;
				}
			}
			const u32 lim = n - p < 64u ? n - p : 64u;
			u64 sel = (u64)otz_readlane ((u32)R, skip) | (u64)otz_readlane ((u32) (R >> 32), skip) << 32;
			if (lim < 64u) {
				sel &= (1ull << lim) - 1ull;
			}
			skip = otz_readlane (J, skip) - 64u;
			if ((sel >> lane) & 1ull) {
				u32 t;
				if (take) {
					u32 xb, xv;
					t = OTZ_TOK_MATCH | (len - 3u) << 16 | (dist - 1u);
					otz_lds_add (&L.hist[otz_enc_len_sym (len, &xb, &xv)], 1u);
					otz_lds_add (&L.hist[OTZ_ENC_DSYM + otz_enc_dist_sym (dist, &xb, &xv)], 1u);
				} else {
					t = (u32)own & 0xFFu;
					otz_lds_add (&L.hist[t], 1u);
				}
				tok[ntok + otz_popc64 (sel & lt_mask)] = t;
			}
			ntok += otz_popc64 (sel);
		}
		otz_wave_sync ();
		// ------------------------------------------------------------------ CODE
		if (lane == 0) {
			L.hist[256] = 1; // the end-of-block code
		}
		otz_wave_sync ();
		otz_enc_build_code<5> (L, L.hist, 286u, 15u, L.lens, L.code);
		otz_enc_build_code<1> (L, L.hist + OTZ_ENC_DSYM, 30u, 15u, L.lens + OTZ_ENC_DSYM, L.code + OTZ_ENC_DSYM);
		otz_wave_sync ();
		// sizes: how many length and distance codes are sent, the run-length tokens, their code
		u32 hlit = 257, hdist = 1;
#pragma unroll
		for (u32 k = 0; k < 5u; k++) {
			const u32 i = lane + 64u * k;
			const u64 nz = otz_ballot (i < 286u && L.lens[i] != 0u);
			if (nz) {
				const u32 top = 64u * k + 64u - otz_clz64 (nz);
				hlit = top > hlit ? top : hlit;
			}
		}
		{
			const u64 nz = otz_ballot (lane < 30u && L.lens[OTZ_ENC_DSYM + lane] != 0u);
			if (nz) {
				hdist = 64u - otz_clz64 (nz);
			}
		}
		if (lane < 32u) {
			L.clhist[lane] = 0;
		}
		otz_wave_sync ();
		if (lane == 0) {
			u32 nt = otz_enc_rle_lengths (L, L.lens, hlit, 0);
			L.n_cltok = otz_enc_rle_lengths (L, L.lens + OTZ_ENC_DSYM, hdist, nt);
		}
		otz_wave_sync ();
		const u32 n_cltok = L.n_cltok;
		for (u32 i = lane; i < n_cltok; i += OTZ_ENC_NT) {
			otz_lds_add (&L.clhist[L.cltok[i] & 31u], 1u);
		}
		otz_wave_sync ();
		otz_enc_build_code<1> (L, L.clhist, 19u, 7u, L.cllens, L.clcode);
		otz_wave_sync ();
		// the code-length code lengths go out in the order of RFC 1951 3.2.7
		const u32 my_cl = lane < 19u ? L.cllens[otz_cl_order[lane]] : 0u;
		const u64 clnz = otz_ballot (my_cl != 0u);
		u32 hclen = clnz ? 64u - otz_clz64 (clnz) : 4u;
		hclen = hclen < 4u ? 4u : hclen;
		// cost of both codes (the extra bits are the same in both)
		u32 dyn = 0, fix = 0;
#pragma unroll
		for (u32 k = 0; k < 5u; k++) {
			const u32 i = lane + 64u * k;
			const u32 f = L.hist[i]; // 0 for the unused slots 286, 287 and beyond the 30 distance symbols
			u32 fl;
			if (i < OTZ_ENC_DSYM) {
				(void)otz_enc_litlen_bits (i, &fl);
			} else {
				fl = 5;
			}
			dyn += f * L.lens[i];
			fix += f * fl;
		}
		for (u32 i = lane; i < n_cltok; i += OTZ_ENC_NT) {
			const u32 t = L.cltok[i] & 31u;
			dyn += L.cllens[t] + (t == 16u ? 2u : t == 17u ? 3u : t == 18u ? 7u : 0u);
		}
		dyn = otz_enc_wave_sum (dyn) + 14u + 3u * hclen;
		fix = otz_enc_wave_sum (fix);
		const bool use_fixed = fix <= dyn;
		// ------------------------------------------------------------------ EMIT
		if (use_fixed) {
			otz_wave_sync ();
			for (u32 i = lane; i < 320u; i += OTZ_ENC_NT) {
				u32 nb, b;
				if (i < OTZ_ENC_DSYM) {
					b = otz_enc_litlen_bits (i, &nb);
				} else {
					b = otz_brev (i - OTZ_ENC_DSYM) >> 27;
					nb = 5;
				}
				L.code[i] = b | nb << 16;
			}
			otz_wave_sync ();
			otz_enc_put (L, S, 2u, lane == 0 ? 3u : 0u); // BFINAL = 0, BTYPE = 01
		} else {
			// BFINAL = 0, BTYPE = 10, HLIT, HDIST, HCLEN, then the code-length code lengths (3 bits each)
			const u32 hdr = 4u | (hlit - 257u) << 3 | (hdist - 1u) << 8 | (hclen - 4u) << 13;
			otz_enc_put (L, S, lane == 0 ? hdr : my_cl, lane == 0 ? 17u : 0u);
			otz_enc_put (L, S, my_cl, lane < hclen ? 3u : 0u);
			for (u32 b = 0; b < n_cltok; b += 64u) {
				const u32 i = b + lane;
				u64 bits = 0;
				u32 nb = 0;
				if (i < n_cltok) {
					const u32 t = L.cltok[i], sym = t & 31u, e = L.clcode[sym];
					nb = e >> 16;
					bits = (e & 0xFFFFu) | (u64) (t >> 5) << nb;
					nb += sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u;
				}
				otz_enc_put (L, S, bits, nb);
			}
		}
		for (u32 b = 0; b < ntok && !S.overflow; b += 64u) {
			const u32 i = b + lane;
			u64 bits = 0;
			u32 nb = 0;
			if (i < ntok) {
				const u32 t = tok[i];
				if (t & OTZ_TOK_MATCH) {
					u32 xb, xv;
					const u32 e = L.code[otz_enc_len_sym (((t >> 16) & 0xFFu) + 3u, &xb, &xv)];
					nb = e >> 16;
					bits = (e & 0xFFFFu) | (u64)xv << nb;
					nb += xb;
					const u32 e2 = L.code[OTZ_ENC_DSYM + otz_enc_dist_sym ((t & 0x7FFFu) + 1u, &xb, &xv)];
					bits |= (u64) (e2 & 0xFFFFu) << nb;
					nb += e2 >> 16;
					bits |= (u64)xv << nb;
					nb += xb;
				} else {
					const u32 e = L.code[t];
					nb = e >> 16;
					bits = e & 0xFFFFu;
				}
			}
			otz_enc_put (L, S, bits, nb);
		}
		{
			const u32 e = L.code[256];
			otz_enc_put (L, S, e & 0xFFFFu, lane == 0 ? e >> 16 : 0u);
		}
		otz_wave_sync ();
	}
	// the final block: stored, empty (BFINAL = 1, BTYPE = 00, pad to a byte, LEN = 0, NLEN = 0xFFFF)
	otz_enc_put (L, S, 1u, lane == 0 ? 3u : 0u);
	S.bitpos = (S.bitpos + 7u) & ~7u;
	otz_enc_put (L, S, 0xFFFF0000u, lane == 0 ? 32u : 0u);
	otz_wave_sync ();
	u32 tail = S.bitpos >> 3; // whole bytes by construction
	if (S.out_pos + tail > S.cap) {
		S.overflow = true;
	}
	if (!S.overflow) {
		const u8 *ob = reinterpret_cast<const u8 *> (L.out_w);
		for (u32 i = lane; i < tail; i += OTZ_ENC_NT) {
			S.dst[S.out_pos + i] = ob[i];
		}
	}
	return S.overflow ? ~0u : S.out_pos + tail;
}

// ---------------------------------------------------------------- method 14
// The reference's private "lzma" token writer (ref src/lib/lzma-enc.inc.c:110-161 inside
// lzmaCompress :220-300): 13-byte header (5D 00 00 00 01 + LE64 size), then per 64 KiB chunk
// runs of one byte, at most 255 long: run >= 4 -> 00 <byte> <n>; shorter -> 01 <n> <byte x n>.
// Byte-exact.  One wavefront: 64 input bytes per step, run starts from a ballot, token offsets
// from a wave prefix sum; a run that crosses steps is carried.  Returns the size, or ~0u where
// the reference fails or can no longer end up smaller than the input (both mean STORE).

OTZ_DEV u32 otz_lzma_emit_run(u8 *out, u32 op, u32 cap, u32 byte, u32 len, bool *overflow) {
	// uniform; lane 0 stores.  A carried run may exceed 255: the reference cuts and restarts.
	while (len > 0) {
		u32 r = len < 255u ? len : 255u;
		u32 need = r >= 4u ? 3u : 2u + r;
		if (cap - op < need) {
			*overflow = true;
			return op;
		}
		if (otz_tid () == 0) {
			if (r >= 4u) {
				out[op] = 0x00;
				out[op + 1] = (u8)byte;
				out[op + 2] = (u8)r;
			} else {
				out[op] = 0x01;
				out[op + 1] = (u8)r;
				for (u32 k = 0; k < r; k++) {
					out[op + 2 + k] = (u8)byte;
				}
			}
		}
		op += need;
		len -= r;
	}
	return op;
}

OTZ_DEV u32 otz_lzma_tok_entry(const u8 *in, u32 n, u8 *out, u32 cap, bool keep_going = false) {
	const u32 lane = otz_tid ();
	if (cap < 13u) {
		return ~0u;
	}
	if (lane < 13u) {
		const u8 props[5] = { 0x5D, 0x00, 0x00, 0x00, 0x01 };
		out[lane] = lane < 5u ? props[lane] : (lane < 9u ? (u8) (n >> (8u * (lane - 5u))) : (u8)0);
	}
	u32 op = 13, ip = 0;
	bool overflow = false;
	while (ip < n) {
		u32 chunk = n - ip < 65536u ? n - ip : 65536u;
		if (cap - op < chunk + 32u) {
			return ~0u; // Z_BUF_ERROR, ref :277-280
		}
		u32 carry_len = 0, carry_byte = 0;
		for (u32 base = 0; base < chunk; base += 64u) {
			u32 idx = base + lane;
			bool valid = idx < chunk;
			u32 b = valid ? in[ip + idx] : 0u;
			u32 prev = otz_shfl_up (b, 1);
			bool same = lane == 0 ? (carry_len > 0 && b == carry_byte) : (b == prev);
			u64 V = otz_ballot (valid);
			u64 B = otz_ballot (valid && !same); // a run starts at this lane
			u32 nvalid = otz_popc64 (V);
			u32 first = B ? (u32)__builtin_ctzll (B) : nvalid;
			carry_len += first;
			if (B) {
				op = otz_lzma_emit_run (out, op, cap, carry_byte, carry_len, &overflow);
				carry_len = 0;
			} else {
				while (carry_len >= 255u && !overflow) {
					op = otz_lzma_emit_run (out, op, cap, carry_byte, 255u, &overflow);
					carry_len -= 255u;
				}
			}
			// runs that start AND end inside this step; the last start becomes the carry
			u64 above = lane < 63u ? (B >> (lane + 1u)) : 0ull;
			bool starts = (B >> lane) & 1ull;
			bool complete = starts && above != 0;
			u32 len = complete ? (u32)__builtin_ctzll (above) + 1u : 0u;
			u32 tsz = complete ? (len >= 4u ? 3u : 2u + len) : 0u;
			u32 incl = tsz;
#pragma unroll
			for (u32 d = 1; d < 64; d <<= 1) {
				u32 o = otz_shfl_up (incl, d);
				if (lane >= d) {
					incl += o;
				}
			}
			u32 total = otz_shfl (incl, 63);
			if (cap - op < total) {
				overflow = true;
			}
			if (overflow) {
				return ~0u; // compressed_size > avail_out, ref :288-290
			}
			if (complete) {
				u32 at = op + incl - tsz;
				if (len >= 4u) {
					out[at] = 0x00;
					out[at + 1] = (u8)b;
					out[at + 2] = (u8)len;
				} else {
					out[at] = 0x01;
					out[at + 1] = (u8)len;
					for (u32 k = 0; k < len; k++) {
						out[at + 2 + k] = (u8)b;
					}
				}
			}
			op += total;
			if (B) {
				u32 last = 63u - (u32)__builtin_clzll (B);
				carry_byte = otz_shfl (b, last);
				carry_len = nvalid - last;
			}
			if (op >= n && !keep_going) {
				return ~0u; // can no longer be smaller than the input: STORE (ref otezip.c:1006-1011)
			}
		}
		op = otz_lzma_emit_run (out, op, cap, carry_byte, carry_len, &overflow); // runs never cross chunks
		if (overflow) {
			return ~0u;
		}
		ip += chunk;
	}
	return op;
}

// One buffer of an encode batch: otezip_compress_data (ref src/lib/otezip.c:788-1076) for it.
//  * method 8: DEFLATE as above; "not smaller than the input" or "does not fit" -> STORE (:846-850)
//  * methods 93 / 100 / 97: the reference's writers only add bytes to the input,
//    so its own fallback rule always stores such entries; STORE is written directly
//  * method 14: the private RLE tokens above, byte-exact, same fallback rule (:1006-1011)
//  * zero-length input -> STORE with size 0 (:793-801)
// The CRC-32 of the input (ref :1124) is taken by otz_enc_crc_kernel.
// ------------------------------------------------- methods 93 / 100 / 97, codec-level calls
// The reference's container WRITERS (zstdCompress ref src/lib/zstd.inc.c:254-414, lzfse_compress
// src/lib/lzfse.inc.c:179-219, brotliCompress src/lib/brotli.inc.c:41-55, 99-128): a few header
// bytes around the unchanged payload.  Inside zip_file_add their output is never kept (it is
// always larger than the input, so the driver stores the entry: SURVEY F9); they are only
// reachable through the z_stream seam (OZG_ENC_F_CODEC_ONLY).  Byte-exact.  One wavefront;
// returns the size, or ~0u where the reference reports Z_BUF_ERROR.  The BROT header's CRC
// field (bytes 13..16) is filled in by otz_enc_crc_kernel, which computes the CRC anyway.
OTZ_DEV void otz_wave_copy(const u8 *src, u8 *dst, u32 n) {
	const u32 lane = otz_tid ();
	for (u32 c = lane; c < (n >> 4); c += OTZ_ENC_NT) {
		otz_st16 (dst + 16ull * c, otz_ld16 (src + 16ull * c));
	}
	for (u32 i = (n & ~15u) + lane; i < n; i += OTZ_ENC_NT) {
		dst[i] = src[i];
	}
}

OTZ_DEV u32 otz_frame_entry(u32 method, const u8 *in, u32 n, u8 *out, u32 cap) {
	const u32 lane = otz_tid ();
	if (method == OZG_METHOD_LZFSE) {
		if ((u64)n + 5u > cap) {
			return ~0u;
		}
		if (lane < 5u) {
			out[lane] = lane == 0 ? (u8)0x06 : (u8) (n >> (8u * (lane - 1u)));
		}
		otz_wave_copy (in, out + 5, n);
		return n + 5u;
	}
	if (method == OZG_METHOD_BROTLI) {
		if (cap < 17u || n > cap - 17u) {
			return ~0u;
		}
		if (lane < 17u) {
			const u8 magic[5] = { 'B', 'R', 'O', 'T', 1 };
			out[lane] = lane < 5u ? magic[lane] : lane < 9u ? (u8) (n >> (8u * (lane - 5u))) : (u8)0; // 64-bit length; CRC later
		}
		otz_wave_copy (in, out + 17, n);
		return n + 17u;
	}
	// method 93: magic as the reference's ENCODER writes it (its own decoder wants the other
	// byte order), one descriptor byte, raw blocks of at most 65 535 bytes, end marker
	const u32 nblk = (n + 65534u) / 65535u;
	const u64 total = 5ull + n + 3ull * nblk + 3ull;
	if (total > cap) {
		return ~0u;
	}
	if (lane < 5u) {
		const u8 hdr[5] = { 0xFD, 0x2F, 0xB5, 0x28, 0x70 };
		out[lane] = hdr[lane];
	}
	for (u32 k = 0; k < nblk; k++) {
		const u32 ip = 65535u * k, bs = n - ip < 65535u ? n - ip : 65535u;
		u8 *b = out + 5u + (u64)k * 65538u;
		if (lane < 3u) {
			b[lane] = lane == 0 ? (u8)0 : (u8) (bs >> (8u * (lane - 1u)));
		}
		otz_wave_copy (in + ip, b + 3, bs);
	}
	if (lane < 3u) {
		out[total - 3u + lane] = lane == 0 ? (u8)1 : (u8)0;
	}
	return (u32)total;
}

OTZ_DEV void otz_encode_entry(otz_enc_lds &L, const u8 *src, u8 *dst, const ozg_enc_entry &e, ozg_enc_result *res, u32 *tok) {
	const u32 lane = otz_tid ();
	const u8 *in = src + e.src_off;
	u8 *out = dst + e.dst_off;
	bool stored = true;
	const bool codec_only = (e.reserved & OZG_ENC_F_CODEC_ONLY) != 0 && e.method != OZG_METHOD_STORE;
	if (((uintptr_t)out & 15u) != 0 || (e.dst_cap < e.src_size && !codec_only)) {
		if (lane == 0) {
			res->status = OZG_ST_INTERNAL;
			res->comp_size = 0;
			res->method = e.method;
			res->flags = 0;
		}
		return;
	}
	if (codec_only) {
		// codec-level call (deflate() of the z_stream seam): the stream itself is the
		// product, there is no "not smaller -> STORE" rule at this level (ref deflate() has none)
		u32 cs;
		if (e.method == OZG_METHOD_LZMA) {
			cs = otz_lzma_tok_entry (in, e.src_size, out, e.dst_cap, true); // lzmaCompress itself has no "not smaller" rule
		} else if (e.method != OZG_METHOD_DEFLATE) {
			cs = otz_frame_entry (e.method, in, e.src_size, out, e.dst_cap);
		} else if (e.src_size > 0) {
			cs = otz_deflate_entry (L, in, e.src_size, out, e.dst_cap, tok);
		} else {
			cs = 5; // an empty final stored block
			if (lane == 0 && e.dst_cap >= 5u) {
				out[0] = 1;
				out[1] = out[2] = 0;
				out[3] = out[4] = 0xFF;
			}
		}
		otz_wave_sync ();
		if (lane == 0) {
			bool fit = cs <= e.dst_cap;
			res->status = fit ? OZG_ST_OK : OZG_ST_BUF;
			res->comp_size = fit ? cs : 0;
			res->method = e.method;
			res->flags = 0;
		}
		return;
	}
	if (e.method == OZG_METHOD_DEFLATE && e.src_size > 0) {
		u32 cs = otz_deflate_entry (L, in, e.src_size, out, e.dst_cap, tok);
		stored = cs >= e.src_size; // did not fit, or not smaller: ref otezip.c:846-850
		otz_wave_sync ();          // the encoder's last stores land before a STORE copy reuses dst
		if (!stored && lane == 0) {
			res->status = OZG_ST_OK;
			res->comp_size = cs;
		}
	}
	if (e.method == OZG_METHOD_LZMA && e.src_size > 0) {
		// the reference's own buffer size decides where its writer gives up (ref otezip.c:975)
		u32 ref_cap = e.src_size + 13u + e.src_size / 8u;
		u32 cs = e.dst_cap >= ref_cap ? otz_lzma_tok_entry (in, e.src_size, out, ref_cap) : ~0u;
		stored = cs >= e.src_size;
		otz_wave_sync ();
		if (!stored && lane == 0) {
			res->status = OZG_ST_OK;
			res->comp_size = cs;
		}
	}
	if (stored) {
		u32 n = e.src_size;
		for (u32 c = lane; c < (n >> 4); c += OTZ_ENC_NT) {
			otz_st16_aligned (out + 16ull * c, otz_ld16 (in + 16ull * c));
		}
		for (u32 i = (n & ~15u) + lane; i < n; i += OTZ_ENC_NT) {
			out[i] = in[i];
		}
		if (lane == 0) {
			res->status = OZG_ST_OK;
			res->comp_size = n;
		}
	}
	if (lane == 0) {
		res->method = stored ? (uint16_t)OZG_METHOD_STORE : e.method;
		res->flags = 0;
	}
}
