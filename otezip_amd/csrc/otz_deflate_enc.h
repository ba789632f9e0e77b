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
//   PARSE  256 input positions per iteration: OTZ_ENC_U = 4 independent steps of 64 positions
//          (one per lane) are in flight together, because one step is a chain of dependent
//          LDS round trips and LDS capacity allows only two wavefronts per SIMD.  Every lane
//          hashes the four bytes at its position, reads its bucket (the most recent earlier
//          position with that hash; two with OTZ_ENC_WAYS = 2), publishes its own position,
//          and measures the candidate eight bytes at a time (dword-aligned LDS reads +
//          v_alignbyte; see the access-cost table in otz_simt.h).  Candidate bytes are read
//          unconditionally -- any address is inside the ring -- and compared without
//          branches, so that all reads of an iteration are in flight at once.  Matches are
//          at least four bytes long (with a dynamic code three-byte matches cost more than
//          their literals on text).  One step of lazy evaluation: a position whose right
//          neighbour has a longer match becomes a literal.  The token chain through the 64
//          positions of a step -- the only serial part of LZ77 parsing -- is found by
//          POINTER DOUBLING: every lane knows where the chain goes from its own position
//          (position + advance), six rounds of ds_bpermute give every lane the set of
//          positions the chain visits from it and where it leaves the step; the lane the
//          previous step ended on holds the answer.  Chosen lanes append their token to
//          the block's token list (global scratch, 4 bytes per token: symbols and extra
//          bits) and count their symbols in LDS histograms.
//   CODE   Huffman codes for the block from the histograms: symbols ranked by weight (all
//          lanes), the two-queue merge (one lane, one step per symbol in use), leaf depths by walking up
//          (all lanes), depth limit 15 by the usual "move a leaf down" repair on the
//          length counts, lengths re-dealt in rank order, canonical codes from ballots.
//          The code lengths are run-length coded (symbols 16/17/18) and get their own
//          7-bit-limited code.  The block is written with the fixed code instead when
//          that is smaller (short inputs).
//   EMIT   64 tokens per step: table lookups give every token its bits (at most 48), a wave
//          prefix sum gives its bit offset, the tokens are OR-ed into a 1 KiB LDS bit
//          buffer that is drained with coalesced 16-byte stores.
//
// LDS: 8 KiB input ring (matches reach 7 150 bytes back), 4096 buckets (the code construction
// borrows their place between the parse and the emit of a block), the bit buffer, histograms
// and code tables: 19.5 KB per wavefront, eight buffers in flight per CU.
#pragma once

#include "otz_simt.h"
#include "../../include/otezip_gpu.h"

#define OTZ_ENC_NT 64u
#ifndef OTZ_ENC_WIN
#define OTZ_ENC_WIN 8192u
#endif
#define OTZ_ENC_WMASK (OTZ_ENC_WIN - 1u)
#define OTZ_ENC_LOAD 512u                 // bytes per refill of the ring
#ifndef OTZ_ENC_U
#define OTZ_ENC_U 4u                      // steps of 64 positions in flight together (instruction-level parallelism)
#endif
#define OTZ_ENC_STEP (64u * OTZ_ENC_U)
#define OTZ_ENC_AHEAD (OTZ_ENC_STEP + 258u + 8u) // an iteration looks this far beyond its first position
#define OTZ_ENC_MAXDIST (OTZ_ENC_WIN - OTZ_ENC_AHEAD - OTZ_ENC_LOAD - 8u)
#ifndef OTZ_ENC_WAYS
#define OTZ_ENC_WAYS 1u                   // candidates per bucket: 1 (4096 buckets) or 2 (2048 buckets; +1.3 % ratio, +25 % work)
#endif
#ifndef OTZ_ENC_HBITS
#define OTZ_ENC_HBITS (OTZ_ENC_WAYS == 2u ? 11u : 12u)
#endif
#define OTZ_ENC_HASH (1u << OTZ_ENC_HBITS)
#define OTZ_ENC_HEAD_WORDS (OTZ_ENC_HASH * OTZ_ENC_WAYS / 2u)
#define OTZ_ENC_OUT 1024u                 // bit buffer: drained by halves, a step adds at most 64 * 48 bits
#define OTZ_ENC_BLOCK 65536u              // input bytes per DEFLATE block
#define OTZ_ENC_MINLEN 4u
#define OTZ_ENC_LAZY_MAX 32u              // a match at least this long is taken without looking at the next position
#define OTZ_ENC_TOK_WORDS (OTZ_ENC_BLOCK + 2048u) // scratch per wavefront: one u32 per input byte of a block (all literals)
                                          // + where the part of the bucket table the code construction borrows is parked
#define OTZ_ENC_DSYM 288u                 // distance symbols live at [288, 320) of hist / code / lens

#define OTZ_TOK_MATCH 0x80000000u         // token: literal = byte; match = this | (length symbol - 257) << 23 | its extra bits << 18
                                          //        | distance symbol << 13 | its extra bits

// scratch of the code construction and the header of a block: it lives where the bucket table is,
// between the parse and the emit of a block (a buffer of more than one block parks that part of the
// table in global memory meanwhile)
struct otz_enc_build {
	u32 sw[288];             // leaf weights, ascending
	u32 iw[288];             // internal-node weights, in creation order
	u16 ssym[288];           // symbol of the leaf with rank r
	u16 lpar[288];           // parent (internal node index) of leaf r
	u16 ipar[288];           // parent of internal node k
	u32 blc[16];             // leaves per code length
	u32 base[16];            // next canonical code per length
	u8 lens[320];            // code lengths, laid out like hist / code
	// the code-length code
	u32 clhist[32];
	u32 clcode[32];
	u8 cllens[32];
	u16 cltok[320];          // run-length tokens: symbol | extra << 5
	u32 n_cltok;
	u32 pad[3];
};
#define OTZ_ENC_BUILD_U4 ((u32) (sizeof (otz_enc_build) / 16u))

struct otz_enc_lds {
	u32 win_w[OTZ_ENC_WIN / 4 + 4]; // the ring; its first 16 bytes are repeated behind its end, so that no read wraps
	union {
		u32 head[OTZ_ENC_HEAD_WORDS]; // two ways: a word per bucket, [15:0] the most recent position with this hash
		                              // (mod 64 Ki), [31:16] the one before; one way: a u16 per bucket
		otz_enc_build b;
	};
	u32 out_w[OTZ_ENC_OUT / 4]; // bit buffer: tokens are OR-ed in at their bit offsets
	u32 hist[320];           // literal/length symbols at 0..287, distance symbols at 288..319
	u32 code[320];           // same layout: bit-reversed code | length << 16
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
		otz_wave_sync_lds ();
		if (S.out_pos + HALF_B > S.cap) {
			S.overflow = true;
		} else {
			for (u32 c = otz_tid (); c < HALF_B / 16u; c += OTZ_ENC_NT) {
				otz_st16_aligned (S.dst + S.out_pos + 16u * c, *reinterpret_cast<const otz_u4 *> (&L.out_w[4u * c]));
			}
		}
		otz_wave_sync_lds ();
		for (u32 i = otz_tid (); i < HALF_W; i += OTZ_ENC_NT) {
			L.out_w[i] = L.out_w[i + HALF_W];
			L.out_w[i + HALF_W] = 0;
		}
		otz_wave_sync_lds ();
		S.out_pos += HALF_B;
		S.bitpos -= HALF_B * 8u;
	}
}

// Every lane contributes nb (0..48) bits: a wave prefix sum places them behind S.bitpos, in lane order.
OTZ_DEV void otz_enc_put(otz_enc_lds &L, otz_enc_state &S, u64 bits, u32 nb) {
	const u32 incl = otz_wave_scan_incl (nb);
	const u32 total = otz_readlane (incl, 63);
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

// eight bytes of the ring starting at byte position q: the three dwords that hold them (one 8-byte
// and one 4-byte read, both dword-aligned) and two v_alignbyte
OTZ_DEV u64 otz_enc_ld8(const otz_enc_lds &L, u32 q) {
	const u32 b = q & OTZ_ENC_WMASK, i = b >> 2, s = b & 3u;
	const u64 w01 = otz_ld64u (&L.win_w[i]);
	const u32 w2 = L.win_w[i + 2u];
	return (u64)otz_alignbyte ((u32) (w01 >> 32), (u32)w01, s) | (u64)otz_alignbyte (w2, (u32) (w01 >> 32), s) << 32;
}

// 16 bytes into the ring at position q (a multiple of 16)
OTZ_DEV void otz_enc_ring_put16(otz_enc_lds &L, u32 q, otz_u4 v) {
	const u32 b = q & OTZ_ENC_WMASK;
	*reinterpret_cast<otz_u4 *> (&L.win_w[b >> 2]) = v;
	if (b == 0u) {
		*reinterpret_cast<otz_u4 *> (&L.win_w[OTZ_ENC_WIN / 4u]) = v;
	}
}

OTZ_DEV u32 otz_enc_wave_sum(u32 v) {
#pragma unroll
	for (u32 d = 1; d < 64; d <<= 1) {
		v += otz_shfl_xor (v, d);
	}
	return v;
}

struct otz_false_c {
	static constexpr bool value = false;
};
struct otz_true_c {
	static constexpr bool value = true;
};

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
		otz_wave_sync_lds ();
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
		otz_wave_sync_lds ();
	}
	// the symbols in use, as keys (weight << 9 | symbol), packed in symbol order (L.b.iw is free until the merge)
	{
		u32 before = 0;
#pragma unroll
		for (u32 k = 0; k < NCH; k++) {
			const u32 i = lane + 64u * k;
			const u32 f = i < nsym ? freq[i] : 0u;
			const u64 bal = otz_ballot (f != 0u);
			if (f) {
				L.b.iw[before + otz_popc64 (bal & ((1ull << lane) - 1ull))] = f << 9 | i;
			}
			if (i < nsym) {
				lens[i] = 0;
			}
			before += otz_popc64 (bal);
		}
		m = before;
	}
	if (lane < 16u) {
		L.b.blc[lane] = 0;
	}
	otz_wave_sync_lds ();
	// rank by (weight, symbol): every key against all m
	{
		u32 key[NCH], rk[NCH];
#pragma unroll
		for (u32 k = 0; k < NCH; k++) {
			const u32 c = lane + 64u * k;
			key[k] = c < m ? L.b.iw[c] : 0u;
			rk[k] = 0;
		}
#pragma unroll 4
		for (u32 j = 0; j < m; j++) {
			const u32 kj = L.b.iw[j];
#pragma unroll
			for (u32 k = 0; k < NCH; k++) {
				rk[k] += kj < key[k] ? 1u : 0u;
			}
		}
#pragma unroll
		for (u32 k = 0; k < NCH; k++) {
			if (key[k]) {
				L.b.sw[rk[k]] = key[k] >> 9;
				L.b.ssym[rk[k]] = (u16) (key[k] & 511u);
			}
		}
	}
	otz_wave_sync_lds ();
	// two-queue merge: leaves in ascending order, internal nodes in creation order (also ascending);
	// the next two of each queue are kept in registers, so a step waits for LDS once
	if (lane == 0) {
		const u32 INF = 0xFFFFFFFFu;
		u32 li = 0, ih = 0;
		u32 l0 = L.b.sw[0], l1 = L.b.sw[1], i0 = INF, i1 = INF;
		for (u32 k = 0; k + 1u < m; k++) {
			u32 sum, nl, ni;
			if (l0 <= i0) { // (a leaf wins a tie)
				if (l1 <= i0) {
					sum = l0 + l1;
					nl = 2;
					ni = 0;
				} else {
					sum = l0 + i0;
					nl = 1;
					ni = 1;
				}
			} else if (l0 <= i1) {
				sum = i0 + l0;
				nl = 1;
				ni = 1;
			} else {
				sum = i0 + i1;
				nl = 0;
				ni = 2;
			}
			for (u32 t = 0; t < nl; t++) {
				L.b.lpar[li + t] = (u16)k;
			}
			for (u32 t = 0; t < ni; t++) {
				L.b.ipar[ih + t] = (u16)k;
			}
			li += nl;
			ih += ni;
			L.b.iw[k] = sum;
			l0 = li < m ? L.b.sw[li] : INF;
			l1 = li + 1u < m ? L.b.sw[li + 1u] : INF;
			i0 = ih < k ? L.b.iw[ih] : (ih == k ? sum : INF);
			i1 = ih + 1u < k ? L.b.iw[ih + 1u] : (ih + 1u == k ? sum : INF);
		}
	}
	otz_wave_sync_lds ();
	// leaf depths, clamped; K = Kraft sum in units of 2^-limit
	u32 K = 0;
#pragma unroll
	for (u32 k = 0; k < NCH; k++) {
		const u32 r = lane + 64u * k;
		if (r < m) {
			u32 d = 1, q = L.b.lpar[r];
			while (q != m - 2u) {
				q = L.b.ipar[q];
				d++;
			}
			d = d < limit ? d : limit;
			otz_lds_add (&L.b.blc[d], 1u);
			K += 1u << (limit - d);
		}
	}
	K = otz_enc_wave_sum (K);
	otz_wave_sync_lds ();
	if (K > (1u << limit) && lane == 0) {
		// too deep a tree was cut off at `limit`: every round turns a leaf above the limit into
		// a node with two leaves below it, one of them taken from the bottom row: Kraft sum - 1
		for (u32 over = K - (1u << limit); over > 0; over--) {
			u32 b = limit - 1u;
			while (L.b.blc[b] == 0u) {
				b--;
			}
			L.b.blc[b]--;
			L.b.blc[b + 1u] += 2u;
			L.b.blc[limit]--;
		}
	}
	otz_wave_sync_lds ();
	// deal the lengths out again in rank order (lightest leaf = longest code), first canonical codes
	{
		u32 c = 0, mine = 0;
		for (u32 b = 1; b <= limit; b++) {
			c = (c + L.b.blc[b - 1u]) << 1;
			if (lane == b) {
				mine = c;
			}
		}
		if (lane >= 1u && lane < 16u) {
			L.b.base[lane] = mine;
		}
	}
#pragma unroll
	for (u32 k = 0; k < NCH; k++) {
		const u32 r = lane + 64u * k;
		if (r < m) {
			u32 cum = 0, len = 0;
			for (u32 b = limit; b >= 1u; b--) {
				const u32 c = L.b.blc[b];
				if (r >= cum && r < cum + c) {
					len = b;
				}
				cum += c;
			}
			lens[L.b.ssym[r]] = (u8)len;
		}
	}
	otz_wave_sync_lds ();
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
				const u32 c = L.b.base[len] + otz_popc64 (same & ((1ull << lane) - 1ull));
				e = (otz_brev (c) >> (32u - len)) | len << 16;
			}
			code[i] = e;
		}
		otz_wave_sync_lds ();
		if (lane >= 1u && lane < 16u) {
			L.b.base[lane] += add;
		}
		otz_wave_sync_lds ();
	}
}

// Run-length tokens (code-length alphabet, RFC 1951 3.2.7) for lens[0..n), appended to L.b.cltok; one lane.
// lens is 8-byte aligned and readable up to the next multiple of eight: eight lengths per LDS read.
OTZ_DEV u32 otz_enc_rle_lengths(otz_enc_lds &L, const u8 *lens, u32 n, u32 nt) {
	u32 v = 0x100u, r = 0; // the run so far
	u64 w = 0;
	for (u32 i = 0; i <= n; i++) {
		if ((i & 7u) == 0u && i < n) {
			w = otz_ld64u (lens + i);
		}
		const u32 c = i < n ? (u32) (w >> (8u * (i & 7u))) & 0xFFu : 0x200u;
		if (c == v) {
			r++;
			continue;
		}
		if (v == 0u) {
			while (r >= 11u) {
				const u32 t = r < 138u ? r : 138u;
				L.b.cltok[nt++] = (u16) (18u | (t - 11u) << 5);
				r -= t;
			}
			if (r >= 3u) {
				L.b.cltok[nt++] = (u16) (17u | (r - 3u) << 5);
				r = 0;
			}
		} else if (r > 0u) {
			L.b.cltok[nt++] = (u16)v;
			r--;
			while (r >= 3u) {
				const u32 t = r < 6u ? r : 6u;
				L.b.cltok[nt++] = (u16) (16u | (t - 3u) << 5);
				r -= t;
			}
		}
		while (r > 0u) {
			L.b.cltok[nt++] = (u16)v;
			r--;
		}
		v = c;
		r = 1;
	}
	return nt;
}

// Encode src[0..n) into dst (capacity cap); tok = OTZ_ENC_TOK_WORDS words of scratch.  All 64
// work-items call together and all get the compressed size, or ~0u when the output does not
// fit (the caller then stores the entry, ref otezip.c:846-850).
// prof (diagnostic builds): 16 x u64 cycle counters -- 0 ring refill + hash + bucket, 1 match measuring,
// 2 chain selection, 3 tokens + histogram, 4 litlen code, 5 distance code, 6 run-length + its code + cost,
// 7 header, 8 token emit, 15 total
#define OTZ_ENC_MARK(i) \
	do { \
		if (PROF) { \
			const u64 _t = otz_cycles (); \
			pc[i] += _t - plast; \
			plast = _t; \
		} \
	} while (0)
template <bool PROF>
OTZ_DEV u32 otz_deflate_entry(otz_enc_lds &L, const u8 *src, u32 n, u8 *dst, u32 cap, u32 *tok, u64 *prof = nullptr) {
	const u32 lane = otz_tid ();
	u64 pc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	u64 plast = PROF ? otz_cycles () : 0;
	const u64 pstart = plast;
	const u32 lane4 = lane << 2;
	otz_enc_state S;
	S.bitpos = 0;
	S.out_pos = 0;
	S.cap = cap;
	S.dst = dst;
	S.overflow = false;
	for (u32 i = lane; i < OTZ_ENC_HEAD_WORDS; i += OTZ_ENC_NT) {
		L.head[i] = 0;
	}
	for (u32 i = lane; i < OTZ_ENC_OUT / 4; i += OTZ_ENC_NT) {
		L.out_w[i] = 0;
	}
	u32 loaded = 0, skip = 0;
	otz_u4 pre = { 0, 0, 0, 0 }; // the next refill of the ring, in flight
	if (lane < OTZ_ENC_LOAD / 16u && 16u * lane + 16u <= n) {
		pre = otz_ld16 (src + 16u * lane);
	}
	otz_wave_sync_lds ();
	for (u32 bstart = 0; bstart < n && !S.overflow; bstart += OTZ_ENC_BLOCK) {
		const u32 bend = n - bstart < OTZ_ENC_BLOCK ? n : bstart + OTZ_ENC_BLOCK;
		for (u32 i = lane; i < 320u; i += OTZ_ENC_NT) {
			L.hist[i] = 0;
		}
		u32 ntok = 0;
		otz_wave_sync ();
		// ------------------------------------------------------------------ PARSE
		for (u32 p = bstart; p < bend; p += OTZ_ENC_STEP) {
			// keep [p - MAXDIST, p + AHEAD) resident in the ring; the next refill is already on its way
			while (loaded < n && p + OTZ_ENC_AHEAD > loaded) {
				const u32 q = loaded + 16u * lane;
				if (lane < OTZ_ENC_LOAD / 16u) {
					if (q + 16u <= n) {
						otz_enc_ring_put16 (L, q, pre);
					} else {
						u8 *wb = reinterpret_cast<u8 *> (L.win_w);
						for (u32 t = q; t < n; t++) {
							wb[t & OTZ_ENC_WMASK] = src[t];
							if ((t & OTZ_ENC_WMASK) < 16u) {
								wb[OTZ_ENC_WIN + (t & OTZ_ENC_WMASK)] = src[t];
							}
						}
					}
				}
				loaded += OTZ_ENC_LOAD;
				if (lane < OTZ_ENC_LOAD / 16u && q + OTZ_ENC_LOAD + 16u <= n) {
					pre = otz_ld16 (src + q + OTZ_ENC_LOAD);
				}
			}
			otz_wave_sync_lds ();
			// (TAIL: the iteration may reach the end of the buffer; all others run without the end checks)
			auto iteration = [&](auto tail_c) {
			constexpr bool TAIL = decltype (tail_c)::value;
			u32 pos[OTZ_ENC_U], h[OTZ_ENC_U], c[OTZ_ENC_U], len[OTZ_ENC_U], dist[OTZ_ENC_U];
			u64 own[OTZ_ENC_U];
			bool can[OTZ_ENC_U];
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				pos[u] = p + 64u * u + lane;
				can[u] = !TAIL || pos[u] + OTZ_ENC_MINLEN <= n;
				own[u] = otz_enc_ld8 (L, pos[u]);
				h[u] = ((u32)own[u] * 0x9E3779B1u) >> (32u - OTZ_ENC_HBITS);
			}
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				c[u] = OTZ_ENC_WAYS == 2u ? L.head[h[u]] : (u32)reinterpret_cast<const u16 *> (L.head)[h[u]];
			}
			otz_wave_sync_lds (); // every lane has read the buckets of the earlier steps
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				// (the last three positions of a buffer, and lanes beyond its end, hash bytes that are
				// not theirs: nothing probes after them, and `can` keeps them from matching)
				if (OTZ_ENC_WAYS == 2u) {
					L.head[h[u]] = c[u] << 16 | (pos[u] & 0xFFFFu);
				} else {
					reinterpret_cast<u16 *> (L.head)[h[u]] = (u16)pos[u];
				}
			}
			OTZ_ENC_MARK (0);
			// all candidates of all steps are measured together: one LDS round trip per eight bytes
			u32 cd[OTZ_ENC_U][OTZ_ENC_WAYS], cl[OTZ_ENC_U][OTZ_ENC_WAYS], maxlen[OTZ_ENC_U];
			bool act[OTZ_ENC_U][OTZ_ENC_WAYS];
			bool any = false;
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				maxlen[u] = TAIL && n - pos[u] < 258u ? n - pos[u] : 258u;
				const bool open = can[u] && 64u * u + lane >= skip;
				const u32 reach = pos[u] < OTZ_ENC_MAXDIST ? pos[u] : OTZ_ENC_MAXDIST; // the farthest a match may lie back
#pragma unroll
				for (u32 w = 0; w < OTZ_ENC_WAYS; w++) {
					const u32 d = (pos[u] - (w ? c[u] >> 16 : c[u])) & 0xFFFFu;
					cd[u][w] = d;
					cl[u][w] = 0;
					act[u][w] = false;
					// (the candidate's bytes are read whether or not it is one: any address is inside the
					// ring, and all the reads of an iteration are in flight together)
					const bool valid = open && d - 1u < reach && (w == 0u || d != cd[u][0]);
					const u64 x = own[u] ^ otz_enc_ld8 (L, pos[u] - d);
					const u32 m8 = x != 0ull ? otz_ctz64 (x) >> 3 : 8u;
					cl[u][w] = valid ? m8 : 0u;
					act[u][w] = valid && m8 == 8u && 8u < maxlen[u];
					any = any || act[u][w];
				}
			}
			for (u32 k = 8; otz_ballot (any) != 0ull; k += 8u) {
				// (branch-free, and every read of the round in flight together: the wavefront goes
				// round as long as any lane has a live match, so a finished lane costs the same)
				u64 o[OTZ_ENC_U], cb[OTZ_ENC_U][OTZ_ENC_WAYS];
#pragma unroll
				for (u32 u = 0; u < OTZ_ENC_U; u++) {
					o[u] = otz_enc_ld8 (L, pos[u] + k);
#pragma unroll
					for (u32 w = 0; w < OTZ_ENC_WAYS; w++) {
						cb[u][w] = otz_enc_ld8 (L, pos[u] - cd[u][w] + k);
					}
				}
				any = false;
#pragma unroll
				for (u32 u = 0; u < OTZ_ENC_U; u++) {
#pragma unroll
					for (u32 w = 0; w < OTZ_ENC_WAYS; w++) {
						const u64 x = o[u] ^ cb[u][w];
						const u32 m8 = x != 0ull ? otz_ctz64 (x) >> 3 : 8u;
						cl[u][w] += act[u][w] ? m8 : 0u;
						act[u][w] = act[u][w] && m8 == 8u && k + 8u < maxlen[u];
						any = any || act[u][w];
					}
				}
			}
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				const u32 l0 = cl[u][0] < maxlen[u] ? cl[u][0] : maxlen[u];
				const u32 l1 = OTZ_ENC_WAYS == 2u ? (cl[u][OTZ_ENC_WAYS - 1u] < maxlen[u] ? cl[u][OTZ_ENC_WAYS - 1u] : maxlen[u]) : 0u;
				len[u] = l1 > l0 ? l1 : l0;
				dist[u] = l1 > l0 ? cd[u][OTZ_ENC_WAYS - 1u] : cd[u][0];
				len[u] = len[u] >= OTZ_ENC_MINLEN ? len[u] : 0u;
			}
			OTZ_ENC_MARK (1);
			// lazy evaluation, then where the chain goes from every position (all steps at once)
			u32 J[OTZ_ENC_U];
			u64 R[OTZ_ENC_U];
			bool take[OTZ_ENC_U];
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				const u32 fill = u + 1u < OTZ_ENC_U ? otz_readlane (len[u + 1u < OTZ_ENC_U ? u + 1u : u], 0) : 0u;
				const u32 lnext = otz_lane_down1 (len[u], fill);
				take[u] = len[u] != 0u && (len[u] >= OTZ_ENC_LAZY_MAX || lnext <= len[u]);
				J[u] = lane + (take[u] ? len[u] : 1u);
				R[u] = 1ull << lane;
			}
			// J is kept premultiplied by four (ds_bpermute takes byte addresses).  A lane whose chain
			// has left the step reads ITSELF from then on, which changes nothing: no selects.
			// (the first round needs no mask from the other lane: one hop reaches J itself)
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				const bool in = J[u] < 64u;
				R[u] |= in ? 1ull << J[u] : 0ull;
				J[u] <<= 2;
				J[u] = otz_bpermute4 (in ? J[u] : lane4, J[u]);
			}
#pragma unroll
			for (u32 k = 1; k < 6u; k++) {
#pragma unroll
				for (u32 u = 0; u < OTZ_ENC_U; u++) {
					const u32 from = J[u] < 256u ? J[u] : lane4;
					const u32 Jn = otz_bpermute4 (from, J[u]);
					const u32 Rlo = otz_bpermute4 (from, (u32)R[u]), Rhi = otz_bpermute4 (from, (u32) (R[u] >> 32));
					R[u] |= (u64)Rhi << 32 | Rlo;
					J[u] = Jn;
				}
			}
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				J[u] >>= 2;
			}
			OTZ_ENC_MARK (2);
#pragma unroll
			for (u32 u = 0; u < OTZ_ENC_U; u++) {
				if (skip >= 64u) {
					skip -= 64u; // the whole step lies inside the previous match
					continue;
				}
				if (TAIL && p + 64u * u >= n) {
					break;
				}
				const u32 left = n - (p + 64u * u), lim = TAIL && left < 64u ? left : 64u;
				u64 sel = (u64)otz_readlane ((u32)R[u], skip) | (u64)otz_readlane ((u32) (R[u] >> 32), skip) << 32;
				skip = otz_readlane (J[u], skip) - 64u;
				if (TAIL && lim < 64u) {
					sel &= (1ull << (lim & 63u)) - 1ull;
				}
				if ((sel >> lane) & 1ull) {
					u32 t;
					if (take[u]) {
						u32 xb, lxv, dxv;
						const u32 ls = otz_enc_len_sym (len[u], &xb, &lxv), ds = otz_enc_dist_sym (dist[u], &xb, &dxv);
						t = OTZ_TOK_MATCH | (ls - 257u) << 23 | lxv << 18 | ds << 13 | dxv;
						otz_lds_add (&L.hist[ls], 1u);
						otz_lds_add (&L.hist[OTZ_ENC_DSYM + ds], 1u);
					} else {
						t = (u32)own[u] & 0xFFu;
						otz_lds_add (&L.hist[t], 1u);
					}
					tok[ntok + otz_mbcnt (sel)] = t;
				}
				ntok += otz_popc64 (sel);
			}
			};
			if (p + OTZ_ENC_STEP + 258u + 8u <= n) {
				iteration (otz_false_c ());
			} else {
				iteration (otz_true_c ());
			}
			OTZ_ENC_MARK (3);
		}
		otz_wave_sync ();
		// ------------------------------------------------------------------ CODE
		const bool more = bend < n;
		if (more) {
			for (u32 i = lane; i < OTZ_ENC_BUILD_U4; i += OTZ_ENC_NT) {
				otz_st16_aligned (tok + OTZ_ENC_BLOCK + 4u * i, *reinterpret_cast<const otz_u4 *> (&L.head[4u * i]));
			}
			otz_wave_sync_lds ();
		}
		if (lane == 0) {
			L.hist[256] = 1; // the end-of-block code
		}
		for (u32 i = lane; i < 320u; i += OTZ_ENC_NT) {
			L.b.lens[i] = 0;
		}
		otz_wave_sync_lds ();
		otz_enc_build_code<5> (L, L.hist, 286u, 15u, L.b.lens, L.code);
		OTZ_ENC_MARK (4);
		otz_enc_build_code<1> (L, L.hist + OTZ_ENC_DSYM, 30u, 15u, L.b.lens + OTZ_ENC_DSYM, L.code + OTZ_ENC_DSYM);
		otz_wave_sync_lds ();
		OTZ_ENC_MARK (5);
		// sizes: how many length and distance codes are sent, the run-length tokens, their code
		u32 hlit = 257, hdist = 1;
#pragma unroll
		for (u32 k = 0; k < 5u; k++) {
			const u32 i = lane + 64u * k;
			const u64 nz = otz_ballot (i < 286u && L.b.lens[i] != 0u);
			if (nz) {
				const u32 top = 64u * k + 64u - otz_clz64 (nz);
				hlit = top > hlit ? top : hlit;
			}
		}
		{
			const u64 nz = otz_ballot (lane < 30u && L.b.lens[OTZ_ENC_DSYM + lane] != 0u);
			if (nz) {
				hdist = 64u - otz_clz64 (nz);
			}
		}
		if (lane < 32u) {
			L.b.clhist[lane] = 0;
		}
		otz_wave_sync_lds ();
		if (lane == 0) {
			u32 nt = otz_enc_rle_lengths (L, L.b.lens, hlit, 0);
			L.b.n_cltok = otz_enc_rle_lengths (L, L.b.lens + OTZ_ENC_DSYM, hdist, nt);
		}
		otz_wave_sync_lds ();
		const u32 n_cltok = L.b.n_cltok;
		for (u32 i = lane; i < n_cltok; i += OTZ_ENC_NT) {
			otz_lds_add (&L.b.clhist[L.b.cltok[i] & 31u], 1u);
		}
		otz_wave_sync_lds ();
		otz_enc_build_code<1> (L, L.b.clhist, 19u, 7u, L.b.cllens, L.b.clcode);
		otz_wave_sync_lds ();
		// the code-length code lengths go out in the order of RFC 1951 3.2.7
		const u32 my_cl = lane < 19u ? L.b.cllens[otz_cl_order[lane]] : 0u;
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
			dyn += f * L.b.lens[i];
			fix += f * fl;
		}
		for (u32 i = lane; i < n_cltok; i += OTZ_ENC_NT) {
			const u32 t = L.b.cltok[i] & 31u;
			dyn += L.b.cllens[t] + (t == 16u ? 2u : t == 17u ? 3u : t == 18u ? 7u : 0u);
		}
		dyn = otz_enc_wave_sum (dyn) + 14u + 3u * hclen;
		fix = otz_enc_wave_sum (fix);
		const bool use_fixed = fix <= dyn;
		OTZ_ENC_MARK (6);
		// ------------------------------------------------------------------ EMIT
		if (use_fixed) {
			otz_wave_sync_lds ();
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
			otz_wave_sync_lds ();
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
					const u32 t = L.b.cltok[i], sym = t & 31u, e = L.b.clcode[sym];
					nb = e >> 16;
					bits = (e & 0xFFFFu) | (u64) (t >> 5) << nb;
					nb += sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u;
				}
				otz_enc_put (L, S, bits, nb);
			}
		}
		OTZ_ENC_MARK (7);
		u32 tnext = lane < ntok ? tok[lane] : 0u;
		for (u32 b = 0; b < ntok && !S.overflow; b += 64u) {
			const u32 i = b + lane;
			u64 bits = 0;
			u32 nb = 0;
			const u32 t = tnext;
			if (i + 64u < ntok) {
				tnext = tok[i + 64u];
			}
			if (i < ntok) {
				if (t & OTZ_TOK_MATCH) {
					const u32 li = (t >> 23) & 31u, ds = (t >> 13) & 31u;
					const u32 e = L.code[257u + li], e2 = L.code[OTZ_ENC_DSYM + ds];
					const u32 xb = li < 8u || li == 28u ? 0u : (li >> 2) - 1u, xb2 = ds < 4u ? 0u : (ds >> 1) - 1u;
					nb = e >> 16;
					bits = (e & 0xFFFFu) | (u64) ((t >> 18) & 31u) << nb;
					nb += xb;
					bits |= (u64) (e2 & 0xFFFFu) << nb;
					nb += e2 >> 16;
					bits |= (u64) (t & 0x1FFFu) << nb;
					nb += xb2;
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
		if (more) {
			for (u32 i = lane; i < OTZ_ENC_BUILD_U4; i += OTZ_ENC_NT) {
				*reinterpret_cast<otz_u4 *> (&L.head[4u * i]) = otz_ld16_aligned (tok + OTZ_ENC_BLOCK + 4u * i);
			}
			otz_wave_sync_lds ();
		}
		OTZ_ENC_MARK (8);
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
	if (PROF && prof && lane == 0) {
		for (u32 i = 0; i < 9u; i++) {
			prof[i] = pc[i];
		}
		prof[15] = otz_cycles () - pstart;
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

template <bool PROF>
OTZ_DEV void otz_encode_entry(otz_enc_lds &L, const u8 *src, u8 *dst, const ozg_enc_entry &e, ozg_enc_result *res, u32 *tok, u64 *prof = nullptr) {
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
	// (one call site: the encoder is by far the largest function of the kernel)
	u32 dcs = 0;
	if (e.method == OZG_METHOD_DEFLATE && e.src_size > 0) {
		dcs = otz_deflate_entry<PROF> (L, in, e.src_size, out, e.dst_cap, tok, prof);
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
			cs = dcs;
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
		const u32 cs = dcs;
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
