// otz_deflate_enc.h -- raw DEFLATE encoder, one wavefront per buffer.  Replaces
// deflateInit2(level 6, -15)/deflate(Z_FINISH)/deflateEnd as otezip_compress_data
// drives them (ref src/lib/otezip.c:817-852, src/lib/deflate-enc.inc.c:199-525).
//
// This is a re-specification, not a mirror: the reference's encoder writes its
// Huffman codes without the bit reversal RFC 1951 section 3.1.1 requires and
// wraps its window position at 32 KiB, so nothing can decode its
// output -- only its SIZE is a meaningful observable (parity rule 5).  What is
// kept is its shape: one fixed-Huffman block, greedy LZ77 with a single hash
// probe per position (ref :32-112, :333-513).  What is different: the output is
// valid DEFLATE, and the stream ends with an EMPTY FINAL STORED BLOCK so that the
// reference's own inflate, which rejects a final end-of-block code that starts at
// bit 1 of the last byte, accepts every stream produced here.
//
// Layout: 8 KiB LDS ring of input (matches reach 7 168 bytes back), 2048-entry hash
// head table, 2 KiB bit buffer drained with coalesced 16-byte stores: 14 KB per
// wavefront, eleven buffers in flight per CU.
// The 64 lanes work on 64 consecutive input positions per step (one hash probe and
// one match measurement each); only the choice of the greedy token chain through
// those 64 positions is serial, and it is a register walk.
#pragma once

#include "otz_simt.h"
#include "../../include/otezip_gpu.h"

#define OTZ_ENC_NT 64u
#define OTZ_ENC_WIN 8192u
#define OTZ_ENC_WMASK (OTZ_ENC_WIN - 1u)
#define OTZ_ENC_MAXDIST 7168u  // ring 8192 - (lookahead 322 + load chunk 512 + slack)
#define OTZ_ENC_HASH 2048u
#define OTZ_ENC_OUT 2048u      // bit buffer: drained by halves, a step adds at most 64 * 31 bits

struct otz_enc_lds {
	u8 win[OTZ_ENC_WIN];
	u16 head[OTZ_ENC_HASH];
	u32 out_w[OTZ_ENC_OUT / 4]; // bit buffer: tokens are OR-ed in at their bit offsets
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

// all bits of one length/distance pair (at most 8+5+5+13 = 31), LSB first
OTZ_DEV u32 otz_enc_match_bits(u32 len, u32 dist, u32 *n) {
	u32 bits, k;
	if (len == 258u) {
		bits = otz_enc_litlen_bits (285u, &k);
	} else {
		u32 l = len - 3u;
		if (l < 8u) {
			bits = otz_enc_litlen_bits (257u + l, &k);
		} else {
			u32 hb = 31u - (u32)__builtin_clz (l), xb = hb - 2u;
			bits = otz_enc_litlen_bits (257u + 4u * hb - 4u + ((l >> xb) & 3u), &k);
			bits |= (l & ((1u << xb) - 1u)) << k;
			k += xb;
		}
	}
	u32 d = dist - 1u;
	if (d < 4u) {
		bits |= (otz_brev (d) >> 27) << k;
		k += 5u;
	} else {
		u32 hb = 31u - (u32)__builtin_clz (d), xb = hb - 1u;
		u32 code = 2u * hb + ((d >> xb) & 1u);
		bits |= (otz_brev (code) >> 27) << k;
		k += 5u;
		bits |= (d & ((1u << xb) - 1u)) << k;
		k += xb;
	}
	*n = k;
	return bits;
}

OTZ_DEV u32 otz_enc_hash(u32 b0, u32 b1, u32 b2) {
	return (((b0 | (b1 << 8) | (b2 << 16)) * 0x9E3779B1u) >> 21) & (OTZ_ENC_HASH - 1u);
}

// Encode src[0..n) into dst (capacity cap). All 64 work-items call together and all
// get the compressed size, or ~0u when the output does not fit (the caller then
// stores the entry, ref otezip.c:846-850).
//
// 64 input positions per step, one per lane: every lane hashes its three bytes, reads the
// hash head (the previous occurrence), all lanes then publish their own position, and each
// measures its match.  The greedy token chain through the 64 positions is picked by a short
// uniform walk (readlane of each step's advance); the chosen lanes build their code bits, a
// wave prefix sum gives every token its bit offset and the tokens are OR-ed into the LDS bit
// buffer in parallel.
OTZ_DEV u32 otz_deflate_entry(otz_enc_lds &L, const u8 *src, u32 n, u8 *dst, u32 cap) {
	const u32 lane = otz_tid ();
	otz_enc_state S;
	S.bitpos = 3; // BFINAL = 0, BTYPE = 01 (fixed Huffman): bits 0 1 0, LSB first
	S.out_pos = 0;
	S.cap = cap;
	S.dst = dst;
	S.overflow = false;
	for (u32 i = lane; i < OTZ_ENC_HASH; i += OTZ_ENC_NT) {
		L.head[i] = 0;
	}
	for (u32 i = lane; i < OTZ_ENC_OUT / 4; i += OTZ_ENC_NT) {
		L.out_w[i] = i == 0 ? 2u : 0u;
	}
	u32 loaded = 0, skip = 0;
	otz_wave_sync ();
	for (u32 p = 0; p < n && !S.overflow; p += 64u) {
		// keep [p - MAXDIST, p + 64 + 258 + 3) resident in the ring
		while (loaded < n && p + 64u + 258u + 3u > loaded) {
			u32 upto = loaded + 512u;
			for (u32 q = loaded + lane; q < upto && q < n; q += OTZ_ENC_NT) {
				L.win[q & OTZ_ENC_WMASK] = src[q];
			}
			loaded = upto < n ? upto : n;
		}
		otz_wave_sync ();
		const u32 pos = p + lane;
		const bool can = pos + 3u <= n;
		u32 h = 0, c16 = 0;
		if (can) {
			h = otz_enc_hash (L.win[pos & OTZ_ENC_WMASK], L.win[(pos + 1u) & OTZ_ENC_WMASK], L.win[(pos + 2u) & OTZ_ENC_WMASK]);
			c16 = L.head[h];
		}
		otz_wave_sync (); // every lane has read the heads of the previous steps
		if (can) {
			L.head[h] = (u16)pos;
		}
		u32 dist = (pos - c16) & 0xFFFFu, len = 0;
		if (can && dist >= 1u && dist <= OTZ_ENC_MAXDIST && dist <= pos) {
			u32 maxlen = n - pos < 258u ? n - pos : 258u;
			while (len < maxlen && L.win[(pos + len) & OTZ_ENC_WMASK] == L.win[(pos - dist + len) & OTZ_ENC_WMASK]) {
				len++;
			}
			if (len < 3u) {
				len = 0;
			}
		}
		// greedy chain through this step's positions, starting after what the last match covered
		u32 adv = len ? len : 1u;
		u64 sel = 0;
		u32 cur = skip;
		u32 lim = n - p < 64u ? n - p : 64u;
		while (cur < lim) {
			sel |= 1ull << cur;
			cur += otz_shfl (adv, cur);
		}
		skip = cur > 64u ? cur - 64u : 0u;
		const bool chosen = (sel >> lane) & 1ull;
		u32 tb = 0, tn = 0;
		if (chosen) {
			if (len) {
				tb = otz_enc_match_bits (len, dist, &tn);
			} else {
				tb = otz_enc_litlen_bits (L.win[pos & OTZ_ENC_WMASK], &tn);
			}
		}
		u32 incl = tn;
#pragma unroll
		for (u32 d = 1; d < 64; d <<= 1) {
			u32 o = otz_shfl_up (incl, d);
			if (lane >= d) {
				incl += o;
			}
		}
		u32 total = otz_shfl (incl, 63);
		if (chosen) {
			otz_enc_or_bits (L, S.bitpos + incl - tn, tb, tn);
		}
		S.bitpos += total;
		otz_enc_drain (L, S);
	}
	otz_wave_sync ();
	// end of the fixed block, then the final block: stored, empty (BFINAL = 1, BTYPE = 00,
	// pad to a byte, LEN = 0, NLEN = 0xFFFF)
	if (lane == 0) {
		u32 n7;
		u32 eob = otz_enc_litlen_bits (256u, &n7);
		otz_enc_or_bits (L, S.bitpos, eob, n7);
		otz_enc_or_bits (L, S.bitpos + n7, 1u, 3);
	}
	S.bitpos += 10u;
	S.bitpos = (S.bitpos + 7u) & ~7u;
	if (lane == 0) {
		otz_enc_or_bits (L, S.bitpos, 0xFFFF0000u, 32);
	}
	S.bitpos += 32u;
	otz_wave_sync ();
	otz_enc_drain (L, S);
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

OTZ_DEV void otz_encode_entry(otz_enc_lds &L, const u8 *src, u8 *dst, const ozg_enc_entry &e, ozg_enc_result *res) {
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
			cs = otz_deflate_entry (L, in, e.src_size, out, e.dst_cap);
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
		u32 cs = otz_deflate_entry (L, in, e.src_size, out, e.dst_cap);
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
