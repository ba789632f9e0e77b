// otz_containers.h -- the copy-shaped codecs of otezip, one entry per 256-lane
// workgroup, CRC-32 fused into the copy.
//
//   STORE      ref src/lib/otezip.c:481-487
//   method 93  private "zstd" raw-block framing, ref src/lib/zstd.inc.c:479-705
//   method 100 private "lzfse" 0x06 raw block,   ref src/lib/lzfse.inc.c:225-257
//   method 97  private "BROT" container + CRC,   ref src/lib/brotli.inc.c:57-80, :152-180
//   method 14  private "lzma" RLE/literal tokens, ref src/lib/lzma-dec.inc.c:102-155, :216-308
//
// None of these is an entropy decoder: the work is a bandwidth
// bound copy plus the CRC the reference takes over every extracted entry
// (ref src/lib/otezip.c:668).  Payloads are read with 16-byte (possibly
// unaligned) global loads, lane l of every 256-lane stripe taking bytes
// [16l, 16l+16); each lane keeps its own CRC register over its interleaved
// 16-byte pieces (advance by 4096 bytes per piece through LDS tables) and the
// 256 registers are aligned with one modular product per lane at the end.
#pragma once

#include "otz_simt.h"
#include "otz_crc.h"
#include "../../include/otezip_gpu.h"

#define OTZ_CNT 256u

struct otz_copy_lds {
	u32 t4[1024]; // slice-by-4
	u32 a4[1024]; // advance by 4096 bytes
	u32 scan[8];
	u32 status;
	u32 seg_src, seg_len, seg_last;
	u32 n_tok, tok_src_next, tok_dst_next;
	u32 tk_src[OTZ_CNT];
	u32 tk_dst[OTZ_CNT];
	u16 tk_len[OTZ_CNT];
	u8 tk_kind[OTZ_CNT];
	u8 tk_byte[OTZ_CNT];
};

OTZ_DEV u32 otz_rd_le32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
OTZ_DEV u64 otz_rd_le64(const u8 *p) { return (u64)otz_rd_le32 (p) | ((u64)otz_rd_le32 (p + 4) << 32); }

// R(M): CRC register reached from zero over src[0..n), optionally copying to dst.
// dst_aligned: dst is 16-byte aligned (full-width aligned stores); otherwise the copy uses
// unaligned 16-byte stores.  All work-items call; all get R.
OTZ_DEV u32 otz_stream_crc(otz_copy_lds &L, const u8 *src, u8 *dst, u32 n, bool do_copy, bool dst_aligned = true) {
	const u32 tid = otz_tid ();
	const u32 nch = n >> 4, tail = n & 15u;
	// front-pad with zero pieces so that the last real piece belongs to lane 255
	const u32 pad = (OTZ_CNT - (nch & (OTZ_CNT - 1u))) & (OTZ_CNT - 1u);
	const u32 iters = (nch + pad) / OTZ_CNT;
	u32 r = 0;
	u32 k = 0;
	// four pieces in flight per lane
	for (; k + 4 <= iters; k += 4) {
		otz_u4 v[4];
		bool ok[4];
#pragma unroll
		for (u32 j = 0; j < 4; j++) {
			u32 vi = tid + OTZ_CNT * (k + j);
			ok[j] = vi >= pad;
			if (ok[j]) {
				v[j] = otz_ld16 (src + 16ull * (vi - pad));
			}
		}
#pragma unroll
		for (u32 j = 0; j < 4; j++) {
			r = otz_crc_adv4k (r, L.a4);
			if (ok[j]) {
				u32 vi = tid + OTZ_CNT * (k + j);
				if (do_copy) {
					if (dst_aligned) {
						otz_st16_aligned (dst + 16ull * (vi - pad), v[j]);
					} else {
						otz_st16 (dst + 16ull * (vi - pad), v[j]);
					}
				}
				r ^= otz_crc_chunk16 (0, v[j], L.t4);
			}
		}
	}
	for (; k < iters; k++) {
		u32 vi = tid + OTZ_CNT * k;
		r = otz_crc_adv4k (r, L.a4);
		if (vi >= pad) {
			otz_u4 v = otz_ld16 (src + 16ull * (vi - pad));
			if (do_copy) {
				if (dst_aligned) {
					otz_st16_aligned (dst + 16ull * (vi - pad), v);
				} else {
					otz_st16 (dst + 16ull * (vi - pad), v);
				}
			}
			r ^= otz_crc_chunk16 (0, v, L.t4);
		}
	}
	if (r) {
		r = otz_gf_mul (r, otz_crc_tab.lane16[tid]);
	}
	u32 R = otz_block_xor (r, L.scan);
	if (tail) {
		R = otz_gf_mul (R, otz_crc_tab.xp[0][tail]);
		u32 rt = 0;
		for (u32 i = 0; i < tail; i++) {
			u8 b = src[16ull * nch + i];
			rt = otz_crc_byte (rt, b, L.t4);
			if (do_copy && tid == i) {
				dst[16ull * nch + i] = b;
			}
		}
		R ^= rt;
	}
	return R;
}

// Plain cooperative byte-range copy, any alignment on either side.
OTZ_DEV void otz_copy_bytes(const u8 *src, u8 *dst, u32 n) {
	const u32 tid = otz_tid ();
	u32 head = (u32) ((16u - ((uintptr_t)dst & 15u)) & 15u);
	if (head > n) {
		head = n;
	}
	if (tid < head) {
		dst[tid] = src[tid];
	}
	u32 body = (n - head) >> 4;
	for (u32 c = tid; c < body; c += OTZ_CNT) {
		otz_st16_aligned (dst + head + 16ull * c, otz_ld16 (src + head + 16ull * c));
	}
	u32 done = head + 16u * body;
	if (tid < n - done) {
		dst[done + tid] = src[done + tid];
	}
}

OTZ_DEV void otz_copy_result(ozg_result *res, int status, u32 total_out, u32 crc, u32 expect_crc) {
	if (otz_tid () == 0) {
		res->status = status;
		res->total_out = total_out;
		res->crc32 = status == OZG_ST_OK ? crc : 0u;
		res->flags = (status == OZG_ST_OK && crc != expect_crc) ? OZG_F_CRC_MISMATCH : 0u;
	}
}

// One entry of any copy-shaped method. All OTZ_CNT work-items call together.
OTZ_DEV void otz_copy_entry(otz_copy_lds &L, const u8 *in, u32 n_in, u8 *out, u32 out_cap, u32 method, u32 expect_crc, ozg_result *res) {
	const u32 tid = otz_tid ();
	otz_crc_load_slice4 (L.t4);
	otz_crc_load_adv4k (L.a4);
	otz_sync ();
	if (((uintptr_t)out & 15u) != 0) {
		otz_copy_result (res, OZG_ST_INTERNAL, 0, 0, 0);
		return;
	}
	const u8 *payload = in;
	u32 plen = 0;
	int status = OZG_ST_OK;
	u32 total = 0;

	if (method == OZG_METHOD_STORE) {
		if (n_in != out_cap) {
			status = OZG_ST_SIZE; // ref otezip.c:482-485
		}
		plen = n_in;
	} else if (method == OZG_METHOD_LZFSE) {
		// ref lzfse.inc.c:225-257: 0x06, LE32 length, raw bytes; empty result is an error
		status = OZG_ST_BUF;
		if (n_in >= 5 && out_cap != 0 && in[0] == 0x06) {
			u32 len = otz_rd_le32 (in + 1);
			if ((u32) (len + 5u) <= n_in && (u64)len + 5u <= n_in && len <= out_cap && len != 0) {
				payload = in + 5;
				plen = len;
				status = len == out_cap ? OZG_ST_OK : OZG_ST_SIZE; // ref otezip.c:584
			}
		}
	} else if (method == OZG_METHOD_BROTLI) {
		// ref brotli.inc.c:57-80, :152-180
		status = OZG_ST_DATA;
		bool framed = false;
		u64 stored = 0;
		if (n_in >= 17 && in[0] == 'B' && in[1] == 'R' && in[2] == 'O' && in[3] == 'T') {
			stored = otz_rd_le64 (in + 5);
			framed = stored <= out_cap && (u64)n_in >= 17u + stored;
		}
		if (framed && stored > 0) {
			payload = in + 17;
			plen = (u32)stored;
			status = plen == out_cap ? OZG_ST_OK : OZG_ST_SIZE; // ref otezip.c:654 (after the CRC check below)
		} else if (n_in >= 13 && otz_rd_le64 (in + 5) == 0) {
			// ":158-171": a zero length field ends the stream with no output, whatever the magic says
			plen = 0;
			status = out_cap == 0 ? OZG_ST_OK : OZG_ST_SIZE;
		}
	} else if (method == OZG_METHOD_ZSTD || method == OZG_METHOD_LZMA) {
		plen = 0; // handled below
	} else {
		status = OZG_ST_METHOD; // ref otezip.c:662-665
	}

	if (method == OZG_METHOD_ZSTD) {
		// ref zstd.inc.c:488-705.  Every block is read ONCE: copied to where it lands (its output
		// offset is rarely 16-byte aligned: unaligned 16-byte stores) while its CRC register is
		// taken from the same loads; the blocks' registers are stitched with x^(8 * length).
		status = OZG_ST_OK;
		u32 ip = 0, op = 0;
		u32 reg = 0xFFFFFFFFu; // running CRC register of out[0, op)
		if (n_in < 5) {
			status = OZG_ST_BUF; // :490-492
		} else if (otz_rd_le32 (in) != 0xFD2FB528u) {
			status = OZG_ST_DATA; // :495-498
		} else {
			ip = 5;
			bool ended = false;
			while (status == OZG_ST_OK && !ended) {
				if (ip >= n_in) {
					status = OZG_ST_NEED_OUT - 1000; // input ran dry without a last block: reference returns Z_OK (:704)
					break;
				}
				if (n_in - ip < 3) {
					status = OZG_ST_BUF; // :698-701
					break;
				}
				u32 h = in[ip];
				u32 type = (h >> 1) & 3u;
				u32 bs = (u32)in[ip + 1] | ((u32)in[ip + 2] << 8);
				if (type == 1u || type == 3u) {
					status = OZG_ST_DATA; // :689-692
					break;
				}
				if (n_in - ip - 3u < bs) {
					status = OZG_ST_BUF; // :570-576, :635-641
					break;
				}
				if (type == 2u && bs == 0) {
					status = OZG_ST_DATA; // decompress_block() returns 0, :189-191, :647-649
					break;
				}
				if (out_cap - op < bs) {
					status = OZG_ST_NEED_OUT - 1000; // partial hand-out, Z_OK with avail_out == 0 (:513-548)
					break;
				}
				if (bs) {
					const u32 Rb = otz_stream_crc (L, in + ip + 3u, out + op, bs, true, (op & 15u) == 0);
					reg = otz_gf_mul (reg, otz_xpow_bytes (bs)) ^ Rb;
				}
				op += bs;
				ip += 3u + bs;
				ended = (h & 1u) != 0;
			}
		}
		total = op;
		if (status == OZG_ST_NEED_OUT - 1000) {
			status = OZG_ST_NEED_OUT;
			otz_copy_result (res, status, total, 0, 0);
			return;
		}
		if (status == OZG_ST_OK && op != out_cap) {
			status = OZG_ST_SIZE; // ref otezip.c:555
		}
		if (status != OZG_ST_OK) {
			otz_copy_result (res, status, total, 0, 0);
			return;
		}
		otz_copy_result (res, OZG_ST_OK, total, ~reg, expect_crc);
		return;
	}

	if (method == OZG_METHOD_LZMA) {
		// ref lzma-dec.inc.c:216-308 around simple_lzma_decompress :102-155.
		// The reference stages at most 64 KiB (:181, :193).
		if (n_in < 13) {
			otz_copy_result (res, OZG_ST_BUF, 0, 0, 0);
			return;
		}
		u64 hdr = otz_rd_le64 (in + 5);
		const u8 *src = in + 13;
		const u32 n = n_in - 13;
		u32 sp = 0, dp = 0;
		int st = OZG_ST_OK;
		bool overflow = false;
		while (sp < n && st == OZG_ST_OK) {
			// work-item 0 parses the next batch of tokens
			if (tid == 0) {
				u32 cnt = 0, s = sp, d = dp;
				L.status = 0;
				while (s < n && cnt < OTZ_CNT) {
					if (s + 2 >= n) {
						L.status = 1;
						break;
					}
					u32 m = src[s];
					if (m == 0x00) {
						u32 len = src[s + 2];
						if (d + len > 65536u) {
							L.status = 1;
							break;
						}
						L.tk_kind[cnt] = 0;
						L.tk_byte[cnt] = src[s + 1];
						L.tk_len[cnt] = (u16)len;
						L.tk_dst[cnt] = d;
						L.tk_src[cnt] = 0;
						s += 3;
						d += len;
					} else if (m == 0x01) {
						u32 len = src[s + 1];
						if (s + 2 + len > n || d + len > 65536u) {
							L.status = 1;
							break;
						}
						L.tk_kind[cnt] = 1;
						L.tk_len[cnt] = (u16)len;
						L.tk_dst[cnt] = d;
						L.tk_src[cnt] = s + 2;
						s += 2 + len;
						d += len;
					} else {
						L.status = 1;
						break;
					}
					cnt++;
				}
				L.n_tok = cnt;
				L.tok_src_next = s;
				L.tok_dst_next = d;
			}
			otz_sync ();
			if (L.status != 0) {
				st = OZG_ST_DATA; // ref :271-273
			}
			u32 cnt = L.n_tok;
			if (st == OZG_ST_OK && tid < cnt) {
				u32 d = L.tk_dst[tid], len = L.tk_len[tid];
				// bytes beyond the caller's buffer are dropped: the reference hands out
				// what fits and returns Z_OK (:286-295)
				for (u32 i = 0; i < len; i++) {
					if (d + i < out_cap) {
						out[d + i] = L.tk_kind[tid] ? src[L.tk_src[tid] + i] : L.tk_byte[tid];
					}
				}
			}
			sp = L.tok_src_next;
			dp = L.tok_dst_next;
			otz_sync ();
		}
		if (st == OZG_ST_OK && n > 0 && dp == 0) {
			st = OZG_ST_DATA; // decomp_size == 0, ref :271-273
		}
		if (st == OZG_ST_OK && dp > out_cap) {
			overflow = true;
		}
		if (st != OZG_ST_OK) {
			otz_copy_result (res, st, 0, 0, 0);
			return;
		}
		if (overflow) {
			otz_copy_result (res, OZG_ST_NEED_OUT, out_cap, 0, 0);
			return;
		}
		// ref :297-310
		if (hdr != 0xFFFFFFFFFFFFFFFFull && (u64)dp < hdr) {
			otz_copy_result (res, OZG_ST_NEED_OUT, dp, 0, 0); // Z_OK: wants more input
			return;
		}
		if (dp != out_cap) {
			otz_copy_result (res, OZG_ST_SIZE, dp, 0, 0); // ref otezip.c:625
			return;
		}
		otz_fence_release ();
		otz_sync ();
		otz_fence_acquire ();
		u32 R = otz_stream_crc (L, out, out, dp, false);
		u32 crc = ~(otz_gf_mul (0xFFFFFFFFu, otz_xpow_bytes (dp)) ^ R);
		otz_copy_result (res, OZG_ST_OK, dp, crc, expect_crc);
		return;
	}

	// single contiguous payload: STORE / lzfse / BROT
	if (status != OZG_ST_OK && status != OZG_ST_SIZE) {
		otz_copy_result (res, status, 0, 0, 0);
		return;
	}
	if (method == OZG_METHOD_BROTLI && plen > 0) {
		// the container's own CRC must hold before anything is handed out (:71-75);
		// it covers the same bytes as the ZIP CRC, so one pass serves both
		u32 want = otz_rd_le32 (in + 13);
		bool copy = status == OZG_ST_OK;
		u32 R = otz_stream_crc (L, payload, out, plen, copy);
		u32 crc = ~(otz_gf_mul (0xFFFFFFFFu, otz_xpow_bytes (plen)) ^ R);
		if (crc != want) {
			otz_copy_result (res, OZG_ST_DATA, 0, 0, 0);
			return;
		}
		otz_copy_result (res, status, plen, crc, expect_crc);
		return;
	}
	if (status == OZG_ST_SIZE) {
		otz_copy_result (res, status, plen, 0, 0);
		return;
	}
	u32 R = otz_stream_crc (L, payload, out, plen, true);
	u32 crc = ~(otz_gf_mul (0xFFFFFFFFu, otz_xpow_bytes (plen)) ^ R);
	otz_copy_result (res, OZG_ST_OK, plen, crc, expect_crc);
}
