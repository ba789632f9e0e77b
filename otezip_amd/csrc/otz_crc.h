// otz_crc.h -- CRC-32 (reflected 0xEDB88320, init/final xor 0xFFFFFFFF) for the
// codec kernels.  Replaces the byte loop of otezip_crc32
// (ref src/lib/crc32.inc.c:40-47), which the reference runs over every decoded
// entry (ref src/lib/otezip.c:668) and every added entry (:1124).
//
// A CRC register is a polynomial over GF(2) modulo P with x^0 in bit 31, so
//   state_after(M) = state_before * x^(8|M|)  xor  R(M)
// where R(M) is the register reached from zero.  That linearity is what lets a
// workgroup cut a buffer into per-lane pieces, run slice-by-4 on each piece and
// stitch the pieces together with one modular multiplication per lane.
// gfx950 has no carry-less multiply, so products are 32 shift/xor steps; they
// are needed once per lane per flush, not per byte.
#pragma once

#include "otz_simt.h"

#define OTZ_CRC_POLY 0xEDB88320u

struct otz_crc_tables {
	u32 slice[4][256]; // slice-by-4 data tables
	u32 xp[4][256];    // xp[k][j] = x^(8 * j * 256^k) mod P
	u32 adv4k[4][256]; // multiply a register by x^(8*4096), one byte lane at a time
	u32 lane16[256];   // lane16[t] = x^(8*16*(255-t)): aligns 256 interleaved 16-byte streams at their end
};

namespace otz_crc_detail {
constexpr u32 step_bit(u32 c) { return (c & 1u) ? (c >> 1) ^ OTZ_CRC_POLY : (c >> 1); }
constexpr u32 step_byte0(u32 c) {
	for (int k = 0; k < 8; k++) {
		c = step_bit (c);
	}
	return c;
}
constexpr u32 mul(u32 a, u32 b) {
	u32 p = 0;
	for (int i = 0; i < 32; i++) {
		if ((a >> (31 - i)) & 1u) {
			p ^= b;
		}
		b = step_bit (b);
	}
	return p;
}
constexpr otz_crc_tables make() {
	otz_crc_tables t{};
	for (u32 n = 0; n < 256; n++) {
		t.slice[0][n] = step_byte0 (n);
	}
	for (int k = 1; k < 4; k++) {
		for (u32 n = 0; n < 256; n++) {
			u32 c = t.slice[k - 1][n];
			t.slice[k][n] = t.slice[0][c & 0xFF] ^ (c >> 8);
		}
	}
	u32 x8 = step_byte0 (0x80000000u); // x^8
	u32 base = x8;
	for (int k = 0; k < 4; k++) {
		t.xp[k][0] = 0x80000000u; // 1
		for (u32 j = 1; j < 256; j++) {
			t.xp[k][j] = mul (t.xp[k][j - 1], base);
		}
		base = mul (t.xp[k][255], base); // x^(8 * 256^(k+1))
	}
	u32 a4k = t.xp[1][16]; // x^(8*4096)
	for (int k = 0; k < 4; k++) {
		for (u32 n = 0; n < 256; n++) {
			t.adv4k[k][n] = mul (n << (8 * k), a4k);
		}
	}
	for (u32 i = 0; i < 256; i++) {
		u32 bytes = 16u * (255u - i);
		t.lane16[i] = mul (t.xp[0][bytes & 255u], t.xp[1][bytes >> 8]);
	}
	return t;
}
} // namespace otz_crc_detail

OTZ_CONST_TABLE otz_crc_tables otz_crc_tab = otz_crc_detail::make ();

// a * b mod P
OTZ_DEV u32 otz_gf_mul(u32 a, u32 b) {
	u32 p = 0;
#pragma unroll 8
	for (int i = 0; i < 32; i++) {
		p ^= b & (0u - ((a >> 31) & 1u));
		a <<= 1;
		b = (b >> 1) ^ (OTZ_CRC_POLY & (0u - (b & 1u)));
	}
	return p;
}

// x^(8n) mod P from the global tables (3 products)
OTZ_DEV u32 otz_xpow_bytes(u32 n) {
	u32 lo = otz_gf_mul (otz_crc_tab.xp[0][n & 255u], otz_crc_tab.xp[1][(n >> 8) & 255u]);
	if (n < 65536u) {
		return lo;
	}
	u32 hi = otz_gf_mul (otz_crc_tab.xp[2][(n >> 16) & 255u], otz_crc_tab.xp[3][n >> 24]);
	return otz_gf_mul (lo, hi);
}

// Copy the slice-by-4 tables (4 KiB) into LDS at t4[1024]; caller syncs.
OTZ_DEV void otz_crc_load_slice4(u32 *t4) {
	const u32 *g = &otz_crc_tab.slice[0][0];
	for (u32 i = otz_tid (); i < 1024u; i += otz_nthreads ()) {
		t4[i] = g[i];
	}
}
OTZ_DEV void otz_crc_load_adv4k(u32 *t4) {
	const u32 *g = &otz_crc_tab.adv4k[0][0];
	for (u32 i = otz_tid (); i < 1024u; i += otz_nthreads ()) {
		t4[i] = g[i];
	}
}

// one little-endian 32-bit word through the register (tables in LDS)
OTZ_DEV u32 otz_crc_word(u32 c, u32 w, const u32 *t4) {
	c ^= w;
	return t4[768 + (c & 0xFF)] ^ t4[512 + ((c >> 8) & 0xFF)] ^ t4[256 + ((c >> 16) & 0xFF)] ^ t4[c >> 24];
}
OTZ_DEV u32 otz_crc_byte(u32 c, u32 b, const u32 *t4) { return t4[(c ^ b) & 0xFF] ^ (c >> 8); }
OTZ_DEV u32 otz_crc_chunk16(u32 c, otz_u4 v, const u32 *t4) {
	c = otz_crc_word (c, v.x, t4);
	c = otz_crc_word (c, v.y, t4);
	c = otz_crc_word (c, v.z, t4);
	c = otz_crc_word (c, v.w, t4);
	return c;
}
// register * x^(8*4096) through the byte-lane tables in LDS
OTZ_DEV u32 otz_crc_adv4k(u32 c, const u32 *a4) {
	return a4[c & 0xFF] ^ a4[256 + ((c >> 8) & 0xFF)] ^ a4[512 + ((c >> 16) & 0xFF)] ^ a4[768 + (c >> 24)];
}

// xor-reduce one value per work-item over the workgroup; scratch = 16 u32 in LDS.
// Every work-item gets the result.  Uniform control flow required.
OTZ_DEV u32 otz_block_xor(u32 v, u32 *scratch) {
#pragma unroll
	for (u32 m = 32; m >= 1; m >>= 1) {
		v ^= otz_shfl_xor (v, m);
	}
	otz_sync ();
	if (otz_lane () == 0) {
		scratch[otz_wave ()] = v;
	}
	otz_sync ();
	u32 r = 0;
	u32 nw = (otz_nthreads () + 63u) >> 6;
	for (u32 i = 0; i < nw; i++) {
		r ^= scratch[i];
	}
	return r;
}

// Adler-32 (RFC 1950) of p[0, n) by a workgroup of NT work-items (NT <= 256); every work-item gets it.
// A = 1 + sum d_i, B = n + sum (n - i) d_i, both mod 65521: each work-item sums a contiguous slice in 64-bit
// registers (no reduction needed inside a slice of < 2^24 bytes: 255 * m^2 / 2 < 2^63), the slices are
// stitched with B += a_t * (bytes behind slice t).  scratch: 2 * NT u64 in LDS.
OTZ_DEV u32 otz_adler32_block(const u8 *p, u32 n, u64 *scratch) {
	const u32 tid = otz_tid (), nt = otz_nthreads ();
	const u32 M = 65521u;
	const u32 per = (n + nt - 1u) / nt;
	const u32 i0 = (u64)tid * per < n ? tid * per : n, i1 = i0 + per < n ? i0 + per : n;
	u64 a = 0, b = 0;
	u32 i = i0;
	for (; i < i1; i++) {
		const u64 d = p[i];
		a += d;
		b += d * (u64) (i1 - i);
		if (((i - i0) & 0xFFFFFu) == 0xFFFFFu) { // (slices longer than a megabyte: keep the sums small)
			a %= M;
			b %= M;
		}
	}
	a %= M;
	b %= M;
	b = (b + a * ((u64) (n - i1) % M)) % M;
	scratch[tid] = a;
	scratch[nt + tid] = b;
	otz_sync ();
	u64 A = 1, B = (u64)n % M;
	for (u32 t = 0; t < nt; t++) {
		A += scratch[t];
		B += scratch[nt + t];
	}
	otz_sync ();
	return (u32) (A % M) | ((u32) (B % M) << 16);
}
