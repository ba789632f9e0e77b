// otz_lzfse.h -- LZFSE streams (Apple's format: blocks bvx- / bvxn / bvx1 / bvx2 / bvx$, LZVN opcodes, the FSE
// coder of LZFSE) under ZIP method 100: ONE WORK-ITEM per entry.
//
// The reference's method 100 is a private framing -- 06, a length, the bytes stored (ref src/lib/lzfse.inc.c:225-257);
// that stays with otz_copy_kernel.  This file is the real format -- SURVEY.md section 8, row f4, and the
// north star's "LZFSE LZVN back-copy" -- which other writers put under the same method id.  Sniffing: a payload
// that starts with "bvx" is tried here first, and what does not decode to exactly uncomp_size bytes is handed to
// the private decoder after all (route[], as for Zstandard, LZMA1 and Brotli).
//
// PARITY UNPINNED.  Nothing on the image implements LZFSE (no library, no Python module, no test vectors), the
// reference does not either, and nothing can be fetched.  This decoder and the Python model it is tested against
// (tools/lzfse_model.py: an encoder and a decoder) were written separately from the same reading of the format;
// that they agree on every stream the model writes, and on damaged ones, shows that the kernel does what the
// model says -- not that the model is Apple's library.  In a ZIP the entry's CRC-32 still decides whether the
// bytes are right.
//
// Shape of the work: as otz_brotli.h.  A block's FSE tables (7 KB) and its literals (<= 40 000 bytes, decoded
// first, in four interleaved states) live in OTZ_LZFSE_ARENA bytes of global scratch per work-item; the entry's
// own output is the window; both payloads of a block are read backwards through a 64-bit accumulator.
#pragma once

#include "otz_simt.h"
#include "../../include/otezip_gpu.h"

typedef int8_t otz_i8;
typedef int16_t otz_i16;
typedef int32_t otz_i32;
typedef int64_t otz_i64;

#define OTZ_LF_LIT_STATES 1024u
#define OTZ_LF_L_STATES 64u
#define OTZ_LF_M_STATES 64u
#define OTZ_LF_D_STATES 256u
#define OTZ_LF_MATCHES 10000u
#define OTZ_LF_LITERALS (4u * OTZ_LF_MATCHES)
// literal table 4 KB, L / M / D tables 0.5 + 0.5 + 2 KB, the block's literals, the 360 frequencies
#define OTZ_LZFSE_ARENA (4096u + 512u + 512u + 2048u + OTZ_LF_LITERALS + 64u + 720u)

struct otz_lf_lit { // literal decoder entry
	otz_i8 k;
	u8 symbol;
	otz_i16 delta;
};
struct otz_lf_val { // L / M / D decoder entry
	u8 total_bits, value_bits;
	otz_i16 delta;
	otz_i32 vbase;
};

// ------------------------------------------------------------------ the backward bit reader (FSE input stream)

struct otz_lf_in {
	const u8 *base; // the whole stream: reads may reach in front of a payload, never in front of this
	otz_i64 pos;
	u64 accum;
	otz_i32 nbits;
	u32 bad;
};

OTZ_DEV u64 otz_lf_ld64(const u8 *p, u32 n) { // n bytes (<= 8), little endian
	u64 v = 0;
	for (u32 i = 0; i < n; i++) {
		v |= (u64)p[i] << (8u * i);
	}
	return v;
}

// end: stream offset one past the payload's last byte; bits in [-7, 0]
OTZ_DEV bool otz_lf_in_init(otz_lf_in &s, const u8 *base, u32 end, otz_i32 bits) {
	const u32 take = bits ? 8u : 7u;
	s.base = base;
	s.bad = 0;
	if (end < take || bits < -7 || bits > 0) {
		return false;
	}
	s.pos = (otz_i64)end - take;
	s.accum = otz_lf_ld64 (base + s.pos, take);
	s.nbits = bits ? bits + 64 : 56;
	return s.nbits >= 56 && s.nbits < 64 && (s.accum >> s.nbits) == 0;
}
OTZ_DEV void otz_lf_flush(otz_lf_in &s) {
	const otz_i32 n = (63 - s.nbits) & -8;
	s.pos -= n >> 3;
	if (s.pos < 0) {
		s.bad = 1;
		s.pos = 0;
	}
	if (n) { // (8 bytes follow the new position: it moved down by at least one from end - 7 or end - 8)
		const u64 incoming = otz_lf_ld64 (s.base + s.pos, 8);
		s.accum = (s.accum << n) | (incoming & (((u64)1 << n) - 1u));
	}
	s.nbits += n;
}
OTZ_DEV u32 otz_lf_pull(otz_lf_in &s, u32 n) { // n <= 32
	if ((otz_i32)n > s.nbits) {
		s.bad = 1;
		return 0;
	}
	s.nbits -= (otz_i32)n;
	const u64 v = s.accum >> s.nbits;
	s.accum &= ((u64)1 << s.nbits) - 1u;
	return (u32)v;
}

// ------------------------------------------------------------------ decoder tables from frequencies

// literal decoder: returns the number of table entries, or 0xFFFFFFFF when the frequencies exceed the states
OTZ_DEV u32 otz_lf_lit_table(const u16 *freq, otz_lf_lit *t) {
	const u32 n_clz = (u32)__builtin_clz (OTZ_LF_LIT_STATES);
	u32 at = 0;
	for (u32 sym = 0; sym < 256u; sym++) {
		const u32 f = freq[sym];
		if (f == 0) {
			continue;
		}
		if (at + f > OTZ_LF_LIT_STATES) {
			return 0xFFFFFFFFu;
		}
		const u32 k = (u32)__builtin_clz (f) - n_clz;
		const u32 j0 = ((2u * OTZ_LF_LIT_STATES) >> k) - f;
		for (u32 j = 0; j < f; j++) {
			otz_lf_lit e;
			e.symbol = (u8)sym;
			if (j < j0) {
				e.k = (otz_i8)k;
				e.delta = (otz_i16) (((f + j) << k) - OTZ_LF_LIT_STATES);
			} else {
				e.k = (otz_i8) (k - 1u);
				e.delta = (otz_i16) ((j - j0) << (k - 1u));
			}
			t[at++] = e;
		}
	}
	return at;
}

OTZ_DEV u32 otz_lf_val_table(u32 nstates, u32 nsym, const u16 *freq, const u8 *extra, const otz_i32 *base, otz_lf_val *t) {
	const u32 n_clz = (u32)__builtin_clz (nstates);
	u32 at = 0;
	for (u32 sym = 0; sym < nsym; sym++) {
		const u32 f = freq[sym];
		if (f == 0) {
			continue;
		}
		if (at + f > nstates) {
			return 0xFFFFFFFFu;
		}
		const u32 k = (u32)__builtin_clz (f) - n_clz;
		const u32 j0 = ((2u * nstates) >> k) - f;
		for (u32 j = 0; j < f; j++) {
			otz_lf_val e;
			e.value_bits = extra[sym];
			e.vbase = base[sym];
			if (j < j0) {
				e.total_bits = (u8) (k + extra[sym]);
				e.delta = (otz_i16) (((f + j) << k) - nstates);
			} else {
				e.total_bits = (u8) (k - 1u + extra[sym]);
				e.delta = (otz_i16) ((j - j0) << (k - 1u));
			}
			t[at++] = e;
		}
	}
	return at;
}

// ------------------------------------------------------------------ the block kinds

struct otz_lf_hdr {
	u32 n_raw, n_literals, n_matches, lit_bytes, lmd_bytes;
	otz_i32 lit_bits, lmd_bits;
	u32 lit_state[4], l_state, m_state, d_state;
};

OTZ_DEV u32 otz_lf_le32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
OTZ_DEV u64 otz_lf_le64(const u8 *p) { return (u64)otz_lf_le32 (p) | ((u64)otz_lf_le32 (p + 4) << 32); }

// one compressed block: its header fields in H, its frequencies in freq[360], its payloads at in + body
OTZ_DEV int otz_lf_block(const u8 *in, u32 n_in, u32 body, const otz_lf_hdr &H, const u16 *freq, u8 *arena, u8 *out, u32 cap, u32 &pos) {
	const u8 l_extra[20] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 5, 8 };
	const otz_i32 l_base[20] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 20, 28, 60 };
	const u8 m_extra[20] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 5, 8, 11 };
	const otz_i32 m_base[20] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 24, 56, 312 };
	u8 d_extra[64];
	otz_i32 d_base[64];
	{
		otz_i32 b = 0;
		for (u32 k = 0; k < 64u; k++) {
			d_extra[k] = (u8) (k >> 2);
			d_base[k] = b;
			b += (otz_i32)1 << (k >> 2);
		}
	}
	otz_lf_lit *lit_t = reinterpret_cast<otz_lf_lit *> (arena);
	otz_lf_val *l_t = reinterpret_cast<otz_lf_val *> (arena + 4096u);
	otz_lf_val *m_t = reinterpret_cast<otz_lf_val *> (arena + 4096u + 512u);
	otz_lf_val *d_t = reinterpret_cast<otz_lf_val *> (arena + 4096u + 1024u);
	u8 *lits = arena + 4096u + 1024u + 2048u;
	if (H.n_literals > OTZ_LF_LITERALS || (H.n_literals & 3u) || H.n_matches > OTZ_LF_MATCHES ||
		(u64)body + H.lit_bytes + H.lmd_bytes > n_in) {
		return OZG_ST_DATA;
	}
	const u32 n_lit_t = otz_lf_lit_table (freq + 104, lit_t);
	const u32 n_l = otz_lf_val_table (OTZ_LF_L_STATES, 20, freq, l_extra, l_base, l_t);
	const u32 n_m = otz_lf_val_table (OTZ_LF_M_STATES, 20, freq + 20, m_extra, m_base, m_t);
	const u32 n_d = otz_lf_val_table (OTZ_LF_D_STATES, 64, freq + 40, d_extra, d_base, d_t);
	if (n_lit_t == 0xFFFFFFFFu || n_l == 0xFFFFFFFFu || n_m == 0xFFFFFFFFu || n_d == 0xFFFFFFFFu) {
		return OZG_ST_DATA;
	}
	// ---- the literals: four interleaved states
	if (H.n_literals) {
		otz_lf_in s;
		if (!otz_lf_in_init (s, in, body + H.lit_bytes, H.lit_bits)) {
			return OZG_ST_DATA;
		}
		u32 st0 = H.lit_state[0], st1 = H.lit_state[1], st2 = H.lit_state[2], st3 = H.lit_state[3];
		for (u32 i = 0; i < H.n_literals; i += 4u) {
			otz_lf_flush (s);
			if (st0 >= n_lit_t || st1 >= n_lit_t || st2 >= n_lit_t || st3 >= n_lit_t) {
				return OZG_ST_DATA;
			}
			const otz_lf_lit e0 = lit_t[st0], e1 = lit_t[st1], e2 = lit_t[st2], e3 = lit_t[st3];
			st0 = (u32) ((otz_i32)e0.delta + (otz_i32)otz_lf_pull (s, (u32)e0.k));
			st1 = (u32) ((otz_i32)e1.delta + (otz_i32)otz_lf_pull (s, (u32)e1.k));
			st2 = (u32) ((otz_i32)e2.delta + (otz_i32)otz_lf_pull (s, (u32)e2.k));
			st3 = (u32) ((otz_i32)e3.delta + (otz_i32)otz_lf_pull (s, (u32)e3.k));
			lits[i] = e0.symbol;
			lits[i + 1u] = e1.symbol;
			lits[i + 2u] = e2.symbol;
			lits[i + 3u] = e3.symbol;
			if (s.bad) {
				return OZG_ST_DATA;
			}
		}
	}
	// ---- the matches
	const u32 start = pos;
	if (H.n_matches) {
		otz_lf_in s;
		if (!otz_lf_in_init (s, in, body + H.lit_bytes + H.lmd_bytes, H.lmd_bits)) {
			return OZG_ST_DATA;
		}
		u32 ls = H.l_state, ms = H.m_state, ds = H.d_state;
		u32 lit_at = 0, D = 0;
		for (u32 k = 0; k < H.n_matches; k++) {
			if (ls >= n_l || ms >= n_m || ds >= n_d) {
				return OZG_ST_DATA;
			}
			otz_lf_flush (s);
			const otz_lf_val el = l_t[ls];
			u32 sv = otz_lf_pull (s, el.total_bits);
			ls = (u32) ((otz_i32)el.delta + (otz_i32) (sv >> el.value_bits));
			const u32 L = (u32)el.vbase + (sv & ((1u << el.value_bits) - 1u));
			otz_lf_flush (s);
			const otz_lf_val em = m_t[ms];
			sv = otz_lf_pull (s, em.total_bits);
			ms = (u32) ((otz_i32)em.delta + (otz_i32) (sv >> em.value_bits));
			const u32 M = (u32)em.vbase + (sv & ((1u << em.value_bits) - 1u));
			otz_lf_flush (s);
			const otz_lf_val ed = d_t[ds];
			sv = otz_lf_pull (s, ed.total_bits);
			ds = (u32) ((otz_i32)ed.delta + (otz_i32) (sv >> ed.value_bits));
			const u32 nd = (u32)ed.vbase + (sv & ((1u << ed.value_bits) - 1u));
			D = nd ? nd : D;
			if (s.bad || lit_at + L > H.n_literals) {
				return OZG_ST_DATA;
			}
			if ((u64)(pos - start) + L + M > H.n_raw) {
				return OZG_ST_DATA; // the block is longer than it says
			}
			if (L + M > cap - pos) {
				return OZG_ST_SIZE;
			}
			for (u32 i = 0; i < L; i++) {
				out[pos++] = lits[lit_at + i];
			}
			lit_at += L;
			if (M) {
				if (D == 0 || D > pos) {
					return OZG_ST_DATA;
				}
				for (u32 i = 0; i < M; i++) { // (byte by byte: the copy may overlap itself)
					out[pos] = out[pos - D];
					pos++;
				}
			}
		}
	}
	return pos - start == H.n_raw ? OZG_ST_OK : OZG_ST_DATA;
}

// LZVN opcodes at in[at, at + n_pay); returns a status, *used = bytes consumed (the end-of-stream opcode included)
OTZ_DEV int otz_lf_lzvn(const u8 *src, u32 n_pay, u32 n_raw, u8 *out, u32 cap, u32 &pos, u32 *used) {
	u32 at = 0, D = 0;
	const u32 start = pos;
	for (;;) {
		if (at >= n_pay) {
			return OZG_ST_DATA;
		}
		const u32 opc = src[at];
		u32 L = 0, M = 0, len = 1;
		if (opc == 0x06u) {
			if (at + 8u > n_pay) {
				return OZG_ST_DATA;
			}
			at += 8u;
			break;
		}
		if (opc == 0x0Eu || opc == 0x16u) {
			at += 1u;
			continue;
		}
		if (opc == 0x1Eu || opc == 0x26u || opc == 0x2Eu || opc == 0x36u || opc == 0x3Eu || (opc >= 0x70u && opc <= 0x7Fu) ||
			(opc >= 0xD0u && opc <= 0xDFu)) {
			return OZG_ST_DATA; // undefined opcodes
		}
		if (opc >= 0xE0u) {
			// literals (E.) or a match with the previous distance (F.): a long form with a count byte
			const bool lng = (opc & 15u) == 0;
			if (lng && at + 2u > n_pay) {
				return OZG_ST_DATA;
			}
			const u32 n = lng ? (u32)src[at + 1u] + 16u : opc & 15u;
			len = lng ? 2u : 1u;
			if (opc >= 0xF0u) {
				M = n;
			} else {
				L = n;
			}
		} else if (opc >= 0xA0u && opc <= 0xBFu) {
			if (at + 3u > n_pay) {
				return OZG_ST_DATA;
			}
			L = (opc >> 3) & 3u;
			M = (((opc & 7u) << 2) | (src[at + 1u] & 3u)) + 3u;
			D = ((u32)src[at + 1u] >> 2) | ((u32)src[at + 2u] << 6);
			len = 3;
		} else {
			L = opc >> 6;
			M = ((opc >> 3) & 7u) + 3u;
			const u32 low = opc & 7u;
			if (low == 7u) {
				if (at + 3u > n_pay) {
					return OZG_ST_DATA;
				}
				D = (u32)src[at + 1u] | ((u32)src[at + 2u] << 8);
				len = 3;
			} else if (low != 6u) {
				if (at + 2u > n_pay) {
					return OZG_ST_DATA;
				}
				D = (low << 8) | src[at + 1u];
				len = 2;
			}
		}
		at += len;
		if (at + L > n_pay) {
			return OZG_ST_DATA;
		}
		if ((u64)(pos - start) + L + M > n_raw) {
			return OZG_ST_DATA;
		}
		if (L + M > cap - pos) {
			return OZG_ST_SIZE;
		}
		for (u32 i = 0; i < L; i++) {
			out[pos++] = src[at + i];
		}
		at += L;
		if (M) {
			if (D == 0 || D > pos) {
				return OZG_ST_DATA;
			}
			for (u32 i = 0; i < M; i++) {
				out[pos] = out[pos - D];
				pos++;
			}
		}
	}
	*used = at;
	return pos - start == n_raw ? OZG_ST_OK : OZG_ST_DATA;
}

// the 360 frequencies behind a v2 header (a variable-length code, 2..14 bits each)
OTZ_DEV bool otz_lf_freq_v2(const u8 *src, u32 n, u16 *freq) {
	const u8 nbits_t[32] = { 2, 3, 2, 5, 2, 3, 2, 8, 2, 3, 2, 5, 2, 3, 2, 14, 2, 3, 2, 5, 2, 3, 2, 8, 2, 3, 2, 5, 2, 3, 2, 14 };
	const otz_i8 value_t[32] = { 0, 2, 1, 4, 0, 3, 1, -1, 0, 2, 1, 5, 0, 3, 1, -1, 0, 2, 1, 6, 0, 3, 1, -1, 0, 2, 1, 7, 0, 3, 1, -1 };
	u32 accum = 0, nb = 0, at = 0;
	for (u32 i = 0; i < 360u; i++) {
		while (at < n && nb + 8u <= 32u) {
			accum |= (u32)src[at++] << nb;
			nb += 8u;
		}
		const u32 b = accum & 31u;
		const u32 k = nbits_t[b];
		u32 v;
		if (k == 8u) {
			v = 8u + ((accum >> 4) & 0xFu);
		} else if (k == 14u) {
			v = 24u + ((accum >> 4) & 0x3FFu);
		} else {
			v = (u32)value_t[b];
		}
		if (k > nb) {
			return false;
		}
		freq[i] = (u16)v;
		accum >>= k;
		nb -= k;
	}
	return true;
}

// ------------------------------------------------------------------ the stream

// One LZFSE stream, one work-item.  arena: OTZ_LZFSE_ARENA bytes of scratch (4-byte aligned).
OTZ_DEV int otz_lzfse_stream(const u8 *in, u32 n_in, u8 *out, u32 cap, u8 *arena, u32 *produced) {
	u32 at = 0, pos = 0;
	u16 *freq = reinterpret_cast<u16 *> (arena + OTZ_LZFSE_ARENA - 720u);
	*produced = 0;
	for (;;) {
		if (at + 4u > n_in || in[at] != 'b' || in[at + 1u] != 'v' || in[at + 2u] != 'x') {
			return OZG_ST_DATA;
		}
		const u32 kind = in[at + 3u];
		if (kind == '$') {
			*produced = pos;
			return OZG_ST_OK;
		}
		if (kind == '-') {
			if (at + 8u > n_in) {
				return OZG_ST_DATA;
			}
			const u32 n = otz_lf_le32 (in + at + 4u);
			if ((u64)at + 8u + n > n_in) {
				return OZG_ST_DATA;
			}
			if (n > cap - pos) {
				return OZG_ST_SIZE;
			}
			for (u32 i = 0; i < n; i++) {
				out[pos++] = in[at + 8u + i];
			}
			at += 8u + n;
		} else if (kind == 'n') {
			if (at + 12u > n_in) {
				return OZG_ST_DATA;
			}
			const u32 n_raw = otz_lf_le32 (in + at + 4u), n_pay = otz_lf_le32 (in + at + 8u);
			if ((u64)at + 12u + n_pay > n_in) {
				return OZG_ST_DATA;
			}
			u32 used = 0;
			const int st = otz_lf_lzvn (in + at + 12u, n_pay, n_raw, out, cap, pos, &used);
			if (st != OZG_ST_OK) {
				return st;
			}
			if (used != n_pay) {
				return OZG_ST_DATA;
			}
			at += 12u + n_pay;
		} else if (kind == '2') {
			if (at + 32u > n_in) {
				return OZG_ST_DATA;
			}
			otz_lf_hdr H;
			H.n_raw = otz_lf_le32 (in + at + 4u);
			const u64 v0 = otz_lf_le64 (in + at + 8u), v1 = otz_lf_le64 (in + at + 16u), v2 = otz_lf_le64 (in + at + 24u);
			H.n_literals = (u32) (v0 & 0xFFFFFu);
			H.lit_bytes = (u32) ((v0 >> 20) & 0xFFFFFu);
			H.n_matches = (u32) ((v0 >> 40) & 0xFFFFFu);
			H.lit_bits = (otz_i32) ((v0 >> 60) & 7u) - 7;
			for (u32 k = 0; k < 4u; k++) {
				H.lit_state[k] = (u32) ((v1 >> (10u * k)) & 0x3FFu);
			}
			H.lmd_bytes = (u32) ((v1 >> 40) & 0xFFFFFu);
			H.lmd_bits = (otz_i32) ((v1 >> 60) & 7u) - 7;
			const u32 hsize = (u32) (v2 & 0xFFFFFFFFu);
			H.l_state = (u32) ((v2 >> 32) & 0x3FFu);
			H.m_state = (u32) ((v2 >> 42) & 0x3FFu);
			H.d_state = (u32) ((v2 >> 52) & 0x3FFu);
			if (hsize < 32u || (u64)at + hsize > n_in) {
				return OZG_ST_DATA;
			}
			if (hsize > 32u) {
				if (!otz_lf_freq_v2 (in + at + 32u, hsize - 32u, freq)) {
					return OZG_ST_DATA;
				}
			} else {
				for (u32 i = 0; i < 360u; i++) {
					freq[i] = 0;
				}
			}
			const int st = otz_lf_block (in, n_in, at + hsize, H, freq, arena, out, cap, pos);
			if (st != OZG_ST_OK) {
				return st;
			}
			at += hsize + H.lit_bytes + H.lmd_bytes;
		} else if (kind == '1') {
			const u32 size = 4u * 7u + 4u + 8u + 4u + 6u + 720u; // the v1 header: every field and the four tables in full
			if ((u64)at + size > n_in) {
				return OZG_ST_DATA;
			}
			otz_lf_hdr H;
			const u8 *h = in + at;
			H.n_raw = otz_lf_le32 (h + 4);
			H.n_literals = otz_lf_le32 (h + 12);
			H.n_matches = otz_lf_le32 (h + 16);
			H.lit_bytes = otz_lf_le32 (h + 20);
			H.lmd_bytes = otz_lf_le32 (h + 24);
			H.lit_bits = (otz_i32)otz_lf_le32 (h + 28);
			for (u32 k = 0; k < 4u; k++) {
				H.lit_state[k] = (u32)h[32u + 2u * k] | ((u32)h[33u + 2u * k] << 8);
			}
			H.lmd_bits = (otz_i32)otz_lf_le32 (h + 40);
			H.l_state = (u32)h[44] | ((u32)h[45] << 8);
			H.m_state = (u32)h[46] | ((u32)h[47] << 8);
			H.d_state = (u32)h[48] | ((u32)h[49] << 8);
			for (u32 i = 0; i < 360u; i++) {
				freq[i] = (u16) ((u32)h[50u + 2u * i] | ((u32)h[51u + 2u * i] << 8));
			}
			const int st = otz_lf_block (in, n_in, at + size, H, freq, arena, out, cap, pos);
			if (st != OZG_ST_OK) {
				return st;
			}
			at += size + H.lit_bytes + H.lmd_bytes;
		} else {
			return OZG_ST_DATA;
		}
	}
}

// does a method-100 payload go to this decoder first?  (see the head of this file)
OTZ_DEV bool otz_lzfse_is_candidate(const u8 *in, u32 n_in) {
	return n_in >= 4u && in[0] == 'b' && in[1] == 'v' && in[2] == 'x';
}

// One method-100 entry (one work-item).
OTZ_DEV void otz_lzfse_run(const u8 *in, u32 n_in, u8 *out, u32 cap, u32 eflags, u8 *arena, ozg_result *res, u8 *route) {
	u32 total = 0;
	const int st = otz_lzfse_stream (in, n_in, out, cap, arena, &total);
	if (st == OZG_ST_OK && (total == cap || (eflags & OZG_EF_SIZE_IS_CAP))) {
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
		*route = 1u; // OTZ_ZS_ROUTE_PRIVATE: the reference's framing decoder has the last word
	}
}
