// otz_brotli.h -- Brotli streams (RFC 7932) under ZIP method 97: ONE WORK-ITEM per entry.
//
// The reference's method 97 is a private container -- "BROT" 01, a length, a CRC, the bytes stored
// (ref src/lib/brotli.inc.c:57-80, :152-180); that stays with otz_copy_kernel.  This file is the real
// format -- SURVEY.md section 8, row f4 -- which other writers put under the same method id.
// Sniffing: a payload that starts with "BROT" is the reference's; anything else is tried here first, and
// what does not decode to exactly uncomp_size bytes is handed to the private decoder after all (route[],
// as for Zstandard and LZMA), so every input the reference rejects is still rejected by its own rules.
// Checked against libbrotlidec 1.0.9 of the image (tests/test_brotli_real.py, tests/golden/brotli_streams.json).
//
// Shape of the work.  A Brotli meta-block is one serial chain of insert-and-copy commands whose prefix
// codes, block types, context maps and distance ring buffer all depend on what was decoded before; the
// batch has one stream per entry.  So each work-item runs a straightforward decoder for its own entry:
// 64 independent streams per wavefront.  Prefix codes are kept in canonical form (symbols sorted by code
// length, counts per length) and decoded bit by bit -- compact (a meta-block may carry 3 x 256 codes) and
// simple; the tables of a meta-block live in global scratch (48 KB per work-item, 832 KB in a second pass for the
// rare meta-block that needs it), the
// entry's own output is the window.  The RFC's constant data (static dictionary, transforms, context
// lookup) comes from otezip_amd/data/brotli_rfc7932.bin, uploaded once per device.
// Large-window Brotli (WBITS > 24) is not RFC 7932 and is rejected.
#pragma once

#include "otz_simt.h"
#include "../../include/otezip_gpu.h"

// ------------------------------------------------------------------ constant data (the blob's sections)

struct otz_br_data {
	const u8 *dict;        // 122 784 bytes
	const u8 *size_bits;   // [32]
	const u32 *offsets;    // [32]
	const u8 *ctx_lut;     // [4][512]
	const u8 *transforms;  // [121][3]: prefix id, type, suffix id
	const u8 *affix;       // length-prefixed strings
	const u16 *affix_map;  // id -> offset into affix
};

OTZ_DEV otz_br_data otz_br_open(const u8 *blob) {
	const u32 *h = reinterpret_cast<const u32 *> (blob);
	otz_br_data d;
	d.dict = blob + h[2];
	d.size_bits = blob + h[4];
	d.offsets = reinterpret_cast<const u32 *> (blob + h[6]);
	d.ctx_lut = blob + h[8];
	d.transforms = blob + h[10];
	d.affix = blob + h[12];
	d.affix_map = reinterpret_cast<const u16 *> (blob + h[14]);
	return d;
}

// ------------------------------------------------------------------ bit reader (RFC 7932 section 1.5.2: LSB first)

struct otz_br_bits {
	const u8 *p;
	u32 n, pos; // bytes, next byte
	u64 acc;
	u32 have; // valid bits in acc (bytes beyond the input read as zero: see otz_br_overrun)
};

OTZ_DEV void otz_br_fill(otz_br_bits &b) {
	while (b.have <= 56u) {
		const u64 v = b.pos < b.n ? b.p[b.pos] : 0u;
		b.pos++;
		b.acc |= v << b.have;
		b.have += 8u;
	}
}
// n <= 32
OTZ_DEV u32 otz_br_peek(otz_br_bits &b, u32 n) {
	if (b.have < n) {
		otz_br_fill (b);
	}
	return (u32) (b.acc & (((u64)1 << n) - 1u));
}
OTZ_DEV void otz_br_drop(otz_br_bits &b, u32 n) {
	b.acc >>= n;
	b.have -= n;
}
OTZ_DEV u32 otz_br_read(otz_br_bits &b, u32 n) {
	if (n == 0) {
		return 0;
	}
	const u32 v = otz_br_peek (b, n);
	otz_br_drop (b, n);
	return v;
}
// to the next byte boundary; false when the skipped bits are not zero
OTZ_DEV bool otz_br_align(otz_br_bits &b) {
	const u32 k = b.have & 7u;
	return otz_br_read (b, k) == 0;
}
// have more bits been consumed than the input holds?  (corrupt or truncated: the decoder checks now and then)
OTZ_DEV bool otz_br_overrun(const otz_br_bits &b) { return 8ull * b.pos - b.have > 8ull * b.n; }

// ------------------------------------------------------------------ prefix codes, canonical form
//
// cnt[l] = codes of length l (1..15), sym[] = the symbols ordered by (length, value).  A code with a single
// symbol takes no bits: cnt[0] = 1.
struct otz_br_code {
	u16 *cnt; // [16]
	u16 *sym;
};

OTZ_DEV u32 otz_br_symbol(otz_br_bits &b, const otz_br_code &c) {
	// the sixteen counts in two 16-byte loads: one trip to memory, then registers (the tables live in global
	// memory, and a work-item has nothing else to do while it waits)
	const otz_u4 lo = otz_ld16_aligned (c.cnt), hi = otz_ld16_aligned (c.cnt + 8);
	const u32 w[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
	if (w[0] & 0xFFFFu) {
		return c.sym[0];
	}
	u32 bits = otz_br_peek (b, 15);
	u32 code = 0, first = 0, index = 0;
#pragma unroll
	for (u32 len = 1; len <= 15u; len++) {
		code |= bits & 1u;
		bits >>= 1;
		const u32 count = (w[len >> 1] >> (16u * (len & 1u))) & 0xFFFFu;
		if (code < first + count) {
			otz_br_drop (b, len);
			return c.sym[index + (code - first)];
		}
		index += count;
		first = (first + count) << 1;
		code <<= 1;
	}
	otz_br_drop (b, 15);
	return 0xFFFFFFFFu; // (cannot happen with a complete code)
}

// lens[0..n) -> canonical form.  Returns false when no symbol has a length.
OTZ_DEV bool otz_br_canon(const u8 *lens, u32 n, otz_br_code &c) {
	for (u32 l = 0; l < 16u; l++) {
		c.cnt[l] = 0;
	}
	u32 used = 0, last = 0;
	for (u32 s = 0; s < n; s++) {
		if (lens[s]) {
			c.cnt[lens[s]]++;
			used++;
			last = s;
		}
	}
	if (used == 0) {
		return false;
	}
	if (used == 1u) {
		for (u32 l = 1; l < 16u; l++) {
			c.cnt[l] = 0;
		}
		c.cnt[0] = 1;
		c.sym[0] = (u16)last;
		return true;
	}
	u16 offs[16];
	u32 o = 0;
	for (u32 l = 1; l < 16u; l++) {
		offs[l] = (u16)o;
		o += c.cnt[l];
	}
	for (u32 s = 0; s < n; s++) {
		if (lens[s]) {
			c.sym[offs[lens[s]]++] = (u16)s;
		}
	}
	return true;
}

// Section 3.4 / 3.5: one prefix code over an alphabet of `alphabet` symbols.  lens: alphabet bytes of scratch.
// Returns 0 or OZG_ST_DATA.
OTZ_DEV int otz_br_read_code(otz_br_bits &b, u32 alphabet, u8 *lens, otz_br_code &c) {
	const u32 hskip = otz_br_read (b, 2);
	for (u32 i = 0; i < alphabet; i++) {
		lens[i] = 0;
	}
	if (hskip == 1u) {
		// simple code: 1..4 symbols
		u32 abits = 0;
		while (((alphabet - 1u) >> abits) != 0) {
			abits++;
		}
		const u32 nsym = otz_br_read (b, 2) + 1u;
		u32 s[4] = { 0, 0, 0, 0 };
		for (u32 i = 0; i < nsym; i++) {
			s[i] = otz_br_read (b, abits);
			if (s[i] >= alphabet) {
				return OZG_ST_DATA;
			}
			for (u32 j = 0; j < i; j++) {
				if (s[j] == s[i]) {
					return OZG_ST_DATA;
				}
			}
		}
		if (nsym == 1u) {
			lens[s[0]] = 1; // (a single symbol: otz_br_canon turns it into the zero-bit code)
		} else if (nsym == 2u) {
			lens[s[0]] = 1;
			lens[s[1]] = 1;
		} else if (nsym == 3u) {
			lens[s[0]] = 1;
			lens[s[1]] = 2;
			lens[s[2]] = 2;
		} else if (otz_br_read (b, 1) == 0) {
			lens[s[0]] = lens[s[1]] = lens[s[2]] = lens[s[3]] = 2;
		} else {
			lens[s[0]] = 1;
			lens[s[1]] = 2;
			lens[s[2]] = 3;
			lens[s[3]] = 3;
		}
		return otz_br_canon (lens, alphabet, c) ? 0 : OZG_ST_DATA;
	}
	// complex code: the code lengths of the code-length alphabet, themselves in a fixed variable-length code
	const u8 order[18] = { 1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15 };
	const u8 fl_len[16] = { 2, 2, 2, 3, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2, 4 };
	const u8 fl_val[16] = { 0, 4, 3, 2, 0, 4, 3, 1, 0, 4, 3, 2, 0, 4, 3, 5 };
	u8 cl[18];
	for (u32 i = 0; i < 18u; i++) {
		cl[i] = 0;
	}
	int space = 32;
	u32 num = 0;
	for (u32 i = hskip; i < 18u && space > 0; i++) {
		const u32 ix = otz_br_peek (b, 4);
		otz_br_drop (b, fl_len[ix]);
		const u32 v = fl_val[ix];
		cl[order[i]] = (u8)v;
		if (v) {
			space -= (int) (32u >> v);
			num++;
		}
	}
	if (!(num == 1u || space == 0)) {
		return OZG_ST_DATA;
	}
	__attribute__ ((aligned (16))) u16 cl_cnt[16];
	u16 cl_sym[18];
	otz_br_code cc = { cl_cnt, cl_sym };
	if (!otz_br_canon (cl, 18, cc)) {
		return OZG_ST_DATA;
	}
	u32 symbol = 0, prev = 8, repeat = 0, repeat_len = 0;
	int room = 32768;
	while (symbol < alphabet && room > 0) {
		const u32 v = otz_br_symbol (b, cc);
		if (v < 16u) {
			repeat = 0;
			lens[symbol++] = (u8)v;
			if (v) {
				prev = v;
				room -= (int) (32768u >> v);
			}
		} else if (v <= 17u) {
			const u32 extra = v - 14u; // 16: 2 bits, 17: 3 bits
			const u32 new_len = v == 16u ? prev : 0u;
			if (repeat_len != new_len) {
				repeat = 0;
				repeat_len = new_len;
			}
			const u32 old = repeat;
			if (repeat > 0) {
				repeat = (repeat - 2u) << extra;
			}
			repeat += otz_br_read (b, extra) + 3u;
			const u32 delta = repeat - old;
			if (symbol + delta > alphabet) {
				return OZG_ST_DATA;
			}
			for (u32 k = 0; k < delta; k++) {
				lens[symbol + k] = (u8)repeat_len;
			}
			symbol += delta;
			if (repeat_len) {
				room -= (int) (delta << (15u - repeat_len));
			}
		} else {
			return OZG_ST_DATA;
		}
		if (otz_br_overrun (b)) {
			return OZG_ST_DATA;
		}
	}
	if (room != 0) {
		return OZG_ST_DATA;
	}
	return otz_br_canon (lens, alphabet, c) ? 0 : OZG_ST_DATA;
}

// ------------------------------------------------------------------ the small variable-length numbers

// section 9.2: 1..256 block types / trees, stored as 0..255
OTZ_DEV u32 otz_br_varlen8(otz_br_bits &b) {
	if (otz_br_read (b, 1) == 0) {
		return 0;
	}
	const u32 n = otz_br_read (b, 3);
	if (n == 0) {
		return 1;
	}
	return (1u << n) + otz_br_read (b, n);
}

OTZ_DEV u32 otz_br_block_count(otz_br_bits &b, const otz_br_code &c) {
	const u16 base[26] = { 1, 5, 9, 13, 17, 25, 33, 41, 49, 65, 81, 97, 113, 145, 177, 209, 241, 305, 369, 497, 753, 1265, 2289, 4337, 8433, 16625 };
	const u8 nb[26] = { 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 13, 24 };
	const u32 s = otz_br_symbol (b, c);
	if (s >= 26u) {
		return 0;
	}
	return base[s] + otz_br_read (b, nb[s]);
}

// ------------------------------------------------------------------ scratch arena (one meta-block's tables)

// Worst case of a meta-block: 256 literal codes, 256 insert-and-copy codes, 256 distance codes, the context
// maps: about 800 KB.  Streams that ask for that much are rare (few block types, few trees is the rule), so the
// batch is decoded in two passes: many work-items with OTZ_BR_ARENA_SMALL each, then the entries that ran
// out of room again with OTZ_BR_ARENA each (otz_brotli_kernel, route value OTZ_BR_ROUTE_AGAIN).
#define OTZ_BR_ARENA (832u * 1024u)
#ifndef OTZ_BR_ARENA_SMALL
#define OTZ_BR_ARENA_SMALL (48u * 1024u)
#endif
#define OTZ_BR_ROUTE_AGAIN 3u

struct otz_br_arena {
	u8 *base; // 16-byte aligned
	u32 used, cap;
};
OTZ_DEV u8 *otz_br_alloc(otz_br_arena &a, u32 bytes) {
	const u32 at = (a.used + 15u) & ~15u;
	if (at + bytes > a.cap) {
		return nullptr;
	}
	a.used = at + bytes;
	return a.base + at;
}
OTZ_DEV bool otz_br_new_code(otz_br_arena &a, u32 alphabet, otz_br_code &c) {
	c.cnt = reinterpret_cast<u16 *> (otz_br_alloc (a, 32u + 2u * alphabet));
	c.sym = c.cnt + 16;
	return c.cnt != nullptr;
}

// section 7.3: a context map of `size` entries; *ntrees out.  Returns 0 or a status.
OTZ_DEV int otz_br_context_map(otz_br_bits &b, otz_br_arena &a, u8 *lens, u8 *map, u32 size, u32 *ntrees) {
	const u32 nt = otz_br_varlen8 (b) + 1u;
	*ntrees = nt;
	if (nt == 1u) {
		for (u32 i = 0; i < size; i++) {
			map[i] = 0;
		}
		return 0;
	}
	const u32 rlemax = otz_br_read (b, 1) ? otz_br_read (b, 4) + 1u : 0u;
	const u32 mark = a.used;
	otz_br_code c;
	if (!otz_br_new_code (a, nt + rlemax, c)) {
		return OZG_ST_INTERNAL;
	}
	int st = otz_br_read_code (b, nt + rlemax, lens, c);
	if (st != 0) {
		return st;
	}
	u32 i = 0;
	while (i < size) {
		const u32 s = otz_br_symbol (b, c);
		if (s == 0) {
			map[i++] = 0;
		} else if (s <= rlemax) {
			u32 reps = (1u << s) + otz_br_read (b, s);
			if (i + reps > size) {
				return OZG_ST_DATA;
			}
			while (reps--) {
				map[i++] = 0;
			}
		} else if (s - rlemax < nt) {
			map[i++] = (u8) (s - rlemax);
		} else {
			return OZG_ST_DATA;
		}
		if (otz_br_overrun (b)) {
			return OZG_ST_DATA;
		}
	}
	a.used = mark; // (the code is not needed any more)
	if (otz_br_read (b, 1)) {
		// inverse move-to-front
		u8 mtf[256];
		for (u32 k = 0; k < 256u; k++) {
			mtf[k] = (u8)k;
		}
		for (u32 k = 0; k < size; k++) {
			const u32 ix = map[k];
			const u8 v = mtf[ix];
			map[k] = v;
			for (u32 j = ix; j > 0; j--) {
				mtf[j] = mtf[j - 1u];
			}
			mtf[0] = v;
		}
	}
	return 0;
}

// ------------------------------------------------------------------ block-switch state of one category

struct otz_br_cat {
	u32 ntypes;
	u32 type, prev_type;
	u32 left; // symbols left in the current block
	otz_br_code type_code, count_code;
};

OTZ_DEV int otz_br_read_cat(otz_br_bits &b, otz_br_arena &a, u8 *lens, otz_br_cat &k) {
	k.ntypes = otz_br_varlen8 (b) + 1u;
	k.type = 0;
	k.prev_type = 1;
	k.left = 1u << 24;
	if (k.ntypes >= 2u) {
		if (!otz_br_new_code (a, k.ntypes + 2u, k.type_code) || !otz_br_new_code (a, 26u, k.count_code)) {
			return OZG_ST_INTERNAL;
		}
		int st = otz_br_read_code (b, k.ntypes + 2u, lens, k.type_code);
		if (st != 0) {
			return st;
		}
		st = otz_br_read_code (b, 26u, lens, k.count_code);
		if (st != 0) {
			return st;
		}
		k.left = otz_br_block_count (b, k.count_code);
	}
	return 0;
}

// a block of the category is used up: the next block's type and length (section 6)
OTZ_DEV bool otz_br_switch(otz_br_bits &b, otz_br_cat &k) {
	if (k.ntypes < 2u) {
		k.left = 1u << 24; // (a single type: libbrotli lets the count run on)
		return true;
	}
	const u32 s = otz_br_symbol (b, k.type_code);
	u32 t;
	if (s == 0) {
		t = k.prev_type;
	} else if (s == 1u) {
		t = k.type + 1u;
		if (t >= k.ntypes) {
			t -= k.ntypes;
		}
	} else {
		t = s - 2u;
	}
	if (t >= k.ntypes) {
		return false;
	}
	k.prev_type = k.type;
	k.type = t;
	k.left = otz_br_block_count (b, k.count_code);
	return true;
}

// ------------------------------------------------------------------ dictionary words (section 8, Appendix B)

OTZ_DEV u32 otz_br_upper(u8 *p, u32 room) {
	if (p[0] < 0xC0u) {
		if (p[0] >= 'a' && p[0] <= 'z') {
			p[0] ^= 32u;
		}
		return 1;
	}
	if (p[0] < 0xE0u) {
		if (room > 1u) {
			p[1] ^= 32u;
		}
		return 2;
	}
	if (room > 2u) {
		p[2] ^= 5u;
	}
	return 3;
}

// Writes the transformed word at out[pos...]; returns its length, or 0xFFFFFFFF when it does not fit below cap.
OTZ_DEV u32 otz_br_word(const otz_br_data &D, u32 len, u32 word_idx, u32 tr, u8 *out, u32 pos, u32 cap) {
	const u8 *w = D.dict + D.offsets[len] + word_idx * len;
	const u8 *t3 = D.transforms + 3u * tr;
	const u8 *pre = D.affix + D.affix_map[t3[0]], *suf = D.affix + D.affix_map[t3[2]];
	const u32 type = t3[1];
	int n = (int)len;
	if (type >= 1u && type <= 9u) {
		n -= (int)type;
	} else if (type >= 12u && type <= 20u) {
		const u32 skip = type - 11u;
		w += skip;
		n -= (int)skip;
	}
	if (n < 0) {
		n = 0;
	}
	const u32 total = pre[0] + (u32)n + suf[0];
	if (total > cap - pos) {
		return 0xFFFFFFFFu;
	}
	u32 at = pos;
	for (u32 i = 0; i < pre[0]; i++) {
		out[at++] = pre[1u + i];
	}
	const u32 w0 = at;
	for (int i = 0; i < n; i++) {
		out[at++] = w[i];
	}
	if (type == 10u && n > 0) {
		(void)otz_br_upper (out + w0, (u32)n);
	} else if (type == 11u) {
		u32 p = w0;
		int left = n;
		while (left > 0) {
			const u32 step = otz_br_upper (out + p, (u32)left);
			p += step;
			left -= (int)step;
		}
	}
	for (u32 i = 0; i < suf[0]; i++) {
		out[at++] = suf[1u + i];
	}
	return total;
}

// ------------------------------------------------------------------ the stream

// One Brotli stream, one work-item.  arena: arena_bytes of scratch (16-byte aligned).  *produced = bytes written.
// OZG_ST_OK, OZG_ST_DATA (not a valid stream), OZG_ST_SIZE (more than cap bytes), OZG_ST_INTERNAL (a meta-block's
// tables do not fit the arena).
OTZ_DEV int otz_brotli_stream(const u8 *in, u32 n_in, u8 *out, u32 cap, u8 *arena_base, u32 arena_bytes, const otz_br_data &D, u32 *produced) {
	otz_br_bits b = { in, n_in, 0, 0, 0 };
	u32 pos = 0;
	u32 p1 = 0, p2 = 0; // the last two bytes written (the literal context): kept in registers, not read back
	*produced = 0;
	// ---- stream header: WBITS
	u32 wbits;
	if (otz_br_read (b, 1) == 0) {
		wbits = 16;
	} else {
		const u32 n = otz_br_read (b, 3);
		if (n != 0) {
			wbits = 17u + n;
		} else {
			const u32 m = otz_br_read (b, 3);
			if (m == 1u) {
				return OZG_ST_DATA; // large-window Brotli is not RFC 7932
			}
			wbits = m == 0 ? 17u : 8u + m;
		}
	}
	const u32 max_back = (1u << wbits) - 16u;
	u32 ring[4] = { 16, 15, 11, 4 };
	u32 ring_at = 0; // the last distance is ring[(ring_at - 1) & 3]
	const u32 ins_base[24] = { 0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594 };
	const u8 ins_extra[24] = { 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24 };
	const u32 cpy_base[24] = { 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118 };
	const u8 cpy_extra[24] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24 };
	const u8 cell_ins[11] = { 0, 0, 0, 0, 8, 8, 0, 16, 8, 16, 16 };
	const u8 cell_cpy[11] = { 0, 8, 0, 8, 0, 8, 16, 0, 16, 8, 16 };

	for (;;) {
		if (otz_br_overrun (b)) {
			return OZG_ST_DATA;
		}
		// ---- meta-block header (section 9.2)
		const u32 is_last = otz_br_read (b, 1);
		if (is_last && otz_br_read (b, 1)) {
			break; // ISLASTEMPTY
		}
		const u32 nib_code = otz_br_read (b, 2);
		if (nib_code == 3u) {
			// metadata: no output
			if (is_last || otz_br_read (b, 1) != 0) {
				return OZG_ST_DATA;
			}
			const u32 nbytes = otz_br_read (b, 2);
			u32 skip = 0;
			for (u32 i = 0; i < nbytes; i++) {
				const u32 v = otz_br_read (b, 8);
				if (i + 1u == nbytes && nbytes > 1u && v == 0) {
					return OZG_ST_DATA;
				}
				skip |= v << (8u * i);
			}
			if (nbytes) {
				skip += 1u;
			}
			if (!otz_br_align (b)) {
				return OZG_ST_DATA;
			}
			for (u32 i = 0; i < skip; i++) {
				(void)otz_br_read (b, 8);
				if (otz_br_overrun (b)) {
					return OZG_ST_DATA;
				}
			}
			continue;
		}
		const u32 nibbles = 4u + nib_code;
		u32 mlen = 0;
		for (u32 i = 0; i < nibbles; i++) {
			const u32 v = otz_br_read (b, 4);
			if (i + 1u == nibbles && nibbles > 4u && v == 0) {
				return OZG_ST_DATA;
			}
			mlen |= v << (4u * i);
		}
		mlen += 1u;
		if (!is_last && otz_br_read (b, 1)) {
			// uncompressed
			if (!otz_br_align (b)) {
				return OZG_ST_DATA;
			}
			if (mlen > cap - pos) {
				return OZG_ST_SIZE;
			}
			for (u32 i = 0; i < mlen; i++) {
				const u32 v = otz_br_read (b, 8);
				out[pos++] = (u8)v;
				p2 = p1;
				p1 = v;
			}
			if (otz_br_overrun (b)) {
				return OZG_ST_DATA;
			}
			continue;
		}
		if (mlen > cap - pos) {
			return OZG_ST_SIZE;
		}
		// ---- the rest of the header: block types, distance parameters, context maps, prefix codes
		otz_br_arena A = { arena_base, 0, arena_bytes };
		u8 *lens = otz_br_alloc (A, 1024u); // code lengths of the code being read (the largest alphabet has 704 symbols)
		otz_br_cat cat[3];
		for (u32 k = 0; k < 3u; k++) {
			const int st = otz_br_read_cat (b, A, lens, cat[k]);
			if (st != 0) {
				return st;
			}
		}
		const u32 npostfix = otz_br_read (b, 2);
		const u32 ndirect = otz_br_read (b, 4) << npostfix;
		u8 *modes = otz_br_alloc (A, cat[0].ntypes);
		u8 *cmap_l = otz_br_alloc (A, 64u * cat[0].ntypes);
		u8 *cmap_d = otz_br_alloc (A, 4u * cat[2].ntypes);
		if (!lens || !modes || !cmap_l || !cmap_d) {
			return OZG_ST_INTERNAL;
		}
		for (u32 i = 0; i < cat[0].ntypes; i++) {
			modes[i] = (u8)otz_br_read (b, 2);
		}
		u32 ntrees_l, ntrees_d;
		int st = otz_br_context_map (b, A, lens, cmap_l, 64u * cat[0].ntypes, &ntrees_l);
		if (st != 0) {
			return st;
		}
		st = otz_br_context_map (b, A, lens, cmap_d, 4u * cat[2].ntypes, &ntrees_d);
		if (st != 0) {
			return st;
		}
		const u32 alpha_d = 16u + ndirect + (48u << npostfix);
		// the codes of one kind are laid out at a fixed stride, so a tree is found by its number
		const u32 stride_l = 32u + 2u * 256u, stride_i = 32u + 2u * 704u, stride_d = 32u + 2u * ((alpha_d + 7u) & ~7u); // (multiples of 16)
		u8 *codes_l = otz_br_alloc (A, ntrees_l * stride_l);
		u8 *codes_i = otz_br_alloc (A, cat[1].ntypes * stride_i);
		u8 *codes_d = otz_br_alloc (A, ntrees_d * stride_d);
		if (!codes_l || !codes_i || !codes_d) {
			return OZG_ST_INTERNAL;
		}
		for (u32 t = 0; t < ntrees_l; t++) {
			otz_br_code c = { reinterpret_cast<u16 *> (codes_l + t * stride_l), reinterpret_cast<u16 *> (codes_l + t * stride_l + 32u) };
			st = otz_br_read_code (b, 256u, lens, c);
			if (st != 0) {
				return st;
			}
		}
		for (u32 t = 0; t < cat[1].ntypes; t++) {
			otz_br_code c = { reinterpret_cast<u16 *> (codes_i + t * stride_i), reinterpret_cast<u16 *> (codes_i + t * stride_i + 32u) };
			st = otz_br_read_code (b, 704u, lens, c);
			if (st != 0) {
				return st;
			}
		}
		for (u32 t = 0; t < ntrees_d; t++) {
			otz_br_code c = { reinterpret_cast<u16 *> (codes_d + t * stride_d), reinterpret_cast<u16 *> (codes_d + t * stride_d + 32u) };
			st = otz_br_read_code (b, alpha_d, lens, c);
			if (st != 0) {
				return st;
			}
		}
		if (otz_br_overrun (b)) {
			return OZG_ST_DATA;
		}
		// ---- the commands (section 10)
		const u32 postfix_mask = (1u << npostfix) - 1u;
		u32 left = mlen;
		while (left > 0) {
			if (cat[1].left == 0 && !otz_br_switch (b, cat[1])) {
				return OZG_ST_DATA;
			}
			cat[1].left--;
			const otz_br_code ci = { reinterpret_cast<u16 *> (codes_i + cat[1].type * stride_i), reinterpret_cast<u16 *> (codes_i + cat[1].type * stride_i + 32u) };
			const u32 sym = otz_br_symbol (b, ci);
			if (sym >= 704u) {
				return OZG_ST_DATA;
			}
			const u32 cell = sym >> 6;
			const u32 icode = cell_ins[cell] + ((sym >> 3) & 7u), ccode = cell_cpy[cell] + (sym & 7u);
			const u32 insert = ins_base[icode] + otz_br_read (b, ins_extra[icode]);
			const u32 copy = cpy_base[ccode] + otz_br_read (b, cpy_extra[ccode]);
			if (insert > left) {
				return OZG_ST_DATA;
			}
			// ---- literals
			for (u32 i = 0; i < insert; i++) {
				if (cat[0].left == 0 && !otz_br_switch (b, cat[0])) {
					return OZG_ST_DATA;
				}
				cat[0].left--;
				const u8 *lut = D.ctx_lut + 512u * modes[cat[0].type];
				const u32 ctx = lut[p1] | lut[256u + p2];
				const u32 tree = cmap_l[64u * cat[0].type + ctx];
				const otz_br_code cl = { reinterpret_cast<u16 *> (codes_l + tree * stride_l), reinterpret_cast<u16 *> (codes_l + tree * stride_l + 32u) };
				const u32 lit = otz_br_symbol (b, cl);
				if (lit > 255u) {
					return OZG_ST_DATA;
				}
				out[pos++] = (u8)lit;
				p2 = p1;
				p1 = lit;
			}
			if (otz_br_overrun (b)) {
				return OZG_ST_DATA;
			}
			left -= insert;
			if (left == 0) {
				break;
			}
			// ---- distance
			u32 dcode = 0;
			if (sym >= 128u) {
				if (cat[2].left == 0 && !otz_br_switch (b, cat[2])) {
					return OZG_ST_DATA;
				}
				cat[2].left--;
				const u32 dctx = copy > 4u ? 3u : copy - 2u;
				const u32 tree = cmap_d[4u * cat[2].type + dctx];
				const otz_br_code cd = { reinterpret_cast<u16 *> (codes_d + tree * stride_d), reinterpret_cast<u16 *> (codes_d + tree * stride_d + 32u) };
				dcode = otz_br_symbol (b, cd);
				if (dcode >= alpha_d) {
					return OZG_ST_DATA;
				}
			}
			u32 dist;
			if (dcode < 16u) {
				const u8 which[16] = { 1, 2, 3, 4, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2 };
				const int delta[16] = { 0, 0, 0, 0, -1, 1, -2, 2, -3, 3, -1, 1, -2, 2, -3, 3 };
				const int v = (int)ring[(ring_at - which[dcode]) & 3u] + delta[dcode];
				if (v <= 0) {
					return OZG_ST_DATA;
				}
				dist = (u32)v;
			} else if (dcode < 16u + ndirect) {
				dist = dcode - 15u;
			} else {
				const u32 x = dcode - ndirect - 16u;
				const u32 nbits = 1u + (x >> (npostfix + 1u));
				const u32 hcode = x >> npostfix, lcode = x & postfix_mask;
				const u32 offset = ((2u + (hcode & 1u)) << nbits) - 4u;
				dist = ((offset + otz_br_read (b, nbits)) << npostfix) + lcode + ndirect + 1u;
			}
			const u32 max_dist = pos < max_back ? pos : max_back;
			if (dist <= max_dist) {
				if (dcode != 0) {
					ring[ring_at & 3u] = dist;
					ring_at++;
				}
				if (copy > left) {
					return OZG_ST_DATA;
				}
				for (u32 i = 0; i < copy; i++) { // (byte by byte: the copy may overlap itself)
					const u32 v = out[pos - dist];
					out[pos++] = (u8)v;
					p2 = p1;
					p1 = v;
				}
				left -= copy;
			} else {
				// a word of the static dictionary, transformed
				if (copy < 4u || copy > 24u) {
					return OZG_ST_DATA;
				}
				const u32 id = dist - max_dist - 1u;
				const u32 shift = D.size_bits[copy];
				const u32 word_idx = id & ((1u << shift) - 1u), tr = id >> shift;
				if (tr >= 121u) {
					return OZG_ST_DATA;
				}
				const u32 n = otz_br_word (D, copy, word_idx, tr, out, pos, pos + left);
				if (n == 0xFFFFFFFFu) {
					return OZG_ST_DATA; // (longer than what the meta-block has left)
				}
				pos += n;
				left -= n;
				p1 = pos > 0 ? out[pos - 1u] : 0u;
				p2 = pos > 1u ? out[pos - 2u] : 0u;
			}
			if (otz_br_overrun (b)) {
				return OZG_ST_DATA;
			}
		}
		if (is_last) {
			break;
		}
	}
	if (!otz_br_align (b) || otz_br_overrun (b)) {
		return OZG_ST_DATA; // non-zero padding, or the stream ran past its bytes
	}
	*produced = pos;
	return OZG_ST_OK;
}

// does a method-97 payload go to this decoder first?  (see the head of this file)
OTZ_DEV bool otz_brotli_is_candidate(const u8 *in, u32 n_in) {
	return n_in >= 1u && !(n_in >= 4u && in[0] == 'B' && in[1] == 'R' && in[2] == 'O' && in[3] == 'T');
}

// One method-97 entry (one work-item).  With a small arena, an entry whose tables do not fit is marked
// OTZ_BR_ROUTE_AGAIN for the second pass.
// ctx_lut (optional): a copy of the context lookup tables in faster memory than the blob (LDS).
OTZ_DEV void otz_brotli_run(const u8 *in, u32 n_in, u8 *out, u32 cap, u32 eflags, u8 *arena, u32 arena_bytes, const u8 *blob, ozg_result *res, u8 *route, const u8 *ctx_lut = nullptr) {
	otz_br_data D = otz_br_open (blob);
	if (ctx_lut) {
		D.ctx_lut = ctx_lut;
	}
	u32 total = 0;
	const int st = otz_brotli_stream (in, n_in, out, cap, arena, arena_bytes, D, &total);
	if (st == OZG_ST_OK && (total == cap || (eflags & OZG_EF_SIZE_IS_CAP))) {
		res->status = OZG_ST_OK;
		res->total_out = total;
		res->crc32 = 0;
		res->flags = 0;
		*route = 2u; // OTZ_ZS_ROUTE_DONE: the CRC pass follows
	} else if (st == OZG_ST_INTERNAL && arena_bytes < OTZ_BR_ARENA) {
		*route = OTZ_BR_ROUTE_AGAIN;
	} else if (st == OZG_ST_SIZE && (eflags & OZG_EF_SIZE_IS_CAP)) {
		// the codec seam: a stream that is valid so far and wants more room than the caller gave -- that is
		// the verdict (Z_BUF_ERROR there), not a reason to ask the reference's container decoder
		res->status = OZG_ST_SIZE;
		res->total_out = 0;
		res->crc32 = 0;
		res->flags = 0;
		*route = 2u;
	} else {
		*route = 1u; // OTZ_ZS_ROUTE_PRIVATE: the reference's container decoder has the last word
	}
}
