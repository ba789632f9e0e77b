// otz_dispatch.h -- per-entry front end shared by the two codec kernels: the part
// of otezip_extract_entry between "compressed bytes are in memory" and the
// method switch (ref src/lib/otezip.c:441-462).
#pragma once

#include "otz_simt.h"
#include "otz_inflate.h"
#include "otz_containers.h"
#include "otz_huff_index.h"
#include "otz_lz_apply.h"

// ref src/lib/otezip.c:454-462: uncomp_size <= comp_size * ratio + slack unless overridden
OTZ_DEV bool otz_zipbomb(const ozg_entry &e, const ozg_limits &lim) {
	if ((e.flags & OZG_EF_IGNORE_ZIPBOMB) || e.comp_size == 0) {
		return false;
	}
	u64 allowed = (u64)e.comp_size * lim.max_expansion_ratio + lim.max_expansion_slack;
	return (u64)e.uncomp_size > allowed;
}

// The descriptor must stay inside the caller's buffers (overflow-safe): a table built from an
// untrusted central directory must not become an out-of-bounds access in HBM.
OTZ_DEV bool otz_entry_in_bounds(const ozg_entry &e, u64 src_len, u64 dst_len) {
	return e.src_off <= src_len && (u64)e.comp_size <= src_len - e.src_off && (e.dst_off & 15u) == 0 && e.dst_off <= dst_len &&
		(u64)e.uncomp_size <= dst_len - e.dst_off;
}

OTZ_DEV void otz_fail(ozg_result *r, int status) {
	if (otz_tid () == 0) {
		r->status = status;
		r->total_out = 0;
		r->crc32 = 0;
		r->flags = 0;
	}
}

// DEFLATE entries (method 8). Entries of other methods are left to the copy kernel.
OTZ_DEV void otz_run_inflate(otz_inflate_lds &L, const u8 *src, u8 *dst, const ozg_entry &e, ozg_result *r, const ozg_limits &lim, u64 *prof) {
	if (e.method != OZG_METHOD_DEFLATE) {
		return;
	}
	if (otz_zipbomb (e, lim)) {
		otz_fail (r, OZG_ST_BOMB);
		return;
	}
	otz_job J;
	J.in = src + e.src_off;
	J.n_in = e.comp_size;
	J.out = dst + e.dst_off;
	J.out_cap = e.uncomp_size;
	J.prof = prof;
	otz_inflate_entry (L, J, e.crc32, r);
}

// DEFLATE entries, first half: the index (one wavefront).  Entries the expansion guard
// rejects get no index; the apply kernel reports them.
// Returns true when the entry has to take the self-contained routine (index overflow).
template <bool PROF = false>
OTZ_DEV bool otz_run_index(otz_k1_lds &L, const u8 *src, u8 *dst, const ozg_entry &e, const ozg_limits &lim, otz_ix_head *head, u8 *region, u32 region_bytes, u64 *prof = nullptr) {
	if (e.method != OZG_METHOD_DEFLATE || otz_zipbomb (e, lim)) {
		return false;
	}
	return otz_index_entry<PROF> (L, src + e.src_off, e.comp_size, e.uncomp_size, ((uintptr_t) (dst + e.dst_off) & 15u) == 0, region, region_bytes, head, prof);
}

// DEFLATE entries, second half: index -> bytes
OTZ_DEV void otz_run_apply(otz_apply_lds &L, const u8 *src, u8 *dst, const ozg_entry &e, ozg_result *r, const ozg_limits &lim, const otz_ix_head *head, const u8 *region, u64 *prof, u64 *end_bit = nullptr) {
	if (e.method != OZG_METHOD_DEFLATE) {
		return;
	}
	if (otz_zipbomb (e, lim)) {
		otz_fail (r, OZG_ST_BOMB);
		return;
	}
	otz_job J;
	J.in = src + e.src_off;
	J.n_in = e.comp_size;
	J.out = dst + e.dst_off;
	J.out_cap = e.uncomp_size;
	J.prof = prof;
	otz_apply_entry (L, J, head, region, r, end_bit);
}

// CRC-32 of an entry's output for the kernels that leave it to a streaming pass (ref
// src/lib/otezip.c:667-679: m bytes: for DEFLATE the whole uncomp_size, a zero-filled tail included).
// keep_flags: the decoder's flags stay (DEFLATE), else they are replaced.  L.t4/L.a4 are loaded.
OTZ_DEV void otz_run_crc(otz_copy_lds &L, const u8 *dst, const ozg_entry &e, ozg_result *r, u32 m, bool keep_flags) {
	const u32 R = otz_stream_crc (L, dst + e.dst_off, nullptr, m, false);
	if (otz_tid () == 0) {
		const u32 crc = ~(otz_gf_mul (0xFFFFFFFFu, otz_xpow_bytes (m)) ^ R);
		r->crc32 = crc;
		r->flags = (keep_flags ? r->flags : 0u) | (crc != e.crc32 ? OZG_F_CRC_MISMATCH : 0u);
	}
}

// every other method, including the unsupported ones (status OZG_ST_METHOD)
OTZ_DEV void otz_run_copy(otz_copy_lds &L, const u8 *src, u8 *dst, const ozg_entry &e, ozg_result *r, const ozg_limits &lim) {
	if (e.method == OZG_METHOD_DEFLATE) {
		return;
	}
	if (otz_zipbomb (e, lim)) {
		otz_fail (r, OZG_ST_BOMB);
		return;
	}
	otz_copy_entry (L, src + e.src_off, e.comp_size, dst + e.dst_off, e.uncomp_size, e.method, e.crc32, r);
}
