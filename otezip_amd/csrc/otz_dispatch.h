// otz_dispatch.h -- per-entry front end shared by the two codec kernels: the part
// of otezip_extract_entry between "compressed bytes are in memory" and the
// method switch (ref src/lib/otezip.c:441-462).
#pragma once

#include "otz_simt.h"
#include "otz_inflate.h"
#include "otz_containers.h"

// ref src/lib/otezip.c:454-462: uncomp_size <= comp_size * ratio + slack unless overridden
OTZ_DEV bool otz_zipbomb(const ozg_entry &e, const ozg_limits &lim) {
	if ((e.flags & OZG_EF_IGNORE_ZIPBOMB) || e.comp_size == 0) {
		return false;
	}
	u64 allowed = (u64)e.comp_size * lim.max_expansion_ratio + lim.max_expansion_slack;
	return (u64)e.uncomp_size > allowed;
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
