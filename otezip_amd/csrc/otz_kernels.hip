// otz_kernels.hip -- gfx950 kernels and the ozg_* C ABI (include/otezip_gpu.h).
//
// Decode kernels:
//   otz_plan_kernel     one work-item per entry: checks the descriptor against the buffers
//                       and carves the entry's region of the index arena
//   otz_index_kernel    DEFLATE entries, first half: ONE WAVEFRONT per entry finds the token
//                       chain and writes the index (otz_huff_index.h); 12 KB of LDS
//   otz_apply_kernel    DEFLATE entries, second half: 512 lanes (8 wavefronts), 77 KB of
//                       LDS per workgroup -> two resident workgroups per CU (<= 128 VGPRs);
//                       index -> bytes, LZ77 in LDS, CRC-32 (otz_lz_apply.h)
//   otz_inflate_kernel  the self-contained workgroup-per-entry DEFLATE decode (otz_inflate.h)
//                       for the entries whose index overflowed its arena region: a small
//                       persistent grid over the overflow list, normally empty
//   otz_copy_kernel     STORE and the private raw framings; 256 lanes, 11 KB of LDS
// Encode kernels, one buffer each: otz_encode_kernel (one wavefront: DEFLATE, the private
// framings' writers, STORE) and otz_enc_crc_kernel (CRC-32 of the input).
// Entries are independent, so the blockIdx -> entry map needs no XCD-aware
// remap: there is no operand shared between workgroups to keep in one L2.
//
// There is no CPU fallback in this library: without a HIP device ozg_init()
// fails with OZG_E_NODEVICE and every other entry point refuses to run.

#include <hip/hip_runtime.h>

#include <mutex>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "otz_dispatch.h"
#include "otz_deflate_enc.h"
#include "otz_zstd.h"
#include "otz_lzma.h"
#include "otz_brotli.h"
#include "otz_lzfse.h"

extern "C" const unsigned char otz_brotli_blob[], otz_brotli_blob_end[]; // otezip_amd/data/brotli_rfc7932.bin (brotli_data.S)

static_assert (sizeof (ozg_entry) == 32, "ozg_entry ABI");
static_assert (sizeof (ozg_result) == 16, "ozg_result ABI");
static_assert (sizeof (ozg_enc_entry) == 32 && sizeof (ozg_enc_result) == 16, "encode ABI");
static_assert (sizeof (otz_inflate_lds) <= 80 * 1024, "two inflate workgroups per CU");

static_assert (OTZ_NT * OTZ_S_BITS / 8 + 512 <= OTZ_STAGE_BYTES, "stage must cover one round of input");
static_assert (sizeof (otz_ix_head) == 32 && sizeof (otz_ix_block) == 32 && sizeof (otz_ix_item) == 8, "index layout");

#ifndef OTZ_K1_WPE
#define OTZ_K1_WPE 5 // 7.6 KB of LDS admit 21 index wavefronts per CU: registers (<= 96) must not cut that to 16
#endif
#ifndef OTZ_K2_WGS
#define OTZ_K2_WGS (OTZ_A_NT == 512u ? 3 : 7) // apply workgroups per CU (18 granules of LDS each; 36 with 512 work-items)
#endif
#ifndef OTZ_K2_WPE
#define OTZ_K2_WPE (OTZ_K2_WGS * (int)(OTZ_A_NT / 64u) / 4)
#endif
static_assert ((sizeof (otz_apply_lds) + 1279u) / 1280u * OTZ_K2_WGS <= 128u, "apply workgroups per CU: LDS comes in 1280-byte granules, 128 of them");
#define OTZ_BR_BIG_THREADS 512u // work-items of the Brotli kernel's second pass (832 KB of scratch each)
#ifndef OTZ_ZS_WPE
#define OTZ_ZS_WPE 4 // the Zstandard kernel's LDS (10.6 KB) admits 15 wavefronts per CU: registers must not halve that
#endif

// the encode-side descriptor check (overflow-safe), cf. otz_entry_in_bounds
OTZ_DEV bool otz_enc_entry_in_bounds(const ozg_enc_entry &e, u64 src_len, u64 dst_len) {
	return e.src_off <= src_len && (u64)e.src_size <= src_len - e.src_off && (e.dst_off & 15u) == 0 && e.dst_off <= dst_len &&
		(u64)e.dst_cap <= dst_len - e.dst_off;
}

struct otz_region {
	u64 off;   // byte offset into the index arena
	u32 bytes; // 0: none (descriptor out of bounds, not DEFLATE, or the arena is full)
	u32 pad;
};

extern "C" __global__ void __launch_bounds__ (256)
otz_plan_kernel(const ozg_entry *entries, u32 n, u64 src_len, u64 dst_len, u64 arena_cap, unsigned long long *cursor, otz_region *regions, otz_ix_head *heads) {
	const u32 i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) {
		return;
	}
	const ozg_entry e = entries[i];
	otz_region r = { 0, 0, 0 };
	if (e.method == OZG_METHOD_DEFLATE && otz_entry_in_bounds (e, src_len, dst_len)) {
		const u64 want = otz_ix_region_bytes (e.comp_size, e.uncomp_size);
		if (want <= 0xFFFFFF00ull) {
			const u64 off = atomicAdd (cursor, (unsigned long long)want);
			if (off + want <= arena_cap) {
				r.off = off;
				r.bytes = (u32)want;
			}
		}
	}
	regions[i] = r;
	heads[i].status = OTZ_IX_OVERFLOW; // until the index kernel says otherwise
}

// One wavefront per entry, no workgroup barrier anywhere: the number of entries in flight
// per CU is set by LDS (10 KB each) and registers alone.
template <bool PROF>
OTZ_DEV void otz_index_body(const u8 *src, u8 *dst, const ozg_entry *entries, u32 first, u32 n, u64 src_len, u64 dst_len, const ozg_limits &lim, u8 *arena,
	const otz_region *regions, otz_ix_head *heads, u32 *ovf, u64 *prof) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	const u32 i = first + blockIdx.x;
	if (i >= n) {
		return;
	}
	otz_k1_lds &L = *reinterpret_cast<otz_k1_lds *> (otz_smem);
	const ozg_entry e = entries[i];
	if (!otz_entry_in_bounds (e, src_len, dst_len)) {
		return;
	}
	const otz_region r = regions[i];
	if (otz_run_index<PROF> (L, src, dst, e, lim, &heads[i], arena + r.off, r.bytes, PROF && prof ? prof + 32ull * i + 16u : nullptr) && otz_lane () == 0) {
		ovf[1u + atomicAdd (&ovf[0], 1u)] = i; // ovf[0]: count, then the entries
	}
}

extern "C" __global__ void __launch_bounds__ (64) __attribute__ ((amdgpu_waves_per_eu (OTZ_K1_WPE, 8)))
otz_index_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, u32 first, u32 n, u64 src_len, u64 dst_len, ozg_limits lim, u8 *arena, const otz_region *regions, otz_ix_head *heads,
	u32 *ovf, u64 *prof) {
	otz_index_body<false> (src, dst, entries, first, n, src_len, dst_len, lim, arena, regions, heads, ovf, prof);
}

// the same with per-phase cycle counters (diagnostics: tools/prof_phases.py)
extern "C" __global__ void __launch_bounds__ (64) __attribute__ ((amdgpu_waves_per_eu (OTZ_K1_WPE, 8)))
otz_index_prof_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, u32 first, u32 n, u64 src_len, u64 dst_len, ozg_limits lim, u8 *arena, const otz_region *regions, otz_ix_head *heads,
	u32 *ovf, u64 *prof) {
	otz_index_body<true> (src, dst, entries, first, n, src_len, dst_len, lim, arena, regions, heads, ovf, prof);
}

extern "C" __global__ void __launch_bounds__ (OTZ_NT) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
otz_inflate_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, ozg_limits lim, const u32 *ovf) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	otz_inflate_lds &L = *reinterpret_cast<otz_inflate_lds *> (otz_smem);
	const u32 count = ovf[0];
	for (u32 k = blockIdx.x; k < count; k += gridDim.x) {
		const u32 i = ovf[1u + k];
		const ozg_entry e = entries[i];
		otz_sync ();
		otz_run_inflate (L, src, dst, e, &results[i], lim, nullptr);
	}
}

// OTZ_K2_WGS workgroups per CU (OTZ_A_NT / 64 wavefronts each): pin the occupancy the LDS footprint
// allows so a register-hungry change fails the resource check in tests/test_abi.py instead of
// silently cutting the number of entries in flight.
extern "C" __global__ void __launch_bounds__ (OTZ_A_NT) __attribute__ ((amdgpu_waves_per_eu (OTZ_K2_WPE, 8)))
otz_apply_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 first, u32 n, u64 src_len, u64 dst_len, ozg_limits lim,
	const u8 *arena, const otz_region *regions, const otz_ix_head *heads, u64 *prof, u64 *end_bits) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	const u32 i = first + blockIdx.x;
	if (i >= n) {
		return;
	}
	otz_apply_lds &L = *reinterpret_cast<otz_apply_lds *> (otz_smem);
	const ozg_entry e = entries[i];
	if (e.method != OZG_METHOD_DEFLATE) {
		return;
	}
	if (!otz_entry_in_bounds (e, src_len, dst_len)) {
		otz_fail (&results[i], OZG_ST_INTERNAL);
		return;
	}
	otz_run_apply (L, src, dst, e, &results[i], lim, &heads[i], arena + regions[i].off, prof ? prof + 32ull * i : nullptr,
		end_bits ? end_bits + i : nullptr);
}

// Zstandard frames under method 93 (otz_zstd.h): one wavefront per entry, a persistent grid (as many
// wavefronts as fit the chip, each taking the next entry off a counter) so that the literal scratch
// (OTZ_ZS_LIT_BYTES per wavefront) is sized by the chip, not by the batch.  Entries that are not
// method 93, not Zstandard candidates, out of bounds or rejected by the expansion guard are left to
// otz_copy_kernel (route stays 0); so are payloads that turn out not to be Zstandard (route 1).
template <bool PROF>
__device__ __forceinline__ void otz_zstd_grid(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, u64 src_len, u64 dst_len, ozg_limits lim,
	u8 *lit_scratch, u32 *cursor, u8 *route, u64 *prof) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	otz_zs_lds &L = *reinterpret_cast<otz_zs_lds *> (otz_smem);
	u8 *lit = lit_scratch + (size_t)blockIdx.x * OTZ_ZS_LIT_BYTES;
	for (;;) {
		u32 i = 0;
		if (otz_tid () == 0) {
			i = atomicAdd (cursor, 1u);
		}
		i = otz_uniform (i);
		if (i >= n) {
			break;
		}
		const ozg_entry e = entries[i];
		u32 take = e.method == OZG_METHOD_ZSTD && otz_entry_in_bounds (e, src_len, dst_len) && !otz_zipbomb (e, lim) &&
			otz_zs_is_candidate (src + e.src_off, e.comp_size) ? 1u : 0u;
		if (otz_uniform (take) == 0u) { // (the same in every lane; saying so keeps the loop's control flow uniform)
			continue;
		}
		otz_zs_run<PROF> (L, src + e.src_off, e.comp_size, dst + e.dst_off, e.uncomp_size, lit, &results[i], &route[i], PROF && prof ? prof + 32ull * i : nullptr);
		otz_wave_sync ();
	}
}

extern "C" __global__ void __launch_bounds__ (64) __attribute__ ((amdgpu_waves_per_eu (OTZ_ZS_WPE, 8)))
otz_zstd_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, u64 src_len, u64 dst_len, ozg_limits lim,
	u8 *lit_scratch, u32 *cursor, u8 *route) {
	otz_zstd_grid<false> (src, dst, entries, results, n, src_len, dst_len, lim, lit_scratch, cursor, route, nullptr);
}

// diagnostic build with per-phase cycle counters (tools/prof_zstd.py)
extern "C" __global__ void __launch_bounds__ (64)
otz_zstd_prof_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, u64 src_len, u64 dst_len, ozg_limits lim,
	u8 *lit_scratch, u32 *cursor, u8 *route, u64 *prof) {
	otz_zstd_grid<true> (src, dst, entries, results, n, src_len, dst_len, lim, lit_scratch, cursor, route, prof);
}

// LZMA1 streams under method 14 (otz_lzma.h): one WORK-ITEM per entry -- an LZMA stream has no
// parallelism inside, the batch has one stream per entry.  A persistent grid: every work-item takes
// the next entry off a counter and owns OTZ_LZMA_PROB_BYTES of scratch for its probability model.
extern "C" __global__ void __launch_bounds__ (64)
otz_lzma_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, u64 src_len, u64 dst_len, ozg_limits lim,
	u8 *prob_scratch, u32 *cursor, u8 *route) {
	u16 *probs = reinterpret_cast<u16 *> (prob_scratch + (size_t) (blockIdx.x * 64u + threadIdx.x) * OTZ_LZMA_PROB_BYTES);
	for (;;) {
		const u32 i = atomicAdd (cursor, 1u);
		if (i >= n) {
			break;
		}
		const ozg_entry e = entries[i];
		if (e.method != OZG_METHOD_LZMA || !otz_entry_in_bounds (e, src_len, dst_len) || otz_zipbomb (e, lim) ||
			!otz_lzma_is_candidate (src + e.src_off, e.comp_size)) {
			continue; // the copy kernel's (route stays 0)
		}
		otz_lzma_run (src + e.src_off, e.comp_size, dst + e.dst_off, e.uncomp_size, e.flags, probs, &results[i], &route[i]);
	}
}

// Brotli streams under method 97 (otz_brotli.h): one work-item per stream, a persistent grid; every work-item
// owns arena_bytes of scratch for the tables of the meta-block it is in.  Launched twice: many work-items with
// small arenas, then (again != 0) few with large ones for the entries the first pass marked.  blob: brotli_rfc7932.bin.
extern "C" __global__ void __launch_bounds__ (64)
otz_brotli_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, u64 src_len, u64 dst_len, ozg_limits lim,
	u8 *arena, u32 arena_bytes, const u8 *blob, u32 *cursor, u8 *route, u32 again) {
	u8 *mine = arena + (size_t) (blockIdx.x * 64u + threadIdx.x) * arena_bytes;
	// the literal context lookup (2 KB, read twice per literal) sits in LDS
	__shared__ __attribute__ ((aligned (16))) u8 ctx_lut[2048];
	{
		const otz_br_data D = otz_br_open (blob);
		for (u32 c = threadIdx.x; c < 2048u / 16u; c += 64u) {
			*reinterpret_cast<otz_u4 *> (&ctx_lut[16u * c]) = otz_ld16 (D.ctx_lut + 16u * c);
		}
		otz_wave_sync_lds ();
	}
	for (;;) {
		const u32 i = atomicAdd (cursor, 1u);
		if (i >= n) {
			break;
		}
		const ozg_entry e = entries[i];
		if (again) {
			if (route[i] != OTZ_BR_ROUTE_AGAIN) {
				continue;
			}
		} else if (e.method != OZG_METHOD_BROTLI || !otz_entry_in_bounds (e, src_len, dst_len) || otz_zipbomb (e, lim) ||
			!otz_brotli_is_candidate (src + e.src_off, e.comp_size)) {
			continue; // the copy kernel's (route stays 0)
		}
		otz_brotli_run (src + e.src_off, e.comp_size, dst + e.dst_off, e.uncomp_size, e.flags, mine, arena_bytes, blob, &results[i], &route[i], ctx_lut);
	}
}

// LZFSE streams under method 100 (otz_lzfse.h): one work-item per stream, a persistent grid; the scratch is the
// Brotli kernel's first-pass arena (OTZ_LZFSE_ARENA <= OTZ_BR_ARENA_SMALL per work-item; the two kernels run one
// after the other on the batch's stream).
static_assert (OTZ_LZFSE_ARENA <= OTZ_BR_ARENA_SMALL, "the LZFSE decoder's scratch fits the Brotli kernel's small arena");
extern "C" __global__ void __launch_bounds__ (64)
otz_lzfse_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, u64 src_len, u64 dst_len, ozg_limits lim,
	u8 *arena, u32 *cursor, u8 *route) {
	u8 *mine = arena + (size_t) (blockIdx.x * 64u + threadIdx.x) * OTZ_BR_ARENA_SMALL;
	for (;;) {
		const u32 i = atomicAdd (cursor, 1u);
		if (i >= n) {
			break;
		}
		const ozg_entry e = entries[i];
		if (e.method != OZG_METHOD_LZFSE || !otz_entry_in_bounds (e, src_len, dst_len) || otz_zipbomb (e, lim) ||
			!otz_lzfse_is_candidate (src + e.src_off, e.comp_size)) {
			continue; // the copy kernel's (route stays 0)
		}
		otz_lzfse_run (src + e.src_off, e.comp_size, dst + e.dst_off, e.uncomp_size, e.flags, mine, &results[i], &route[i]);
	}
}

// How many entries of a batch each of the real-format kernels would take (the same tests those kernels make):
// counts[0..3] = Zstandard, LZMA1, Brotli, LZFSE.  For batches whose payloads are only on the device, where the
// host cannot look: a kernel none of whose entries is in the batch is not launched and gets no scratch, and the
// scratch of the others is sized by what the batch holds, not by its number of entries.
extern "C" __global__ void __launch_bounds__ (256)
otz_sniff_kernel(const u8 *src, const ozg_entry *entries, u32 n, u64 src_len, u64 dst_len, ozg_limits lim, u32 *counts) {
	const u32 i = blockIdx.x * 256u + threadIdx.x;
	u32 which = 4u;
	if (i < n) {
		const ozg_entry e = entries[i];
		if (otz_entry_in_bounds (e, src_len, dst_len) && !otz_zipbomb (e, lim)) {
			const u8 *p = src + e.src_off;
			if (e.method == OZG_METHOD_ZSTD && otz_zs_is_candidate (p, e.comp_size)) {
				which = 0u;
			} else if (e.method == OZG_METHOD_LZMA && otz_lzma_is_candidate (p, e.comp_size)) {
				which = 1u;
			} else if (e.method == OZG_METHOD_BROTLI && otz_brotli_is_candidate (p, e.comp_size)) {
				which = 2u;
			} else if (e.method == OZG_METHOD_LZFSE && otz_lzfse_is_candidate (p, e.comp_size)) {
				which = 3u;
			}
		}
	}
	for (u32 k = 0; k < 4u; k++) { // (one atomic per wavefront and format)
		const u64 m = otz_ballot (which == k);
		if (m != 0 && otz_lane () == otz_ctz64 (m)) {
			atomicAdd (&counts[k], otz_popc64 (m));
		}
	}
}

// CRC-32 of the entries otz_zstd_kernel / otz_lzma_kernel / otz_brotli_kernel / otz_lzfse_kernel decoded (ref src/lib/otezip.c:667-679)
extern "C" __global__ void __launch_bounds__ (256)
otz_zstd_crc_kernel(u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, const u8 *route) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	const u32 i = blockIdx.x;
	if (i >= n || route[i] != OTZ_ZS_ROUTE_DONE || results[i].status != OZG_ST_OK) {
		return;
	}
	otz_copy_lds &L = *reinterpret_cast<otz_copy_lds *> (otz_smem);
	otz_crc_load_slice4 (L.t4);
	otz_crc_load_adv4k (L.a4);
	otz_sync ();
	otz_run_crc (L, dst, entries[i], &results[i], results[i].total_out, false);
}

// CRC-32 of the DEFLATE entries otz_apply_kernel decoded
extern "C" __global__ void __launch_bounds__ (256)
otz_out_crc_kernel(const u8 *dst, const ozg_entry *entries, ozg_result *results, u32 first, u32 n) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	const u32 i = first + blockIdx.x;
	if (i >= n || entries[i].method != OZG_METHOD_DEFLATE || results[i].status != OZG_ST_OK) {
		return;
	}
	otz_copy_lds &L = *reinterpret_cast<otz_copy_lds *> (otz_smem);
	otz_crc_load_slice4 (L.t4);
	otz_crc_load_adv4k (L.a4);
	otz_sync ();
	otz_run_crc (L, dst, entries[i], &results[i], entries[i].uncomp_size, true);
}

extern "C" __global__ void __launch_bounds__ (256)
otz_copy_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, u64 src_len, u64 dst_len, ozg_limits lim, const u8 *route) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	u32 i = blockIdx.x;
	if (i >= n) {
		return;
	}
	otz_copy_lds &L = *reinterpret_cast<otz_copy_lds *> (otz_smem);
	ozg_entry e = entries[i];
	if (e.method == OZG_METHOD_DEFLATE) {
		return;
	}
	if (route && route[i] == OTZ_ZS_ROUTE_DONE) {
		return; // a Zstandard frame: otz_zstd_kernel has decoded it
	}
	if (!otz_entry_in_bounds (e, src_len, dst_len)) {
		otz_fail (&results[i], OZG_ST_INTERNAL);
		return;
	}
	otz_run_copy (L, src, dst, e, &results[i], lim);
}

// CRC-32 of n independent byte ranges (otezip_crc32, ref src/lib/crc32.inc.c:40-47)
extern "C" __global__ void __launch_bounds__ (256)
otz_crc32_kernel(const u8 *buf, const u64 *off, const u32 *len, u32 *crc, u32 n) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	u32 i = blockIdx.x;
	if (i >= n) {
		return;
	}
	otz_copy_lds &L = *reinterpret_cast<otz_copy_lds *> (otz_smem);
	otz_crc_load_slice4 (L.t4);
	otz_crc_load_adv4k (L.a4);
	otz_sync ();
	u32 m = len[i];
	u32 R = otz_stream_crc (L, buf + off[i], nullptr, m, false);
	if (otz_tid () == 0) {
		crc[i] = ~(otz_gf_mul (0xFFFFFFFFu, otz_xpow_bytes (m)) ^ R);
	}
}

// CRC-32 (kind 0) or Adler-32 (kind 1) of one buffer
extern "C" __global__ void __launch_bounds__ (256)
otz_checksum_kernel(const u8 *buf, u32 n, u32 kind, u32 *out) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	otz_copy_lds &L = *reinterpret_cast<otz_copy_lds *> (otz_smem);
	u32 v;
	if (kind == 1u) {
		v = otz_adler32_block (buf, n, reinterpret_cast<u64 *> (L.t4));
	} else {
		otz_crc_load_slice4 (L.t4);
		otz_crc_load_adv4k (L.a4);
		otz_sync ();
		const u32 R = otz_stream_crc (L, buf, nullptr, n, false);
		v = ~(otz_gf_mul (0xFFFFFFFFu, otz_xpow_bytes (n)) ^ R);
	}
	if (otz_tid () == 0) {
		*out = v;
	}
}

// DEFLATE encode / STORE of n independent buffers: one wavefront per buffer (20 KB LDS).  The grid is
// persistent -- as many wavefronts as fit the chip, each taking the next buffer off a counter -- so
// that the token scratch (OTZ_ENC_TOK_WORDS words per wavefront) is sized by the chip, not the batch.
template <bool PROF>
__device__ __forceinline__ void otz_encode_grid(const u8 *src, u8 *dst, const ozg_enc_entry *entries, ozg_enc_result *results, u32 n, u64 src_len,
	u64 dst_len, u32 *tok, u32 *cursor, u64 *prof) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	otz_enc_lds &L = *reinterpret_cast<otz_enc_lds *> (otz_smem);
	u32 *my_tok = tok + (size_t)blockIdx.x * OTZ_ENC_TOK_WORDS;
	for (;;) {
		u32 i = 0;
		if (otz_tid () == 0) {
			i = atomicAdd (cursor, 1u);
		}
		i = otz_uniform (i);
		if (i >= n) {
			break;
		}
		ozg_enc_entry e = entries[i];
		// (the verdict is the same in every lane; saying so keeps the loop's control flow uniform)
		if (otz_uniform (otz_enc_entry_in_bounds (e, src_len, dst_len) ? 1u : 0u) == 0u) {
			if (otz_tid () == 0) {
				results[i].status = OZG_ST_INTERNAL;
				results[i].comp_size = 0;
				results[i].crc32 = 0;
				results[i].method = e.method;
				results[i].flags = 0;
			}
			otz_wave_sync ();
			continue;
		}
		otz_encode_entry<PROF> (L, src, dst, e, &results[i], my_tok, PROF && prof ? prof + 32ull * i : nullptr);
		otz_wave_sync ();
	}
}

extern "C" __global__ void __launch_bounds__ (64)
otz_encode_kernel(const u8 *src, u8 *dst, const ozg_enc_entry *entries, ozg_enc_result *results, u32 n, u64 src_len, u64 dst_len,
	u32 *tok, u32 *cursor, u64 *prof) {
	otz_encode_grid<false> (src, dst, entries, results, n, src_len, dst_len, tok, cursor, prof);
}

// diagnostic build of the same kernel with the per-phase cycle counters (tools/prof_encode.py)
extern "C" __global__ void __launch_bounds__ (64)
otz_encode_prof_kernel(const u8 *src, u8 *dst, const ozg_enc_entry *entries, ozg_enc_result *results, u32 n, u64 src_len, u64 dst_len,
	u32 *tok, u32 *cursor, u64 *prof) {
	otz_encode_grid<true> (src, dst, entries, results, n, src_len, dst_len, tok, cursor, prof);
}

// CRC-32 of every input buffer of an encode batch (ref src/lib/otezip.c:1124)
extern "C" __global__ void __launch_bounds__ (256)
otz_enc_crc_kernel(const u8 *src, u8 *dst, const ozg_enc_entry *entries, ozg_enc_result *results, u32 n, u64 src_len, u64 dst_len) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	u32 i = blockIdx.x;
	if (i >= n) {
		return;
	}
	if (!otz_enc_entry_in_bounds (entries[i], src_len, dst_len)) {
		return;
	}
	otz_copy_lds &L = *reinterpret_cast<otz_copy_lds *> (otz_smem);
	otz_crc_load_slice4 (L.t4);
	otz_crc_load_adv4k (L.a4);
	otz_sync ();
	u32 m = entries[i].src_size;
	if (entries[i].reserved & OZG_ENC_F_ADLER32) {
		const u32 ad = otz_adler32_block (src + entries[i].src_off, m, reinterpret_cast<u64 *> (L.t4));
		if (otz_tid () == 0) {
			results[i].crc32 = ad;
		}
		return;
	}
	u32 R = otz_stream_crc (L, src + entries[i].src_off, nullptr, m, false);
	if (otz_tid () == 0) {
		const u32 crc = ~(otz_gf_mul (0xFFFFFFFFu, otz_xpow_bytes (m)) ^ R);
		results[i].crc32 = crc;
		// a BROT container written by the codec-level call carries this CRC (ref brotli.inc.c:49-52);
		// the encode kernel (same stream, earlier) left the field blank
		if (entries[i].method == OZG_METHOD_BROTLI && (entries[i].reserved & OZG_ENC_F_CODEC_ONLY) &&
			results[i].status == OZG_ST_OK && results[i].method == OZG_METHOD_BROTLI) {
			u8 *h = dst + entries[i].dst_off + 13;
			h[0] = (u8)crc;
			h[1] = (u8) (crc >> 8);
			h[2] = (u8) (crc >> 16);
			h[3] = (u8) (crc >> 24);
		}
	}
}

// Packing the outputs of an encode batch before they cross PCIe: the encode kernel writes every
// payload at its worst-case slot (dst_off, dst_cap = compressBound), so the batch's output buffer is
// 2-3x the bytes that matter.  otz_pack_scan_kernel (one workgroup) gives entry i the running
// 16-byte-aligned offset pack_off[i] (pack_off[n] = total); otz_pack_copy_kernel moves the payloads.
extern "C" __global__ void __launch_bounds__ (256)
otz_pack_scan_kernel(const ozg_enc_result *results, u32 n, u64 *pack_off) {
	__shared__ u64 part[256];
	const u32 t = threadIdx.x;
	const u32 per = (n + 255u) / 256u;
	const u32 i0 = t * per < n ? t * per : n, i1 = i0 + per < n ? i0 + per : n;
	u64 sum = 0;
	for (u32 i = i0; i < i1; i++) {
		sum += ((u64)(results[i].status == OZG_ST_OK ? results[i].comp_size : 0u) + 15u) & ~(u64)15u;
	}
	part[t] = sum;
	__syncthreads ();
	if (t == 0) {
		u64 run = 0;
		for (u32 k = 0; k < 256u; k++) {
			const u64 v = part[k];
			part[k] = run;
			run += v;
		}
		pack_off[n] = run;
	}
	__syncthreads ();
	u64 at = part[t];
	for (u32 i = i0; i < i1; i++) {
		pack_off[i] = at;
		at += ((u64)(results[i].status == OZG_ST_OK ? results[i].comp_size : 0u) + 15u) & ~(u64)15u;
	}
}

extern "C" __global__ void __launch_bounds__ (256)
otz_pack_copy_kernel(const u8 *dst, const ozg_enc_entry *entries, const ozg_enc_result *results, const u64 *pack_off, u8 *packed, u32 n, u64 dst_len) {
	const u32 i = blockIdx.x;
	if (i >= n || results[i].status != OZG_ST_OK) {
		return;
	}
	const ozg_enc_entry e = entries[i];
	const u32 m = results[i].comp_size;
	if ((e.dst_off & 15u) != 0 || e.dst_off > dst_len || (u64)m > dst_len - e.dst_off) {
		return; // (the encode kernel refused such a descriptor: status is not OK; belt and braces)
	}
	const u8 *from = dst + e.dst_off;
	u8 *to = packed + pack_off[i];
	const u32 chunks = (m + 15u) >> 4; // both sides are 16-byte aligned; the slot is at least that long rounded up
	for (u32 c = threadIdx.x; c < chunks; c += 256u) {
		otz_st16_aligned (to + 16u * c, otz_ld16_aligned (from + 16u * c));
	}
}

// ------------------------------------------------------------------ host side
//
// One context per HIP device (staging buffers, the index arena, the slots of the asynchronous
// staged batches): nothing here is process-global but the table of contexts, so a host can
// drive several GPUs of a node from one process -- ozg_init(d) once per device, ozg_use_device(d)
// to pick the one the following calls use (per thread).  Every entry point selects its device
// first: hipSetDevice is per host thread.

namespace {

constexpr int MAX_DEV = 16;
#define OTZ_BATCH_HOST_SNIFFED 0x80000000u // (internal batch flag: the host has looked at the payloads' first bytes; see check_entries)
constexpr int N_SLOTS = 8; // asynchronous staged batches in flight per device (a full table is back-pressure for the caller: OZG_E_NOMEM)

std::mutex g_mu; // contexts, their buffers and slots
#define OTZ_AUX_STREAMS 4
u32 g_chunks = 1; // chunks of a large DEFLATE batch on two internal streams (OTEZIP_GPU_CHUNKS; 0 = cut at a whole number of index-kernel generations): neither measured faster than one launch pair (4.13 ms; 4 chunks 4.43, generation cut 4.14), kept for experiments
ozg_limits g_limits = { 1000ull, 1024ull * 1024ull };
bool g_dbg_index_only = false; // diagnostics (OTEZIP_DEBUG_INDEX_ONLY=1): a DEFLATE batch stops behind its index kernel -- for timing experimental
                               // builds of that kernel whose index the apply kernel could not use; results are NOT produced
u64 *g_prof = nullptr; // diagnostics: per-entry phase cycle counters (ozg_debug_set_profile_buffer)
thread_local char g_err[256] = "";
thread_local int g_cur = -1; // this thread's device (ozg_init / ozg_use_device)
int g_default = -1;          // the device of the first successful ozg_init: what a thread starts with
// Device numbers of this ABI are LOGICAL: OTEZIP_GPU_DEVICE_MAP="0,0,1" makes logical devices 0 and 1
// two independent contexts (buffers, arena, job slots, streams) on physical device 0 and logical 2 one
// on physical 1.  Without the variable logical = physical.  (What it is for: running the multi-device
// paths of the zip layer -- dealing, per-device arenas, job ownership -- on a box with one GPU.)
int g_phys[16];
int g_nlogical = -1; // -1: map not read yet; 0: no map
int phys_of(int d) { return g_nlogical > 0 ? g_phys[d] : d; }
void read_device_map() {
	if (g_nlogical >= 0) {
		return;
	}
	g_nlogical = 0;
	const char *m = getenv ("OTEZIP_GPU_DEVICE_MAP");
	if (!m || !*m) {
		return;
	}
	int n = 0;
	while (*m && n < 16) {
		char *end = nullptr;
		long v = strtol (m, &end, 10);
		if (end == m || v < 0 || v > 1023) {
			break;
		}
		g_phys[n++] = (int)v;
		m = *end == ',' ? end + 1 : end;
		if (*end != ',' ) {
			break;
		}
	}
	g_nlogical = n;
}

struct staging {
	void *d_src = nullptr, *d_dst = nullptr, *d_ent = nullptr, *d_res = nullptr, *d_pack = nullptr;
	size_t src_cap = 0, dst_cap = 0, ent_cap = 0, res_cap = 0, pack_cap = 0;
};

// Index arena of the DEFLATE path; batches on different streams are ordered through `done`.
struct workspace {
	u8 *arena = nullptr;
	size_t arena_cap = 0;
	otz_region *regions = nullptr;
	otz_ix_head *heads = nullptr;
	size_t n_cap = 0;
	unsigned long long *cursor = nullptr;
	u32 *ovf = nullptr; // overflow list: count, then entry numbers
	hipEvent_t done = nullptr;
	bool busy = false;
	hipStream_t aux[OTZ_AUX_STREAMS] = {};
	hipEvent_t fork = nullptr, join[OTZ_AUX_STREAMS] = {};
	// Zstandard: literal scratch of the persistent grid, its entry counter, the routing bytes
	u8 *zs_lit = nullptr;
	u32 zs_grid = 0;
	u32 *zs_cursor = nullptr;
	u8 *zs_route = nullptr;
	size_t zs_route_cap = 0;
	// LZMA1: probability-model scratch of the persistent grid
	u8 *lz_probs = nullptr;
	u32 lz_threads = 0;
	u32 *lz_cursor = nullptr;
	u8 *br_arena = nullptr; // Brotli: OTZ_BR_ARENA_SMALL bytes per work-item of the first pass,
	u32 br_threads = 0;
	u8 *br_big = nullptr;   // OTZ_BR_ARENA for each of the OTZ_BR_BIG_THREADS of the second
	u32 *br_cursor = nullptr;
	u8 *br_blob = nullptr;  // the RFC's constant data on this device
	// encode side: token scratch of the persistent grid, its buffer counter
	u32 *enc_tok = nullptr;
	u32 enc_grid = 0;
	u32 *enc_cursor = nullptr;
	hipEvent_t enc_done = nullptr;
	bool enc_busy = false;
};

} // namespace

// an asynchronous staged batch: H2D, kernels, D2H on a stream of its own
struct ozg_job {
	int dev = -1;
	bool busy = false;
	hipStream_t stream = nullptr;
	hipEvent_t done = nullptr;
	hipEvent_t up = nullptr; // decode jobs: the input is on the device
	staging st;
	void *h_ent = nullptr; // pinned copy of the descriptor table (+ 64 bytes: the packed total comes back there)
	size_t h_ent_cap = 0;
	// packed encode batches finish in two steps: results + total first (mid), then exactly `total` bytes
	bool two_step = false;
	hipEvent_t mid = nullptr;
	void *h_dst = nullptr;
	uint64_t dst_len = 0;
	size_t total_at = 0; // byte offset of the u64 total inside h_ent
};

namespace {

struct dev_ctx {
	bool ready = false;
	u32 gen = 4096; // index wavefronts resident on the chip at once
	u32 enc_grid = 2048; // encode wavefronts resident on the chip at once (LDS-bound)
	u32 zs_grid = 3840;  // Zstandard wavefronts resident on the chip at once
	staging stage; // synchronous host entry points
	workspace ws;
	ozg_job slots[N_SLOTS];
	// The copies of the decode jobs of a device go first come, first served: two uploads (or two downloads) of
	// jobs in flight would otherwise share the link, and the job submitted first -- the one its caller waits
	// for -- would finish as late as the second.
	hipEvent_t h2d_tail = nullptr, d2h_tail = nullptr;
} g_dev[MAX_DEV];

int fail_hip(const char *what, hipError_t e) {
	snprintf (g_err, sizeof (g_err), "%s: %s", what, hipGetErrorString (e));
	return OZG_E_HIP;
}

#define OZG_HIP(call) \
	do { \
		hipError_t _e = (call); \
		if (_e != hipSuccess) { \
			return fail_hip (#call, _e); \
		} \
	} while (0)

// the calling thread's context; selects its device
int need_ctx(dev_ctx **out) {
	int d = g_cur >= 0 ? g_cur : g_default;
	if (d < 0 || !g_dev[d].ready) {
		snprintf (g_err, sizeof (g_err), "ozg_init() has not succeeded: no HIP device, and this library has no CPU path");
		return OZG_E_NODEVICE;
	}
	OZG_HIP (hipSetDevice (phys_of (d)));
	g_cur = d;
	*out = &g_dev[d];
	return OZG_OK;
}

int grow(void **p, size_t *cap, size_t want) {
	if (want <= *cap && *p) {
		return OZG_OK;
	}
	if (*p) {
		(void)hipFree (*p);
		*p = nullptr;
		*cap = 0;
	}
	size_t sz = want < 4096 ? 4096 : want;
	OZG_HIP (hipMalloc (p, sz));
	*cap = sz;
	return OZG_OK;
}

void staging_free(staging &s) {
	(void)hipFree (s.d_src);
	(void)hipFree (s.d_dst);
	(void)hipFree (s.d_ent);
	(void)hipFree (s.d_res);
	(void)hipFree (s.d_pack);
	s = staging ();
}

int staging_reserve(staging &s, size_t src, size_t dst, size_t ent, size_t res) {
	int rc;
	if ((rc = grow (&s.d_src, &s.src_cap, src)) != OZG_OK || (rc = grow (&s.d_dst, &s.dst_cap, dst)) != OZG_OK ||
		(rc = grow (&s.d_ent, &s.ent_cap, ent)) != OZG_OK || (rc = grow (&s.d_res, &s.res_cap, res)) != OZG_OK) {
		return rc;
	}
	return OZG_OK;
}

// Room for the index of a batch: every entry asks for otz_ix_region_bytes(); their sum is
// bounded by the buffer sizes (dst regions do not overlap).  Entries that find the arena
// full are decoded by the self-contained kernel instead.
int ws_reserve(workspace &ws, uint64_t src_len, uint64_t dst_len, uint32_t n) {
	// (the checkpoint rows of an entry: 16 bytes per 64 bits of its payload and a fixed part, at most OTZ_K1_TEMP_MAX)
	const size_t per_entry = 4096u + 256u + (OTZ_K1_ROWS_B + 3u) * 512u;
	const size_t temp_max = (size_t)n * OTZ_K1_TEMP_MAX, temp_est = (size_t) (src_len << 1) + (size_t)n * 4096u;
	size_t want = (size_t)dst_len + (size_t) (src_len >> 1) + (size_t)n * per_entry + (temp_est < temp_max ? temp_est : temp_max);
	if (want > ((size_t)8 << 30)) {
		want = (size_t)8 << 30; // beyond that, the overflow path takes what does not fit
	}
	if (!ws.done) {
		OZG_HIP (hipEventCreateWithFlags (&ws.done, hipEventDisableTiming));
	}
	if (!ws.cursor) {
		OZG_HIP (hipMalloc ((void **)&ws.cursor, 256));
	}
	if (!ws.fork) {
		OZG_HIP (hipEventCreateWithFlags (&ws.fork, hipEventDisableTiming));
		for (int k = 0; k < OTZ_AUX_STREAMS; k++) {
			OZG_HIP (hipStreamCreateWithFlags (&ws.aux[k], hipStreamNonBlocking));
			OZG_HIP (hipEventCreateWithFlags (&ws.join[k], hipEventDisableTiming));
		}
	}
	if (want > ws.arena_cap || n > ws.n_cap) {
		OZG_HIP (hipDeviceSynchronize ()); // earlier batches may still read what is about to be freed
		ws.busy = false;
	}
	if (want > ws.arena_cap) {
		if (ws.arena) {
			(void)hipFree (ws.arena);
			ws.arena = nullptr;
			ws.arena_cap = 0;
		}
		want += want >> 2; // (some head room: the next, slightly larger batch should not reallocate)
		OZG_HIP (hipMalloc ((void **)&ws.arena, want));
		ws.arena_cap = want;
	}
	if (n > ws.n_cap) {
		(void)hipFree (ws.regions);
		(void)hipFree (ws.heads);
		(void)hipFree (ws.ovf);
		ws.regions = nullptr;
		ws.heads = nullptr;
		ws.ovf = nullptr;
		ws.n_cap = 0;
		const size_t cap = n < 1024u ? 1024u : n + (n >> 2);
		OZG_HIP (hipMalloc ((void **)&ws.regions, cap * sizeof (otz_region)));
		OZG_HIP (hipMalloc ((void **)&ws.heads, cap * sizeof (otz_ix_head)));
		OZG_HIP (hipMalloc ((void **)&ws.ovf, (cap + 1u) * sizeof (u32)));
		ws.n_cap = cap;
	}
	return OZG_OK;
}

void ws_free(workspace &ws) {
	void *const owned[] = { ws.arena, ws.regions, ws.heads, ws.cursor, ws.ovf, ws.enc_tok, ws.enc_cursor, ws.zs_lit, ws.zs_cursor, ws.zs_route, ws.lz_probs, ws.lz_cursor, ws.br_arena, ws.br_big, ws.br_cursor, ws.br_blob };
	for (size_t k = 0; k < sizeof (owned) / sizeof (owned[0]); k++) {
		const hipError_t e = hipFree (owned[k]);
		if (e != hipSuccess && getenv ("OTEZIP_TRACE")) {
			fprintf (stderr, "[otezip-amd] hipFree of workspace block %d (%p): %s\n", (int)k, owned[k], hipGetErrorString (e));
		}
	}
	if (ws.enc_done) {
		(void)hipEventDestroy (ws.enc_done);
	}
	if (ws.done) {
		(void)hipEventDestroy (ws.done);
	}
	if (ws.fork) {
		(void)hipEventDestroy (ws.fork);
		for (int k = 0; k < OTZ_AUX_STREAMS; k++) {
			(void)hipStreamDestroy (ws.aux[k]);
			(void)hipEventDestroy (ws.join[k]);
		}
	}
	ws = workspace ();
}

hipEvent_t ws_done_event(dev_ctx &C) { return C.ws.done; }

// the launches of one decode batch on stream s (device pointers); the caller holds g_mu
int launch_decode(dev_ctx &C, const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len, const ozg_entry *d_entries,
	ozg_result *d_results, uint32_t n, hipStream_t s, uint32_t batch_flags, u64 *d_end_bits = nullptr) {
	int rc;
	{ // an error some earlier, unchecked call of this thread left behind (here or in another library) is not this batch's
		const hipError_t stale = hipGetLastError ();
		if (stale != hipSuccess && getenv ("OTEZIP_TRACE")) {
			fprintf (stderr, "[otezip-amd] stale HIP error before a decode batch: %s\n", hipGetErrorString (stale));
		}
	}
	if (!(batch_flags & OZG_BATCH_NO_DEFLATE)) {
		workspace &ws = C.ws;
		if ((rc = ws_reserve (ws, src_len, dst_len, n)) != OZG_OK) {
			return rc;
		}
		if (ws.busy) {
			OZG_HIP (hipStreamWaitEvent (s, ws.done, 0)); // the arena is shared between streams
		}
		OZG_HIP (hipMemsetAsync (ws.cursor, 0, 8, s));
		OZG_HIP (hipMemsetAsync (ws.ovf, 0, 4, s));
		hipLaunchKernelGGL (otz_plan_kernel, dim3 ((n + 255u) / 256u), dim3 (256), 0, s, d_entries, n, (u64)src_len, (u64)dst_len,
			(u64)ws.arena_cap, ws.cursor, ws.regions, ws.heads);
		OZG_HIP (hipGetLastError ());
		// OTEZIP_GPU_CHUNKS=0: cut the batch where a whole number of index-kernel "generations" ends
		// (one wavefront per entry, g_gen of them resident on the chip): the remainder's index
		// kernel, which cannot fill the chip, then runs beside the first part's apply kernel
		u32 n_chunks = n >= 4096u ? g_chunks : 1u;
		u32 cut = 0;
		if (g_chunks == 0) {
			cut = n / C.gen * C.gen;
			n_chunks = cut > 0 && cut < n && n - cut >= 256u ? 2u : 1u;
		}
		if (n_chunks <= 1u) {
			hipLaunchKernelGGL (g_prof ? otz_index_prof_kernel : otz_index_kernel, dim3 (n), dim3 (64), sizeof (otz_k1_lds), s, (const u8 *)d_src, (u8 *)d_dst, d_entries, 0u, n,
				(u64)src_len, (u64)dst_len, g_limits, ws.arena, ws.regions, ws.heads, ws.ovf, g_prof);
			OZG_HIP (hipGetLastError ());
			if (g_dbg_index_only) {
				OZG_HIP (hipEventRecord (ws.done, s));
				ws.busy = true;
				return OZG_OK;
			}
			hipLaunchKernelGGL (otz_apply_kernel, dim3 (n), dim3 (OTZ_A_NT), sizeof (otz_apply_lds), s, (const u8 *)d_src, (u8 *)d_dst,
				d_entries, d_results, 0u, n, (u64)src_len, (u64)dst_len, g_limits, (const u8 *)ws.arena, ws.regions, ws.heads, g_prof, d_end_bits);
			OZG_HIP (hipGetLastError ());
			hipLaunchKernelGGL (otz_out_crc_kernel, dim3 (n), dim3 (256), sizeof (otz_copy_lds), s, (const u8 *)d_dst, d_entries, d_results, 0u, n);
			OZG_HIP (hipGetLastError ());
		} else {
			OZG_HIP (hipEventRecord (ws.fork, s));
			for (int k = 0; k < OTZ_AUX_STREAMS; k++) {
				OZG_HIP (hipStreamWaitEvent (ws.aux[k], ws.fork, 0));
			}
			for (u32 c = 0; c < n_chunks; c++) {
				u32 e0 = (u32) ((u64)n * c / n_chunks), e1 = (u32) ((u64)n * (c + 1u) / n_chunks);
				if (cut) {
					e0 = c ? cut : 0u;
					e1 = c ? n : cut;
				}
				hipStream_t cs = ws.aux[c % OTZ_AUX_STREAMS];
				hipLaunchKernelGGL (otz_index_kernel, dim3 (e1 - e0), dim3 (64), sizeof (otz_k1_lds), cs, (const u8 *)d_src, (u8 *)d_dst, d_entries, e0, n,
					(u64)src_len, (u64)dst_len, g_limits, ws.arena, ws.regions, ws.heads, ws.ovf, g_prof);
				OZG_HIP (hipGetLastError ());
				hipLaunchKernelGGL (otz_apply_kernel, dim3 (e1 - e0), dim3 (OTZ_A_NT), sizeof (otz_apply_lds), cs, (const u8 *)d_src, (u8 *)d_dst,
					d_entries, d_results, e0, n, (u64)src_len, (u64)dst_len, g_limits, (const u8 *)ws.arena, ws.regions, ws.heads, g_prof, d_end_bits);
				OZG_HIP (hipGetLastError ());
				hipLaunchKernelGGL (otz_out_crc_kernel, dim3 (e1 - e0), dim3 (256), sizeof (otz_copy_lds), cs, (const u8 *)d_dst, d_entries, d_results, e0, n);
				OZG_HIP (hipGetLastError ());
			}
			for (int k = 0; k < OTZ_AUX_STREAMS; k++) {
				OZG_HIP (hipEventRecord (ws.join[k], ws.aux[k]));
				OZG_HIP (hipStreamWaitEvent (s, ws.join[k], 0));
			}
		}
		hipLaunchKernelGGL (otz_inflate_kernel, dim3 (n < 256u ? n : 256u), dim3 (OTZ_NT), sizeof (otz_inflate_lds), s, (const u8 *)d_src,
			(u8 *)d_dst, d_entries, d_results, g_limits, (const u32 *)ws.ovf);
		OZG_HIP (hipGetLastError ());
		OZG_HIP (hipEventRecord (ws.done, s));
		ws.busy = true;
	}
	if (!(batch_flags & OZG_BATCH_ONLY_DEFLATE)) {
		const u8 *route = nullptr;
		bool want_zs = !(batch_flags & OZG_BATCH_NO_ZSTD), want_lz = !(batch_flags & OZG_BATCH_NO_LZMA), want_br = !(batch_flags & OZG_BATCH_NO_BROTLI);
		bool want_lf = !(batch_flags & OZG_BATCH_NO_LZFSE);
		u32 have[4] = { n, n, n, n }; // entries each real-format kernel would take, at most
		if ((want_zs || want_lz || want_br || want_lf) && !(batch_flags & OTZ_BATCH_HOST_SNIFFED)) {
			// What the batch holds is counted where the payloads are (a caller that staged them from host memory has
			// said so through batch_flags; for the rest this costs one small kernel and 16 bytes back): a format
			// none of whose streams is in the batch gets no kernel and no scratch, the others scratch for what there is.
			workspace &ws = C.ws;
			if (!ws.zs_cursor) {
				OZG_HIP (hipMalloc ((void **)&ws.zs_cursor, 256));
				OZG_HIP (hipMalloc ((void **)&ws.lz_cursor, 256));
				OZG_HIP (hipMalloc ((void **)&ws.br_cursor, 256));
			}
			if (ws.busy) {
				OZG_HIP (hipStreamWaitEvent (s, ws.done, 0)); // (the counters are shared between streams)
			}
			OZG_HIP (hipMemsetAsync (ws.zs_cursor + 16, 0, 16, s));
			hipLaunchKernelGGL (otz_sniff_kernel, dim3 ((n + 255u) / 256u), dim3 (256), 0, s, (const u8 *)d_src, d_entries, n, (u64)src_len, (u64)dst_len, g_limits,
				ws.zs_cursor + 16);
			OZG_HIP (hipGetLastError ());
			OZG_HIP (hipMemcpyAsync (have, ws.zs_cursor + 16, sizeof (have), hipMemcpyDeviceToHost, s));
			OZG_HIP (hipStreamSynchronize (s));
			want_zs = want_zs && have[0] != 0;
			want_lz = want_lz && have[1] != 0;
			want_br = want_br && have[2] != 0;
			want_lf = want_lf && have[3] != 0;
		}
		if (want_zs || want_lz || want_br || want_lf) {
			// methods 93 and 14: the real formats first (the rest of the batch, and what turns out not to
			// be one of them, is the copy kernel's)
			workspace &ws = C.ws;
			const u32 grid = have[0] < C.zs_grid ? have[0] : C.zs_grid;
			const u32 lz_threads = ((have[1] < 32768u ? have[1] : 32768u) + 63u) & ~63u;
			const u32 br_n = have[2] > have[3] ? have[2] : have[3];
			const u32 br_threads = ((br_n < 16384u ? br_n : 16384u) + 63u) & ~63u;
			if (!ws.done) {
				OZG_HIP (hipEventCreateWithFlags (&ws.done, hipEventDisableTiming));
			}
			if (!ws.zs_cursor) {
				OZG_HIP (hipMalloc ((void **)&ws.zs_cursor, 256));
				OZG_HIP (hipMalloc ((void **)&ws.lz_cursor, 256));
				OZG_HIP (hipMalloc ((void **)&ws.br_cursor, 256));
			}
			if ((want_zs && grid > ws.zs_grid) || n > ws.zs_route_cap || (want_lz && lz_threads > ws.lz_threads) || ((want_br || want_lf) && br_threads > ws.br_threads)) {
				OZG_HIP (hipDeviceSynchronize ()); // an earlier batch may still use what is about to be freed
				ws.busy = false;
			}
			if (want_zs && grid > ws.zs_grid) {
				(void)hipFree (ws.zs_lit);
				ws.zs_lit = nullptr;
				ws.zs_grid = 0;
				const u32 g = grid < 64u ? 64u : (grid + (grid >> 1) < C.zs_grid ? grid + (grid >> 1) : C.zs_grid);
				OZG_HIP (hipMalloc ((void **)&ws.zs_lit, (size_t)g * OTZ_ZS_LIT_BYTES));
				ws.zs_grid = g;
			}
			if (want_lz && lz_threads > ws.lz_threads) {
				(void)hipFree (ws.lz_probs);
				ws.lz_probs = nullptr;
				ws.lz_threads = 0;
				const u32 t = lz_threads < 1024u ? 1024u : (lz_threads + (lz_threads >> 1) < 32768u ? (lz_threads + (lz_threads >> 1) + 63u) & ~63u : 32768u);
				OZG_HIP (hipMalloc ((void **)&ws.lz_probs, (size_t)t * OTZ_LZMA_PROB_BYTES));
				ws.lz_threads = t;
			}
			if ((want_br || want_lf) && br_threads > ws.br_threads) {
				(void)hipFree (ws.br_arena);
				ws.br_arena = nullptr;
				ws.br_threads = 0;
				const u32 t = br_threads < 1024u ? 1024u : (br_threads + (br_threads >> 1) < 16384u ? (br_threads + (br_threads >> 1) + 63u) & ~63u : 16384u);
				OZG_HIP (hipMalloc ((void **)&ws.br_arena, (size_t)t * OTZ_BR_ARENA_SMALL));
				ws.br_threads = t;
			}
			if (want_br && !ws.br_big) {
				OZG_HIP (hipMalloc ((void **)&ws.br_big, (size_t)OTZ_BR_BIG_THREADS * OTZ_BR_ARENA));
			}
			if (want_br && !ws.br_blob) {
				const size_t bytes = (size_t) (otz_brotli_blob_end - otz_brotli_blob);
				OZG_HIP (hipMalloc ((void **)&ws.br_blob, bytes));
				OZG_HIP (hipMemcpy (ws.br_blob, otz_brotli_blob, bytes, hipMemcpyHostToDevice));
			}
			if (n > ws.zs_route_cap) {
				(void)hipFree (ws.zs_route);
				ws.zs_route = nullptr;
				ws.zs_route_cap = 0;
				const size_t cap = n < 1024u ? 1024u : n + (n >> 2);
				OZG_HIP (hipMalloc ((void **)&ws.zs_route, cap));
				ws.zs_route_cap = cap;
			}
			if (ws.busy) {
				OZG_HIP (hipStreamWaitEvent (s, ws.done, 0)); // the scratch is shared between streams
			}
			OZG_HIP (hipMemsetAsync (ws.zs_cursor, 0, 4, s));
			OZG_HIP (hipMemsetAsync (ws.lz_cursor, 0, 4, s));
			OZG_HIP (hipMemsetAsync (ws.br_cursor, 0, 12, s)); // (one counter per Brotli pass, one for LZFSE)
			OZG_HIP (hipMemsetAsync (ws.zs_route, 0, n, s));
			if (want_zs) {
				if (g_prof) {
					hipLaunchKernelGGL (otz_zstd_prof_kernel, dim3 (grid), dim3 (64), sizeof (otz_zs_lds), s, (const u8 *)d_src, (u8 *)d_dst, d_entries, d_results, n,
						(u64)src_len, (u64)dst_len, g_limits, ws.zs_lit, ws.zs_cursor, ws.zs_route, g_prof);
				} else {
					hipLaunchKernelGGL (otz_zstd_kernel, dim3 (grid), dim3 (64), sizeof (otz_zs_lds), s, (const u8 *)d_src, (u8 *)d_dst, d_entries, d_results, n,
						(u64)src_len, (u64)dst_len, g_limits, ws.zs_lit, ws.zs_cursor, ws.zs_route);
				}
				OZG_HIP (hipGetLastError ());
			}
			if (want_lz) {
				hipLaunchKernelGGL (otz_lzma_kernel, dim3 (lz_threads / 64u), dim3 (64), 0, s, (const u8 *)d_src, (u8 *)d_dst, d_entries, d_results, n,
					(u64)src_len, (u64)dst_len, g_limits, ws.lz_probs, ws.lz_cursor, ws.zs_route);
				OZG_HIP (hipGetLastError ());
			}
			if (want_br) {
				hipLaunchKernelGGL (otz_brotli_kernel, dim3 (br_threads / 64u), dim3 (64), 0, s, (const u8 *)d_src, (u8 *)d_dst, d_entries, d_results, n,
					(u64)src_len, (u64)dst_len, g_limits, ws.br_arena, OTZ_BR_ARENA_SMALL, (const u8 *)ws.br_blob, ws.br_cursor, ws.zs_route, 0u);
				OZG_HIP (hipGetLastError ());
				hipLaunchKernelGGL (otz_brotli_kernel, dim3 (OTZ_BR_BIG_THREADS / 64u), dim3 (64), 0, s, (const u8 *)d_src, (u8 *)d_dst, d_entries, d_results, n,
					(u64)src_len, (u64)dst_len, g_limits, ws.br_big, OTZ_BR_ARENA, (const u8 *)ws.br_blob, ws.br_cursor + 1, ws.zs_route, 1u);
				OZG_HIP (hipGetLastError ());
			}
			if (want_lf) {
				hipLaunchKernelGGL (otz_lzfse_kernel, dim3 (br_threads / 64u), dim3 (64), 0, s, (const u8 *)d_src, (u8 *)d_dst, d_entries, d_results, n,
					(u64)src_len, (u64)dst_len, g_limits, ws.br_arena, ws.br_cursor + 2, ws.zs_route);
				OZG_HIP (hipGetLastError ());
			}
			route = ws.zs_route;
		}
		hipLaunchKernelGGL (otz_copy_kernel, dim3 (n), dim3 (256), sizeof (otz_copy_lds), s, (const u8 *)d_src,
			(u8 *)d_dst, d_entries, d_results, n, (u64)src_len, (u64)dst_len, g_limits, route);
		OZG_HIP (hipGetLastError ());
		if (route) {
			hipLaunchKernelGGL (otz_zstd_crc_kernel, dim3 (n), dim3 (256), sizeof (otz_copy_lds), s, (u8 *)d_dst, d_entries, d_results, n, route);
			OZG_HIP (hipGetLastError ());
			OZG_HIP (hipEventRecord (ws_done_event (C), s));
			C.ws.busy = true;
		}
	}
	return OZG_OK;
}

// host-side view of a descriptor table: bounds, and which launches it needs
// h_src (optional): the payloads, when they are in host memory -- then the kernels of the real formats (Zstandard,
// LZMA1, Brotli, LZFSE) are only asked for when some payload LOOKS like one (the same first-bytes tests the
// kernels make: otz_*_is_candidate), not merely because an entry has the method id: an archive of the
// reference's own framings launches none of them and allocates none of their scratch.
int check_entries(const ozg_entry *entries, uint32_t n, uint64_t src_len, uint64_t dst_len, uint32_t *flags, const uint8_t *h_src = nullptr) {
	uint32_t f = OZG_BATCH_NO_DEFLATE | OZG_BATCH_ONLY_DEFLATE | OZG_BATCH_NO_ZSTD | OZG_BATCH_NO_LZMA | OZG_BATCH_NO_BROTLI | OZG_BATCH_NO_LZFSE;
	for (uint32_t i = 0; i < n; i++) {
		const ozg_entry &e = entries[i];
		if (e.src_off > src_len || e.comp_size > src_len - e.src_off || (e.dst_off & 15u) || e.dst_off > dst_len ||
			e.uncomp_size > dst_len - e.dst_off) {
			snprintf (g_err, sizeof (g_err), "entry %u: offsets outside the buffers or dst_off not a multiple of 16", i);
			return OZG_E_ARG;
		}
		f &= e.method == OZG_METHOD_DEFLATE ? ~OZG_BATCH_NO_DEFLATE : ~OZG_BATCH_ONLY_DEFLATE;
		const uint8_t *p = h_src ? h_src + e.src_off : nullptr;
		const uint32_t m = e.comp_size;
		if (e.method == OZG_METHOD_ZSTD) {
			const uint32_t magic = p && m >= 5u ? (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24) : 0u;
			if (!p || (m >= 5u && ((magic & 0xFFFFFFF0u) == 0x184D2A50u || (magic == 0xFD2FB528u && (p[4] & 0x10u) == 0)))) {
				f &= ~OZG_BATCH_NO_ZSTD;
			}
		}
		if (e.method == OZG_METHOD_LZMA && (!p || (m >= 14u && p[2] == 5u && p[3] == 0 && p[4] < 225u))) {
			f &= ~OZG_BATCH_NO_LZMA;
		}
		if (e.method == OZG_METHOD_BROTLI && (!p || (m >= 1u && !(m >= 4u && p[0] == 'B' && p[1] == 'R' && p[2] == 'O' && p[3] == 'T')))) {
			f &= ~OZG_BATCH_NO_BROTLI;
		}
		if (e.method == OZG_METHOD_LZFSE && (!p || (m >= 4u && p[0] == 'b' && p[1] == 'v' && p[2] == 'x'))) {
			f &= ~OZG_BATCH_NO_LZFSE;
		}
	}
	*flags = f | (h_src ? OTZ_BATCH_HOST_SNIFFED : 0u);
	return OZG_OK;
}

} // namespace

extern "C" {

const char *ozg_last_error(void) { return g_err; }

int ozg_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount (&n) != hipSuccess) {
		return 0;
	}
	std::lock_guard<std::mutex> lk (g_mu);
	read_device_map ();
	if (g_nlogical > 0) {
		for (int k = 0; k < g_nlogical; k++) {
			if (g_phys[k] >= n) {
				return k; // (logical devices up to the first one that names a GPU the node does not have)
			}
		}
		return g_nlogical;
	}
	return n;
}

int ozg_init(int device) {
	std::lock_guard<std::mutex> lk (g_mu);
	int n = 0;
	hipError_t e = hipGetDeviceCount (&n);
	if (e != hipSuccess || n <= 0) {
		snprintf (g_err, sizeof (g_err), "no HIP device available (%s); otezip-amd has no CPU fallback",
			e == hipSuccess ? "device count 0" : hipGetErrorString (e));
		return OZG_E_NODEVICE;
	}
	read_device_map ();
	if (device < 0 || device >= MAX_DEV || (g_nlogical > 0 ? device >= g_nlogical || g_phys[device] >= n : device >= n)) {
		snprintf (g_err, sizeof (g_err), "device %d out of range (%d HIP device(s)%s)", device, n, g_nlogical > 0 ? ", OTEZIP_GPU_DEVICE_MAP in force" : "");
		return OZG_E_ARG;
	}
	OZG_HIP (hipSetDevice (phys_of (device)));
	if (!g_dev[device].ready) {
		// (function attributes are per device)
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_apply_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_out_crc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_copy_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_inflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_inflate_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_index_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_k1_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_index_prof_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_k1_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_copy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_copy_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_crc32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_copy_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_enc_crc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_copy_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_enc_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_encode_prof_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			(int)sizeof (otz_enc_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_zstd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof (otz_zs_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_zstd_prof_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof (otz_zs_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_zstd_crc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof (otz_copy_lds)));
		OZG_HIP (hipFuncSetAttribute ((const void *)otz_checksum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof (otz_copy_lds)));
		int cus = 256, per_cu = 16;
		(void)hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, phys_of (device));
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, otz_index_kernel, 64, sizeof (otz_k1_lds));
		g_dev[device].gen = (u32) (cus > 0 ? cus : 256) * (u32) (per_cu > 0 ? per_cu : 16);
		per_cu = 8;
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, otz_encode_kernel, 64, sizeof (otz_enc_lds));
		g_dev[device].enc_grid = (u32) (cus > 0 ? cus : 256) * (u32) (per_cu > 0 ? per_cu : 8);
		per_cu = 12;
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, otz_zstd_kernel, 64, sizeof (otz_zs_lds));
		g_dev[device].zs_grid = (u32) (cus > 0 ? cus : 256) * (u32) (per_cu > 0 ? per_cu : 12);
		g_dev[device].ready = true;
	}
	if (const char *c = getenv ("OTEZIP_DEBUG_INDEX_ONLY")) {
		g_dbg_index_only = atoi (c) != 0;
	}
	if (const char *c = getenv ("OTEZIP_GPU_CHUNKS")) {
		const int v = atoi (c);
		g_chunks = v < 0 ? 1u : v > 64 ? 64u : (u32)v;
	}
	g_cur = device;
	if (g_default < 0) {
		g_default = device;
	}
	g_err[0] = 0;
	return OZG_OK;
}

int ozg_use_device(int device) {
	if (device < 0 || device >= MAX_DEV || !g_dev[device].ready) {
		snprintf (g_err, sizeof (g_err), "device %d has not been initialised with ozg_init()", device);
		return OZG_E_ARG;
	}
	OZG_HIP (hipSetDevice (phys_of (device)));
	g_cur = device;
	return OZG_OK;
}

int ozg_current_device(void) { return g_cur >= 0 ? g_cur : g_default; }

void ozg_shutdown(void) {
	std::lock_guard<std::mutex> lk (g_mu);
	for (int d = 0; d < MAX_DEV; d++) {
		dev_ctx &C = g_dev[d];
		if (!C.ready) {
			continue;
		}
		if (hipSetDevice (phys_of (d)) != hipSuccess) {
			continue;
		}
		(void)hipDeviceSynchronize ();
		staging_free (C.stage);
		ws_free (C.ws);
		for (int k = 0; k < N_SLOTS; k++) {
			ozg_job &j = C.slots[k];
			staging_free (j.st);
			if (j.h_ent) {
				(void)hipHostFree (j.h_ent);
			}
			if (j.stream) {
				(void)hipStreamDestroy (j.stream);
			}
			if (j.done) {
				(void)hipEventDestroy (j.done);
			}
			if (j.mid) {
				(void)hipEventDestroy (j.mid);
			}
			if (j.up) {
				(void)hipEventDestroy (j.up);
			}
			j = ozg_job ();
		}
		C.h2d_tail = C.d2h_tail = nullptr; // (they were events of the slots)
		C.ready = false;
	}
	g_default = -1;
	g_cur = -1;
}

void ozg_set_limits(const ozg_limits *lim) {
	if (lim) {
		g_limits = *lim;
	}
}

int ozg_decode_batch_device_ex(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_entry *d_entries, ozg_result *d_results, uint32_t n, void *stream, uint32_t batch_flags) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	if (n == 0) {
		return OZG_OK;
	}
	if (!d_src || !d_dst || !d_entries || !d_results) {
		snprintf (g_err, sizeof (g_err), "null device pointer");
		return OZG_E_ARG;
	}
	if (((uintptr_t)d_dst & 15u) != 0) {
		snprintf (g_err, sizeof (g_err), "d_dst must be 16-byte aligned");
		return OZG_E_ARG;
	}
	std::lock_guard<std::mutex> lk (g_mu);
	return launch_decode (*C, d_src, src_len, d_dst, dst_len, d_entries, d_results, n, (hipStream_t)stream, batch_flags & ~OTZ_BATCH_HOST_SNIFFED); // (nobody has looked at device-resident payloads)
}

int ozg_decode_batch_device(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_entry *d_entries, ozg_result *d_results, uint32_t n, void *stream) {
	return ozg_decode_batch_device_ex (d_src, src_len, d_dst, dst_len, d_entries, d_results, n, stream, 0);
}

int ozg_decode_batch_host(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *results, uint32_t n) {
	return ozg_decode_batch_host_ex (src, src_len, dst, dst_len, entries, results, n, nullptr);
}

int ozg_decode_batch_host_ex(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *results, uint32_t n, uint64_t *end_bits) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	if (n == 0) {
		return OZG_OK;
	}
	if (!src || !dst || !entries || !results) {
		snprintf (g_err, sizeof (g_err), "null pointer");
		return OZG_E_ARG;
	}
	uint32_t flags;
	if ((rc = check_entries (entries, n, src_len, dst_len, &flags, (const uint8_t *)src)) != OZG_OK) {
		return rc;
	}
	std::lock_guard<std::mutex> lk (g_mu);
	staging &S = C->stage;
	// (results and, behind them, the optional end positions share one staging buffer)
	const size_t res_bytes = (size_t)n * sizeof (ozg_result);
	if ((rc = staging_reserve (S, src_len + 64, dst_len + 64, (size_t)n * sizeof (ozg_entry), res_bytes + (size_t)n * 8u)) != OZG_OK) {
		return rc;
	}
	u64 *d_end = end_bits ? (u64 *) ((u8 *)S.d_res + res_bytes) : nullptr;
	if (d_end) {
		OZG_HIP (hipMemsetAsync (d_end, 0, (size_t)n * 8u, nullptr));
	}
	OZG_HIP (hipMemcpyAsync (S.d_src, src, src_len, hipMemcpyHostToDevice, nullptr));
	OZG_HIP (hipMemcpyAsync (S.d_ent, entries, (size_t)n * sizeof (ozg_entry), hipMemcpyHostToDevice, nullptr));
	rc = launch_decode (*C, S.d_src, src_len, S.d_dst, dst_len, (const ozg_entry *)S.d_ent, (ozg_result *)S.d_res, n, nullptr, flags, d_end);
	if (rc != OZG_OK) {
		return rc;
	}
	if (d_end) {
		OZG_HIP (hipMemcpyAsync (end_bits, d_end, (size_t)n * 8u, hipMemcpyDeviceToHost, nullptr));
	}
	OZG_HIP (hipMemcpyAsync (dst, S.d_dst, dst_len, hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipMemcpyAsync (results, S.d_res, (size_t)n * sizeof (ozg_result), hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipStreamSynchronize (nullptr));
	return OZG_OK;
}

// ---- asynchronous staged batches: the ingest / egress pipeline of the zip layer

static int check_enc_entries(const ozg_enc_entry *entries, uint32_t n, uint64_t src_len, uint64_t dst_len) {
	for (uint32_t i = 0; i < n; i++) {
		const ozg_enc_entry &e = entries[i];
		const bool codec_only = (e.reserved & OZG_ENC_F_CODEC_ONLY) != 0;
		if (e.src_off > src_len || e.src_size > src_len - e.src_off || (e.dst_off & 15u) || e.dst_off > dst_len ||
			e.dst_cap > dst_len - e.dst_off || (e.dst_cap < e.src_size && !codec_only)) {
			snprintf (g_err, sizeof (g_err), "encode entry %u: offsets outside the buffers, dst_off not a multiple of 16, or dst_cap < src_size", i);
			return OZG_E_ARG;
		}
	}
	return OZG_OK;
}


int ozg_decode_submit(const void *h_src, uint64_t src_len, void *h_dst, uint64_t dst_len, const ozg_entry *entries,
	ozg_result *h_results, uint32_t n, ozg_job **job) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	if (!job || !h_src || !h_dst || !entries || !h_results || n == 0) {
		snprintf (g_err, sizeof (g_err), "null pointer or empty batch");
		return OZG_E_ARG;
	}
	uint32_t flags;
	if ((rc = check_entries (entries, n, src_len, dst_len, &flags, (const uint8_t *)h_src)) != OZG_OK) {
		return rc;
	}
	std::lock_guard<std::mutex> lk (g_mu);
	ozg_job *J = nullptr;
	for (int k = 0; k < N_SLOTS; k++) {
		if (!C->slots[k].busy) {
			J = &C->slots[k];
			break;
		}
	}
	if (!J) {
		snprintf (g_err, sizeof (g_err), "all %d staged-batch slots of the device are in flight", N_SLOTS);
		return OZG_E_NOMEM;
	}
	if (!J->stream) {
		OZG_HIP (hipStreamCreateWithFlags (&J->stream, hipStreamNonBlocking));
		OZG_HIP (hipEventCreateWithFlags (&J->done, hipEventDisableTiming));
	}
	J->dev = g_cur;
	const size_t ent_bytes = (size_t)n * sizeof (ozg_entry);
	if ((rc = staging_reserve (J->st, src_len + 64, dst_len + 64, ent_bytes, (size_t)n * sizeof (ozg_result))) != OZG_OK) {
		return rc;
	}
	if (ent_bytes > J->h_ent_cap) {
		if (J->h_ent) {
			(void)hipHostFree (J->h_ent);
			J->h_ent = nullptr;
			J->h_ent_cap = 0;
		}
		OZG_HIP (hipHostMalloc (&J->h_ent, ent_bytes + (ent_bytes >> 1), hipHostMallocDefault));
		J->h_ent_cap = ent_bytes + (ent_bytes >> 1);
	}
	memcpy (J->h_ent, entries, ent_bytes); // (the caller's table may be gone before the copy runs)
	hipStream_t s = J->stream;
	if (!J->up) {
		OZG_HIP (hipEventCreateWithFlags (&J->up, hipEventDisableTiming));
	}
	if (C->h2d_tail) {
		OZG_HIP (hipStreamWaitEvent (s, C->h2d_tail, 0));
	}
	OZG_HIP (hipMemcpyAsync (J->st.d_src, h_src, src_len, hipMemcpyHostToDevice, s));
	OZG_HIP (hipMemcpyAsync (J->st.d_ent, J->h_ent, ent_bytes, hipMemcpyHostToDevice, s));
	OZG_HIP (hipEventRecord (J->up, s));
	C->h2d_tail = J->up;
	rc = launch_decode (*C, J->st.d_src, src_len, J->st.d_dst, dst_len, (const ozg_entry *)J->st.d_ent, (ozg_result *)J->st.d_res, n, s, flags);
	if (rc != OZG_OK) {
		return rc;
	}
	if (C->d2h_tail) {
		OZG_HIP (hipStreamWaitEvent (s, C->d2h_tail, 0));
	}
	OZG_HIP (hipMemcpyAsync (h_dst, J->st.d_dst, dst_len, hipMemcpyDeviceToHost, s));
	OZG_HIP (hipMemcpyAsync (h_results, J->st.d_res, (size_t)n * sizeof (ozg_result), hipMemcpyDeviceToHost, s));
	OZG_HIP (hipEventRecord (J->done, s));
	C->d2h_tail = J->done;
	J->busy = true;
	*job = J;
	return OZG_OK;
}

// the launches of one encode batch on stream s (device pointers); the caller holds g_mu
static int launch_encode(dev_ctx &C, const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len, const ozg_enc_entry *d_entries,
	ozg_enc_result *d_results, uint32_t n, hipStream_t s);

// the encode side of the same pipeline: H2D, encode + CRC kernels, D2H on the job's stream.
// packed: the payloads come back back to back (entry i at the running sum of its predecessors'
// comp_size, each rounded up to 16) instead of at dst_off[i]: the device packs them first and only
// the bytes that matter cross PCIe (in two steps: results and the total, then the payloads;
// ozg_job_wait runs the second step).
static int encode_submit(const void *h_src, uint64_t src_len, void *h_dst, uint64_t dst_len, const ozg_enc_entry *entries,
	ozg_enc_result *h_results, uint32_t n, ozg_job **job, bool packed) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	if (!job || !h_src || !h_dst || !entries || !h_results || n == 0) {
		snprintf (g_err, sizeof (g_err), "null pointer or empty batch");
		return OZG_E_ARG;
	}
	if ((rc = check_enc_entries (entries, n, src_len, dst_len)) != OZG_OK) {
		return rc;
	}
	std::lock_guard<std::mutex> lk (g_mu);
	ozg_job *J = nullptr;
	for (int k = 0; k < N_SLOTS; k++) {
		if (!C->slots[k].busy) {
			J = &C->slots[k];
			break;
		}
	}
	if (!J) {
		snprintf (g_err, sizeof (g_err), "all %d staged-batch slots of the device are in flight", N_SLOTS);
		return OZG_E_NOMEM;
	}
	if (!J->stream) {
		OZG_HIP (hipStreamCreateWithFlags (&J->stream, hipStreamNonBlocking));
		OZG_HIP (hipEventCreateWithFlags (&J->done, hipEventDisableTiming));
	}
	if (!J->mid) {
		OZG_HIP (hipEventCreateWithFlags (&J->mid, hipEventDisableTiming));
	}
	J->dev = g_cur;
	J->two_step = false;
	const size_t ent_bytes = (size_t)n * sizeof (ozg_enc_entry);
	if ((rc = staging_reserve (J->st, src_len + 64, dst_len + 64, ent_bytes, (size_t)n * sizeof (ozg_enc_result))) != OZG_OK) {
		return rc;
	}
	const size_t off_bytes = ((size_t)n + 1u) * sizeof (u64);
	if (packed && (rc = grow (&J->st.d_pack, &J->st.pack_cap, dst_len + 64 + off_bytes + 64)) != OZG_OK) {
		return rc;
	}
	if (ent_bytes + 64 > J->h_ent_cap) {
		if (J->h_ent) {
			(void)hipHostFree (J->h_ent);
			J->h_ent = nullptr;
			J->h_ent_cap = 0;
		}
		OZG_HIP (hipHostMalloc (&J->h_ent, ent_bytes + (ent_bytes >> 1) + 64, hipHostMallocDefault));
		J->h_ent_cap = ent_bytes + (ent_bytes >> 1) + 64;
	}
	memcpy (J->h_ent, entries, ent_bytes);
	hipStream_t s = J->stream;
	OZG_HIP (hipMemcpyAsync (J->st.d_src, h_src, src_len, hipMemcpyHostToDevice, s));
	OZG_HIP (hipMemcpyAsync (J->st.d_ent, J->h_ent, ent_bytes, hipMemcpyHostToDevice, s));
	rc = launch_encode (*C, J->st.d_src, src_len, J->st.d_dst, dst_len, (const ozg_enc_entry *)J->st.d_ent, (ozg_enc_result *)J->st.d_res, n, s);
	if (rc != OZG_OK) {
		return rc;
	}
	if (packed) {
		// offsets live behind the packed payload area (16-byte aligned)
		u8 *pk = (u8 *)J->st.d_pack;
		u64 *d_off = (u64 *) (pk + ((dst_len + 64u + 15u) & ~(uint64_t)15u));
		hipLaunchKernelGGL (otz_pack_scan_kernel, dim3 (1), dim3 (256), 0, s, (const ozg_enc_result *)J->st.d_res, n, d_off);
		OZG_HIP (hipGetLastError ());
		hipLaunchKernelGGL (otz_pack_copy_kernel, dim3 (n), dim3 (256), 0, s, (const u8 *)J->st.d_dst, (const ozg_enc_entry *)J->st.d_ent,
			(const ozg_enc_result *)J->st.d_res, (const u64 *)d_off, pk, n, (u64)dst_len);
		OZG_HIP (hipGetLastError ());
		J->total_at = (ent_bytes + 15u) & ~(size_t)15u;
		if (J->total_at + 8u > J->h_ent_cap) {
			J->total_at = J->h_ent_cap - 8u;
		}
		OZG_HIP (hipMemcpyAsync ((u8 *)J->h_ent + J->total_at, d_off + n, sizeof (u64), hipMemcpyDeviceToHost, s));
		OZG_HIP (hipMemcpyAsync (h_results, J->st.d_res, (size_t)n * sizeof (ozg_enc_result), hipMemcpyDeviceToHost, s));
		OZG_HIP (hipEventRecord (J->mid, s));
		J->two_step = true;
		J->h_dst = h_dst;
		J->dst_len = dst_len;
	} else {
		OZG_HIP (hipMemcpyAsync (h_dst, J->st.d_dst, dst_len, hipMemcpyDeviceToHost, s));
		OZG_HIP (hipMemcpyAsync (h_results, J->st.d_res, (size_t)n * sizeof (ozg_enc_result), hipMemcpyDeviceToHost, s));
		OZG_HIP (hipEventRecord (J->done, s));
	}
	J->busy = true;
	*job = J;
	return OZG_OK;
}

int ozg_encode_submit(const void *h_src, uint64_t src_len, void *h_dst, uint64_t dst_len, const ozg_enc_entry *entries,
	ozg_enc_result *h_results, uint32_t n, ozg_job **job) {
	return encode_submit (h_src, src_len, h_dst, dst_len, entries, h_results, n, job, false);
}

int ozg_encode_submit_packed(const void *h_src, uint64_t src_len, void *h_dst, uint64_t dst_len, const ozg_enc_entry *entries,
	ozg_enc_result *h_results, uint32_t n, ozg_job **job) {
	return encode_submit (h_src, src_len, h_dst, dst_len, entries, h_results, n, job, true);
}

// Second step of a packed encode job, once job->mid has been reached: the packed payload's size is known now,
// so its device-to-host copy can be issued (and job->done recorded behind it).  The job's device is current.
static hipError_t job_second_step(ozg_job *job) {
	uint64_t total = 0;
	memcpy (&total, (const u8 *)job->h_ent + job->total_at, sizeof (total));
	if (total > job->dst_len) {
		total = job->dst_len; // (cannot happen: every payload fits its slot)
	}
	hipError_t e = hipSuccess;
	if (total) {
		e = hipMemcpyAsync (job->h_dst, job->st.d_pack, (size_t)total, hipMemcpyDeviceToHost, job->stream);
	}
	if (e == hipSuccess) {
		e = hipEventRecord (job->done, job->stream);
	}
	if (e == hipSuccess) {
		std::lock_guard<std::mutex> lk (g_mu);
		job->two_step = false;
	}
	return e;
}

int ozg_job_done(ozg_job *job) {
	if (!job || !job->busy) {
		return 1;
	}
	if (job->two_step) {
		// a packed encode job: its payload copy can only be issued once the sizes are back.  A poll that finds
		// them back issues it; the job is done when that copy is.
		if (hipEventQuery (job->mid) != hipSuccess) {
			return 0;
		}
		const int keep = g_cur;
		if (hipSetDevice (phys_of (job->dev)) != hipSuccess) {
			return 0;
		}
		const hipError_t e = job_second_step (job);
		if (keep >= 0 && keep != job->dev) {
			(void)hipSetDevice (phys_of (keep));
		}
		if (e != hipSuccess) {
			return 1; // (ozg_job_wait reports the error)
		}
	}
	return hipEventQuery (job->done) == hipSuccess ? 1 : 0;
}

int ozg_job_wait(ozg_job *job) {
	if (!job || !job->busy) {
		return OZG_OK;
	}
	const int keep = g_cur;
	if (hipSetDevice (phys_of (job->dev)) != hipSuccess) {
		return OZG_E_HIP;
	}
	hipError_t e = hipSuccess;
	if (job->two_step) {
		e = hipEventSynchronize (job->mid);
		if (e == hipSuccess) {
			e = job_second_step (job);
		}
	}
	if (e == hipSuccess) {
		e = hipEventSynchronize (job->done);
	}
	{
		std::lock_guard<std::mutex> lk (g_mu);
		job->busy = false;
		job->two_step = false;
	}
	if (keep >= 0 && keep != job->dev) {
		(void)hipSetDevice (phys_of (keep));
	}
	if (e != hipSuccess) {
		return fail_hip ("ozg_job_wait", e);
	}
	return OZG_OK;
}

uint64_t ozg_compress_bound(uint16_t method, uint64_t n) {
	uint64_t b;
	switch (method) {
	case OZG_METHOD_DEFLATE: b = n + (n >> 3) + 75u; break; /* compressBound, ref deflate.inc.c:72-77 */
	case OZG_METHOD_ZSTD: b = n + 1024u; break;             /* ref otezip.c:860 */
	case OZG_METHOD_LZFSE: b = n + n / 10u + 100u; break;   /* :931 */
	case OZG_METHOD_LZMA: b = n + 13u + n / 8u; break;      /* :975 */
	case OZG_METHOD_BROTLI: b = 2u * n + 64u; break;        /* :1019 */
	default: b = n; break;
	}
	return b < 16u ? 16u : b;
}

static int launch_encode(dev_ctx &C, const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len, const ozg_enc_entry *d_entries,
	ozg_enc_result *d_results, uint32_t n, hipStream_t s) {
	workspace &ws = C.ws;
	const u32 ENC_GRID = C.enc_grid;
	const u32 grid = n < ENC_GRID ? n : ENC_GRID;
	if (!ws.enc_done) {
		OZG_HIP (hipEventCreateWithFlags (&ws.enc_done, hipEventDisableTiming));
		OZG_HIP (hipMalloc ((void **)&ws.enc_cursor, 256));
	}
	if (grid > ws.enc_grid) {
		OZG_HIP (hipDeviceSynchronize ()); // an earlier batch may still use the scratch that is about to be freed
		ws.enc_busy = false;
		(void)hipFree (ws.enc_tok);
		ws.enc_tok = nullptr;
		ws.enc_grid = 0;
		const u32 g = grid < 64u ? 64u : (grid + (grid >> 1) < ENC_GRID ? grid + (grid >> 1) : ENC_GRID);
		OZG_HIP (hipMalloc ((void **)&ws.enc_tok, (size_t)g * OTZ_ENC_TOK_WORDS * sizeof (u32)));
		ws.enc_grid = g;
	}
	if (ws.enc_busy) {
		OZG_HIP (hipStreamWaitEvent (s, ws.enc_done, 0)); // the scratch is shared between streams
	}
	OZG_HIP (hipMemsetAsync (ws.enc_cursor, 0, 4, s));
	hipLaunchKernelGGL (g_prof ? otz_encode_prof_kernel : otz_encode_kernel, dim3 (grid), dim3 (64), sizeof (otz_enc_lds), s, (const u8 *)d_src,
		(u8 *)d_dst, d_entries, d_results, n, (u64)src_len, (u64)dst_len, ws.enc_tok, ws.enc_cursor, g_prof);
	OZG_HIP (hipGetLastError ());
	OZG_HIP (hipEventRecord (ws.enc_done, s));
	ws.enc_busy = true;
	hipLaunchKernelGGL (otz_enc_crc_kernel, dim3 (n), dim3 (256), sizeof (otz_copy_lds), s, (const u8 *)d_src, (u8 *)d_dst,
		d_entries, d_results, n, (u64)src_len, (u64)dst_len);
	OZG_HIP (hipGetLastError ());
	return OZG_OK;
}

int ozg_encode_batch_device(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_enc_entry *d_entries, ozg_enc_result *d_results, uint32_t n, void *stream) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	if (n == 0) {
		return OZG_OK;
	}
	if (!d_src || !d_dst || !d_entries || !d_results || ((uintptr_t)d_dst & 15u)) {
		snprintf (g_err, sizeof (g_err), "null device pointer or d_dst not 16-byte aligned");
		return OZG_E_ARG;
	}
	std::lock_guard<std::mutex> lk (g_mu); // (the token scratch and its event belong to the device's workspace)
	return launch_encode (*C, d_src, src_len, d_dst, dst_len, d_entries, d_results, n, (hipStream_t)stream);
}

int ozg_encode_batch_host(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_enc_entry *entries, ozg_enc_result *results, uint32_t n) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	if (n == 0) {
		return OZG_OK;
	}
	if (!src || !dst || !entries || !results) {
		snprintf (g_err, sizeof (g_err), "null pointer");
		return OZG_E_ARG;
	}
	if ((rc = check_enc_entries (entries, n, src_len, dst_len)) != OZG_OK) {
		return rc;
	}
	std::lock_guard<std::mutex> lk (g_mu);
	staging &S = C->stage;
	if ((rc = staging_reserve (S, src_len + 64, dst_len + 64, (size_t)n * sizeof (ozg_enc_entry), (size_t)n * sizeof (ozg_enc_result))) != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (S.d_src, src, src_len, hipMemcpyHostToDevice, nullptr));
	OZG_HIP (hipMemcpyAsync (S.d_ent, entries, (size_t)n * sizeof (ozg_enc_entry), hipMemcpyHostToDevice, nullptr));
	rc = launch_encode (*C, S.d_src, src_len, S.d_dst, dst_len, (const ozg_enc_entry *)S.d_ent, (ozg_enc_result *)S.d_res, n, nullptr);
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (dst, S.d_dst, dst_len, hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipMemcpyAsync (results, S.d_res, (size_t)n * sizeof (ozg_enc_result), hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipStreamSynchronize (nullptr));
	return OZG_OK;
}

int ozg_checksum_host(int kind, const void *buf, uint64_t len, uint32_t *out) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	if (!out || (!buf && len) || len > 0xFFFFFFFFull || (kind != 0 && kind != 1)) {
		snprintf (g_err, sizeof (g_err), "ozg_checksum_host: bad argument");
		return OZG_E_ARG;
	}
	std::lock_guard<std::mutex> lk (g_mu);
	staging &S = C->stage;
	if ((rc = staging_reserve (S, len + 64, 64, 64, 64)) != OZG_OK) {
		return rc;
	}
	if (len) {
		OZG_HIP (hipMemcpyAsync (S.d_src, buf, len, hipMemcpyHostToDevice, nullptr));
	}
	hipLaunchKernelGGL (otz_checksum_kernel, dim3 (1), dim3 (256), sizeof (otz_copy_lds), nullptr, (const u8 *)S.d_src, (u32)len, (u32)kind, (u32 *)S.d_res);
	OZG_HIP (hipGetLastError ());
	OZG_HIP (hipMemcpyAsync (out, S.d_res, 4, hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipStreamSynchronize (nullptr));
	return OZG_OK;
}

int ozg_crc32_batch_device(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t *d_crc,
	uint32_t n, void *stream) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	if (n == 0) {
		return OZG_OK;
	}
	hipLaunchKernelGGL (otz_crc32_kernel, dim3 (n), dim3 (256), sizeof (otz_copy_lds), (hipStream_t)stream,
		(const u8 *)d_buf, (const u64 *)d_off, d_len, d_crc, n);
	OZG_HIP (hipGetLastError ());
	return OZG_OK;
}

void *ozg_malloc(size_t bytes) {
	dev_ctx *C;
	if (need_ctx (&C) != OZG_OK) {
		return nullptr;
	}
	void *p = nullptr;
	hipError_t e = hipMalloc (&p, bytes ? bytes : 1);
	if (e != hipSuccess) {
		fail_hip ("hipMalloc", e);
		return nullptr;
	}
	return p;
}
void ozg_free(void *d_ptr) {
	if (d_ptr) {
		(void)hipFree (d_ptr);
	}
}
void *ozg_malloc_pinned(size_t bytes) {
	dev_ctx *C;
	if (need_ctx (&C) != OZG_OK) {
		return nullptr;
	}
	void *p = nullptr;
	hipError_t e = hipHostMalloc (&p, bytes ? bytes : 1, hipHostMallocPortable); // (portable: every device of the node may copy from it)
	if (e != hipSuccess) {
		fail_hip ("hipHostMalloc", e);
		return nullptr;
	}
	return p;
}
void ozg_free_pinned(void *h_ptr) {
	if (h_ptr) {
		(void)hipHostFree (h_ptr);
	}
}
int ozg_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
	return OZG_OK;
}
int ozg_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
	return OZG_OK;
}
int ozg_stream_sync(void *stream) {
	dev_ctx *C;
	int rc = need_ctx (&C);
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipStreamSynchronize ((hipStream_t)stream));
	return OZG_OK;
}

/* Diagnostics only: d_buf = n x 32 uint64 device words: per entry 16 phase cycle counters of the
 * apply kernel, then 16 of the index kernel (NULL switches it off). */
void ozg_debug_set_profile_buffer(void *d_buf) { g_prof = (u64 *)d_buf; }

const char *ozg_kernel_name_inflate(void) { return "otz_apply_kernel"; }
const char *ozg_kernel_name_index(void) { return "otz_index_kernel"; }
const char *ozg_kernel_name_copy(void) { return "otz_copy_kernel"; }

} // extern "C"
