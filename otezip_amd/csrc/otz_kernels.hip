// otz_kernels.hip -- gfx950 kernels and the ozg_* C ABI (include/otezip_gpu.h).
//
// Decode kernels, one workgroup per archive entry:
//   otz_inflate_kernel  DEFLATE entries; 512 lanes (8 wavefronts), 77 KB of LDS per
//                       workgroup -> two resident workgroups per CU (160 KB LDS,
//                       <= 128 VGPRs), 512 entries in flight on the chip
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

#include "otz_dispatch.h"
#include "otz_deflate_enc.h"

static_assert (sizeof (ozg_entry) == 32, "ozg_entry ABI");
static_assert (sizeof (ozg_result) == 16, "ozg_result ABI");
static_assert (sizeof (ozg_enc_entry) == 32 && sizeof (ozg_enc_result) == 16, "encode ABI");
static_assert (sizeof (otz_inflate_lds) <= 80 * 1024, "two inflate workgroups per CU");

static_assert (OTZ_NT * OTZ_S_BITS / 8 + 512 <= OTZ_STAGE_BYTES, "stage must cover one round of input");

// Two workgroups per CU (16 waves, 4 per SIMD) need <= 128 VGPRs: pin that occupancy so a
// register-hungry change fails the resource check in tests/test_abi.py instead of
// silently halving the number of entries in flight.
extern "C" __global__ void __launch_bounds__ (OTZ_NT) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
otz_inflate_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, ozg_limits lim, u64 *prof) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	u32 i = blockIdx.x;
	if (i >= n) {
		return;
	}
	otz_inflate_lds &L = *reinterpret_cast<otz_inflate_lds *> (otz_smem);
	ozg_entry e = entries[i];
	otz_run_inflate (L, src, dst, e, &results[i], lim, prof ? prof + 16ull * i : nullptr);
}

extern "C" __global__ void __launch_bounds__ (256)
otz_copy_kernel(const u8 *src, u8 *dst, const ozg_entry *entries, ozg_result *results, u32 n, ozg_limits lim) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	u32 i = blockIdx.x;
	if (i >= n) {
		return;
	}
	otz_copy_lds &L = *reinterpret_cast<otz_copy_lds *> (otz_smem);
	ozg_entry e = entries[i];
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

// DEFLATE encode / STORE of n independent buffers: one wavefront each (26 KB LDS)
extern "C" __global__ void __launch_bounds__ (64)
otz_encode_kernel(const u8 *src, u8 *dst, const ozg_enc_entry *entries, ozg_enc_result *results, u32 n) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	u32 i = blockIdx.x;
	if (i >= n) {
		return;
	}
	otz_enc_lds &L = *reinterpret_cast<otz_enc_lds *> (otz_smem);
	ozg_enc_entry e = entries[i];
	otz_encode_entry (L, src, dst, e, &results[i]);
}

// CRC-32 of every input buffer of an encode batch (ref src/lib/otezip.c:1124)
extern "C" __global__ void __launch_bounds__ (256)
otz_enc_crc_kernel(const u8 *src, u8 *dst, const ozg_enc_entry *entries, ozg_enc_result *results, u32 n) {
	extern __shared__ __attribute__ ((aligned (16))) u8 otz_smem[];
	u32 i = blockIdx.x;
	if (i >= n) {
		return;
	}
	otz_copy_lds &L = *reinterpret_cast<otz_copy_lds *> (otz_smem);
	otz_crc_load_slice4 (L.t4);
	otz_crc_load_adv4k (L.a4);
	otz_sync ();
	u32 m = entries[i].src_size;
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

// ------------------------------------------------------------------ host side

namespace {

std::mutex g_mu;
bool g_ready = false;
int g_device = -1;
ozg_limits g_limits = { 1000ull, 1024ull * 1024ull };
u64 *g_prof = nullptr; // diagnostics: per-entry phase cycle counters (ozg_debug_set_profile_buffer)
thread_local char g_err[256] = "";

struct staging {
	void *d_src = nullptr, *d_dst = nullptr, *d_ent = nullptr, *d_res = nullptr;
	size_t src_cap = 0, dst_cap = 0, ent_cap = 0, res_cap = 0;
} g_stage;

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

int need_ready() {
	if (!g_ready) {
		snprintf (g_err, sizeof (g_err), "ozg_init() has not succeeded: no HIP device, and this library has no CPU path");
		return OZG_E_NODEVICE;
	}
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

} // namespace

extern "C" {

const char *ozg_last_error(void) { return g_err; }

int ozg_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount (&n) != hipSuccess) {
		return 0;
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
	if (device < 0 || device >= n) {
		snprintf (g_err, sizeof (g_err), "device %d out of range (0..%d)", device, n - 1);
		return OZG_E_ARG;
	}
	OZG_HIP (hipSetDevice (device));
	OZG_HIP (hipFuncSetAttribute ((const void *)otz_inflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
		(int)sizeof (otz_inflate_lds)));
	OZG_HIP (hipFuncSetAttribute ((const void *)otz_copy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
		(int)sizeof (otz_copy_lds)));
	OZG_HIP (hipFuncSetAttribute ((const void *)otz_crc32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
		(int)sizeof (otz_copy_lds)));
	OZG_HIP (hipFuncSetAttribute ((const void *)otz_enc_crc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
		(int)sizeof (otz_copy_lds)));
	OZG_HIP (hipFuncSetAttribute ((const void *)otz_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
		(int)sizeof (otz_enc_lds)));
	g_device = device;
	g_ready = true;
	g_err[0] = 0;
	return OZG_OK;
}

void ozg_shutdown(void) {
	std::lock_guard<std::mutex> lk (g_mu);
	if (!g_ready) {
		return;
	}
	(void)hipDeviceSynchronize ();
	if (g_stage.d_src) {
		(void)hipFree (g_stage.d_src);
	}
	if (g_stage.d_dst) {
		(void)hipFree (g_stage.d_dst);
	}
	if (g_stage.d_ent) {
		(void)hipFree (g_stage.d_ent);
	}
	if (g_stage.d_res) {
		(void)hipFree (g_stage.d_res);
	}
	g_stage = staging ();
	g_ready = false;
}

void ozg_set_limits(const ozg_limits *lim) {
	if (lim) {
		g_limits = *lim;
	}
}

int ozg_decode_batch_device_ex(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_entry *d_entries, ozg_result *d_results, uint32_t n, void *stream, uint32_t batch_flags) {
	int rc = need_ready ();
	if (rc != OZG_OK) {
		return rc;
	}
	(void)src_len;
	(void)dst_len;
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
	hipStream_t s = (hipStream_t)stream;
	if (!(batch_flags & OZG_BATCH_NO_DEFLATE)) {
		hipLaunchKernelGGL (otz_inflate_kernel, dim3 (n), dim3 (OTZ_NT), sizeof (otz_inflate_lds), s, (const u8 *)d_src,
			(u8 *)d_dst, d_entries, d_results, n, g_limits, g_prof);
		OZG_HIP (hipGetLastError ());
	}
	if (!(batch_flags & OZG_BATCH_ONLY_DEFLATE)) {
		hipLaunchKernelGGL (otz_copy_kernel, dim3 (n), dim3 (256), sizeof (otz_copy_lds), s, (const u8 *)d_src,
			(u8 *)d_dst, d_entries, d_results, n, g_limits);
		OZG_HIP (hipGetLastError ());
	}
	return OZG_OK;
}

int ozg_decode_batch_device(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_entry *d_entries, ozg_result *d_results, uint32_t n, void *stream) {
	return ozg_decode_batch_device_ex (d_src, src_len, d_dst, dst_len, d_entries, d_results, n, stream, 0);
}

int ozg_decode_batch_host(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *results, uint32_t n) {
	int rc = need_ready ();
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
	uint32_t flags = OZG_BATCH_NO_DEFLATE | OZG_BATCH_ONLY_DEFLATE;
	for (uint32_t i = 0; i < n; i++) {
		const ozg_entry &e = entries[i];
		if (e.src_off + e.comp_size > src_len || (e.dst_off & 15u) || e.dst_off + e.uncomp_size > dst_len) {
			snprintf (g_err, sizeof (g_err), "entry %u: offsets outside the buffers or dst_off not a multiple of 16", i);
			return OZG_E_ARG;
		}
		flags &= e.method == OZG_METHOD_DEFLATE ? ~OZG_BATCH_NO_DEFLATE : ~OZG_BATCH_ONLY_DEFLATE;
	}
	std::lock_guard<std::mutex> lk (g_mu);
	if ((rc = grow (&g_stage.d_src, &g_stage.src_cap, src_len + 64)) != OZG_OK) {
		return rc;
	}
	if ((rc = grow (&g_stage.d_dst, &g_stage.dst_cap, dst_len + 64)) != OZG_OK) {
		return rc;
	}
	if ((rc = grow (&g_stage.d_ent, &g_stage.ent_cap, (size_t)n * sizeof (ozg_entry))) != OZG_OK) {
		return rc;
	}
	if ((rc = grow (&g_stage.d_res, &g_stage.res_cap, (size_t)n * sizeof (ozg_result))) != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (g_stage.d_src, src, src_len, hipMemcpyHostToDevice, nullptr));
	OZG_HIP (hipMemcpyAsync (g_stage.d_ent, entries, (size_t)n * sizeof (ozg_entry), hipMemcpyHostToDevice, nullptr));
	rc = ozg_decode_batch_device_ex (g_stage.d_src, src_len, g_stage.d_dst, dst_len, (const ozg_entry *)g_stage.d_ent,
		(ozg_result *)g_stage.d_res, n, nullptr, flags);
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (dst, g_stage.d_dst, dst_len, hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipMemcpyAsync (results, g_stage.d_res, (size_t)n * sizeof (ozg_result), hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipStreamSynchronize (nullptr));
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

int ozg_encode_batch_device(const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len,
	const ozg_enc_entry *d_entries, ozg_enc_result *d_results, uint32_t n, void *stream) {
	int rc = need_ready ();
	if (rc != OZG_OK) {
		return rc;
	}
	(void)src_len;
	(void)dst_len;
	if (n == 0) {
		return OZG_OK;
	}
	if (!d_src || !d_dst || !d_entries || !d_results || ((uintptr_t)d_dst & 15u)) {
		snprintf (g_err, sizeof (g_err), "null device pointer or d_dst not 16-byte aligned");
		return OZG_E_ARG;
	}
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL (otz_encode_kernel, dim3 (n), dim3 (64), sizeof (otz_enc_lds), s, (const u8 *)d_src, (u8 *)d_dst,
		d_entries, d_results, n);
	OZG_HIP (hipGetLastError ());
	hipLaunchKernelGGL (otz_enc_crc_kernel, dim3 (n), dim3 (256), sizeof (otz_copy_lds), s, (const u8 *)d_src, (u8 *)d_dst,
		d_entries, d_results, n);
	OZG_HIP (hipGetLastError ());
	return OZG_OK;
}

int ozg_encode_batch_host(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_enc_entry *entries, ozg_enc_result *results, uint32_t n) {
	int rc = need_ready ();
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
	for (uint32_t i = 0; i < n; i++) {
		const ozg_enc_entry &e = entries[i];
		const bool codec_only = (e.reserved & OZG_ENC_F_CODEC_ONLY) != 0;
		if (e.src_off + e.src_size > src_len || (e.dst_off & 15u) || e.dst_off + e.dst_cap > dst_len || (e.dst_cap < e.src_size && !codec_only)) {
			snprintf (g_err, sizeof (g_err), "encode entry %u: offsets outside the buffers, dst_off not a multiple of 16, or dst_cap < src_size", i);
			return OZG_E_ARG;
		}
	}
	std::lock_guard<std::mutex> lk (g_mu);
	if ((rc = grow (&g_stage.d_src, &g_stage.src_cap, src_len + 64)) != OZG_OK) {
		return rc;
	}
	if ((rc = grow (&g_stage.d_dst, &g_stage.dst_cap, dst_len + 64)) != OZG_OK) {
		return rc;
	}
	if ((rc = grow (&g_stage.d_ent, &g_stage.ent_cap, (size_t)n * sizeof (ozg_enc_entry))) != OZG_OK) {
		return rc;
	}
	if ((rc = grow (&g_stage.d_res, &g_stage.res_cap, (size_t)n * sizeof (ozg_enc_result))) != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (g_stage.d_src, src, src_len, hipMemcpyHostToDevice, nullptr));
	OZG_HIP (hipMemcpyAsync (g_stage.d_ent, entries, (size_t)n * sizeof (ozg_enc_entry), hipMemcpyHostToDevice, nullptr));
	rc = ozg_encode_batch_device (g_stage.d_src, src_len, g_stage.d_dst, dst_len, (const ozg_enc_entry *)g_stage.d_ent,
		(ozg_enc_result *)g_stage.d_res, n, nullptr);
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (dst, g_stage.d_dst, dst_len, hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipMemcpyAsync (results, g_stage.d_res, (size_t)n * sizeof (ozg_enc_result), hipMemcpyDeviceToHost, nullptr));
	OZG_HIP (hipStreamSynchronize (nullptr));
	return OZG_OK;
}

int ozg_crc32_batch_device(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t *d_crc,
	uint32_t n, void *stream) {
	int rc = need_ready ();
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
	if (need_ready () != OZG_OK) {
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
	if (need_ready () != OZG_OK) {
		return nullptr;
	}
	void *p = nullptr;
	hipError_t e = hipHostMalloc (&p, bytes ? bytes : 1, hipHostMallocDefault);
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
	int rc = need_ready ();
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
	return OZG_OK;
}
int ozg_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
	int rc = need_ready ();
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipMemcpyAsync (h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
	return OZG_OK;
}
int ozg_stream_sync(void *stream) {
	int rc = need_ready ();
	if (rc != OZG_OK) {
		return rc;
	}
	OZG_HIP (hipStreamSynchronize ((hipStream_t)stream));
	return OZG_OK;
}

/* Diagnostics only: d_buf = n x 16 uint64 device words that the inflate kernel fills with
 * per-phase cycle counts of every entry (NULL switches it off). */
void ozg_debug_set_profile_buffer(void *d_buf) { g_prof = (u64 *)d_buf; }

const char *ozg_kernel_name_inflate(void) { return "otz_inflate_kernel"; }
const char *ozg_kernel_name_copy(void) { return "otz_copy_kernel"; }

} // extern "C"
