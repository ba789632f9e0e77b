// Microbenchmark (diagnostic, not product): cost of LDS accesses at unaligned byte addresses on
// gfx950, 16 wavefronts per CU all hammering the LDS.  Prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

template <int MODE>
__global__ void __launch_bounds__ (1024) k(uint32_t *out, uint32_t iters, uint32_t seed) {
	extern __shared__ __attribute__ ((aligned (16))) uint8_t lds[];
	const uint32_t tid = threadIdx.x;
	for (uint32_t i = tid; i < 16384; i += 1024) {
		reinterpret_cast<uint32_t *> (lds)[i] = i * 2654435761u;
	}
	__syncthreads ();
	uint32_t x = (tid * 2654435761u + seed) | 1u;
	uint64_t acc = 0;
	const uint64_t t0 = __builtin_readcyclecounter ();
	for (uint32_t it = 0; it < iters; it++) {
		x = x * 1664525u + 1013904223u;
		const uint32_t r = (x >> 8) & 0xFFFFu; // random offset within 64 KiB - 64
		const uint32_t a = r % 65000u;
		if (MODE == 0) { // aligned b32 random
			uint32_t v;
			memcpy (&v, lds + (a & ~3u), 4);
			acc += v;
		} else if (MODE == 1) { // aligned b64 random
			uint64_t v;
			memcpy (&v, lds + (a & ~7u), 8);
			acc += v;
		} else if (MODE == 2) { // b64 at any byte
			uint64_t v;
			memcpy (&v, lds + a, 8);
			acc += v;
		} else if (MODE == 3) { // b64 4-byte aligned
			uint64_t v;
			memcpy (&v, lds + (a & ~3u), 8);
			acc += v;
		} else if (MODE == 4) { // b128 aligned
			uint4 v = *reinterpret_cast<const uint4 *> (lds + (a & ~15u));
			acc += v.x + v.w;
		} else if (MODE == 5) { // b128 any byte (two b64)
			uint64_t v, w;
			memcpy (&v, lds + a, 8);
			memcpy (&w, lds + a + 8, 8);
			acc += v + w;
		} else if (MODE == 6) { // u16 random (aligned 2)
			uint16_t v;
			memcpy (&v, lds + (a & ~1u), 2);
			acc += v;
		} else if (MODE == 7) { // write b64 any byte
			uint64_t v = x;
			memcpy (lds + a, &v, 8);
		} else if (MODE == 8) { // write b32 any byte
			uint32_t v = x;
			memcpy (lds + a, &v, 4);
		} else if (MODE == 9) { // write b8
			lds[a] = (uint8_t)x;
		} else if (MODE == 10) { // atomic or b32
			__hip_atomic_fetch_or (reinterpret_cast<uint32_t *> (lds) + (a >> 2), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else if (MODE == 11) { // b64 any byte, lane-strided (lane * 92 + small offset): the index kernel's slot reads
			uint64_t v;
			memcpy (&v, lds + (tid & 63u) * 92u + (tid >> 6) * 5900u % 60000u + (r & 63u), 8);
			acc += v;
		} else if (MODE == 12) { // aligned b64, consecutive lanes (ideal)
			uint64_t v;
			memcpy (&v, lds + (((tid & 63u) * 8u + (r & ~7u)) % 65000u & ~7u), 8);
			acc += v;
		} else if (MODE == 13) { // write b16 any byte
			uint16_t v = (uint16_t)x;
			memcpy (lds + a, &v, 2);
		}
	}
	const uint64_t t1 = __builtin_readcyclecounter ();
	if (acc == 0x1234567u) {
		out[tid] = 1;
	}
	if (tid == 0) {
		out[blockIdx.x * 2 + 0] = (uint32_t) (t1 - t0);
	}
}

template <int MODE>
static void run(const char *name) {
	uint32_t *d;
	hipMalloc (&d, 1 << 20);
	const uint32_t iters = 4096;
	hipFuncSetAttribute ((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
	k<MODE><<<256, 1024, 65536>>> (d, iters, 1);
	hipDeviceSynchronize ();
	k<MODE><<<256, 1024, 65536>>> (d, iters, 2);
	hipDeviceSynchronize ();
	uint32_t h[512];
	hipMemcpy (h, d, sizeof (h), hipMemcpyDeviceToHost);
	double s = 0;
	for (int i = 0; i < 256; i++) {
		s += h[2 * i];
	}
	// 16 waves per CU (one 1024-thread block per CU... 2 may fit by LDS: 64 KB each) -> per-CU LDS time per wave-instruction
	printf ("%-42s %7.1f cycles/iter per wave (16 waves/WG) -> %5.1f LDS cycles per wave-instruction\n", name, s / 256 / iters, s / 256 / iters / 16.0);
	hipFree (d);
}

int main() {
	run<0> ("ds_read_b32 aligned random");
	run<1> ("ds_read_b64 aligned random");
	run<2> ("ds_read_b64 any byte random");
	run<3> ("ds_read_b64 4-byte aligned random");
	run<4> ("ds_read_b128 aligned random");
	run<5> ("2 x ds_read_b64 any byte (16 B)");
	run<6> ("ds_read_u16 random");
	run<7> ("ds_write_b64 any byte random");
	run<8> ("ds_write_b32 any byte random");
	run<9> ("ds_write_b8 random");
	run<13> ("ds_write_b16 any byte random");
	run<10> ("ds_or_b32 random");
	run<11> ("ds_read_b64 any byte, lane stride 92 B");
	run<12> ("ds_read_b64 aligned, consecutive lanes");
	return 0;
}
