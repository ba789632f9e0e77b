#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32;
__device__ __forceinline__ u32 lds_addr(const void *p) {
	return (u32)(size_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void lds_mskor(void *p, u32 mask, u32 data) {
	asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(lds_addr(p)), "v"(mask), "v"(data) : "memory");
}
extern "C" __global__ void k(unsigned *out) {
	extern __shared__ unsigned buf[];
	buf[threadIdx.x] = 0x11223344u;
	__syncthreads();
	lds_mskor(&buf[threadIdx.x ^ 1], 0x00FFFF00u, 0x00ABCD00u | threadIdx.x << 24);
	__syncthreads();
	out[threadIdx.x] = buf[threadIdx.x];
}
int main() {
	unsigned *d, h[64];
	hipMalloc(&d, 256);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 256, 0, d);
	hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 64; i++) if (h[i] != 0x11ABCD44u) bad++;
	printf("mskor: %08x %08x bad=%d\n", h[0], h[1], bad);
	return bad;
}
