// otz_simt.h -- the handful of SIMT primitives the codec kernels are written against.
//
// Product build (hipcc, gfx950): everything below maps 1:1 onto CDNA4 wave64 /
// LDS / workgroup-barrier instructions.  There is no CPU code path in the
// shipped library.
//
// tests/simt_emu/ provides a second definition of the same names on top of
// pthreads so that the *same kernel bodies* can be run under g++/ASan on a
// machine without a GPU (GPU AddressSanitizer is not available on the pool).
// That emulation is test infrastructure; it is only reachable when a test
// build passes -DOTZ_SIMT_EMU and puts tests/simt_emu on the include path.
#pragma once

#include <stdint.h>

#ifdef OTZ_SIMT_EMU
#include "otz_simt_emu.h"
#else

#include <hip/hip_runtime.h>

#define OTZ_DEV __device__ __forceinline__
#define OTZ_DEV_NOINLINE __device__ __noinline__
#define OTZ_CONST_TABLE __device__ const
#define OTZ_WAVE 64

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint16_t u16;
typedef uint8_t u8;

struct otz_u4 {
	u32 x, y, z, w;
};

OTZ_DEV u32 otz_tid() { return threadIdx.x; }
OTZ_DEV u32 otz_lane() { return threadIdx.x & 63u; }
OTZ_DEV u32 otz_wave() { return threadIdx.x >> 6; }
OTZ_DEV u32 otz_nthreads() { return blockDim.x; }

OTZ_DEV void otz_sync() { __syncthreads (); }
// For single-wavefront workgroups: LDS/global accesses of the wave issue in order, so making
// them visible across lanes only needs the outstanding ones to drain -- no s_barrier.
OTZ_DEV void otz_wave_sync() {
	__builtin_amdgcn_fence (__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier ();
}

// Same, when only LDS traffic has to be ordered: the LDS pipeline executes one wavefront's DS
// instructions in issue order, so nothing has to drain -- the compiler just must not move them.
OTZ_DEV void otz_wave_sync_lds() {
	asm volatile ("" ::: "memory");
	__builtin_amdgcn_wave_barrier ();
	asm volatile ("" ::: "memory");
}

// wave64 cross-lane
OTZ_DEV u64 otz_ballot(bool p) { return __ballot (p); }
OTZ_DEV u32 otz_shfl_up(u32 v, u32 d) { return (u32)__shfl_up ((int)v, d, 64); }
OTZ_DEV u32 otz_shfl(u32 v, u32 src) { return (u32)__shfl ((int)v, (int)src, 64); }
OTZ_DEV u32 otz_shfl_xor(u32 v, u32 m) { return (u32)__shfl_xor ((int)v, (int)m, 64); }
// DPP cross-lane moves: VALU only, no trip through the LDS crossbar (__shfl_up is a ds_bpermute_b32,
// ~100 cycles of latency each; a six-step scan built from it is 600+ cycles of dependent waiting).
// value of lane - 1 (lane 0: fill) / of lane + 1 (lane 63: fill): v_mov_b32 wave_shr:1 / wave_shl:1
OTZ_DEV u32 otz_lane_up1(u32 v, u32 fill) { return (u32)__builtin_amdgcn_update_dpp ((int)fill, (int)v, 0x138, 0xf, 0xf, false); }
OTZ_DEV u32 otz_lane_down1(u32 v, u32 fill) { return (u32)__builtin_amdgcn_update_dpp ((int)fill, (int)v, 0x130, 0xf, 0xf, false); }
// inclusive prefix sum over the wavefront: row_shr 1/2/4/8 inside the rows of 16, then row_bcast:15 / :31
OTZ_DEV u32 otz_wave_scan_incl(u32 v) {
	v += (u32)__builtin_amdgcn_update_dpp (0, (int)v, 0x111, 0xf, 0xf, true);
	v += (u32)__builtin_amdgcn_update_dpp (0, (int)v, 0x112, 0xf, 0xf, true);
	v += (u32)__builtin_amdgcn_update_dpp (0, (int)v, 0x114, 0xf, 0xf, true);
	v += (u32)__builtin_amdgcn_update_dpp (0, (int)v, 0x118, 0xf, 0xf, true);
	v += (u32)__builtin_amdgcn_update_dpp (0, (int)v, 0x142, 0xa, 0xf, false);
	v += (u32)__builtin_amdgcn_update_dpp (0, (int)v, 0x143, 0xc, 0xf, false);
	return v;
}
// value of the lane whose number * 4 is byte_addr (ds_bpermute_b32 takes a byte address: callers that
// keep lane numbers premultiplied save the shift __shfl would add)
OTZ_DEV u32 otz_bpermute4(u32 byte_addr, u32 v) { return (u32)__builtin_amdgcn_ds_bpermute ((int)byte_addr, (int)v); }
// number of set bits of mask below this lane's own bit: v_mbcnt_lo + v_mbcnt_hi
OTZ_DEV u32 otz_mbcnt(u64 mask) { return __builtin_amdgcn_mbcnt_hi ((u32) (mask >> 32), __builtin_amdgcn_mbcnt_lo ((u32)mask, 0u)); }
// value of lane `src`, where src is the same in every lane: v_readlane_b32 (no LDS round trip)
OTZ_DEV u32 otz_readlane(u32 v, u32 src) { return (u32)__builtin_amdgcn_readlane ((int)v, (int)__builtin_amdgcn_readfirstlane ((int)src)); }
// a value known to be the same in every lane, moved to a scalar register
OTZ_DEV u32 otz_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane ((int)v); }

// LDS atomics (workgroup scope)
OTZ_DEV void otz_lds_or(u32 *p, u32 v) { __hip_atomic_fetch_or (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
OTZ_DEV void otz_lds_min(u32 *p, u32 v) { __hip_atomic_fetch_min (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
OTZ_DEV void otz_lds_max(u32 *p, u32 v) { __hip_atomic_fetch_max (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
OTZ_DEV u32 otz_lds_add(u32 *p, u32 v) { return __hip_atomic_fetch_add (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
OTZ_DEV u32 otz_lds_load(const u32 *p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// "Flag, then data" without draining the LDS queue in between: the LDS pipeline of a CU
// executes one wavefront's DS instructions in issue order, so a flag word loaded with
// otz_lds_load followed -- after otz_order_loads(), which only stops the COMPILER from
// moving memory accesses -- by data loads observes the data at least as new as the
// flag.  That is acquire ordering in one LDS round trip instead of two.  (A volatile
// load is not an option: it is lowered to a system-coherent FLAT load.)
OTZ_DEV void otz_lds_store(u32 *p, u32 v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// LDS is one in-order pipeline per CU; these only stop the compiler (and the
// issuing wave) from reordering around a hand-off inside the workgroup.
OTZ_DEV void otz_fence_release() { __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup"); }
OTZ_DEV void otz_fence_acquire() { __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup"); }
OTZ_DEV void otz_backoff() { __builtin_amdgcn_s_sleep (2); }
// Issue priority of this wavefront (0 = default .. 3): raised around dependency-bound phases
// so that the few wavefronts on the critical path are not queued behind the bulk work of
// the other workgroup on the CU.
#ifndef OTZ_PRIO_HIGH
#define OTZ_PRIO_HIGH 2
#endif
OTZ_DEV void otz_set_priority_high() { __builtin_amdgcn_s_setprio (OTZ_PRIO_HIGH); }
OTZ_DEV void otz_set_priority_normal() { __builtin_amdgcn_s_setprio (0); }

// shader clock, for the diagnostic phase counters only
OTZ_DEV u64 otz_cycles() { return __builtin_readcyclecounter (); }
// where this wavefront runs (diagnostics): HW_REG_HW_ID in the low word (wave, SIMD, CU, SH, SE), HW_REG_XCC_ID in the high one
OTZ_DEV u64 otz_where() {
	return (u64)__builtin_amdgcn_s_getreg ((31 << 11) | 4) | ((u64)__builtin_amdgcn_s_getreg ((31 << 11) | 20) << 32);
}

OTZ_DEV u32 otz_popc(u32 v) { return (u32)__popc (v); }
OTZ_DEV u32 otz_popc64(u64 v) { return (u32)__popcll (v); }
OTZ_DEV u32 otz_brev(u32 v) { return __brev (v); }
OTZ_DEV u32 otz_ctz(u32 v) { return (u32)__builtin_ctz (v); }
OTZ_DEV u32 otz_ctz64(u64 v) { return (u32)__builtin_ctzll (v); }
OTZ_DEV u32 otz_clz64(u64 v) { return (u32)__builtin_clzll (v); }
// width (<32) bits of v starting at bit off: v_bfe_u32
OTZ_DEV u32 otz_bfe(u32 v, u32 off, u32 width) { return __builtin_amdgcn_ubfe (v, off, width); }
// low 32 bits of (hi:lo) >> (sh & 31): v_alignbit_b32
OTZ_DEV u32 otz_alignbit(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbit (hi, lo, sh); }

// low 32 bits of (hi:lo) >> (8 * (q & 3)): v_alignbyte_b32
OTZ_DEV u32 otz_alignbyte(u32 hi, u32 lo, u32 q) { return __builtin_amdgcn_alignbyte (hi, lo, q); }

// Unaligned 2/4/8-byte LDS accesses: gfx950 executes DS instructions at any byte address
// (the compiler lowers these to single ds_read_b64 / ds_write_b32 / ... instructions).
// MEASURED (tools/ubench/lds_unaligned.hip, cycles of LDS time per wave-instruction, random
// addresses): ds_read_b64 at an arbitrary BYTE address 56 (aligned: 7.6, 4-byte aligned: 8.4);
// ds_write_b32/_b64 at an arbitrary byte 16, ds_write_b16 5.2, ds_write_b8 3.1.  So reads on a
// hot path fetch dword-aligned words and shift (v_alignbit / v_alignbyte); these helpers are
// for 4-byte-aligned 8-byte reads and for stores.
OTZ_DEV u64 otz_ld64u(const void *p) {
	u64 v;
	__builtin_memcpy (&v, p, 8);
	return v;
}
OTZ_DEV u32 otz_ld32u(const void *p) {
	u32 v;
	__builtin_memcpy (&v, p, 4);
	return v;
}
// An aligned word of GLOBAL memory through a pointer the compiler no longer knows the address space of (one that
// came out of a struct): said explicitly, it is a global_load; left generic it is a flat_load, which counts on the
// LDS counter as well -- the next wait for an LDS access then waits for the global round trip too.
OTZ_DEV u32 otz_ldg32(const void *p) {
	return *reinterpret_cast<const __attribute__ ((address_space (1))) u32 *> (reinterpret_cast<uintptr_t> (p));
}
// 16 bytes of global memory at any byte address, said to be global likewise (one global_load_dwordx4).
typedef u32 otz_v4u __attribute__ ((ext_vector_type (4)));
typedef otz_v4u otz_v4u_any __attribute__ ((aligned (1)));
OTZ_DEV void otz_ldg16u(const void *p, u64 *lo, u64 *hi) {
	const otz_v4u v = *reinterpret_cast<const __attribute__ ((address_space (1))) otz_v4u_any *> (reinterpret_cast<uintptr_t> (p));
	*lo = (u64)v.x | ((u64)v.y << 32);
	*hi = (u64)v.z | ((u64)v.w << 32);
}
OTZ_DEV void otz_st64u(void *p, u64 v) { __builtin_memcpy (p, &v, 8); }
OTZ_DEV void otz_st32u(void *p, u32 v) { __builtin_memcpy (p, &v, 4); }
OTZ_DEV void otz_st16u(void *p, u16 v) { __builtin_memcpy (p, &v, 2); }
// Keeps the compiler from moving memory accesses across this point; emits nothing.  With
// the in-order LDS pipeline that is all "flag, then data" needs (see otz_lds_load_inorder).
OTZ_DEV void otz_order_loads() { asm volatile ("" ::: "memory"); }
// Same for stores: data stores, then the flag (an LDS atomic), without waiting for the stores to
// complete in between -- the in-order LDS queue already keeps the flag behind the data.
OTZ_DEV void otz_order_stores() { asm volatile ("" ::: "memory"); }
// Forces a loaded value to be materialised HERE: keeps the compiler from sinking an LDS load
// below a branch, where it would start a second dependent round trip.
OTZ_DEV void otz_touch64(u64 &v) { asm volatile ("" : "+v"(v)); }
OTZ_DEV void otz_touch32(u32 &v) { asm volatile ("" : "+v"(v)); }
// Hides a value's origin from the optimiser: stops it from computing per-lane constants
// once in the kernel prologue and keeping them alive (spilled) across everything else.
// "Uses" a value without doing anything: the load that produced it has to have returned by here.
OTZ_DEV void otz_consume(u32 v) { asm volatile ("" ::"v"(v)); }
OTZ_DEV u32 otz_opaque(u32 v) {
	asm volatile ("" : "+v"(v));
	return v;
}

// 16-byte global accesses.  gfx950 runs with unaligned global access enabled,
// so an under-aligned source still lowers to one global_load_dwordx4.
OTZ_DEV otz_u4 otz_ld16(const void *p) {
	otz_u4 v;
	__builtin_memcpy (&v, p, 16);
	return v;
}
OTZ_DEV void otz_st16(void *p, otz_u4 v) { __builtin_memcpy (p, &v, 16); }
OTZ_DEV otz_u4 otz_ld16_aligned(const void *p) { return *reinterpret_cast<const otz_u4 *> (p); }
OTZ_DEV void otz_st16_aligned(void *p, otz_u4 v) { *reinterpret_cast<otz_u4 *> (p) = v; }

#endif // OTZ_SIMT_EMU
