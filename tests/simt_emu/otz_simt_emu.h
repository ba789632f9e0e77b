// otz_simt_emu.h -- pthread emulation of the SIMT primitives in
// otezip_amd/csrc/otz_simt.h.  TEST INFRASTRUCTURE ONLY: lets the kernel
// bodies run under g++ (-fsanitize=address,undefined) where no GPU exists.
// One OS thread per work-item, one pthread barrier per workgroup and per wave.
// Never compiled into the product library.
#pragma once

#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>

#define OTZ_DEV static inline
#define OTZ_DEV_NOINLINE static
#define OTZ_CONST_TABLE static const
#define OTZ_WAVE 64

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint16_t u16;
typedef uint8_t u8;

struct otz_u4 {
	u32 x, y, z, w;
};

#ifndef OTZ_EMU_PTHREADS
// Default: one OS thread per WAVEFRONT, its 64 work-items are cooperative fibers
// (ucontext) that switch at every wave-level operation.  A wave barrier costs 64 user-space
// context switches instead of 64 futex waits, which makes the emulation ~10x faster.
// -DOTZ_EMU_PTHREADS (the sanitizer build) keeps one OS thread per work-item.
#include <ucontext.h>

struct otz_emu_wave {
	ucontext_t sched;        // the wave's OS thread
	ucontext_t fib[64];
	unsigned char done[64];
	u32 nfib, alive, cur;
	u32 bar_count, bar_gen;
	void *stacks;
};

struct otz_emu_block {
	u32 nthreads;
	pthread_barrier_t block_bar; // one participant per wavefront
	otz_emu_wave wave[16];
	u32 xchg[1024];
};

struct otz_emu_thread {
	u32 tid;
	otz_emu_block *blk;
};

extern thread_local otz_emu_thread otz_emu_self;

// switch to the next live fiber of this wave (round robin); returns when this one is resumed
void otz_emu_yield();

OTZ_DEV void otz_emu_wave_sync() {
	otz_emu_wave &w = otz_emu_self.blk->wave[otz_emu_self.tid >> 6];
	const u32 gen = w.bar_gen;
	if (++w.bar_count >= w.alive) {
		w.bar_count = 0;
		w.bar_gen = gen + 1u;
		return;
	}
	while (w.bar_gen == gen) {
		otz_emu_yield ();
	}
}
OTZ_DEV void otz_wave_sync() { otz_emu_wave_sync (); }
// workgroup barrier: every wave gathers its fibers, ONE fiber per wave meets the other
// waves' OS threads, then the wave's fibers are released together
OTZ_DEV void otz_sync() {
	otz_emu_wave &w = otz_emu_self.blk->wave[otz_emu_self.tid >> 6];
	const u32 gen = w.bar_gen;
	if (++w.bar_count >= w.alive) {
		pthread_barrier_wait (&otz_emu_self.blk->block_bar);
		w.bar_count = 0;
		w.bar_gen = gen + 1u;
		return;
	}
	while (w.bar_gen == gen) {
		otz_emu_yield ();
	}
}
OTZ_DEV void otz_backoff() { otz_emu_yield (); }

#else

struct otz_emu_block {
	u32 nthreads;
	pthread_barrier_t block_bar;
	pthread_barrier_t wave_bar[16];
	u32 xchg[1024];
};

struct otz_emu_thread {
	u32 tid;
	otz_emu_block *blk;
};

extern thread_local otz_emu_thread otz_emu_self;

OTZ_DEV void otz_sync() { pthread_barrier_wait (&otz_emu_self.blk->block_bar); }
OTZ_DEV void otz_wave_sync() { pthread_barrier_wait (&otz_emu_self.blk->wave_bar[otz_emu_self.tid >> 6]); }
OTZ_DEV void otz_emu_wave_sync() { pthread_barrier_wait (&otz_emu_self.blk->wave_bar[otz_emu_self.tid >> 6]); }
OTZ_DEV void otz_backoff() { sched_yield (); }

#endif

OTZ_DEV u32 otz_tid() { return otz_emu_self.tid; }
OTZ_DEV u32 otz_lane() { return otz_emu_self.tid & 63u; }
OTZ_DEV u32 otz_wave() { return otz_emu_self.tid >> 6; }
OTZ_DEV u32 otz_nthreads() { return otz_emu_self.blk->nthreads; }

OTZ_DEV u64 otz_ballot(bool p) {
	otz_emu_block *b = otz_emu_self.blk;
	u32 base = otz_wave () * 64;
	__atomic_store_n (&b->xchg[otz_tid ()], p ? 1u : 0u, __ATOMIC_RELAXED);
	otz_emu_wave_sync ();
	u64 m = 0;
	for (u32 i = 0; i < 64 && base + i < b->nthreads; i++) {
		if (__atomic_load_n (&b->xchg[base + i], __ATOMIC_RELAXED)) {
			m |= 1ull << i;
		}
	}
	otz_emu_wave_sync ();
	return m;
}

OTZ_DEV u32 otz_shfl(u32 v, u32 src) {
	otz_emu_block *b = otz_emu_self.blk;
	u32 base = otz_wave () * 64;
	__atomic_store_n (&b->xchg[otz_tid ()], v, __ATOMIC_RELAXED);
	otz_emu_wave_sync ();
	u32 r = __atomic_load_n (&b->xchg[base + (src & 63u)], __ATOMIC_RELAXED);
	otz_emu_wave_sync ();
	return r;
}
OTZ_DEV u32 otz_shfl_up(u32 v, u32 d) {
	u32 l = otz_lane ();
	u32 r = otz_shfl (v, l >= d ? l - d : l);
	return r;
}
OTZ_DEV u32 otz_shfl_xor(u32 v, u32 m) { return otz_shfl (v, otz_lane () ^ m); }
OTZ_DEV u32 otz_lane_up1(u32 v, u32 fill) {
	const u32 l = otz_lane (), r = otz_shfl (v, l ? l - 1u : 0u);
	return l ? r : fill;
}
OTZ_DEV u32 otz_lane_down1(u32 v, u32 fill) {
	const u32 l = otz_lane (), r = otz_shfl (v, l < 63u ? l + 1u : 63u);
	return l < 63u ? r : fill;
}
OTZ_DEV u32 otz_wave_scan_incl(u32 v) {
	const u32 l = otz_lane ();
	for (u32 d = 1; d < 64; d <<= 1) {
		const u32 o = otz_shfl (v, l >= d ? l - d : l);
		if (l >= d) {
			v += o;
		}
	}
	return v;
}
OTZ_DEV void otz_wave_sync_lds() { otz_wave_sync (); }
OTZ_DEV u32 otz_bpermute4(u32 byte_addr, u32 v) { return otz_shfl (v, (byte_addr >> 2) & 63u); }
OTZ_DEV u32 otz_mbcnt(u64 mask) { return (u32)__builtin_popcountll (mask & ((1ull << otz_lane ()) - 1ull)); }
OTZ_DEV u32 otz_readlane(u32 v, u32 src) { return otz_shfl (v, src); }
OTZ_DEV u32 otz_uniform(u32 v) { return v; }

OTZ_DEV void otz_lds_or(u32 *p, u32 v) { __atomic_fetch_or (p, v, __ATOMIC_RELAXED); }
OTZ_DEV void otz_lds_min(u32 *p, u32 v) {
	u32 cur = __atomic_load_n (p, __ATOMIC_RELAXED);
	while (v < cur && !__atomic_compare_exchange_n (p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
	}
}
OTZ_DEV void otz_lds_max(u32 *p, u32 v) {
	u32 cur = __atomic_load_n (p, __ATOMIC_RELAXED);
	while (v > cur && !__atomic_compare_exchange_n (p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
	}
}
OTZ_DEV u32 otz_lds_add(u32 *p, u32 v) { return __atomic_fetch_add (p, v, __ATOMIC_RELAXED); }
OTZ_DEV u32 otz_lds_load(const u32 *p) { return __atomic_load_n (p, __ATOMIC_RELAXED); }
OTZ_DEV void otz_lds_store(u32 *p, u32 v) { __atomic_store_n (p, v, __ATOMIC_RELAXED); }

OTZ_DEV void otz_fence_release() { __atomic_thread_fence (__ATOMIC_RELEASE); }
OTZ_DEV void otz_fence_acquire() { __atomic_thread_fence (__ATOMIC_ACQUIRE); }
OTZ_DEV void otz_set_priority_high() {}
OTZ_DEV void otz_set_priority_normal() {}

OTZ_DEV u64 otz_cycles() { return 0; }
OTZ_DEV u64 otz_where() { return 0; }
OTZ_DEV u32 otz_popc(u32 v) { return (u32)__builtin_popcount (v); }
OTZ_DEV u32 otz_popc64(u64 v) { return (u32)__builtin_popcountll (v); }
OTZ_DEV u32 otz_brev(u32 v) {
	v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	return __builtin_bswap32 (v);
}
OTZ_DEV u32 otz_ctz(u32 v) { return (u32)__builtin_ctz (v); }
OTZ_DEV u32 otz_ctz64(u64 v) { return (u32)__builtin_ctzll (v); }
OTZ_DEV u32 otz_clz64(u64 v) { return (u32)__builtin_clzll (v); }
OTZ_DEV u32 otz_bfe(u32 v, u32 off, u32 width) { return (v >> (off & 31u)) & ((1u << (width & 31u)) - 1u); }
OTZ_DEV u32 otz_alignbit(u32 hi, u32 lo, u32 sh) { return (u32) ((((u64)hi << 32) | lo) >> (sh & 31u)); }

OTZ_DEV u32 otz_alignbyte(u32 hi, u32 lo, u32 q) { return (u32) ((((u64)hi << 32) | lo) >> (8u * (q & 3u))); }

OTZ_DEV u64 otz_ld64u(const void *p) {
	u64 v;
	memcpy (&v, p, 8);
	return v;
}
OTZ_DEV u32 otz_ld32u(const void *p) {
	u32 v;
	memcpy (&v, p, 4);
	return v;
}
OTZ_DEV u32 otz_ldg32(const void *p) { return *reinterpret_cast<const u32 *> (p); }
OTZ_DEV void otz_ldg16u(const void *p, u64 *lo, u64 *hi) {
	memcpy (lo, p, 8);
	memcpy (hi, static_cast<const u8 *> (p) + 8, 8);
}
OTZ_DEV void otz_st64u(void *p, u64 v) { memcpy (p, &v, 8); }
OTZ_DEV void otz_st32u(void *p, u32 v) { memcpy (p, &v, 4); }
OTZ_DEV void otz_st16u(void *p, u16 v) { memcpy (p, &v, 2); }
OTZ_DEV void otz_order_loads() { __atomic_thread_fence (__ATOMIC_ACQUIRE); }
OTZ_DEV void otz_order_stores() { __atomic_thread_fence (__ATOMIC_RELEASE); }
OTZ_DEV void otz_touch64(u64 &v) { (void)v; }
OTZ_DEV void otz_touch32(u32 &v) { (void)v; }
OTZ_DEV void otz_consume(u32 v) { (void)v; }
OTZ_DEV u32 otz_opaque(u32 v) { return v; }

OTZ_DEV otz_u4 otz_ld16(const void *p) {
	otz_u4 v;
	memcpy (&v, p, 16);
	return v;
}
OTZ_DEV void otz_st16(void *p, otz_u4 v) { memcpy (p, &v, 16); }
OTZ_DEV otz_u4 otz_ld16_aligned(const void *p) { return otz_ld16 (p); }
OTZ_DEV void otz_st16_aligned(void *p, otz_u4 v) { memcpy (p, &v, 16); }
