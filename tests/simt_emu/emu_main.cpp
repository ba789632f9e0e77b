// emu_main.cpp -- runs the codec kernel bodies of otezip_amd/csrc on CPU threads.
// TEST INFRASTRUCTURE ONLY (see otz_simt_emu.h): one OS thread per work-item of a
// 256-lane workgroup, one workgroup at a time.  Built by tests/simt_emu/Makefile
// into tests/simt_emu/libotz_emu.so, optionally with -fsanitize=address.
#include "otz_dispatch.h"
#include "otz_deflate_enc.h"
#include "otz_huff_index.h"
#include "otz_lz_apply.h"
#include "otz_zstd.h"
#include "otz_lzma.h"
#include "otz_brotli.h"
#include "otz_lzfse.h"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

thread_local otz_emu_thread otz_emu_self;

#ifndef OTZ_EMU_PTHREADS
void otz_emu_yield() {
	otz_emu_wave &w = otz_emu_self.blk->wave[otz_emu_self.tid >> 6];
	const u32 me = w.cur;
	u32 nx = me;
	do {
		nx = (nx + 1u) % w.nfib;
	} while (w.done[nx] && nx != me);
	if (nx == me) {
		return;
	}
	w.cur = nx;
	otz_emu_self.tid = (otz_emu_self.tid & ~63u) | nx;
	swapcontext (&w.fib[me], &w.fib[nx]);
}
#endif

// the RFC 7932 constant data (otezip_amd/data/brotli_rfc7932.bin), set by the test before method-97 streams are decoded
static const u8 *g_brotli_blob = nullptr;
extern "C" void emu_set_brotli_blob(const void *p) { g_brotli_blob = (const u8 *)p; }

namespace {

struct block_args {
	otz_emu_block *blk;
	u32 tid;
	void *lds;
	const u8 *src;
	u8 *dst;
	const ozg_entry *e;
	ozg_result *r;
	ozg_limits lim;
	int phase; // 0: whole-entry routines; 1: index kernel body; 2: apply kernel body
	otz_ix_head *head;
	u8 *region;
	u32 region_bytes;
	u64 *end_bit;
	const ozg_enc_entry *enc_e; // phase 3: encode kernel body
	ozg_enc_result *enc_r;
	u32 *enc_tok;
	u8 *zs_lit;   // phase 4: Zstandard wavefront body; phase 5: its CRC pass
	u8 *zs_route;
};

void *thread_main(void *p) {
	block_args *a = (block_args *)p;
	otz_emu_self.tid = a->tid;
	otz_emu_self.blk = a->blk;
	if (a->phase == 3) {
		otz_encode_entry<false> (*(otz_enc_lds *)a->lds, a->src, a->dst, *a->enc_e, a->enc_r, a->enc_tok);
	} else if (a->phase == 4) {
		// what otz_zstd_kernel does for one entry
		const ozg_entry &e = *a->e;
		if (!otz_zipbomb (e, a->lim) && otz_zs_is_candidate (a->src + e.src_off, e.comp_size)) {
			otz_zs_run<false> (*(otz_zs_lds *)a->lds, a->src + e.src_off, e.comp_size, a->dst + e.dst_off, e.uncomp_size, a->zs_lit, a->r, a->zs_route);
		}
	} else if (a->phase == 6) {
		// what otz_lzma_kernel does for one entry (one work-item)
		const ozg_entry &e = *a->e;
		if (otz_tid () == 0 && !otz_zipbomb (e, a->lim) && otz_lzma_is_candidate (a->src + e.src_off, e.comp_size)) {
			otz_lzma_run (a->src + e.src_off, e.comp_size, a->dst + e.dst_off, e.uncomp_size, e.flags, (u16 *)a->zs_lit, a->r, a->zs_route);
		}
	} else if (a->phase == 8) {
		// what otz_brotli_kernel does for one entry (one work-item)
		const ozg_entry &e = *a->e;
		if (otz_tid () == 0 && !otz_zipbomb (e, a->lim) && otz_brotli_is_candidate (a->src + e.src_off, e.comp_size)) {
			otz_brotli_run (a->src + e.src_off, e.comp_size, a->dst + e.dst_off, e.uncomp_size, e.flags, a->zs_lit, a->region_bytes, g_brotli_blob, a->r, a->zs_route);
		}
	} else if (a->phase == 9) {
		// what otz_lzfse_kernel does for one entry (one work-item)
		const ozg_entry &e = *a->e;
		if (otz_tid () == 0 && !otz_zipbomb (e, a->lim) && otz_lzfse_is_candidate (a->src + e.src_off, e.comp_size)) {
			otz_lzfse_run (a->src + e.src_off, e.comp_size, a->dst + e.dst_off, e.uncomp_size, e.flags, a->zs_lit, a->r, a->zs_route);
		}
	} else if (a->phase == 5) {
		// what otz_zstd_crc_kernel does
		otz_copy_lds &L = *(otz_copy_lds *)a->lds;
		otz_crc_load_slice4 (L.t4);
		otz_crc_load_adv4k (L.a4);
		otz_sync ();
		otz_run_crc (L, a->dst, *a->e, a->r, a->r->total_out, false);
	} else if (a->phase == 7) {
		// what otz_out_crc_kernel does
		otz_copy_lds &L = *(otz_copy_lds *)a->lds;
		otz_crc_load_slice4 (L.t4);
		otz_crc_load_adv4k (L.a4);
		otz_sync ();
		otz_run_crc (L, a->dst, *a->e, a->r, a->e->uncomp_size, true);
	} else if (a->phase == 1) {
		otz_run_index<false> (*(otz_k1_lds *)a->lds, a->src, a->dst, *a->e, a->lim, a->head, a->region, a->region_bytes);
	} else if (a->phase == 2) {
		otz_run_apply (*(otz_apply_lds *)a->lds, a->src, a->dst, *a->e, a->r, a->lim, a->head, a->region, nullptr, a->end_bit);
	} else if (a->e->method == OZG_METHOD_DEFLATE) {
		otz_run_inflate (*(otz_inflate_lds *)a->lds, a->src, a->dst, *a->e, a->r, a->lim, nullptr);
	} else {
		otz_run_copy (*(otz_copy_lds *)a->lds, a->src, a->dst, *a->e, a->r, a->lim);
	}
	return nullptr;
}

// one workgroup of NT emulated work-items running thread_main
#ifdef OTZ_EMU_PTHREADS
void run_block(u32 NT, block_args proto, size_t lds_bytes) {
	otz_emu_block blk;
	blk.nthreads = NT;
	pthread_barrier_init (&blk.block_bar, nullptr, NT);
	for (u32 w = 0; w < (NT + 63) / 64; w++) {
		pthread_barrier_init (&blk.wave_bar[w], nullptr, NT < 64 ? NT : 64);
	}
	// garbage-filled LDS: the kernels must not rely on zero initialisation
	void *lds = aligned_alloc (64, (lds_bytes + 63) & ~(size_t)63);
	memset (lds, 0xA5, lds_bytes);
	std::vector<block_args> args (NT);
	std::vector<pthread_t> th (NT);
	pthread_attr_t attr;
	pthread_attr_init (&attr);
	pthread_attr_setstacksize (&attr, 256 * 1024);
	for (u32 t = 0; t < NT; t++) {
		args[t] = proto;
		args[t].blk = &blk;
		args[t].tid = t;
		args[t].lds = lds;
		if (pthread_create (&th[t], &attr, thread_main, &args[t]) != 0) {
			fprintf (stderr, "emu: pthread_create failed\n");
			abort ();
		}
	}
	for (u32 t = 0; t < NT; t++) {
		pthread_join (th[t], nullptr);
	}
	pthread_attr_destroy (&attr);
	free (lds);
	pthread_barrier_destroy (&blk.block_bar);
	for (u32 w = 0; w < (NT + 63) / 64; w++) {
		pthread_barrier_destroy (&blk.wave_bar[w]);
	}
}
#else
constexpr size_t FIBER_STACK = 192 * 1024;

struct wave_args {
	otz_emu_block *blk;
	u32 wave;
	block_args *lane_args; // 64 entries
};

thread_local otz_emu_wave *g_wave = nullptr;
thread_local block_args *g_lane_args = nullptr;

void fiber_entry(int lane) {
	thread_main (&g_lane_args[lane]);
	otz_emu_wave &w = *g_wave;
	w.done[lane] = 1;
	w.alive--;
	if (w.alive && w.bar_count >= w.alive) { // the others were only waiting for this one
		w.bar_count = 0;
		w.bar_gen++;
	}
	// returning resumes w.sched (uc_link)
}

void *wave_main(void *p) {
	wave_args *a = (wave_args *)p;
	otz_emu_wave &w = a->blk->wave[a->wave];
	g_wave = &w;
	g_lane_args = a->lane_args;
	otz_emu_self.blk = a->blk;
	for (u32 l = 0; l < w.nfib; l++) {
		getcontext (&w.fib[l]);
		w.fib[l].uc_stack.ss_sp = (char *)w.stacks + (size_t)l * FIBER_STACK;
		w.fib[l].uc_stack.ss_size = FIBER_STACK;
		w.fib[l].uc_link = &w.sched;
		makecontext (&w.fib[l], (void (*) ())fiber_entry, 1, (int)l);
	}
	u32 next = 0;
	while (w.alive) {
		while (w.done[next]) {
			next = (next + 1u) % w.nfib;
		}
		w.cur = next;
		otz_emu_self.tid = a->wave * 64u + next;
		swapcontext (&w.sched, &w.fib[next]);
		next = (w.cur + 1u) % w.nfib;
	}
	return nullptr;
}

void run_block(u32 NT, block_args proto, size_t lds_bytes) {
	static otz_emu_block blk; // (large: ucontexts) one workgroup at a time
	const u32 nw = (NT + 63) / 64;
	blk.nthreads = NT;
	pthread_barrier_init (&blk.block_bar, nullptr, nw);
	void *lds = aligned_alloc (64, (lds_bytes + 63) & ~(size_t)63);
	memset (lds, 0xA5, lds_bytes); // garbage-filled LDS: the kernels must not rely on zero initialisation
	std::vector<block_args> args (NT);
	std::vector<wave_args> wa (nw);
	std::vector<pthread_t> th (nw);
	static void *stacks[16] = {};
	for (u32 t = 0; t < NT; t++) {
		args[t] = proto;
		args[t].blk = &blk;
		args[t].tid = t;
		args[t].lds = lds;
	}
	for (u32 wv = 0; wv < nw; wv++) {
		otz_emu_wave &w = blk.wave[wv];
		w.nfib = NT - 64 * wv < 64 ? NT - 64 * wv : 64;
		w.alive = w.nfib;
		w.cur = 0;
		w.bar_count = w.bar_gen = 0;
		memset (w.done, 0, sizeof (w.done));
		if (!stacks[wv]) {
			stacks[wv] = aligned_alloc (4096, 64 * FIBER_STACK);
		}
		w.stacks = stacks[wv];
		wa[wv] = { &blk, wv, &args[64 * wv] };
		if (pthread_create (&th[wv], nullptr, wave_main, &wa[wv]) != 0) {
			fprintf (stderr, "emu: pthread_create failed\n");
			abort ();
		}
	}
	for (u32 wv = 0; wv < nw; wv++) {
		pthread_join (th[wv], nullptr);
	}
	free (lds);
	pthread_barrier_destroy (&blk.block_bar);
}
#endif

} // namespace

// mode 0: the product path (index kernel + apply kernel for DEFLATE, copy kernel otherwise);
// mode 1: the self-contained workgroup-per-entry DEFLATE routine (what the apply kernel
//         falls back to when an index does not fit its arena region);
// mode 2: product path with NO arena region (forces that fallback through the real plumbing);
// region_scale_pct (mode 0): size of the arena regions relative to otz_ix_region_bytes().
static u64 *g_end_bits = nullptr; // optional: n words, filled by the product path (emu_set_end_bits)
extern "C" void emu_set_end_bits(uint64_t *p) { g_end_bits = (u64 *)p; }

extern "C" int emu_decode_batch_mode(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *results, uint32_t n, uint64_t ratio, uint64_t slack, int mode, uint32_t region_scale_pct) {
	(void)src_len;
	(void)dst_len;
	size_t lds_bytes = sizeof (otz_inflate_lds) > sizeof (otz_copy_lds) ? sizeof (otz_inflate_lds) : sizeof (otz_copy_lds);
	for (uint32_t i = 0; i < n; i++) {
		results[i].status = -999;
		block_args a = {};
		a.src = (const u8 *)src;
		a.dst = (u8 *)dst;
		a.e = &entries[i];
		a.r = &results[i];
		a.lim = { ratio, slack };
		if (entries[i].method == OZG_METHOD_ZSTD) {
			// product order: Zstandard kernel, copy kernel (private framing) unless it decoded the entry, CRC pass
			std::vector<u8> lit (OTZ_ZS_LIT_BYTES, 0xA5);
			u8 route = OTZ_ZS_ROUTE_NONE;
			a.zs_lit = lit.data ();
			a.zs_route = &route;
			a.phase = 4;
			run_block (64, a, sizeof (otz_zs_lds));
			a.phase = 0;
			if (route != OTZ_ZS_ROUTE_DONE) {
				run_block (OTZ_CNT, a, lds_bytes);
			} else if (results[i].status == OZG_ST_OK) {
				a.phase = 5;
				run_block (OTZ_CNT, a, lds_bytes);
			}
			continue;
		}
		if (entries[i].method == OZG_METHOD_LZMA) {
			// product order: LZMA1 kernel, copy kernel (private tokens) unless it decoded the entry, CRC pass
			std::vector<u8> probs (OTZ_LZMA_PROB_BYTES, 0xA5);
			u8 route = 0;
			a.zs_lit = probs.data ();
			a.zs_route = &route;
			a.phase = 6;
			run_block (64, a, 64);
			a.phase = 0;
			if (route != OTZ_ZS_ROUTE_DONE) {
				run_block (OTZ_CNT, a, lds_bytes);
			} else if (results[i].status == OZG_ST_OK) {
				a.phase = 5;
				run_block (OTZ_CNT, a, lds_bytes);
			}
			continue;
		}
		if (entries[i].method == OZG_METHOD_LZFSE) {
			// product order: LZFSE kernel, copy kernel (the reference's 06 framing) unless it decoded the entry, CRC pass
			std::vector<u8> arena (OTZ_BR_ARENA_SMALL + 16, 0xA5);
			u8 route = 0;
			a.zs_lit = (u8 *) (((uintptr_t)arena.data () + 15u) & ~(uintptr_t)15u);
			a.zs_route = &route;
			a.phase = 9;
			run_block (64, a, 64);
			a.phase = 0;
			if (route != OTZ_ZS_ROUTE_DONE) {
				run_block (OTZ_CNT, a, lds_bytes);
			} else if (results[i].status == OZG_ST_OK) {
				a.phase = 5;
				run_block (OTZ_CNT, a, lds_bytes);
			}
			continue;
		}
		if (entries[i].method == OZG_METHOD_BROTLI && g_brotli_blob) {
			// product order: Brotli kernel, copy kernel ("BROT" container) unless it decoded the entry, CRC pass
			// (two passes: a small arena, then a large one for the entry that ran out of room)
			std::vector<u8> arena (OTZ_BR_ARENA + 16, 0xA5);
			u8 route = 0;
			a.zs_lit = (u8 *) (((uintptr_t)arena.data () + 15u) & ~(uintptr_t)15u);
			a.zs_route = &route;
			a.phase = 8;
			a.region_bytes = OTZ_BR_ARENA_SMALL;
			run_block (64, a, 64);
			if (route == OTZ_BR_ROUTE_AGAIN) {
				a.region_bytes = OTZ_BR_ARENA;
				run_block (64, a, 64);
			}
			a.phase = 0;
			if (route != OTZ_ZS_ROUTE_DONE) {
				run_block (OTZ_CNT, a, lds_bytes);
			} else if (results[i].status == OZG_ST_OK) {
				a.phase = 5;
				run_block (OTZ_CNT, a, lds_bytes);
			}
			continue;
		}
		if (entries[i].method != OZG_METHOD_DEFLATE) {
			run_block (OTZ_CNT, a, lds_bytes);
			continue;
		}
		if (mode == 1) {
			run_block (OTZ_NT, a, lds_bytes);
			continue;
		}
		otz_ix_head head;
		memset (&head, 0xA5, sizeof (head));
		u64 want = mode == 2 ? 0 : otz_ix_region_bytes (entries[i].comp_size, entries[i].uncomp_size) * region_scale_pct / 100u;
		want &= ~(u64)255u;
		std::vector<u8> region (want + 512, 0xA5);
		a.head = &head;
		a.end_bit = g_end_bits ? g_end_bits + i : nullptr;
		a.region = region.data ();
		a.region_bytes = (u32)want;
		a.phase = 1;
		run_block (64, a, sizeof (otz_k1_lds));
		if (getenv ("OTZ_EMU_TRACE")) {
			fprintf (stderr, "[emu] entry %u: index status %d, %u blocks, %u bytes\n", i, head.status, head.n_blocks, head.total_out);
		}
		a.phase = 2;
		run_block (OTZ_A_NT, a, sizeof (otz_apply_lds));
		if (results[i].status == OZG_ST_OK) {
			a.phase = 7;
			run_block (OTZ_CNT, a, lds_bytes);
		}
		if (head.status == OTZ_IX_OVERFLOW) { // what otz_inflate_kernel does with the overflow list
			a.phase = 0;
			run_block (OTZ_NT, a, lds_bytes);
		}
	}
	return 0;
}

extern "C" int emu_decode_batch(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *results, uint32_t n, uint64_t ratio, uint64_t slack) {
	return emu_decode_batch_mode (src, src_len, dst, dst_len, entries, results, n, ratio, slack, 0, 100);
}

extern "C" uint32_t emu_lds_bytes_index(void) { return (uint32_t)sizeof (otz_k1_lds); }
extern "C" uint32_t emu_lds_bytes_inflate(void) { return (uint32_t)sizeof (otz_inflate_lds); }
extern "C" uint32_t emu_lds_bytes_apply(void) { return (uint32_t)sizeof (otz_apply_lds); }
extern "C" uint32_t emu_lds_bytes_copy(void) { return (uint32_t)sizeof (otz_copy_lds); }
extern "C" uint32_t emu_lds_bytes_zstd(void) { return (uint32_t)sizeof (otz_zs_lds); }

// ---- encode side: one 64-thread workgroup per buffer (CRC is left to the caller in emulation)
extern "C" int emu_encode_batch(const void *src, void *dst, const ozg_enc_entry *entries, ozg_enc_result *results, uint32_t n) {
	u32 *tok = (u32 *)malloc (sizeof (u32) * OTZ_ENC_TOK_WORDS);
	for (uint32_t i = 0; i < n; i++) {
		results[i].status = -999;
		block_args a = {};
		a.src = (const u8 *)src;
		a.dst = (u8 *)dst;
		a.phase = 3;
		a.enc_tok = tok;
		a.enc_e = &entries[i];
		a.enc_r = &results[i];
		run_block (OTZ_ENC_NT, a, sizeof (otz_enc_lds));
	}
	free (tok);
	return 0;
}
