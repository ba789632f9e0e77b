// emu_main.cpp -- runs the codec kernel bodies of otezip_amd/csrc on CPU threads.
// TEST INFRASTRUCTURE ONLY (see otz_simt_emu.h): one OS thread per work-item of a
// 256-lane workgroup, one workgroup at a time.  Built by tests/simt_emu/Makefile
// into tests/simt_emu/libotz_emu.so, optionally with -fsanitize=address.
#include "otz_dispatch.h"
#include "otz_deflate_enc.h"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

thread_local otz_emu_thread otz_emu_self;

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
};

void *thread_main(void *p) {
	block_args *a = (block_args *)p;
	otz_emu_self.tid = a->tid;
	otz_emu_self.blk = a->blk;
	if (a->e->method == OZG_METHOD_DEFLATE) {
		otz_run_inflate (*(otz_inflate_lds *)a->lds, a->src, a->dst, *a->e, a->r, a->lim, nullptr);
	} else {
		otz_run_copy (*(otz_copy_lds *)a->lds, a->src, a->dst, *a->e, a->r, a->lim);
	}
	return nullptr;
}

} // namespace

extern "C" int emu_decode_batch(const void *src, uint64_t src_len, void *dst, uint64_t dst_len,
	const ozg_entry *entries, ozg_result *results, uint32_t n, uint64_t ratio, uint64_t slack) {
	(void)src_len;
	(void)dst_len;
	size_t lds_bytes = sizeof (otz_inflate_lds) > sizeof (otz_copy_lds) ? sizeof (otz_inflate_lds) : sizeof (otz_copy_lds);
	for (uint32_t i = 0; i < n; i++) {
		const u32 NT = entries[i].method == OZG_METHOD_DEFLATE ? OTZ_NT : OTZ_CNT;
		otz_emu_block blk;
		blk.nthreads = NT;
		pthread_barrier_init (&blk.block_bar, nullptr, NT);
		for (u32 w = 0; w < NT / 64; w++) {
			pthread_barrier_init (&blk.wave_bar[w], nullptr, 64);
		}
		// garbage-filled LDS: the kernels must not rely on zero initialisation
		void *lds = aligned_alloc (64, (lds_bytes + 63) & ~(size_t)63);
		memset (lds, 0xA5, lds_bytes);
		results[i].status = -999;
		std::vector<block_args> args (NT);
		std::vector<pthread_t> th (NT);
		pthread_attr_t attr;
		pthread_attr_init (&attr);
		pthread_attr_setstacksize (&attr, 256 * 1024);
		for (u32 t = 0; t < NT; t++) {
			args[t] = { &blk, t, lds, (const u8 *)src, (u8 *)dst, &entries[i], &results[i], { ratio, slack } };
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
		for (u32 w = 0; w < NT / 64; w++) {
			pthread_barrier_destroy (&blk.wave_bar[w]);
		}
	}
	return 0;
}

extern "C" uint32_t emu_lds_bytes_inflate(void) { return (uint32_t)sizeof (otz_inflate_lds); }
extern "C" uint32_t emu_lds_bytes_copy(void) { return (uint32_t)sizeof (otz_copy_lds); }

// ---- encode side: one 64-thread workgroup per buffer (CRC is left to the caller in emulation)
namespace {
struct enc_args {
	otz_emu_block *blk;
	u32 tid;
	void *lds;
	const u8 *src;
	u8 *dst;
	const ozg_enc_entry *e;
	ozg_enc_result *r;
};
void *enc_thread(void *p) {
	enc_args *a = (enc_args *)p;
	otz_emu_self.tid = a->tid;
	otz_emu_self.blk = a->blk;
	otz_encode_entry (*(otz_enc_lds *)a->lds, a->src, a->dst, *a->e, a->r);
	return nullptr;
}
} // namespace

extern "C" int emu_encode_batch(const void *src, void *dst, const ozg_enc_entry *entries, ozg_enc_result *results, uint32_t n) {
	const u32 NT = OTZ_ENC_NT;
	for (uint32_t i = 0; i < n; i++) {
		otz_emu_block blk;
		blk.nthreads = NT;
		pthread_barrier_init (&blk.block_bar, nullptr, NT);
		pthread_barrier_init (&blk.wave_bar[0], nullptr, NT);
		void *lds = aligned_alloc (64, (sizeof (otz_enc_lds) + 63) & ~(size_t)63);
		memset (lds, 0xA5, sizeof (otz_enc_lds));
		results[i].status = -999;
		std::vector<enc_args> args (NT);
		std::vector<pthread_t> th (NT);
		for (u32 t = 0; t < NT; t++) {
			args[t] = { &blk, t, lds, (const u8 *)src, (u8 *)dst, &entries[i], &results[i] };
			pthread_create (&th[t], nullptr, enc_thread, &args[t]);
		}
		for (u32 t = 0; t < NT; t++) {
			pthread_join (th[t], nullptr);
		}
		free (lds);
		pthread_barrier_destroy (&blk.block_bar);
		pthread_barrier_destroy (&blk.wave_bar[0]);
	}
	return 0;
}
