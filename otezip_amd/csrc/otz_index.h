// otz_index.h -- the hand-off between the two halves of the DEFLATE decode.
//
// Raw DEFLATE is one serial chain of variable-length tokens, and the LZ77 window needs
// 32 KiB of fast memory per stream.  Those two facts want opposite launch shapes, so the
// decode of an entry is split (ref path: inflate(), src/lib/deflate-dec.inc.c:547-831):
//
//   otz_index_kernel (otz_huff_index.h): ONE WAVEFRONT per entry, no workgroup barriers,
//     8 KB of LDS -> many entries per CU.  It finds the token chain -- where every token
//     starts and how many output bytes precede it -- and writes that down as an INDEX:
//     per DEFLATE block the code lengths and a flat list of items, one every <= OTZ_IX_G
//     tokens: (bit position, output position).  It produces no output bytes.
//   otz_apply_kernel (otz_lz_apply.h): one 512-lane workgroup per entry with the 56 KiB
//     LDS ring.  With the index every lane starts at an exact token boundary with an exact
//     output position: no speculation, no fixpoint, and every lane decodes the same number
//     of tokens.  It places literals, resolves the LZ77 copies in LDS, takes the CRC-32
//     and streams the bytes out.
//
// The index lives in a device arena the library owns (one region per entry, carved by
// otz_plan_kernel).  An entry whose index does not fit its region (pathological streams:
// thousands of empty blocks, ...) is marked OTZ_IX_OVERFLOW and the apply kernel decodes
// it with the self-contained workgroup-per-entry routine of otz_inflate.h instead.
#pragma once

#include "otz_simt.h"

#ifndef OTZ_IX_G
#define OTZ_IX_G 8u          // steps per item (a step: one token, or two literals); at most 8
#endif
#define OTZ_IX_OVERFLOW 77   // head.status: index did not fit; decode with otz_inflate_entry
#define OTZ_IX_KIND_STORED 0u
#define OTZ_IX_KIND_HUFF 1u
#define OTZ_IX_LENS 320u     // code lengths stored per Huffman block (288 + 32)

#define OTZ_IX_FIRST_BLOCK 32u  // region offset of the first block record
#define OTZ_IX_NEED_OUT (-100)  // head.status: the stream wants more output than the entry holds (OZG_ST_NEED_OUT is 0)

// one per entry (its own array, not inside the region)
struct otz_ix_head {
	int32_t status;     // 0: index complete, the apply kernel decodes; OTZ_IX_OVERFLOW; OTZ_IX_NEED_OUT: index
	                    // complete up to the token that does not fit; else the entry's final status
	u32 total_out;      // bytes the indexed tokens produce
	u32 flags;          // OZG_F_REF_FALSE_REJECT
	u32 n_blocks;
	u32 end_lo, end_hi; // status 0: bit position behind the final end-of-block code (what the reference's
	                    // total_in is the byte count of); OTZ_IX_NEED_OUT: behind the token that does not fit
	u32 clip_lo, clip_hi; // OTZ_IX_NEED_OUT: bit position of that token (it belongs to the last block record)
};

// block record: header, then (Huffman blocks) OTZ_IX_LENS code-length bytes and n_items + 1 items
struct otz_ix_block {
	u32 kind;
	u32 n_items;        // Huffman: items, not counting the end sentinel
	u32 bit_lo, bit_hi; // Huffman: entry bit position that item.bit is relative to; stored: source byte offset
	u32 out_base;       // output position of the block's first byte
	u32 out_len;        // bytes the block produces
	u32 hlit_hdist;     // hlit | hdist << 16
	u32 next;           // region offset of the next block record
};

// One item: the token that starts at bit `bit` (relative to the block's bit base) writes its
// first byte at output position `out` (relative to the block's out_base).  An item ends where
// the next one starts; the list is closed by a sentinel (bit = start of the end-of-block
// code, out = out_len).
struct otz_ix_item {
	u32 bit;
	u32 out;
};

#define OTZ_IX_ITEMS_OFF (32u + OTZ_IX_LENS)

// Geometry of the index kernel's speculation (otz_huff_index.h): a span of the input is cut into 64
// segments of S bits, one per lane, S a multiple of 32 between OTZ_K1_SMIN and OTZ_K1_SMAX.
#define OTZ_K1_SMIN 256u
#ifndef OTZ_K1_SMAX
#define OTZ_K1_SMAX 4096u
#endif
OTZ_DEV u32 otz_k1_seg_bits(u64 span_bits) {
	u64 s = (((span_bits + 63u) >> 6) + 31u) & ~(u64)31u;
	return s < OTZ_K1_SMIN ? OTZ_K1_SMIN : s > OTZ_K1_SMAX ? OTZ_K1_SMAX : (u32)s;
}
// What a region must hold besides the records: the checkpoint rows of the index kernel's two walks
// ([row][lane], 64 lanes x 8 bytes per row), at the region's end.  A lane notes a checkpoint every
// OTZ_IX_G tokens: rows for one token per 4 bits of the longest segment the entry can have (a lane
// that needs more closes its span early -- that costs time, never correctness); the re-walks of a
// span's fixpoint rarely go beyond a few dozen tokens.
#define OTZ_K1_ROWS_B 32u
OTZ_DEV u32 otz_k1_rows_a(u32 comp_size) { return otz_k1_seg_bits ((u64)comp_size * 8u) / 32u + 2u; }
OTZ_DEV u32 otz_k1_rows_b(u32 comp_size) {
	const u32 a = otz_k1_rows_a (comp_size);
	return a < OTZ_K1_ROWS_B ? a : OTZ_K1_ROWS_B;
}
OTZ_DEV u32 otz_k1_temp_bytes(u32 comp_size) { return (otz_k1_rows_a (comp_size) + otz_k1_rows_b (comp_size)) * 512u; }
#define OTZ_K1_TEMP_MAX ((OTZ_K1_SMAX / 32u + 2u + OTZ_K1_ROWS_B) * 512u)

// bytes of arena an entry asks for: records for one item per OTZ_IX_G output bytes in the
// worst case (a stream of literals), one lane head per 64 bytes of input, block records, temp
OTZ_DEV u64 otz_ix_region_bytes(u32 comp_size, u32 uncomp_size) {
	u64 b = (u64)uncomp_size + (comp_size >> 1) + 4096u + otz_k1_temp_bytes (comp_size);
	return (b + 255u) & ~(u64)255u;
}
