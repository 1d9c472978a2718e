/*
 * magicdec_hip_dev.h -- DEVELOPMENT knobs of libmagicdec_hip.so (kernel tuning sweeps and same-process A/Bs only).
 *
 * Not part of the drop-in boundary (include/magicdec_hip.h).  These functions set process-global, non-thread-safe
 * host state that changes which kernel variant / decomposition the next calls use; they exist only in a library built
 * with -DMD_DEV_KNOBS (the in-tree Makefile's default, DEV=1, because tools/ and the variant-parity tests use them;
 * `make DEV=0` builds the library without them).  No knob selects a kernel that returns wrong results.
 */
#ifndef MAGICDEC_HIP_DEV_H
#define MAGICDEC_HIP_DEV_H

#ifdef __cplusplus
extern "C" {
#endif

/* md_paged_attn: target number of workgroups of the split-KV decomposition; n <= 0 restores the default (256) */
void md_debug_set_attn_target_wgs(int n);
/* md_paged_attn, decode / verify: wavefronts per workgroup, 4 | 8 forced; 0 = the rule of make_plan (8 for the fp8 two-M-tile
 * kernel when the launch leaves one workgroup per CU and >= 8 tiles per wave) */
void md_debug_set_attn_waves(int nw);
/* md_paged_attn, prefill: 32x32x16-MFMA kernel: -1 = the measured rule (default), 0 = off (the 16x16x32 kernel),
 * 32 | 64 | 128 = keys per shared tile (halved until it divides the page size); 129 = 128 keys, first V sub-tile pairing */
void md_debug_set_prefill_mfma32(int kt);
/* md_paged_attn, prefill: keys per shared tile of the 16x16 kernel (32 | 64 = default), waves per workgroup (4 | 8; 0 = rule) */
void md_debug_set_prefill_kt(int kt, int nw);
/* md_linear: split-K policy (workgroups to aim for) */
void md_debug_set_gemm_target_blocks(int n);
/* md_linear: wavefronts per workgroup, 4 | 6 | 7 forced (bit-identical results); 0 = the balance rule (plan_of) */
void md_debug_set_gemm_waves(int nw);
/* md_linear: W prefetch ring depth of the four-wave, <= 64-row, bf16 form: 16 = the deep ring (bit-identical results, measured
 * equal or 1-3 % slower: profiles/r06_ring_depth_ab.txt); 0 / 8 = the rule (8) */
void md_debug_set_gemm_ring(int rd);
/* md_linear_fused: wavefronts (K slices) per workgroup, 8 | 16; 11 / 22 = 1 x 1 / 2 x 2 MFMA tiles per workgroup with 8 K
 * slices (bit-identical results); 0 = the measured rules */
void md_debug_set_fused_nw(int nw);
/* md_linear_fused_split: K slices over workgroups, forced where (K / 16) % (8 S) == 0; 0 = the rule (about one workgroup per CU) */
void md_debug_set_fused_split(int S);
/* md_linear_fused / md_linear_fused_split: phase timestamps.  buf = device pointer to [workgroups x wavefronts][6] uint64, or
 * NULL = off: every wavefront records the 100 MHz wall clock at (0) entry, (1) first loads issued (after the deferred-norm
 * prologue), (2) first activation chunk staged, (3) K slice consumed, (4) partial tiles in LDS (after the barrier),
 * (5) epilogue stores issued -- tools/tile_timing.py */
void md_debug_set_tile_timing(void* buf);
/* md_linear_block: workgroups to aim for when K is split (0 = default 256); non-temporal weight DMA (1 = default) */
void md_debug_set_block_gemm(int target_blocks, int weights_nontemporal);

#ifdef __cplusplus
}
#endif
#endif
