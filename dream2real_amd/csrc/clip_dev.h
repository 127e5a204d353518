// clip_dev.h — measurement switches of clip.hip.  DEVELOPMENT BUILDS ONLY (make DEV=1, GEMM_ABLATE=<mask>, ATTN_ABLATE=<mask>,
// or DEV=1 EXTRA=-DD2R_GEMM_STAMPS — the stamps need the development build): the product library is built with every
// switch here off.  The Makefile keeps the flags of the last build in _build/flags.stamp, so changing any of them
// rebuilds every object.
//
// D2R_GEMM_ABLATE (bitmask) — k_gemm8 and the shared epilogue: 1 no LDS-DMA, 2 no fragment ds_reads, 4 no epilogue,
//   8 no MFMA, 128 no global loads/stores in the epilogue, 256 no LDS transposes in the epilogue.
// D2R_ATTN_ABLATE (bitmask) — k_attention_s: 16 no DMA requests, 32 no softmax, 64 no PV (V reads + MFMAs), 128 no
//   S MFMAs, 256 no per-tile barriers, 512 the partial last query tile computes nothing (prices a cheaper tail tile: -2.7 % of the kernel at
//   197 tokens, profiles/r06_attn_price.txt).  (1024, round 6's pricing of the shifted accumulator, is gone: it is what the kernel now does.)
// Results are garbage when a mask is set; tools/gemm_ablate.sh / tools/attn_ablate.sh rebuild with each mask to see
// what a tile's time is made of.
// D2R_GEMM_STAMPS — shader-clock cycles wave 0 of every workgroup spends per tile section of k_gemm8, read back with
//   d2r_debug_gemm_stamps (tools/gemm_stamps.py): [EPI][0 drain wait, 1 K loop, 2 epilogue, 3 tiles].
#pragma once
#ifndef D2R_GEMM_ABLATE
#define D2R_GEMM_ABLATE 0
#endif
#ifndef D2R_ATTN_ABLATE
#define D2R_ATTN_ABLATE 0
#endif
// D2R_F8_EXP (bitmask) — k_gemm8f and the fp8 epilogues: 1 no scale-byte stores, 2 no e4m3 stores, 4 no GELU, 8 no scale loads in the K loop,
//   16 no MFMA, 32 no LDS-DMA, 64 no fragment reads.  Garbage results; tools/f8_ablate.sh.
#ifndef D2R_F8_EXP
#define D2R_F8_EXP 0
#endif
// D2R_F8_VAR (bitmask) — schedule variants of k_gemm8f's phase: 1 no s_setprio, 2 s_setprio 1 around the MFMA section of every wave (instead of
//   a static priority for wave row 0), 4 staging requests before the fragment reads.  Same results.
#ifndef D2R_F8_VAR
#define D2R_F8_VAR 0
#endif
#ifdef D2R_GEMM_STAMPS
#define D2R_GEMM_STAMP_KINDS 13          /* EPI_KINDS of clip.hip */
__device__ unsigned long long d2r_gemm_stamps[D2R_GEMM_STAMP_KINDS][4];
extern "C" __attribute__((visibility("default"))) int d2r_debug_gemm_stamps(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(d2r_gemm_stamps), sizeof(unsigned long long) * 4 * D2R_GEMM_STAMP_KINDS) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[4 * D2R_GEMM_STAMP_KINDS] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(d2r_gemm_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define STAMP(var)
#endif
