// clip.hip — CLIP image scoring for gfx950 (MI355X), hand-written HIP.
//
// Replaces np.rot90 + CLIPProcessor + CLIPModel (vision tower, projection, logits) of
// reference clip_scoring.py:145,177-181.  Specification: oracle/d2r_oracle.c
// (d2r_oracle_clip_preprocess) and oracle/clip_ref.py.
//
//   k_preprocess   rot90 + Pillow antialiased bicubic resize (both passes in 22-bit fixed
//                  point with the uint8 intermediate staged in LDS) + centre crop +
//                  rescale/normalise, written patch-major in bf16 so patch embedding is a
//                  plain GEMM; optional fp32 pixel_values for parity
//   k_gemm8        C = A[M,K] * W[N,K]^T on v_mfma_f32_32x32x16_bf16: persistent 256x256x64 tiles,
//                  operands staged by LDS-DMA into a ring of eight 16 KiB half-tile slots that
//                  never drains (counted vmcnt, 8 phases per K-tile pair), XOR-swizzled LDS
//                  (conflict-free ds_read_b128), XCD-aware tile order, fused epilogues (bias,
//                  quick_gelu, fp32 residual); k_gemm is the plain-K-loop variant for small outputs
//   k_embed_ln     [class | patches] + position embedding + pre_layrnorm -> fp32 residual
//   k_layernorm    fp32 residual -> bf16 GEMM operand, one wave per token
//   k_attention    flash-style attention per (image, head): S^T = K Q^T so each query's
//                  scores are lane-local, online softmax, P fed back as the MFMA B operand
//                  without leaving registers, V^T staged in LDS
//   k_head         post_layernorm(CLS) -> visual_projection -> L2 normalise -> logits
#include "d2r_internal.h"

#include <type_traits>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <deque>
#include <mutex>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ClipWeights {
    // bf16 GEMM operands, row-major [N][K]
    const uint16_t *w_patch;   // [d][Kp_pad]
    const float *cls, *pos;    // [d], [T][d]
    const float *pre_w, *pre_b;
    struct Layer {
        const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
        const uint16_t *w_qkv, *w_o, *w_fc1, *w_fc2;
        const float *b_qkv, *b_o, *b_fc1, *b_fc2;
        // LayerNorm-folded operands of the two Linears that follow a LayerNorm (vision tower; see EPI_LN_*)
        const uint16_t *wf_qkv = nullptr, *wf_fc1 = nullptr;       // bf16(W * gamma)
        const float *cs_qkv = nullptr, *cs_fc1 = nullptr;          // column sums of the folded weight
        const float *bf_qkv = nullptr, *bf_fc1 = nullptr;          // b + W beta
    };
    const float *post_w, *post_b;
    const float *proj;         // fp32 [D][d]
};

struct ResampleTables {
    const int *bounds_h;     // [rw][2]  (xmin, count) in rotated-source columns
    const int *kk_h;         // [rw][ks_h]
    const int *bounds_v;     // [rh][2]
    const int *kk_v;         // [rh][ks_v]
    int ks_h, ks_v;
    int need_h, need_v;
    int iw, ih;              // (rotated) source size
    int rw, rh;              // resized size
    int left, top;           // crop offset in the resized image
    int max_rows;            // LDS rows per band
};

struct PrepCache {
    uint32_t w, h;
    int rot90;
    ResampleTables R;
};

struct d2r_clip {
    d2r_ctx *ctx;
    d2r_clip_desc desc;
    uint32_t T, Kp, Kp_pad;
    std::vector<void *> allocs;
    ClipWeights w;
    std::vector<ClipWeights::Layer> layers;
    std::deque<PrepCache> prep;      // resampling tables per (w, h, rot90), built on first use (deque: stable references)
    std::mutex prep_mu;              // two threads may score different frame sizes with one model
    // "fp8 blocks": e4m3 copies of the four Linear weights of every layer + their per-matrix scales, built on first use
    struct F8Layer { const uint8_t *w_qkv, *w_o, *w_fc1, *w_fc2; float s_qkv, s_o, s_fc1, s_fc2; };
    std::vector<F8Layer> f8;
    std::mutex f8_mu;
};

__device__ __forceinline__ uint16_t f2bf(float x)
{
    union { __bf16 b; uint16_t u; } c;
    c.b = (__bf16)x;
    return c.u;
}
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    // the two-element vector form compiles to ONE v_cvt_pk_bf16_f32; converting the halves separately
    // and or-ing them costs two conversions plus shift/or fix-ups
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    union { bf16x2 v; uint32_t u; } r;
    r.v[0] = (__bf16)a;
    r.v[1] = (__bf16)b;
    return r.u;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------ preprocess

#define PRECISION_BITS 22
__device__ __forceinline__ uint32_t clip8(int v)
{
    v >>= PRECISION_BITS;
    return (uint32_t)min(max(v, 0), 255);
}

// grid (bands of P output rows, n); block 256.  dynamic LDS: 3 planes of [S][max_rows] bytes
// (column-major: the vertical pass then reads consecutive bytes).  Thread order in pass 1 follows
// the SOURCE memory order — with rot90 consecutive source bytes are consecutive rotated rows —
// so the uint8 frame is read coalesced in both orientations.
__global__ __launch_bounds__(256) void k_preprocess(const uint8_t *__restrict__ frames, uint32_t w, uint32_t h,
                                                    int rot90, ResampleTables R, uint32_t S, uint32_t P,
                                                    uint32_t band, uint16_t *__restrict__ patches,
                                                    uint32_t Kp_pad, float *__restrict__ pixel_values,
                                                    const int4 *__restrict__ rects, const uint16_t *__restrict__ bg_patches, int touched_only)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tmp[];
    const uint32_t img = blockIdx.y;
    const uint32_t row0 = blockIdx.x * band;                 // first output row (cropped coords)
    const uint32_t nrows = min(band, S - row0);
    uint32_t x_lo = 0, x_n = S;                              // output columns this workgroup produces
    // Composited candidates differ from the background frame only inside the rectangle their rays were generated in
    // (rects[img] = x0, y0, x1, y1 in frame pixels, inclusive; k_raygen_rect).  A band (= one row of patches) whose
    // source rows do not meet that rectangle resamples background pixels only: it is the background's own patch row
    // (computed once per view with this same kernel), copied.  Exact by construction; ~3/4 of the bands of a
    // candidate with a 60-pixel object.
    if (rects && bg_patches && patches && !pixel_values && band == P) {
        int ys, ye;                                          // rotated-source rows this band reads: [ys, ye)
        if (R.need_v) {
            ys = R.bounds_v[2 * (R.top + row0)];
            const int lastrow = R.top + row0 + nrows - 1;
            ye = R.bounds_v[2 * lastrow] + R.bounds_v[2 * lastrow + 1];
        } else {
            ys = R.top + row0;
            ye = ys + nrows;
        }
        const int4 rc = rects[img];
        // the rectangle's rows in the (rotated) source: rot90 maps frame column x to row w-1-x
        const int rlo = rot90 ? (int)w - 1 - rc.z : rc.y, rhi = rot90 ? (int)w - 1 - rc.x : rc.w;
        const bool untouched = rc.x > rc.z || rc.y > rc.w || rhi < ys || rlo >= ye;
        if (touched_only) {
            // layer-0 reuse (k_touch_list): only the TOUCHED patches' rows are read downstream — an untouched band needs
            // nothing, a touched band only the patch columns whose footprint meets the rectangle (an interval: the same test)
            if (untouched) return;
            const int clo = rot90 ? rc.y : rc.x, chi = rot90 ? rc.w : rc.z;
            uint32_t p_lo = S / P, p_hi = 0;
            for (uint32_t pcol = 0; pcol < S / P; pcol++) {
                int xs, xe;
                if (R.need_h) {
                    xs = R.bounds_h[2 * (R.left + pcol * P)];
                    const int last = R.left + pcol * P + P - 1;
                    xe = R.bounds_h[2 * last] + R.bounds_h[2 * last + 1];
                } else { xs = R.left + pcol * P; xe = xs + P; }
                if (!(chi < xs || clo >= xe)) { p_lo = min(p_lo, pcol); p_hi = max(p_hi, pcol + 1); }
            }
            if (p_lo >= p_hi) return;
            x_lo = p_lo * P;
            x_n = (p_hi - p_lo) * P;
        } else if (untouched) {
            const uint32_t g = S / P;
            const size_t n16 = (size_t)g * Kp_pad * 2 / 16;                         // one patch row, in 16-byte words
            const uint4 *src16 = (const uint4 *)(bg_patches + (size_t)blockIdx.x * g * Kp_pad);
            uint4 *dst16 = (uint4 *)(patches + ((size_t)img * g * g + (size_t)blockIdx.x * g) * Kp_pad);
            for (size_t i = threadIdx.x; i < n16; i += blockDim.x) dst16[i] = src16[i];
            return;
        }
    }
    const uint8_t *src = frames + (size_t)img * w * h * 3;
    // rows of the horizontally-resampled image this band needs
    int y_first, y_last;
    if (R.need_v) {
        y_first = R.bounds_v[2 * (R.top + row0)];
        int lastrow = R.top + row0 + nrows - 1;
        y_last = R.bounds_v[2 * lastrow] + R.bounds_v[2 * lastrow + 1];
    } else {
        y_first = R.top + row0;
        y_last = y_first + nrows;
    }
    const uint32_t trows = (uint32_t)(y_last - y_first);
    const uint32_t ld = (uint32_t)R.max_rows;                // bytes per LDS column
    const uint32_t plane = S * ld;
    // pass 1: horizontal filter (or copy) into LDS, columns [left, left+S)
    for (uint32_t i = threadIdx.x; i < trows * x_n; i += blockDim.x) {
        uint32_t ty, tx;
        if (rot90) { tx = i / trows; ty = i - tx * trows; } else { ty = i / x_n; tx = i - ty * x_n; }
        tx += x_lo;
        const int sy = y_first + (int)ty;      // row in rotated source
        const int ox = R.left + (int)tx;       // column in resized image
        uint32_t o0, o1, o2;
        if (R.need_h) {
            const int xmin = R.bounds_h[2 * ox], cnt = R.bounds_h[2 * ox + 1];
            const int *k = R.kk_h + (size_t)ox * R.ks_h;
            int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
            for (int x = 0; x < cnt; x++) {
                const int sx = xmin + x;
                const uint8_t *p = rot90 ? src + ((size_t)sx * w + (w - 1 - sy)) * 3
                                         : src + ((size_t)sy * w + sx) * 3;
                const int kv = k[x];
                uint32_t px;                     // one unaligned 4-byte load per RGB pixel (the 4th byte is unused;
                __builtin_memcpy(&px, p, 4);     //  every device buffer has slack behind its last element)
                s0 += (int)(px & 0xffu) * kv;
                s1 += (int)((px >> 8) & 0xffu) * kv;
                s2 += (int)((px >> 16) & 0xffu) * kv;
            }
            o0 = clip8(s0); o1 = clip8(s1); o2 = clip8(s2);
        } else {
            const uint8_t *p = rot90 ? src + ((size_t)ox * w + (w - 1 - sy)) * 3
                                     : src + ((size_t)sy * w + ox) * 3;
            o0 = p[0]; o1 = p[1]; o2 = p[2];
        }
        uint8_t *d = tmp + tx * ld + ty;
        d[0] = (uint8_t)o0;
        d[plane] = (uint8_t)o1;
        d[2 * plane] = (uint8_t)o2;
    }
    __syncthreads();
    // pass 2: vertical filter, normalise, scatter patch-major
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    const uint32_t g = S / P;
    for (uint32_t i = threadIdx.x; i < nrows * x_n; i += blockDim.x) {
        const uint32_t ry = i / x_n, rx = x_lo + i % x_n;
        const uint32_t oy = row0 + ry;
        uint32_t q[3];
        const uint8_t *col = tmp + rx * ld;
        if (R.need_v) {
            const int ymin = R.bounds_v[2 * (R.top + oy)] - y_first, cnt = R.bounds_v[2 * (R.top + oy) + 1];
            const int *k = R.kk_v + (size_t)(R.top + oy) * R.ks_v;
            int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
            for (int y = 0; y < cnt; y++) {
                const uint8_t *p = col + ymin + y;
                const int kv = k[y];
                s0 += p[0] * kv;
                s1 += p[plane] * kv;
                s2 += p[2 * plane] * kv;
            }
            q[0] = clip8(s0); q[1] = clip8(s1); q[2] = clip8(s2);
        } else {
            q[0] = col[ry]; q[1] = col[plane + ry]; q[2] = col[2 * plane + ry];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float f = (float)((double)q[c] * (1.0 / 255.0));
            float v = (f - mean[c]) / stdv[c];
            if (pixel_values) pixel_values[(((size_t)img * 3 + c) * S + oy) * S + rx] = v;
            if (patches) {
                uint32_t pr = oy / P, pc = rx / P, iy = oy % P, ix = rx % P;
                size_t rowi = (size_t)img * g * g + pr * g + pc;
                patches[rowi * Kp_pad + (c * P + iy) * P + ix] = f2bf(v);
            }
        }
    }
}

// pixel_values [n][3][S][S] fp32 -> patch-major bf16 (parity entry for the ViT alone)
__global__ void k_patchify(const float *__restrict__ pv, uint32_t n, uint32_t S, uint32_t P,
                           uint16_t *__restrict__ patches, uint32_t Kp_pad)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n * 3 * S * S;
    if (i >= total) return;
    uint32_t x = i % S, y = (i / S) % S, c = (i / ((size_t)S * S)) % 3, img = i / ((size_t)3 * S * S);
    uint32_t g = S / P;
    size_t rowi = (size_t)img * g * g + (y / P) * g + (x / P);
    patches[rowi * Kp_pad + (c * P + (y % P)) * P + (x % P)] = f2bf(pv[i]);
}

// ------------------------------------------------------------------ GEMM

// EPI_RESID_STATS_SPLIT8: byte offset of the lo byte of residual element (row, col) — [col / 64][M_pad / 64][row % 32]
// [(col % 64) / 8][(row / 32) % 2][col % 8]: the 8 columns a lane of the residual epilogue owns, for the row pair
// (r, r + 32) it handles in one group, are 16 contiguous bytes
__device__ __forceinline__ size_t lo8_off(uint32_t M_pad, size_t row, uint32_t col)
{
    return ((((size_t)(col >> 6) * (M_pad >> 6)) + (row >> 6)) << 12) + ((row & 31) << 7) + (((col & 63) >> 3) << 4) + (((row >> 5) & 1) << 3) + (col & 7);
}
__device__ __forceinline__ float split8_value(uint32_t hi16, int32_t lo8) { return __uint_as_float((hi16 << 16) + (uint32_t)(lo8 << 8)); }
// bits 8..15 of the result are the lo byte (the caller picks byte 1): round to nearest, saturating at +127
__device__ __forceinline__ int32_t split8_round(float x, uint32_t hi_bits /* hi16 << 16 */)
{
    const int32_t dlt = (int32_t)__float_as_uint(x) - (int32_t)hi_bits;
    return min(dlt + 0x80, 0x7fff);
}

// Epilogues.  0-3: the plain ones.  4-7: the LayerNorm-folded transformer block —
//   LN(x) W^T + b = rstd (x (gamma o W)^T - mu colsum(gamma o W)) + (b + W beta)
// so the GEMM that follows a LayerNorm takes the RAW residual row (bf16) as its A operand and applies
// the row statistics in its epilogue (EPI_LN_*), and the GEMM that produces the residual writes the bf16
// operand copy and per-row partial sums (EPI_RESID_STATS_*): no LayerNorm kernel, no LN round trip
// through HBM.  _F32X keeps the fp32 residual stream next to the bf16 copy, _BF16 keeps only bf16, _SPLIT keeps
// the residual as TWO bf16 arrays, x ~ hi + lo with hi = bf16(x) (the operand copy) and lo = bf16(x - hi): 16
// mantissa bits for the bytes of one fp32 array, i.e. a third less residual traffic than _F32X.  _SPLIT8 keeps lo in ONE
// byte: with hi = bf16(x) rounded to nearest, the fp32 BIT PATTERN of x lies within +-0x8000 of (hi << 16), and lo8 =
// round((bits(x) - (hi << 16)) / 256) in [-128, 127]; x' = as_float((hi << 16) + (lo8 << 8)) is x rounded to 16 significant
// bits (relative error <= 2^-17, the fp16-lo pair's is comparable), integer arithmetic only on both sides, and a quarter
// less residual traffic again (hi 2 + lo 1 bytes read and written per element instead of 2 + 2).
enum { EPI_F32 = 0, EPI_BIAS_BF16 = 1, EPI_BIAS_GELU_BF16 = 2, EPI_BIAS_RESID_F32 = 3,
       EPI_LN_BIAS_BF16 = 4, EPI_LN_BIAS_GELU_BF16 = 5, EPI_RESID_STATS_F32X = 6, EPI_RESID_STATS_BF16 = 7,
       EPI_RESID_STATS_SPLIT = 8, EPI_RESID_STATS_SPLIT8 = 9,
       // fp8 operands (k_gemm8 only; see "fp8 blocks" below): the accumulator is multiplied by the weight matrix's scale first
       EPI_F8_BIAS_BF16 = 10, EPI_F8_BIAS_GELU_Q8 = 11, EPI_F8_RESID_STATS_SPLIT8 = 12, EPI_KINDS = 13 };
#define EPI_IS_F8(E) ((E) >= EPI_F8_BIAS_BF16)
#define EPI_IS_LN(E) ((E) == EPI_LN_BIAS_BF16 || (E) == EPI_LN_BIAS_GELU_BF16)
#define EPI_IS_STATS(E) ((E) == EPI_RESID_STATS_F32X || (E) == EPI_RESID_STATS_BF16 || (E) == EPI_RESID_STATS_SPLIT || (E) == EPI_RESID_STATS_SPLIT8 || (E) == EPI_F8_RESID_STATS_SPLIT8)
#define EPI_IS_F32_LAYOUT(E) ((E) == EPI_F32 || (E) == EPI_BIAS_RESID_F32)

// operands of the folded epilogues
struct EpiAux {
    const float *cs;       // EPI_LN_*: [N] column sums of the folded (bf16-rounded) weight
    const float2 *ab;      // EPI_LN_*: [M_pad] per row (rstd, -rstd * mean) of the residual row
    uint16_t *xb;          // EPI_RESID_STATS_*: bf16 residual, tile-major [N / 64][M_pad][64] (read + written by _BF16 / _SPLIT, written by _F32X)
    float2 *part;          // EPI_RESID_STATS_*: [N / 64][M_pad] partial (sum, sum of squares) per 64-column group (needs hm_rows = M_pad)
    uint16_t *xlo;         // EPI_RESID_STATS_SPLIT: low half of the split residual, same layout; _SPLIT8: the lo BYTES in the
                           // layout of lo8_off() below (a lane's 16-byte access = its 8 columns of rows r and r + 32)
    // Layout of bf16 activations.  "Tile-major" = [cols / 64][M_pad][64]: the 64-column group a wave tile produces
    // (one attention head; one K-tile of the GEMM that consumes it) is a contiguous plane, so an epilogue writes
    // whole 128-byte rows back to back and the consumer's LDS-DMA reads 8 KiB runs, instead of 128-byte pieces
    // at a row stride of 1.5-6 KiB (HBM delivers about half its streaming rate on those).
    uint32_t hm_rows;      // bf16 output (and xb / xlo) tile-major with this many rows per plane (M_pad); 0 = row-major
    uint32_t a_rs, a_ks;   // A operand: elements between rows, and between K-tiles (row-major: K, 64; tile-major: 64, 64 M_pad); 0,0 = row-major
    const uint32_t *m_dev; // k_gemm only: the number of live rows, in DEVICE memory (a product over a compacted row list whose length the
                           // host does not know: the grid covers the worst case and workgroups whose row tile starts at or beyond it leave at once)
    // EPI_F8_*: A = e4m3 bytes in planes [K / 64][M_pad][64], W = e4m3 bytes [N][K]
    const uint32_t *a_scale; // E8M0 scale bytes of A per (row, 64-column group), in the layout of q8_scale_off()
    float w_scale;           // the weight matrix's (per-tensor, power-of-two) scale: result = acc * w_scale + bias
    uint8_t *q8;             // EPI_F8_BIAS_GELU_Q8: e4m3 output planes [N / 64][hm_rows][64] ...
    uint8_t *q8_scale;       // ... and its scale bytes (q8_scale_off)
};

// ---- fp8 activations (OCP e4m3): one E8M0 scale byte per (row, group of 64 columns), 2^(byte - 127) >= amax / 448 ----
__device__ __forceinline__ uint32_t q8_scale_byte(float amax)
{
    const int e = (int)((__float_as_uint(amax) + 0x200000u) >> 23);          // biased exponent of amax, one more when its mantissa is >= 1.75
    return (uint32_t)min(max(e - 8, 1), 253);
}
__device__ __forceinline__ float q8_inv_scale(uint32_t byte) { return __uint_as_float((254u - byte) << 23); }     // 2^(127 - byte)
// scale bytes: [group / 2][M_pad / 64][32][4] — the dword of (K-tile = group pair, row % 32 of a 64-row block) holds the bytes of rows
// (r, r + 32) x groups (even, odd): what one lane of the GEMM's 32x32x64 MFMAs needs for its two m-tiles and a K-tile's two MFMAs
__device__ __forceinline__ size_t q8_scale_off(uint32_t M_pad, size_t row, uint32_t group)
{
    return ((((size_t)(group >> 1) * (M_pad >> 6) + (row >> 6)) * 32 + (row & 31)) << 2) + (((row >> 5) & 1) << 1) + (group & 1u);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 q8_pack8(const float (&f)[8], float inv)
{
    const f32x2 i2 = f32x2{inv, inv};
    const f32x2 q0 = f32x2{f[0], f[1]} * i2, q1 = f32x2{f[2], f[3]} * i2, q2 = f32x2{f[4], f[5]} * i2, q3 = f32x2{f[6], f[7]} * i2;      // v_pk_mul_f32
    uint32_t a = __builtin_amdgcn_cvt_pk_fp8_f32(q0[0], q0[1], 0, false);
    a = __builtin_amdgcn_cvt_pk_fp8_f32(q1[0], q1[1], a, true);
    uint32_t b = __builtin_amdgcn_cvt_pk_fp8_f32(q2[0], q2[1], 0, false);
    b = __builtin_amdgcn_cvt_pk_fp8_f32(q3[0], q3[1], b, true);
    return make_uint2(a, b);
}

// Measurement switches (ablation masks, cycle stamps) live in clip_dev.h, which only development builds (make DEV=1,
// or any ablation mask) include; a product build sees the masks as the constant 0 and no stamp code at all.
#ifdef D2R_DEV
#include "clip_dev.h"
#else
#define D2R_GEMM_ABLATE 0
#define D2R_ATTN_ABLATE 0
#define D2R_F8_EXP 0
#define D2R_F8_VAR 0
#define STAMP(var)
#endif
#ifndef D2R_GEMM_LATE_DRAIN
#define D2R_GEMM_LATE_DRAIN 1  /* 1 (default since round 6): k_gemm8 does not drain the previous tile's epilogue stores at the top of a tile (see there); 0: the round-5 drain */
#endif
#ifndef D2R_GEMM_PRIO
#define D2R_GEMM_PRIO 2        /* 0: s_setprio 1 around every MFMA section; 1 / 2: static priority for wave row 1 / 0; 3: none */
#endif
#define BM 256                 /* row padding of every GEMM operand buffer (largest tile height) */
#define BK 64
#define GEMM_THREADS 512

__device__ __forceinline__ uint32_t lds_off(uint32_t row, uint32_t chunk)
{
    return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4);
}

// Direct global->LDS copy of one 16-byte chunk per lane (1 KiB per wave instruction).  The LDS
// destination is wave-uniform base (M0) + lane*16, so the XOR swizzle is applied on the SOURCE
// side.  Issued from inline asm so that hipcc does not see an in-flight LDS write and drain it
// (vmcnt(0)) in front of every ds_read of the tile being computed; the waits are placed by hand
// (cdna_hip_programming.md 5.7).  lds_byte_addr must be wave-uniform.  M0 is not preserved:
// nothing else in these kernels uses it.
__device__ __forceinline__ void glds16(const void *g, uint32_t lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const void *)p;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue shared by the GEMM kernels: each wave transposes its fp32 tile through LDS 32 rows at a
// time (the staging ring is free: the main loop ended on a barrier with no DMA in flight) so that a
// lane owns consecutive columns of a row: bias / quick_gelu / fp32 residual on float4, 16-byte stores.
// acc[i][j][r] <-> row 32i+(r&3)+8(r>>2)+4hi, col 32j+li of the wave tile whose origin is (row0, col0).
#define EP_LD 68u                          /* floats per row of the epilogue's LDS transpose buffer (pad 4) */
#define EP_WAVE_FLOATS (8u * EP_LD)        /* 8 rows per wave: 2176 bytes */
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// Every output buffer has M_pad rows, so rows >= M_real are computed and stored like the others
// (they only ever feed rows >= M_real downstream): the epilogue is straight-line code, which lets
// hipcc place exact counted vmcnt waits instead of a vmcnt(0) in every predicated block.
// `hook` runs once: the persistent kernel requests the next tile's operands there.  It sits where no
// later compiler-placed wait can also wait for those requests: after the bias values have been
// consumed (bf16 variants), after the LAST batch of residual loads has been issued (fp32 residual).
// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in every lane: four rotate-adds
__device__ __forceinline__ float row16_sum(float x)
{
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));   // row_ror:8
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false));   // row_ror:4
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122, 0xf, 0xf, false));   // row_ror:2
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121, 0xf, 0xf, false));   // row_ror:1
    return x;
}
// sum over the 8 lanes of an aligned group (lanes 8g .. 8g+7), result in every lane: xor 1, xor 2 inside the quad,
// then the other quad of the group through row_half_mirror (lane i <-> 7 - i)
__device__ __forceinline__ float row8_sum(float x)
{
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));    // quad_perm:[1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));    // quad_perm:[2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, false));   // row_half_mirror
    return x;
}
__device__ __forceinline__ float row8_max(float x)
{
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, false)));
    return x;
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// ab_lds (EPI_LN_* only): this wave's LDS copy of (rstd, -rstd*mean) for the MT*32 rows of its tile.
template <int EPI, int MT, typename Hook = NoHook>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[MT][2], float *ep, uint32_t lane,
                                              uint32_t row0, uint32_t col0, const float *__restrict__ bias,
                                              void *__restrict__ Cout, uint32_t N, const EpiAux &aux,
                                              const float2 *ab_lds, Hook hook = Hook())
{
    // ep: this wave's private 8 x EP_LD float buffer.  Eight rows of the wave tile at a time:
    // acc[i][j][4p..4p+3] of both lane halves are rows 8p..8p+7 of m-tile i.  Register (j, r) of all 64 lanes is two
    // rows (r and r + 4, one per lane half) of 32 consecutive columns: exactly the lane-linear image
    // ds_write_addtid_b32 stores (address = M0 + offset + 4 lane, no address VGPR, 128 B/clk where ds_write_b32 does
    // 64 — the transposes' LDS writes were what an epilogue's time was made of: 256 KiB per tile at 64 B/clk).
    // Buffer layout: dword (4 j + r) * EP_LD + 32 hi + li, i.e. logical row R = r + 4 hi of the 8-row group sits in
    // row-pair slot R & 3, half R >> 2.
    const uint32_t ep_addr = __builtin_amdgcn_readfirstlane(lds_addr(ep));
    auto transpose_in = [&](int i, int p8) {
        if (D2R_GEMM_ABLATE & 256) return;
        asm volatile("s_mov_b32 m0, %8\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:272\n\t"
                     "ds_write_addtid_b32 %2 offset:544\n\tds_write_addtid_b32 %3 offset:816\n\t"
                     "ds_write_addtid_b32 %4 offset:1088\n\tds_write_addtid_b32 %5 offset:1360\n\t"
                     "ds_write_addtid_b32 %6 offset:1632\n\tds_write_addtid_b32 %7 offset:1904"
                     :
                     : "v"(acc[i][0][4 * p8 + 0]), "v"(acc[i][0][4 * p8 + 1]), "v"(acc[i][0][4 * p8 + 2]), "v"(acc[i][0][4 * p8 + 3]),
                       "v"(acc[i][1][4 * p8 + 0]), "v"(acc[i][1][4 * p8 + 1]), "v"(acc[i][1][4 * p8 + 2]), "v"(acc[i][1][4 * p8 + 3]),
                       "s"(ep_addr)
                     : "memory");
    };
    static_assert(EP_LD == 68u, "the ds_write_addtid offsets above are (4 j + r) * EP_LD * 4 bytes");
    // float offset of logical row R (0..7), column c (0..63) of the 8-row group in ep
    auto ep_at = [](uint32_t R, uint32_t c) -> uint32_t { return ((c >> 5) * 4u + (R & 3u)) * EP_LD + (R >> 2) * 32u + (c & 31u); };
    if (EPI_IS_F32_LAYOUT(EPI)) {
        // fp32 outputs: a lane owns 4 columns of a row, 16 lanes (one DPP row) a 256-byte row segment
        const uint32_t c4 = (lane & 15) * 4, rl0 = lane >> 4;
        const uint32_t col = col0 + c4;
        constexpr bool RESID_F32 = EPI == EPI_BIAS_RESID_F32;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI != EPI_F32) {
            bv = *(const float4 *)(bias + col);
            asm volatile("" : "+v"(bv.x), "+v"(bv.y), "+v"(bv.z), "+v"(bv.w));      // loaded before anything else is issued
        }
        if (EPI == EPI_F32) hook();
        // the residual rows of one m-tile (8 loads per lane) are requested together, before its first transpose
#pragma unroll
        for (int ih = 0; ih < MT; ih++) {
            float4 xr[8];
            if (RESID_F32) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t row = row0 + ih * 32 + rl0 + 4 * k;
                    xr[k] = (D2R_GEMM_ABLATE & 128) ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4 *)((const float *)Cout + row * N + col);
                }
                if (ih == MT - 1) hook();
            }
#pragma unroll
            for (int p8 = 0; p8 < 4; p8++) {
                transpose_in(ih, p8);
#pragma unroll
                for (int qq = 0; qq < 2; qq++) {
                    const int k = 2 * p8 + qq;
                    const uint32_t row = row0 + ih * 32 + rl0 + 4 * k;
                    float4 v = (D2R_GEMM_ABLATE & 256) ? make_float4(acc[ih][0][k], acc[ih][0][k + 8], acc[ih][1][k], acc[ih][1][k + 8])
                                                       : *(const float4 *)(ep + ep_at(rl0 + 4 * qq, c4));
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (RESID_F32) {
                        v.x += xr[k].x; v.y += xr[k].y; v.z += xr[k].z; v.w += xr[k].w;
                    }
#if (D2R_GEMM_ABLATE & 128) && defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                    continue;
#endif
                    *(float4 *)((float *)Cout + row * N + col) = v;       // < 2^32 elements (checked on host)
                }
            }
        }
    } else if (EPI_IS_STATS(EPI)) {
        // Residual + LayerNorm statistics (bf16 residual as one array or hi + lo; or fp32 residual + bf16 operand copy):
        // same lane layout as the bf16 outputs below — a lane owns 8 columns of a row, so the residual comes in and
        // goes out in 16-byte pieces, 1 KiB per wave instruction (in the 4-column layout of the plain fp32 outputs a
        // tile took 160 half-width memory instructions per wave and its epilogue 32 k cycles, against 5-8 k for the
        // bf16 outputs).  The residual rows of two m-tiles (16 loads per lane) are requested before the first transpose.
        constexpr bool SPLIT = EPI == EPI_RESID_STATS_SPLIT;
        constexpr bool SPLIT8 = EPI == EPI_RESID_STATS_SPLIT8 || EPI == EPI_F8_RESID_STATS_SPLIT8;
        constexpr bool F32X = EPI == EPI_RESID_STATS_F32X;
        const float ws = EPI_IS_F8(EPI) ? aux.w_scale : 1.0f;
        const uint32_t c8 = (lane & 7) * 8, rl0 = lane >> 3;
        const uint32_t col = col0 + c8;
        float4 b0 = *(const float4 *)(bias + col), b1 = *(const float4 *)(bias + col + 4);
        asm volatile("" : "+v"(b0.x), "+v"(b0.y), "+v"(b0.z), "+v"(b0.w), "+v"(b1.x), "+v"(b1.y), "+v"(b1.z), "+v"(b1.w));
      if constexpr (SPLIT8) {
        // hi as in _SPLIT; lo bytes: ONE 16-byte access per (group of two m-tiles, k) covers this lane's 8 columns of rows
        // r (m-tile ih) and r + 32 (m-tile ih + 1), so the loop runs k outside, the m-tile pair inside
        static_assert(MT % 2 == 0, "the lo-byte layout pairs rows r and r + 32");
        constexpr uint32_t xs = 128u;
        const uint32_t x0 = (((col0 >> 6) * aux.hm_rows + row0 + rl0) * 64u + c8) * 2u;
        const uint32_t p0 = ((col0 >> 6) * aux.hm_rows + row0 + rl0) * 8u;
        const uint32_t l0 = (((col0 >> 6) * (aux.hm_rows >> 6) + (row0 >> 6)) << 12) + rl0 * 128u + (lane & 7) * 16u;
        char *xb_b = (char *)aux.xb, *xlo_b = (char *)aux.xlo, *part_b = (char *)aux.part;
#pragma unroll
        for (int ih = 0; ih < MT; ih += 2) {
            uint4 xh[2][4], xl[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                xh[0][k] = *(const uint4 *)(xb_b + x0 + (ih * 32 + 8 * k) * xs);
                xh[1][k] = *(const uint4 *)(xb_b + x0 + ((ih + 1) * 32 + 8 * k) * xs);
                xl[k] = *(const uint4 *)(xlo_b + l0 + (ih >> 1) * 4096u + k * 1024u);
            }
            if (ih == MT - 2) hook();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t lo_out[4];
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    transpose_in(ih + ii, k);
                    const uint32_t rr = (ih + ii) * 32 + 8 * k;
                    const float4 u = *(const float4 *)(ep + ep_at(rl0, c8));
                    const float4 w = *(const float4 *)(ep + ep_at(rl0, c8) + 4);
                    float f[8] = {u.x + b0.x, u.y + b0.y, u.z + b0.z, u.w + b0.w, w.x + b1.x, w.y + b1.y, w.z + b1.z, w.w + b1.w};
                    if constexpr (EPI_IS_F8(EPI)) {
                        f[0] = fmaf(u.x, ws, b0.x); f[1] = fmaf(u.y, ws, b0.y); f[2] = fmaf(u.z, ws, b0.z); f[3] = fmaf(u.w, ws, b0.w);
                        f[4] = fmaf(w.x, ws, b1.x); f[5] = fmaf(w.y, ws, b1.y); f[6] = fmaf(w.z, ws, b1.z); f[7] = fmaf(w.w, ws, b1.w);
                    }
                    const uint32_t hw[4] = {xh[ii][k].x, xh[ii][k].y, xh[ii][k].z, xh[ii][k].w};
                    const uint32_t lw[2] = {ii ? xl[k].z : xl[k].x, ii ? xl[k].w : xl[k].y};
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const uint32_t hb = (e & 1) ? (hw[e >> 1] & 0xffff0000u) : (hw[e >> 1] << 16);
                        const int32_t t = __builtin_amdgcn_sbfe((int32_t)lw[e >> 2], 8 * (e & 3), 8);
                        f[e] += __uint_as_float(hb + (uint32_t)(t << 8));
                    }
                    const uint4 hv = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
                    *(uint4 *)(xb_b + x0 + rr * xs) = hv;
                    const uint32_t nh[4] = {hv.x, hv.y, hv.z, hv.w};
                    int32_t q[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) q[e] = split8_round(f[e], (e & 1) ? (nh[e >> 1] & 0xffff0000u) : (nh[e >> 1] << 16));
                    // byte 1 of each q: two v_perm_b32 + one shift-or per four elements
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t a = __builtin_amdgcn_perm((uint32_t)q[4 * h + 1], (uint32_t)q[4 * h], 0x0c0c0501u);
                        const uint32_t b = __builtin_amdgcn_perm((uint32_t)q[4 * h + 3], (uint32_t)q[4 * h + 2], 0x0c0c0501u);
                        lo_out[2 * ii + h] = a | (b << 16);
                    }
                    const float sm = row8_sum(((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7])));
                    const float sq = row8_sum((fmaf(f[0], f[0], f[1] * f[1]) + fmaf(f[2], f[2], f[3] * f[3])) +
                                              (fmaf(f[4], f[4], f[5] * f[5]) + fmaf(f[6], f[6], f[7] * f[7])));
                    if ((lane & 7) == 0) *(float2 *)(part_b + p0 + rr * 8u) = make_float2(sm, sq);
                }
                *(uint4 *)(xlo_b + l0 + (ih >> 1) * 4096u + k * 1024u) = make_uint4(lo_out[0], lo_out[1], lo_out[2], lo_out[3]);
            }
        }
      } else {
        // BYTE offsets in 32 bits on top of the uniform array bases (SGPR base + VGPR offset addressing, no 64-bit
        // address per access; launch_gemm checks that the arrays are < 4 GiB and tile-major): this lane's 8 columns of
        // the wave tile's row rl0 in plane col0 / 64; rows are 128 B apart, so the row of an access is an immediate
        constexpr uint32_t xs = 128u;
        const uint32_t x0 = (((col0 >> 6) * aux.hm_rows + row0 + rl0) * 64u + c8) * 2u;
        const uint32_t p0 = ((col0 >> 6) * aux.hm_rows + row0 + rl0) * 8u;              // partial statistics [N / 64][M_pad] float2
        const uint32_t f0 = ((row0 + rl0) * N + col) * 4u, fs = N * 4u;                 // fp32 residual, row-major
        char *xb_b = (char *)aux.xb, *xlo_b = (char *)aux.xlo, *part_b = (char *)aux.part, *c_b = (char *)Cout;
        constexpr int G = F32X ? 1 : 2;          // fp32 rows are twice the registers
        static_assert(MT % G == 0, "epilogue handles m-tiles in groups");
#pragma unroll
        for (int ih = 0; ih < MT; ih += G) {
            uint4 xh[G][4], xl[G][4];
#pragma unroll
            for (int ii = 0; ii < G; ii++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t rr = (ih + ii) * 32 + 8 * k;            // row of the wave tile, minus rl0
                    if (F32X) {
                        xh[ii][k] = *(const uint4 *)(c_b + f0 + rr * fs);
                        xl[ii][k] = *(const uint4 *)(c_b + f0 + rr * fs + 16);
                    } else {
                        xh[ii][k] = *(const uint4 *)(xb_b + x0 + rr * xs);
                        if (SPLIT) xl[ii][k] = *(const uint4 *)(xlo_b + x0 + rr * xs);
                    }
                }
            if (ih == MT - G) hook();
#pragma unroll
            for (int ii = 0; ii < G; ii++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    transpose_in(ih + ii, k);
                    const uint32_t rr = (ih + ii) * 32 + 8 * k;
                    const float4 u = (D2R_GEMM_ABLATE & 256) ? make_float4(acc[ih + ii][0][k], acc[ih + ii][0][k + 8], acc[ih + ii][1][k], acc[ih + ii][1][k + 8])
                                                             : *(const float4 *)(ep + ep_at(rl0, c8));
                    const float4 w = (D2R_GEMM_ABLATE & 256) ? make_float4(acc[ih + ii][0][k + 4], acc[ih + ii][0][k + 12], acc[ih + ii][1][k + 4], acc[ih + ii][1][k + 12])
                                                             : *(const float4 *)(ep + ep_at(rl0, c8) + 4);
                    float f[8] = {u.x + b0.x, u.y + b0.y, u.z + b0.z, u.w + b0.w, w.x + b1.x, w.y + b1.y, w.z + b1.z, w.w + b1.w};
                    const uint32_t hw[4] = {xh[ii][k].x, xh[ii][k].y, xh[ii][k].z, xh[ii][k].w};
                    const uint32_t lw[4] = {xl[ii][k].x, xl[ii][k].y, xl[ii][k].z, xl[ii][k].w};
                    if (F32X) {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            f[e] += __uint_as_float(hw[e]);
                            f[4 + e] += __uint_as_float(lw[e]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            f[2 * e] += bf_lo(hw[e]);
                            f[2 * e + 1] += bf_hi(hw[e]);
                        }
                        if (SPLIT) {
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                f[2 * e] += bf_lo(lw[e]);
                                f[2 * e + 1] += bf_hi(lw[e]);
                            }
                        }
                    }
#if (D2R_GEMM_ABLATE & 128) && defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]));
                    continue;
#endif
                    // the bf16 copy the next GEMM reads as its A operand (hi), what it leaves of the fp32 value (lo) or
                    // the fp32 value itself, and this 64-column group's share of the row's LayerNorm statistics (of the
                    // fp32 values)
                    const uint4 hv = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
                    *(uint4 *)(xb_b + x0 + rr * xs) = hv;
                    if (SPLIT)
                        *(uint4 *)(xlo_b + x0 + rr * xs) = make_uint4(pack2(f[0] - bf_lo(hv.x), f[1] - bf_hi(hv.x)), pack2(f[2] - bf_lo(hv.y), f[3] - bf_hi(hv.y)),
                                                                      pack2(f[4] - bf_lo(hv.z), f[5] - bf_hi(hv.z)), pack2(f[6] - bf_lo(hv.w), f[7] - bf_hi(hv.w)));
                    if (F32X) {
                        *(float4 *)(c_b + f0 + rr * fs) = make_float4(f[0], f[1], f[2], f[3]);
                        *(float4 *)(c_b + f0 + rr * fs + 16) = make_float4(f[4], f[5], f[6], f[7]);
                    }
                    const float sm = row8_sum(((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7])));
                    const float sq = row8_sum((fmaf(f[0], f[0], f[1] * f[1]) + fmaf(f[2], f[2], f[3] * f[3])) +
                                              (fmaf(f[4], f[4], f[5] * f[5]) + fmaf(f[6], f[6], f[7] * f[7])));
                    if ((lane & 7) == 0) *(float2 *)(part_b + p0 + rr * 8u) = make_float2(sm, sq);      // coalesced for k_rowstats
                }
        }
      }
    } else {
        // bf16 outputs: a lane owns 8 columns (one 16-byte store), 8 lanes a 128-byte row segment
        constexpr bool LN = EPI_IS_LN(EPI);
        constexpr bool GELU = EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_LN_BIAS_GELU_BF16 || EPI == EPI_F8_BIAS_GELU_Q8;
        constexpr bool Q8 = EPI == EPI_F8_BIAS_GELU_Q8;
        const float ws = EPI_IS_F8(EPI) ? aux.w_scale : 1.0f;
        const uint32_t c8 = (lane & 7) * 8, rl0 = lane >> 3;
        const uint32_t col = col0 + c8;
        float4 b0 = *(const float4 *)(bias + col), b1 = *(const float4 *)(bias + col + 4);
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        if (LN) {
            s0 = *(const float4 *)(aux.cs + col);
            s1 = *(const float4 *)(aux.cs + col + 4);
            asm volatile("" : "+v"(s0.x), "+v"(s0.y), "+v"(s0.z), "+v"(s0.w), "+v"(s1.x), "+v"(s1.y), "+v"(s1.z), "+v"(s1.w));
        }
        asm volatile("" : "+v"(b0.x), "+v"(b0.y), "+v"(b0.z), "+v"(b0.w), "+v"(b1.x), "+v"(b1.y), "+v"(b1.z), "+v"(b1.w));
        hook();
        // output addressing: a wave-uniform 64-bit base (the wave tile's first row in its plane, or in the row-major
        // array) + a 32-bit lane offset + a wave-uniform row distance: no per-store 64-bit arithmetic, no layout branch
        const uint32_t rs = aux.hm_rows ? 128u : N * 2u;                                   // bytes between rows
        char *const cb = (char *)Cout + (aux.hm_rows ? ((size_t)(col0 >> 6) * aux.hm_rows + row0) * 128u : ((size_t)row0 * N + col0) * 2u);
        const uint32_t loff = rl0 * rs + c8 * 2u;
        // The transposes are software-pipelined: group g + 1 is written to the wave's buffer and read back into a second
        // register set BEFORE group g's values are consumed.  One buffer suffices — a wave's LDS instructions execute in
        // order, so the reads of group g (issued earlier) see group g's data and the writes of group g + 1 land behind
        // them; only the arithmetic waits (lgkmcnt) for its own reads.  Unpipelined, every group paid two LDS round trips
        // back to back (write -> read -> use): 16 groups, ~4 k of a tile's 5 k epilogue cycles.
        // (EPI_F8_BIAS_GELU_Q8: see the store below; row0 is a multiple of 128, so row >> 6 = row0 >> 6 + half, row & 31 = 8 (j & 3) + rl0,
        //  (row >> 5) & 1 = j >> 2 for the group 8 half + j: q8_scale_off() with the lane's part separated)
        uint8_t *q8_base = nullptr, *q8s_base = nullptr;
        uint32_t q8_lane = 0, q8s_lane = 0, q8_mine = 0;
        uint2 q8_prev = make_uint2(0u, 0u);
        if constexpr (Q8) {
            static_assert(MT == 4, "the scale-byte gather is written for a 128-row wave tile");
            q8_base = aux.q8 + (((size_t)(col0 >> 6) * aux.hm_rows + row0) << 6);
            q8_lane = (lane & 1) ? (rl0 + 8u) * 64u + c8 - 8u : rl0 * 64u + c8;
            q8s_base = aux.q8_scale + ((((size_t)(col0 >> 7) * (aux.hm_rows >> 6) + (row0 >> 6)) * 32u) << 2) + ((col0 >> 6) & 1u);
            q8s_lane = ((8u * (lane & 3u) + rl0) << 2) + (((lane >> 2) & 1u) << 1);
        }
        constexpr int NG = MT * 4;
        // The buffer reads are issued from inline asm like the writes: hipcc cannot count the LDS operations inside an asm
        // statement, so with compiler-visible reads its own lgkmcnt waits came out too strict (a group's values were
        // waited for together with the NEXT group's eight writes).  Here the compiler sees no LDS traffic at all and the
        // waits are counted by hand: when group g is consumed, exactly the operations of fetch(g + 1) may be outstanding.
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr int FETCH_OPS = 8 + 2 + (LN ? 1 : 0);
        f32x4 ub[2], wb[2];
        f32x2 abv[2];
        const uint32_t rd_addr = lds_addr(ep) + ep_at(rl0, c8) * 4u;
        const uint32_t ab_addr = LN ? lds_addr(ab_lds) + rl0 * 8u : 0u;
        auto fetch = [&](int g, int slot) {
            const int i = g >> 2, k = g & 3;
            transpose_in(i, k);
            if (D2R_GEMM_ABLATE & 256) {
                ub[slot] = f32x4{acc[i][0][k], acc[i][0][k + 8], acc[i][1][k], acc[i][1][k + 8]};
                wb[slot] = f32x4{acc[i][0][k + 4], acc[i][0][k + 12], acc[i][1][k + 4], acc[i][1][k + 12]};
                abv[slot] = f32x2{1.f, 0.f};
                return;
            }
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(ub[slot]), "=&v"(wb[slot]) : "v"(rd_addr) : "memory");
            if (LN) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(abv[slot]) : "v"(ab_addr), "n"((i * 32 + 8 * k) * 8) : "memory");   // (rstd, -rstd * mean) of this row
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int cur = g & 1;
            if (g + 1 < NG) fetch(g + 1, cur ^ 1);
            // group g's values have landed once at most fetch(g + 1)'s operations are outstanding (LDS operations of a
            // wave complete in order); the operands are tied to the wait so that no use is scheduled above it
            if (!(D2R_GEMM_ABLATE & 256)) {
                if (g + 1 < NG) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(ub[cur]), "+v"(wb[cur]), "+v"(abv[cur]) : "n"(FETCH_OPS) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ub[cur]), "+v"(wb[cur]), "+v"(abv[cur]) : : "memory");
            }
            // The arithmetic runs on PAIRS (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 results per instruction slot; the same
            // roundings as the scalar instructions) — these epilogues are VALU-bound (fc1: ~1 800 instructions per wave and tile, a wave64
            // instruction is four cycles, two waves share a SIMD), only exp2 / rcp stay scalar.
            const f32x2 p_in[4] = {f32x2{ub[cur][0], ub[cur][1]}, f32x2{ub[cur][2], ub[cur][3]}, f32x2{wb[cur][0], wb[cur][1]}, f32x2{wb[cur][2], wb[cur][3]}};
            const f32x2 bb[4] = {f32x2{b0.x, b0.y}, f32x2{b0.z, b0.w}, f32x2{b1.x, b1.y}, f32x2{b1.z, b1.w}};
            f32x2 pf[4];
            if (LN) {
                const f32x2 ax = f32x2{abv[cur][0], abv[cur][0]}, ay = f32x2{abv[cur][1], abv[cur][1]};
                const f32x2 ss[4] = {f32x2{s0.x, s0.y}, f32x2{s0.z, s0.w}, f32x2{s1.x, s1.y}, f32x2{s1.z, s1.w}};
#pragma unroll
                for (int e = 0; e < 4; e++) pf[e] = __builtin_elementwise_fma(ax, p_in[e], __builtin_elementwise_fma(ay, ss[e], bb[e]));
            } else if (EPI_IS_F8(EPI)) {
                const f32x2 w2 = f32x2{ws, ws};
#pragma unroll
                for (int e = 0; e < 4; e++) pf[e] = __builtin_elementwise_fma(p_in[e], w2, bb[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) pf[e] = p_in[e] + bb[e];
            }
            if (GELU && !(D2R_F8_EXP & 4)) {
                // quick_gelu: x * sigmoid(1.702 x) = x / (1 + exp2(-1.702 log2(e) x))
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const f32x2 t = pf[e] * f32x2{-2.4554669595930156f, -2.4554669595930156f};
                    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2{1.0f, 1.0f};
                    pf[e] = pf[e] * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                }
            }
            float f[8] = {pf[0][0], pf[0][1], pf[1][0], pf[1][1], pf[2][0], pf[2][1], pf[3][0], pf[3][1]};
            if constexpr (Q8) {
                // e4m3 + one scale byte per (row, this wave tile's 64 columns): the 8 lanes of a row agree on the largest magnitude.
                // Addressing: wave-uniform bases + one lane offset each + immediates (group g is rows 32 (g >> 2) + 8 (g & 3) + rl0 of the
                // wave tile).  The scale bytes of eight groups are collected across the row's eight lanes — lane j keeps the byte of group
                // 8 half + j — and leave in ONE all-lanes byte store per half (a predicated store per group is a branch per group).
                const float am = row8_max(fmaxf(fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))),
                                                fmaxf(fmaxf(fabsf(f[4]), fabsf(f[5])), fmaxf(fabsf(f[6]), fabsf(f[7])))));
                const uint32_t sb = q8_scale_byte(am);
                // 16-byte stores: groups come in pairs 8 rows apart; neighbouring lanes swap halves so that the even lane holds 16 columns of
                // the first group's row and the odd lane 16 columns of the second's (half as many store instructions, each 1 KiB)
                const uint2 pk8 = q8_pack8(f, q8_inv_scale(sb));
                if ((g & 1) == 0) q8_prev = pk8;
                else if (!(D2R_F8_EXP & 2)) {
                    const bool odd = lane & 1;
                    const uint32_t sx = odd ? q8_prev.x : pk8.x, sy = odd ? q8_prev.y : pk8.y;
                    const uint32_t rx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xf, 0xf, false);       // quad_perm:[1,0,3,2]
                    const uint32_t ry = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xf, 0xf, false);
                    const uint4 o = odd ? make_uint4(rx, ry, pk8.x, pk8.y) : make_uint4(q8_prev.x, q8_prev.y, rx, ry);
                    *(uint4 *)(q8_base + (q8_lane + (uint32_t)((g >> 2) * 32 + ((g - 1) & 3) * 8) * 64u)) = o;
                }
                q8_mine = (lane & 7) == (uint32_t)(g & 7) ? sb : q8_mine;
                if ((g & 7) == 7 && !(D2R_F8_EXP & 1)) q8s_base[q8s_lane + (uint32_t)(g >> 3) * 128u] = (uint8_t)q8_mine;
                continue;
            }
            const uint4 pk = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
#if (D2R_GEMM_ABLATE & 128) && defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(pk.x), "v"(pk.y), "v"(pk.z), "v"(pk.w));
            continue;
#endif
            *(uint4 *)(cb + (loff + (uint32_t)((g >> 2) * 32 + (g & 3) * 8) * rs)) = pk;
        }
    }
}

// (Round 6 measured the alternative to the LDS transposes above: the TRANSPOSED product — W fragment as the MFMA's A operand, so that a
// lane holds 4 consecutive columns of one row, pairs its halves with v_permlane32_swap and stores 16 bytes with no LDS traffic.  Correct
// and bit-identical, but a store instruction then writes 32 rows x 32 bytes instead of 8 full 128-byte rows, and the write path answers
// four times the requests slower than the LDS round trip costs: QKV epilogue 4.4 -> 5.5 k cycles, drain 1.6 -> 4.8 k; residual epilogue
// 16.5 -> 28.3 k.  Commit bdc5381, profiles/r06_gemm_stamps_transposed.txt.)
// C = A[M,K] * W[N,K]^T.  A [M_pad][K] bf16, W [N][K] bf16, K % 64 == 0.  The plain-K-loop GEMM that
// serves what k_gemm8 below does not (outputs with few 256x256 tiles, K not a multiple of 128).
// 8 waves as WGM x WGN, each wave MT x 2 MFMA 32x32x16 tiles:
//   <WGM=4, WGN=2, MT=2, STAGES=3>  256x128 tile, 3-stage ring, counted vmcnt
//   <WGM=2, WGN=4, MT=4, STAGES=2>  256x256 tile, 2-stage
// LDS-DMA staging: tile kt+STAGES-1 is issued before tile kt is computed; with 3 stages the wait at
// the end of the iteration is COUNTED (this wave's copies of the newest tile stay in flight across
// the barrier) -- the loads are never drained inside the loop.
template <int EPI, int WGM, int WGN, int MT, int STAGES>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void k_gemm(const uint16_t *__restrict__ A,
                                                          const uint16_t *__restrict__ W,
                                                          const float *__restrict__ bias, void *__restrict__ Cout,
                                                          uint32_t M_pad, uint32_t N, uint32_t K, uint32_t n_xcd,
                                                          EpiAux aux)
{
    constexpr uint32_t TBM = WGM * MT * 32, TBN = WGN * 64;
    constexpr uint32_t STAGE_BYTES = (TBM + TBN) * BK * 2;
    constexpr int NWAVE = WGM * WGN;
    constexpr int A_PER_WAVE = TBM / 8 / NWAVE, B_PER_WAVE = TBN / 8 / NWAVE;   // 1 KiB copies per wave per stage
    constexpr int PER_STAGE = A_PER_WAVE + B_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // XCD-aware tile order (bijective): blocks b, b+n_xcd, ... share an L2; give each XCD a
    // contiguous run of tiles, n fastest so neighbours reuse the same A row panel.
    const uint32_t nwg = gridDim.x, tiles_n = N / TBN;
    const uint32_t xcd = blockIdx.x % n_xcd, loc = blockIdx.x / n_xcd, q = nwg / n_xcd, rr = nwg % n_xcd;
    uint32_t tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    // Device-sized products (aux.m_dev: a compacted row list whose length the host does not know): the grid is a FIXED set of
    // workgroups that walk the live tiles — launching one workgroup per worst-case tile and letting the idle ones leave costs a
    // dispatch per tile (tens of thousands of 144 KiB-LDS workgroups at one per CU: milliseconds of nothing).
    uint32_t live_tiles = 0xffffffffu;
    if (aux.m_dev) {
        live_tiles = ((*aux.m_dev + TBM - 1) / TBM) * tiles_n;
        tile = blockIdx.x;
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..NWAVE-1
    const uint32_t wm = (wave / WGN) * (MT * 32), wn = (wave % WGN) * 64;
    const uint32_t li = lane & 31, hi = lane >> 5;
  for (;; tile += gridDim.x) {                                         // ONE pass unless the row count is device-sized
    if (tile >= live_tiles) break;
    const uint32_t m0 = (tile / tiles_n) * TBM, n0 = (tile % tiles_n) * TBN;

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // staging: an 8-row x 128-byte block per wave instruction.  lane -> (row in block, physical
    // chunk); it fetches the swizzle-inverse logical chunk.
    const uint32_t r_in = lane >> 3, pc = lane & 7;
    const size_t a_rs = aux.a_rs ? aux.a_rs : K, a_ks = aux.a_rs ? aux.a_ks : BK;       // A operand layout (EpiAux)
    const uint16_t *ag[A_PER_WAVE], *wg[B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; i++) {
        const uint32_t row = (wave * A_PER_WAVE + i) * 8 + r_in;
        ag[i] = A + (size_t)(m0 + row) * a_rs + (pc ^ ((row >> 1) & 7u)) * 8;
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; i++) {
        const uint32_t row = (wave * B_PER_WAVE + i) * 8 + r_in;
        wg[i] = W + (size_t)(n0 + row) * K + (pc ^ ((row >> 1) & 7u)) * 8;
    }
    const uint32_t nk = K / BK;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // copy `idx` (0 .. PER_STAGE-1) of this wave's share of tile kt into ring buffer buf
    auto stage_one = [&](uint32_t buf, uint32_t kt, int idx) {
        const uint32_t base = lds0 + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; i++)
            if (idx == i) glds16(ag[i] + (size_t)kt * a_ks, base + (wave * A_PER_WAVE + i) * 1024);
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; i++)
            if (idx == A_PER_WAVE + i)
                glds16(wg[i] + (size_t)kt * BK, base + TBM * BK * 2 + (wave * B_PER_WAVE + i) * 1024);
    };
    auto stage = [&](uint32_t buf, uint32_t kt) {
#pragma unroll
        for (int c = 0; c < PER_STAGE; c++) stage_one(buf, kt, c);
    };

    // prologue: STAGES-1 tiles in flight, the first one landed
    stage(0, 0);
    if (STAGES == 3 && nk > 1) {
        stage(1, 1);
        wait_vmcnt<PER_STAGE>();
    } else {
        wait_vmcnt<0>();
    }
    __syncthreads();

    uint32_t cur = 0;
    for (uint32_t kt = 0; kt < nk; kt++) {
        // refill the buffer that was computed in the previous iteration (everyone left it at the barrier)
        const uint32_t ahead = kt + STAGES - 1;
        const uint32_t nbuf = cur >= 1 ? cur - 1 : STAGES - 1;
        const bool fill = ahead < nk;                  // block-uniform
        const uint8_t *Ab = smem + cur * STAGE_BYTES;
        const uint8_t *Bb = Ab + TBM * BK * 2;
        // fragments are double-buffered in registers: k-step s+1 is read from LDS while the MFMAs of
        // k-step s issue; the MFMA groups run at raised priority so the partner wave's loads yield
        uint4 fa[2][MT], fb[2][2];
#pragma unroll
        for (int i = 0; i < MT; i++) fa[0][i] = *(const uint4 *)(Ab + lds_off(wm + i * 32 + li, hi));
#pragma unroll
        for (int j = 0; j < 2; j++) fb[0][j] = *(const uint4 *)(Bb + lds_off(wn + j * 32 + li, hi));
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int cb = s & 1, nb = cb ^ 1;
            if (s + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MT; i++) fa[nb][i] = *(const uint4 *)(Ab + lds_off(wm + i * 32 + li, 2 * (s + 1) + hi));
#pragma unroll
                for (int j = 0; j < 2; j++) fb[nb][j] = *(const uint4 *)(Bb + lds_off(wn + j * 32 + li, 2 * (s + 1) + hi));
            }
            // this k-step's share of the next tile's LDS-DMA: issued between the fragment reads and
            // the MFMA group instead of in one burst ahead of the first MFMA of the iteration
            if (fill) {
#pragma unroll
                for (int c = s * ((PER_STAGE + 3) / 4); c < (s + 1) * ((PER_STAGE + 3) / 4) && c < PER_STAGE; c++)
                    stage_one(nbuf, ahead, c);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    union { uint4 u; bf16x8 v; } a, b;
                    a.u = fa[cb][i];
                    b.u = fb[cb][j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
        }
        // tile kt+1 must have landed; with 3 stages this wave's copies of tile kt+2 stay in flight
        if (STAGES == 3 && ahead < nk)
            wait_vmcnt<PER_STAGE>();
        else
            wait_vmcnt<0>();
        __syncthreads();
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
    // the ring is free (the loop ended on a barrier with nothing in flight): its first bytes serve as
    // the epilogue's transpose buffers
    // (EPI_LN_*: the rows' (rstd, -rstd*mean) pairs go into a private LDS strip behind the transpose buffers:
    // written and read by this wave only, LDS operations of one wave execute in order)
    float2 *abw = (float2 *)(smem + NWAVE * EP_WAVE_FLOATS * 4) + wave * (MT * 32);
    if (EPI_IS_LN(EPI)) {
#pragma unroll
        for (int h = 0; h < MT * 32; h += 64)
            if (h + lane < MT * 32) abw[h + lane] = aux.ab[m0 + wm + h + lane];
    }
    gemm_epilogue<EPI, MT>(acc, (float *)smem + wave * EP_WAVE_FLOATS, lane, m0 + wm, n0 + wn, bias, Cout, N, aux, abw);
    if (!aux.m_dev) break;
    __syncthreads();                  // the ring and the transpose buffers serve the next tile
  }
}

// ---- 256x256x64 GEMM with a half-tile staging ring that never drains ("8-phase" K loop) ----
//
// Same operands, tile order and epilogue as k_gemm<EPI, 2, 4, 4, 2>, different K loop.  The 128 KiB of
// LDS are 8 slots of 16 KiB, one per half-tile kind and K-tile parity:
//     A half h = rows {128 wm + 64 h + 0..63 : wm = 0,1},  B half h = W rows {64 wn + 32 h + 0..31 : wn = 0..3}
// so a wave's 128x64 output splits into four 64x32 quadrants (mh, nh), each needing ONE A half and ONE B
// half.  A K-tile pair is 8 phases; every phase
//     1. reads one half-tile from LDS into fragment registers that the current MFMAs do not use,
//     2. issues the LDS-DMA of one half-tile 6 phases ahead (2 x 1 KiB per wave) into the slot that was
//        read two phases ago,
//     3. runs the 8 MFMAs of one quadrant from fragments read in earlier phases,
//     4. waits until all but its 10 newest DMAs have landed, lgkmcnt(0), workgroup barrier.
// Position s of the half-tile sequence (kind s mod 8: A0 B0 B1 A1 | A0 B1 B0 A1, the second group for
// the odd K-tile) lives in slot s mod 8, is staged in phase s-6, has landed by the barrier that ends
// phase s-1 and is read in phase s; its slot is restaged in phase s+2.  Five half-tiles (80 KiB per CU)
// are in flight at every barrier and no wait ever drains the queue until the last K-tile pair.
// Quadrant order (0,0)(0,1)(1,1)(1,0) | (0,1)(0,0)(1,0)(1,1) makes the register set that a phase
// overwrites the one its MFMAs do not read.  Requires K % 128 == 0, N % 256 == 0.
typedef int v8i32 __attribute__((ext_vector_type(8)));
#define F8_B_SCALE 0x7f7f7f7f         /* the W-side scale operand: 1 (its scale is applied in the epilogue) */
__device__ __forceinline__ void glds16s(uint32_t voff, const void *sbase, uint32_t lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void k_gemm8(const uint16_t *__restrict__ A, const uint16_t *__restrict__ W,
                                                 const float *__restrict__ bias, void *__restrict__ Cout,
                                                 uint32_t M_pad, uint32_t N, uint32_t K, uint32_t n_xcd, EpiAux aux,
                                                 uint32_t stagger_sleeps, uint32_t gn_split)
{
    constexpr uint32_t SLOT = 128 * BK * 2;          // 16 KiB
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // persistent: gridDim.x workgroups (one per CU) walk the tiles.  XCD x (blockIdx mod n_xcd: one L2) owns the row
    // panels [x MP/n_xcd, (x+1) MP/n_xcd) with all their column tiles, and walks them column GROUP by column group
    // (`gn` column tiles at a time, the row panels swept inside a group, column fastest), its workgroups taking
    // consecutive tiles of that order in every round.  With gn = 4 a round of 32 workgroups is 8 row panels x 4
    // column tiles: the group's W panels (gn x K x 512 B = 1.5 MB at K = 768) are touched every round and stay in
    // the 4 MiB L2 while the A panels stream past — with the plain column-fastest order every round cycled ALL of W
    // (3.5-4.7 MB for the QKV / fc1 products) through the L2 and a quarter of the operand reads missed it.
    // Column sections (`gn_split` >> 16 = n_split, 1 = none): the XCDs form n_split sets; set c owns the column tiles
    // [c, c + 1) * tiles / n_split of EVERY row panel, its XCDs share the row panels among them.  A set's W panels
    // (half or a quarter of W) stay in its XCDs' L2s across rounds; the A panels are then read by n_split XCDs, at about
    // the same time (the sets walk the row panels in step), i.e. once from HBM and otherwise from the Infinity Cache.
    const uint32_t n_split = max(1u, gn_split >> 16), gn = gn_split & 0xffffu;
    const uint32_t tiles_n = N / 256 / n_split, mp_all = M_pad / 256;          // column tiles of this XCD's section
    const uint32_t xcd = blockIdx.x % n_xcd, loc = blockIdx.x / n_xcd, per_xcd = gridDim.x / n_xcd;
    const uint32_t csec = xcd % n_split, xr = xcd / n_split, n_xr = n_xcd / n_split;
    const uint32_t mp_lo = (uint32_t)((uint64_t)mp_all * xr / n_xr), mp_cnt = (uint32_t)((uint64_t)mp_all * (xr + 1) / n_xr) - mp_lo;
    const uint32_t t_begin = 0, t_end = mp_cnt * tiles_n;          // local tile ids of this XCD
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave >> 2, wn = wave & 3;
    const uint32_t li = lane & 31, hi = lane >> 5;

    f32x16 acc[4][2];

    // staging: a piece of a half-tile = 8 slot rows x 128 B per wave instruction
    const uint32_t r_in = lane >> 3, pc = lane & 7;
    const uint32_t a_rs = aux.a_rs ? aux.a_rs : K;                                      // A operand layout (EpiAux)
    const size_t a_ks = aux.a_rs ? aux.a_ks : BK;
    // A wave stages slot rows 8 wave + r_in (piece 0) and the same + 64 (piece 1) of every half-tile: 64 slot rows apart the
    // swizzle term ((row >> 1) & 7) is the same and the global rows lie 128 operand rows apart for A and for W, so the
    // two pieces share one lane-offset register per operand and differ by a wave-uniform distance on the scalar base
    uint32_t voffA, voffB;
    {
        const uint32_t sr = wave * 8 + r_in;                                  // slot row of piece 0: 0..63
        const uint32_t chunk = pc ^ ((sr >> 1) & 7u);
        voffA = (sr * a_rs + chunk * 8) * 2;                                  // bytes from the half's first row
        voffB = (((sr >> 5) * 64 + (sr & 31)) * K + chunk * 8) * 2;
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const uint32_t nk = K / BK;
    // kind -> operand / half
    uint32_t m0 = 0, n0 = 0;                           // tile whose K loop runs
    // Running source pointers, one per kind (scalar registers): where the kind's next request reads.  A request
    // advances its pointer by two K-tiles; the last K-tile pair of a tile re-seats the pointers on the next tile first.
    const uint16_t *src[8];
    auto kind_is_a = [](int kind) { return kind == 0 || kind == 3 || kind == 4 || kind == 7; };
    auto seat = [&](int kind, uint32_t tm, uint32_t tn) -> const uint16_t * {       // K-tile (kind >= 4) of tile (tm, tn)
        const uint32_t h = (kind == 2 || kind == 3 || kind == 5 || kind == 7) ? 1u : 0u;
        return kind_is_a(kind) ? A + (size_t)(tm + h * 64) * a_rs + (kind >= 4 ? a_ks : (size_t)0)
                               : W + (size_t)(tn + h * 32) * K + (kind >= 4 ? (size_t)BK : (size_t)0);
    };
    auto stage = [&](int kind) {
        const bool isA = kind_is_a(kind);
        const uint16_t *base = src[kind];
        src[kind] = base + 2 * (isA ? a_ks : (size_t)BK);
        if (D2R_GEMM_ABLATE & 1) return;
#pragma unroll
        for (int qq = 0; qq < 2; qq++)
            glds16s(isA ? voffA : voffB, base + (size_t)qq * 128 * (isA ? a_rs : K), lds0 + kind * SLOT + (wave + qq * 8) * 1024);
    };
    auto tile_origin = [&](uint32_t t, uint32_t &tm, uint32_t &tn) {
        const uint32_t per_group = mp_cnt * gn;                    // tiles of a full column group
        const uint32_t g = t / per_group, r = t - g * per_group;
        const uint32_t gw = min(gn, tiles_n - g * gn);             // the last group may be narrower
        const uint32_t mi = r / gw;
        tm = (mp_lo + mi) * 256;
        tn = (csec * tiles_n + g * gn + (r - mi * gw)) * 256;
    };
    // fragment read addresses: the swizzle term is the same for every row this lane reads
    // ((row >> 1) & 7 == (li >> 1) & 7), so four byte offsets per operand serve all kinds; the slot and
    // m-tile go into the instruction's immediate offset (kinds 4-7 lie beyond its 64 KiB reach and
    // add 65536 on the VALU instead of keeping eight more address registers alive)
    // (the B addresses are the A addresses plus a wave-uniform distance, added per read from an SGPR: four address
    // registers instead of eight — the kernel sits at the 256-register limit and a spill inside the counted-vmcnt K
    // loop is not an option)
    uint32_t aoff[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
        const uint32_t sw = ((2 * s4 + hi) ^ ((li >> 1) & 7u)) << 4;
        aoff[s4] = (wm * 64 + li) * 128 + sw;
    }
    const uint32_t d_ab = (wn * 32 - wm * 64) * 128;          // wave-uniform (modulo 2^32)
    uint4 fa[2][2][4], fb[2][4];
    auto read_pos = [&](int kind) {
        const bool isA = kind == 0 || kind == 3 || kind == 4 || kind == 7;
        const int h = (kind == 2 || kind == 3 || kind == 5 || kind == 7) ? 1 : 0;
        uint32_t far = 0;
        if (kind >= 4) {
            far = 65536;
            asm volatile("" : "+v"(far));            // not loop-invariant for the optimiser
        }
        const uint8_t *sb = smem + (kind & 3) * SLOT;
        if (D2R_GEMM_ABLATE & 2) {
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                if (isA) fa[h][0][s4] = fa[h][1][s4] = make_uint4(lane, kind, s4, far);
                else fb[h][s4] = make_uint4(lane, kind, s4, far);
            }
            return;
        }
        if (isA) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) fa[h][mt][s4] = *(const uint4 *)(sb + (aoff[s4] + far) + mt * 4096);
        } else {
            uint32_t db = d_ab + (kind >= 4 ? 65536u : 0u);
            asm volatile("" : "+s"(db));             // an SGPR operand of the adds below, not four loop-invariant VGPRs
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) fb[h][s4] = *(const uint4 *)(sb + (aoff[s4] + db));
        }
    };
    auto mfma_quadrant = [&](int mh, int nh) {
#if D2R_GEMM_PRIO == 0
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                union { uint4 u; bf16x8 v; } a, b;
                a.u = fa[mh][mt][s4];
                b.u = fb[nh][s4];
#if (D2R_GEMM_ABLATE & 8) && defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(a.u.x), "v"(b.u.x));
#else
                acc[mh * 2 + mt][nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[mh * 2 + mt][nh], 0, 0, 0);
#endif
            }
#if D2R_GEMM_PRIO == 0
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // positions 0..7 (both K-tiles of the first pair) of the first tile are requested at once; later tiles
    // get theirs from the previous tile's last K-tile pair
    uint32_t t = t_begin + loc;
    if (t >= t_end) return;
    tile_origin(t, m0, n0);
#pragma unroll
    for (int kind = 0; kind < 8; kind++) {
        src[kind] = seat(kind, m0, n0);
        stage(kind);
    }
    // Every tile of a launch costs the same, so workgroups that start together stay together: the whole chip runs K
    // loops (HBM nearly idle), then the whole chip runs epilogues (HBM saturated: the epilogues' loads and stores
    // took 30 of a step's 145 GEMM ms at exactly the HBM rate).  Workgroup `loc` of each XCD starts (loc mod 8) / 8
    // of a tile period late, once; equal tile times keep the eight phases apart for the rest of the launch, so at
    // any moment an eighth of the CUs are in their epilogue and the memory system sees a steady stream.
    for (uint32_t i = 0, n = (loc & 7u) * stagger_sleeps; i < n; i++) __builtin_amdgcn_s_sleep(127);
    float *ep = (float *)(smem + 8 * SLOT) + wave * EP_WAVE_FLOATS;      // transpose buffer behind the ring
    // EPI_LN_*: 1 KiB per wave behind the transpose buffers for the (rstd, -rstd*mean) pairs of its 128 rows,
    // fetched by ONE LDS-DMA instruction per tile (16 B = two rows per lane) so that it is ordered by the same
    // hand-counted vmcnt waits as the operand ring (a compiler-visible load would be waited for with vmcnt(0),
    // i.e. for the whole next tile's first K-tile pair)
    const float2 *ab_lds = (const float2 *)(smem + 8 * SLOT + 8 * EP_WAVE_FLOATS * 4) + wave * 128;

#if D2R_GEMM_LATE_DRAIN
    bool first_tile = true;                           // block-uniform
#endif
    for (;;) {
    STAMP(ts0);               // (the first bucket of the cycle stamps: accumulator reset + drain + first fragment reads)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // everything in flight is drained ONCE per tile: the eight half-tiles staged during the previous tile's
    // last K-tile pair and that tile's epilogue stores.  (Stores share the vmcnt counter with the loads.  The counted
    // waits below would stay sound with stores outstanding — vmcnt(N) with N loads younger than the awaited one bounds
    // the pending LOADS, stores only make it wait longer — but then the first of them wait for the same store
    // acknowledgements: measured, no difference; the drain keeps the compiler's own bookkeeping trivial.)
    // (the builtin, not inline asm: hipcc's own wait-count bookkeeping must also learn that the
    // epilogue's loads and stores have retired, or it protects their registers with a vmcnt(0) of its
    // own inside the K loop)
#if D2R_GEMM_LATE_DRAIN
    // No drain (same-box A/B, profiles/r06_ab_gemm.md: CLIP 140.1 -> 138.9 ms per configs[1] step, bit-identical results).  The eight half-tiles this tile starts on were requested BEFORE the previous tile's epilogue, whose first act is
    // to load and consume bias values: loads retire in order, so those half-tiles had landed before the first epilogue store was even
    // issued.  What the drain really waits for are the store acknowledgements (2-7 k cycles per tile, profiles/r06_gemm_stamps.txt: the
    // whole chip writes its tiles at the same moment) — and nothing needs them before the first request of THIS tile is awaited, in
    // phase 5 of the first K-tile pair (phases 0..4 read positions 3..7).  So: only a workgroup's first tile waits here, and the first
    // pair's phases 0..4 carry no vmcnt wait (below); the stores get ~1.5 k cycles of MFMA work to retire behind.
    // (EPI_F32 loads nothing in its epilogue: no such guarantee, it keeps the drain)
    if (first_tile || EPI == EPI_F32) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        wait_vmcnt<0>();
    }
    first_tile = false;
    bar();
#else
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), lgkmcnt / expcnt untouched
    wait_vmcnt<0>();
    bar();
#endif
    read_pos(0);
    read_pos(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();                                            // slots 0, 1 may be restaged from phase 2 on
    STAMP(ts1);
    // The two wave rows run half a phase apart: wm = 1 starts one barrier late, so that on every SIMD
    // one wave is in its MFMA section while the other reads LDS and issues DMA.  A wave waits for the
    // DMA that the NEXT phase reads at the end of its section that precedes the barrier in front of
    // group 0's next read section: after the MFMAs for wm = 0, after the DMA issue for wm = 1.
    if (wm == 1) bar();
#if D2R_GEMM_PRIO == 1
    if (wm == 1) __builtin_amdgcn_s_setprio(1);      // static priority for the second-dispatched half
#elif D2R_GEMM_PRIO == 2
    if (wm == 0) __builtin_amdgcn_s_setprio(1);
#endif

    // The ring runs on across tile boundaries: the last K-tile pair of a tile stages the first pair of the
    // workgroup's NEXT tile, so the memory pipe (the resource this kernel is bound by) is not left idle
    // while the last pair is consumed, and the epilogue's stores do not share it with a burst of requests.
    const uint32_t t_next = t + per_xcd;
    const bool has_next = t_next < t_end;             // block-uniform
    uint32_t m0n = 0, n0n = 0;
    if (has_next) tile_origin(t_next, m0n, n0n);
    const uint32_t n_iter = nk / 2;
    // Every K-tile pair stages eight positions, the last pair of a workgroup's last tile too (it re-requests the first
    // pair of the launch's first tile: valid addresses, never read), so the waits are the same vmcnt(10) everywhere and
    // the loop carries no end-of-work special case; the queue is drained before the workgroup exits.
    for (uint32_t u = 0; u < n_iter; u++) {
        const bool last = u + 1 == n_iter;            // block-uniform
#pragma unroll
        for (int P = 0; P < 8; P++) {
            // read section: position 8u+P+2 (kind (P+2) mod 8) for the next phase's MFMAs; stage position
            // 8u+P+8 (kind P): K-tile 2(u+1) + (P >= 4) of this tile, or K-tile (P >= 4) of the next one
            if (EPI_IS_LN(EPI) && last && P == 0) {     // older than every request that follows: landed by the last counted wait
                const uint32_t l16 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) * 16u;   // not a value kept live across the loop
                glds16s(l16, aux.ab + m0 + wm * 128, lds0 + 8 * SLOT + 8 * EP_WAVE_FLOATS * 4 + wave * 1024);
            }
            if (!(last && P >= 6)) read_pos((P + 2) & 7);
            if (last) src[P] = seat(P, m0n, n0n);
            stage(P);
            // position 8u+P+3 must have landed before the barrier in front of the read section that takes it; younger
            // requests stay in flight.  That barrier is this one for wave row 1 and the one behind the MFMAs for wave row
            // 0 (half a phase apart); both rows wait at both — the second wait of a row finds its count already met, and
            // two unconditional waits are cheaper than a wave-uniform branch around each
            if (!(D2R_GEMM_LATE_DRAIN && P < 5) || u != 0) wait_vmcnt<10>();
            __builtin_amdgcn_sched_barrier(0);
            bar();
            // MFMA section
            const int mh = (P == 2 || P == 3 || P == 6 || P == 7) ? 1 : 0;
            const int nh = (P == 1 || P == 2 || P == 4 || P == 7) ? 1 : 0;
            mfma_quadrant(mh, nh);
            __builtin_amdgcn_sched_barrier(0);
            if (!(D2R_GEMM_LATE_DRAIN && P < 5) || u != 0) wait_vmcnt<10>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
        }
    }
    if (wm == 0) bar();
    STAMP(ts2);
    // every wave is past its last fragment read of this tile; the ring already holds (or is receiving)
    // the next tile's first K-tile pair
    const uint32_t em = m0 + wm * 128, en = n0 + wn * 64;
#if (D2R_GEMM_ABLATE & 4) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) asm volatile("" ::"v"(acc[i][j][r]));
#else
    // a freshly computed lane id (mbcnt) instead of the one derived from threadIdx at kernel entry: that
    // one would stay live across the K loop for the epilogue's sake, and at 250+ registers it gets spilled
    const uint32_t lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    gemm_epilogue<EPI, 4>(acc, ep, lane_e, em, en, bias, Cout, N, aux, ab_lds);
#endif
#ifdef D2R_GEMM_STAMPS
    {
        const unsigned long long ts3 = __builtin_readcyclecounter();
        if (tid == 0) {
            atomicAdd(&d2r_gemm_stamps[EPI][0], ts1 - ts0);
            atomicAdd(&d2r_gemm_stamps[EPI][1], ts2 - ts1);
            atomicAdd(&d2r_gemm_stamps[EPI][2], ts3 - ts2);
            atomicAdd(&d2r_gemm_stamps[EPI][3], 1ull);
        }
    }
#endif
    if (!has_next) break;
    t = t_next;
    m0 = m0n;
    n0 = n0n;
    }
    wait_vmcnt<0>();          // the last pair's (unused) requests must have landed before the LDS is given back
}

// ---- the persistent 256x256 kernel on fp8 operands ("fp8 blocks", below) ----
//
// Same tile walk, staging ring, phase schedule and epilogues as k_gemm8; what differs:
//   operands   a slot row is still 128 BYTES = now 128 elements of K.  A comes in planes [K / 64][M_pad][64 bytes] (chunks 0-3 of a slot
//              row from one plane, 4-7 from the next: what the producers' 64-column wave tiles write), W row-major [N][K bytes].
//   MFMA       v_mfma_scale_f32_32x32x64_f8f6f4 (16 passes, 64 elements of K: twice the bf16 instruction's work per cycle).  A lane's 32
//              operand bytes are two of its 16-byte chunks — for MFMA s2 of a K-tile the chunks (4 s2 + hi, 4 s2 + 2 + hi), for A and for W
//              alike, so the pairing of K elements is the same on both sides (a dot product does not care in which order) — i.e. the
//              fragment reads are EXACTLY k_gemm8's conflict-free 16-byte reads; MFMA s2 covers the row's 64-column group 2 kt + s2.
//   fragments  live in the FIXED registers v[160:255], named explicitly in inline asm: the instruction wants 8 consecutive registers per
//              operand, and with 128 accumulators + 96 fragment registers of a 256-register budget hipcc cannot place 8-register tuples
//              assembled from two 16-byte reads (it spilled 220-330 registers inside the counted-vmcnt loop; AGPRs are no way out — a
//              kernel that names one gets its budget split 128 + 128).  fa[h][mt][s4] = v[160 + 32 h + 16 mt + 4 s4 ..+3], fb[h][s4] =
//              v[224 + 16 h + 4 s4 ..+3].  EVERY asm statement of the K loop (reads, MFMAs, staging requests, scale loads) lists
//              v160-v255 as clobbered, so nothing of the compiler's that is live across or feeds any of them can sit there: it keeps the
//              accumulators, addresses and scales in v0-v159, and the epilogue has all 256 again.  tests/test_isa.py checks that no
//              compiler-generated instruction between the loop's first and last asm statement touches v160-v255, and that the loop spills nothing.
//   scales     one E8M0 byte per (row of A, 64-column group), applied by the MFMA (its scale operand is per lane = per row, op_sel picks the
//              byte): one dword per (A half, K-tile) and lane, bytes (m-tile, MFMA of the K-tile) — see q8_scale_off.  The two dwords of a
//              K-tile are re-requested (global_load_dword from inline asm, hand-counted like the staging requests) as soon as the K-tile's
//              last quadrant has used them — K-tile 1's at the top of phase 0, K-tile 0's (of the next pair / tile) at the top of phase 4 —
//              and are needed four phases later: every five-phase window of requests holds exactly one such batch, so every counted wait
//              is vmcnt(12) where k_gemm8 has 10.  W's scale operand is 1; the matrix's scale is applied by the epilogue.
#define F8_CL_FRAG "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define F8_READ_A(name, r)                                                                                                                   \
    template <int OFF> __device__ __forceinline__ void name(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3)                             \
    {                                                                                                                                        \
        asm volatile("ds_read_b128 v[" #r "+0:" #r "+3], %0 offset:%4\n\tds_read_b128 v[" #r "+4:" #r "+7], %1 offset:%4\n\t"               \
                     "ds_read_b128 v[" #r "+8:" #r "+11], %2 offset:%4\n\tds_read_b128 v[" #r "+12:" #r "+15], %3 offset:%4\n\t"            \
                     "ds_read_b128 v[" #r "+16:" #r "+19], %0 offset:%5\n\tds_read_b128 v[" #r "+20:" #r "+23], %1 offset:%5\n\t"           \
                     "ds_read_b128 v[" #r "+24:" #r "+27], %2 offset:%5\n\tds_read_b128 v[" #r "+28:" #r "+31], %3 offset:%5"               \
                     : : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(OFF), "n"(OFF + 4096) : "memory", F8_CL_FRAG);                             \
    }
#define F8_READ_B(name, r)                                                                                                                   \
    template <int OFF> __device__ __forceinline__ void name(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3)                             \
    {                                                                                                                                        \
        asm volatile("ds_read_b128 v[" #r "+0:" #r "+3], %0 offset:%4\n\tds_read_b128 v[" #r "+4:" #r "+7], %1 offset:%4\n\t"               \
                     "ds_read_b128 v[" #r "+8:" #r "+11], %2 offset:%4\n\tds_read_b128 v[" #r "+12:" #r "+15], %3 offset:%4"                \
                     : : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(OFF) : "memory", F8_CL_FRAG);                                              \
    }
F8_READ_A(f8_read_a0, 160)
F8_READ_A(f8_read_a1, 192)
F8_READ_B(f8_read_b0, 224)
F8_READ_B(f8_read_b1, 240)
// the four MFMAs of quadrant (mh, nh): m-tiles 0 / 1 alternate (two independent chains), op_sel = 2 mt + s2 picks the scale byte
#define F8_MFMA_Q(name, ra, rb)                                                                                                              \
    __device__ __forceinline__ void name(f32x16 &c0, f32x16 &c1, int sa, int sb)                                                            \
    {                                                                                                                                        \
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, v[" #ra "+0:" #ra "+7], v[" #rb "+0:" #rb "+7], %0, %2, %3 op_sel_hi:[0,0,0]\n\t"  \
                     "v_mfma_scale_f32_32x32x64_f8f6f4 %1, v[" #ra "+16:" #ra "+23], v[" #rb "+0:" #rb "+7], %1, %2, %3 op_sel_hi:[1,0,0]\n\t" \
                     "v_mfma_scale_f32_32x32x64_f8f6f4 %0, v[" #ra "+8:" #ra "+15], v[" #rb "+8:" #rb "+15], %0, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,0]\n\t" \
                     "v_mfma_scale_f32_32x32x64_f8f6f4 %1, v[" #ra "+24:" #ra "+31], v[" #rb "+8:" #rb "+15], %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" \
                     : "+v"(c0), "+v"(c1) : "v"(sa), "v"(sb) : F8_CL_FRAG);                                                                  \
    }
F8_MFMA_Q(f8_mfma_00, 160, 224)
F8_MFMA_Q(f8_mfma_01, 160, 240)
F8_MFMA_Q(f8_mfma_10, 192, 224)
F8_MFMA_Q(f8_mfma_11, 192, 240)

template <int EPI>
__global__ __launch_bounds__(512, 2) void k_gemm8f(const uint8_t *__restrict__ A, const uint8_t *__restrict__ W, const float *__restrict__ bias,
                                                  void *__restrict__ Cout, uint32_t M_pad, uint32_t N, uint32_t K, uint32_t n_xcd, EpiAux aux,
                                                  uint32_t gn_split)
{
    static_assert(EPI_IS_F8(EPI), "fp8 epilogues only");
    constexpr uint32_t SLOT = 128 * 128;             // 16 KiB: 128 rows x 128 bytes
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // (tile order: see k_gemm8)
    const uint32_t n_split = max(1u, gn_split >> 16), gn = gn_split & 0xffffu;
    const uint32_t tiles_n = N / 256 / n_split, mp_all = M_pad / 256;
    const uint32_t xcd = blockIdx.x % n_xcd, loc = blockIdx.x / n_xcd, per_xcd = gridDim.x / n_xcd;
    const uint32_t csec = xcd % n_split, xr = xcd / n_split, n_xr = n_xcd / n_split;
    const uint32_t mp_lo = (uint32_t)((uint64_t)mp_all * xr / n_xr), mp_cnt = (uint32_t)((uint64_t)mp_all * (xr + 1) / n_xr) - mp_lo;
    const uint32_t t_end = mp_cnt * tiles_n;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wm = wave >> 2, wn = wave & 3;

    f32x16 acc[4][2];

    // Per-lane state of the K loop (staging offsets, fragment addresses, scale offset): set up afresh for every tile from a lane id the
    // optimiser cannot see through, so that none of it is live across the epilogue — which needs all 128 registers for the accumulators
    // and more, and would otherwise push these values to scratch and reload them INSIDE the loop (scratch loads share vmcnt with the
    // hand-counted requests).
    // staging: a piece = 8 slot rows x 128 bytes per wave instruction (two pieces, 64 slot rows apart, per wave and half-tile)
    const uint32_t a_plane = M_pad * 64u;                                     // bytes per plane of A (4 planes < 2^32: checked on the host)
    uint32_t voffA = 0, voffB = 0, aoff[4] = {0, 0, 0, 0};
    auto lane_setup = [&]() {
        uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane));
        const uint32_t li = lane & 31, hi = lane >> 5, r_in = lane >> 3, pc = lane & 7;
        const uint32_t sr = wave * 8 + r_in;
        const uint32_t chunk = pc ^ ((sr >> 1) & 7u);
        voffA = sr * 64u + (chunk >> 2) * a_plane + (chunk & 3u) * 16u;
        voffB = ((sr >> 5) * 64 + (sr & 31)) * K + chunk * 16u;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) aoff[s4] = (wm * 64 + li) * 128 + (((2 * s4 + hi) ^ ((li >> 1) & 7u)) << 4);
    };
    lane_setup();
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const uint32_t nk = K / 128u;
    uint32_t m0 = 0, n0 = 0;
    const uint8_t *src[8];
    auto kind_is_a = [](int kind) { return kind == 0 || kind == 3 || kind == 4 || kind == 7; };
    auto seat = [&](int kind, uint32_t tm, uint32_t tn) -> const uint8_t * {
        const uint32_t h = (kind == 2 || kind == 3 || kind == 5 || kind == 7) ? 1u : 0u;
        return kind_is_a(kind) ? A + (size_t)(tm + h * 64) * 64u + (kind >= 4 ? 2 * (size_t)a_plane : (size_t)0)
                               : W + (size_t)(tn + h * 32) * K + (kind >= 4 ? (size_t)128 : (size_t)0);
    };
    auto stage = [&](int kind) {
        const bool isA = kind_is_a(kind);
        const uint8_t *base = src[kind];
        src[kind] = base + (isA ? 4 * (size_t)a_plane : (size_t)256);        // two K-tiles on
        if (D2R_F8_EXP & 32) return;
#pragma unroll
        for (int qq = 0; qq < 2; qq++)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         : : "v"(isA ? voffA : voffB), "s"(base + (size_t)qq * 128 * (isA ? (size_t)64 : (size_t)K)), "s"(lds0 + kind * SLOT + (wave + qq * 8) * 1024)
                         : "memory", F8_CL_FRAG);
    };
    // scale dwords [K-tile parity][A half]; "+v": a loop-carried value that the asm rewrites keeps ONE register
    int sc[2][2] = {{0, 0}, {0, 0}};
    const uint32_t sc_kt = M_pad >> 1;                // dwords between K-tiles
    const uint32_t *sc_src[2] = {nullptr, nullptr};
    auto load_scales = [&](int kt) {
        // dword (64-row block 2 wm + mh, lane li) of a K-tile's array, mh: + 128 bytes; (wm * 64 + li) * 4 from a fragment address (the
        // swizzle term is < 128) rather than one more register held through the loop
        const uint32_t sc_voff = (aoff[0] >> 5) & ~3u;
        if (D2R_F8_EXP & 8) { sc_src[kt] += 2 * (size_t)sc_kt; return; }
        asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:128"
                     : "+v"(sc[kt][0]), "+v"(sc[kt][1]) : "v"(sc_voff), "s"(sc_src[kt]) : "memory", F8_CL_FRAG);
        sc_src[kt] += 2 * (size_t)sc_kt;
    };
    auto tile_origin = [&](uint32_t t, uint32_t &tm, uint32_t &tn) {
        const uint32_t per_group = mp_cnt * gn;
        const uint32_t g = t / per_group, r = t - g * per_group;
        const uint32_t gw = min(gn, tiles_n - g * gn);
        const uint32_t mi = r / gw;
        tm = (mp_lo + mi) * 256;
        tn = (csec * tiles_n + g * gn + (r - mi * gw)) * 256;
    };
    const uint32_t d_ab = (wn * 32 - wm * 64) * 128;          // B addresses = A addresses + this (wave-uniform, modulo 2^32)
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using std::integral_constant;
    // fragment reads of half-tile `kind` (slot kind: kinds 4-7 lie beyond the 64 KiB reach of the offset field)
    auto read_pos = [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool isA = KIND == 0 || KIND == 3 || KIND == 4 || KIND == 7;
        constexpr int h = (KIND == 2 || KIND == 3 || KIND == 5 || KIND == 7) ? 1 : 0;
        constexpr int OFF = (KIND & 3) * (int)SLOT;
        uint32_t add = (isA ? 0u : d_ab) + (KIND >= 4 ? 65536u : 0u);
        if (D2R_F8_EXP & 64) return;
        if (isA) asm volatile("" : "+v"(add));       // not loop-invariant for the optimiser: four temporaries, not sixteen address registers
        else asm volatile("" : "+s"(add));
        if constexpr (isA && h == 0) f8_read_a0<OFF>(aoff[0] + add, aoff[1] + add, aoff[2] + add, aoff[3] + add);
        else if constexpr (isA) f8_read_a1<OFF>(aoff[0] + add, aoff[1] + add, aoff[2] + add, aoff[3] + add);
        else if constexpr (h == 0) f8_read_b0<OFF>(aoff[0] + add, aoff[1] + add, aoff[2] + add, aoff[3] + add);
        else f8_read_b1<OFF>(aoff[0] + add, aoff[1] + add, aoff[2] + add, aoff[3] + add);
    };

    uint32_t t = loc;
    if (t >= t_end) return;
    tile_origin(t, m0, n0);
#pragma unroll
    for (int kind = 0; kind < 8; kind++) {
        src[kind] = seat(kind, m0, n0);
        stage(kind);
    }
    float *ep = (float *)(smem + 8 * SLOT) + wave * EP_WAVE_FLOATS;

    for (;;) {
    STAMP(ts0);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    if (t != loc) lane_setup();                       // (the first tile's was needed by the prologue's requests)
    int b_scale = 0x7f7f7f7f;                         // W-side scale operand: 1 (scale operands must be VGPRs)
    asm volatile("" : "+v"(b_scale));
    // the scale registers start afresh too (nothing of the K loop's is live across an epilogue): the first K-tile's are requested here
    // and drained with everything else just below
    sc[0][0] = sc[0][1] = sc[1][0] = sc[1][1] = 0;
    asm volatile("" : "+v"(sc[0][0]), "+v"(sc[0][1]), "+v"(sc[1][0]), "+v"(sc[1][1]));
    sc_src[0] = aux.a_scale + (m0 >> 1);              // (m0 / 64) blocks x 32 dwords
    sc_src[1] = sc_src[0] + sc_kt;
    load_scales(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the previous tile's epilogue stores, the staged first K-tile pair
    wait_vmcnt<0>();
    bar();
    read_pos(integral_constant<int, 0>{});
    read_pos(integral_constant<int, 1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    STAMP(ts1);
    if (wm == 1) bar();                               // the wave rows run half a phase apart (k_gemm8)
    if (wm == 0 && !(D2R_F8_VAR & 3)) __builtin_amdgcn_s_setprio(1);

    const uint32_t t_next = t + per_xcd;
    const bool has_next = t_next < t_end;
    uint32_t m0n = 0, n0n = 0;
    if (has_next) tile_origin(t_next, m0n, n0n);
    const uint32_t n_iter = nk / 2;
    for (uint32_t u = 0; u < n_iter; u++) {
        const bool last = u + 1 == n_iter;
        auto phase = [&](auto p_tag) {
            constexpr int P = decltype(p_tag)::value;
            // K-tile 1's scales of THIS pair at the top of phase 0, K-tile 0's of the NEXT pair at the top of phase 4: each into the
            // registers whose last reader was the phase before
            if constexpr (P == 0) load_scales(1);
            if constexpr (P == 4) {
                if (last) sc_src[0] = aux.a_scale;    // two requests all the same (the waits count them): valid, never used, and kept
                load_scales(0);                       // "live" until they have landed (below) so that nothing else is given their registers
            }
            if (D2R_F8_VAR & 4) {
                if (last) src[P] = seat(P, m0n, n0n);
                stage(P);
                if (!(last && P >= 6)) read_pos(integral_constant<int, (P + 2) & 7>{});
            } else {
            if (!(last && P >= 6)) read_pos(integral_constant<int, (P + 2) & 7>{});
            if (last) src[P] = seat(P, m0n, n0n);
            stage(P);
            }
            wait_vmcnt<12>();
            __builtin_amdgcn_sched_barrier(0);
            bar();
            constexpr int mh = (P == 2 || P == 3 || P == 6 || P == 7) ? 1 : 0;
            constexpr int nh = (P == 1 || P == 2 || P == 4 || P == 7) ? 1 : 0;
            constexpr int kt = P >= 4 ? 1 : 0;
            if (D2R_F8_VAR & 2) __builtin_amdgcn_s_setprio(1);
            if (D2R_F8_EXP & 16) {}
            else if constexpr (mh == 0 && nh == 0) f8_mfma_00(acc[0][0], acc[1][0], sc[kt][0], b_scale);
            else if constexpr (mh == 0) f8_mfma_01(acc[0][1], acc[1][1], sc[kt][0], b_scale);
            else if constexpr (nh == 0) f8_mfma_10(acc[2][0], acc[3][0], sc[kt][1], b_scale);
            else f8_mfma_11(acc[2][1], acc[3][1], sc[kt][1], b_scale);
            if (D2R_F8_VAR & 2) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<12>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
        };
        phase(integral_constant<int, 0>{});
        phase(integral_constant<int, 1>{});
        phase(integral_constant<int, 2>{});
        phase(integral_constant<int, 3>{});
        phase(integral_constant<int, 4>{});
        phase(integral_constant<int, 5>{});
        phase(integral_constant<int, 6>{});
        phase(integral_constant<int, 7>{});
    }
    if (wm == 0) bar();
    STAMP(ts2);
    __builtin_amdgcn_s_setprio(0);
    wait_vmcnt<8>();          // every scale request has landed (only the last four phases' staging requests are younger)
    asm volatile("" : : "v"(sc[0][0]), "v"(sc[0][1]), "v"(sc[1][0]), "v"(sc[1][1]));
    // the MFMAs above are inline asm: hipcc does not know that they wrote the accumulators and pads no MFMA -> VALU / LDS hazard
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const uint32_t em = m0 + wm * 128, en = n0 + wn * 64;
    uint32_t lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane_e));          // nothing of the epilogue's per-lane addressing is hoisted above the K loop (it has 128 registers)
    gemm_epilogue<EPI, 4>(acc, ep, lane_e, em, en, bias, Cout, N, aux, (const float2 *)nullptr);
#ifdef D2R_GEMM_STAMPS
    {
        const unsigned long long ts3 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) {
            atomicAdd(&d2r_gemm_stamps[EPI][0], ts1 - ts0);
            atomicAdd(&d2r_gemm_stamps[EPI][1], ts2 - ts1);
            atomicAdd(&d2r_gemm_stamps[EPI][2], ts3 - ts2);
            atomicAdd(&d2r_gemm_stamps[EPI][3], 1ull);
        }
    }
#endif
    if (!has_next) break;
    t = t_next;
    m0 = m0n;
    n0 = n0n;
    }
    wait_vmcnt<0>();          // the last pair's (unused) requests must have landed before the LDS is given back
}

// -------------------------------------------------------- embeddings + LN

// X[b*T + t] = pre_layrnorm( (t==0 ? class_embedding : patch_out[b*(T-1)+t-1]) + position[t] )
__global__ __launch_bounds__(256) void k_embed_ln(const float *__restrict__ patch_out, const float *__restrict__ cls,
                                                  const float *__restrict__ pos, const float *__restrict__ w,
                                                  const float *__restrict__ b, float *__restrict__ X, uint32_t rows,
                                                  uint32_t T, uint32_t d, uint16_t *__restrict__ Xb,
                                                  float2 *__restrict__ AB, uint16_t *__restrict__ Xlo, uint32_t M_pad, int lo8 = 0,
                                                  const uint32_t *__restrict__ list = nullptr, const uint32_t *__restrict__ list_n = nullptr)
{
    // X (fp32 residual stream) and Xb/AB (LayerNorm-folded path: bf16 operand copy of the row and the
    // (rstd, -rstd*mean) of the row for the first block's layer_norm1) are each optional.
    // list (layer-0 reuse): entry i = image * (T - 1) + patch of a token whose patch embedding is row i of patch_out; only
    // those *list_n tokens are written (the others hold the background's own rows already)
    const uint32_t lane = threadIdx.x & 63;
  // (list mode: a fixed grid walks the list — one workgroup per worst-case entry would be a dispatch per idle workgroup)
  for (uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);; item += gridDim.x * 4) {
    uint32_t row = item;
    const float *src;
    uint32_t t;
    if (list) {
        if (row >= *list_n) return;
        const uint32_t e = list[row], bi = e / (T - 1);
        t = 1 + (e - bi * (T - 1));
        src = patch_out + (size_t)row * d;
        row = bi * T + t;
    } else {
        if (row >= rows) return;
        const uint32_t bi = row / T;
        t = row % T;
        src = t == 0 ? cls : patch_out + ((size_t)bi * (T - 1) + (t - 1)) * d;
    }
    float4 v[4], o[4];
    for (int i = 0; i < 4; i++) {
        uint32_t c0 = (lane + 64 * i) * 4;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 < d) {
            float4 a = *(const float4 *)(src + c0), p = *(const float4 *)(pos + (size_t)t * d + c0);
            v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        }
    }
    // padded lanes hold zeros but must not bias the variance: handle via masked deviation
    float s = 0.f;
    for (int i = 0; i < 4; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int i = 0; i < 4; i++) {
        uint32_t c0 = (lane + 64 * i) * 4;
        if (c0 < d) {
            float a = v[i].x - mu, bb = v[i].y - mu, c = v[i].z - mu, e = v[i].w - mu;
            q += a * a + bb * bb + c * c + e * e;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    for (int i = 0; i < 4; i++) {
        uint32_t c0 = (lane + 64 * i) * 4;
        if (c0 < d) {
            float4 ww = *(const float4 *)(w + c0), bb = *(const float4 *)(b + c0);
            o[i].x = (v[i].x - mu) * rstd * ww.x + bb.x;
            o[i].y = (v[i].y - mu) * rstd * ww.y + bb.y;
            o[i].z = (v[i].z - mu) * rstd * ww.z + bb.z;
            o[i].w = (v[i].w - mu) * rstd * ww.w + bb.w;
            if (X) *(float4 *)(X + (size_t)row * d + c0) = o[i];
            if (Xb) {
                const uint2 hv = make_uint2(pack2(o[i].x, o[i].y), pack2(o[i].z, o[i].w));
                const size_t xo = ((size_t)(c0 >> 6) * M_pad + row) * 64 + (c0 & 63);       // tile-major [d/64][M_pad][64]
                *(uint2 *)(Xb + xo) = hv;
                if (Xlo && lo8) {              // EPI_RESID_STATS_SPLIT8's lo bytes: 4 columns of this row = one 4-byte store
                    const uint32_t q0 = (uint32_t)split8_round(o[i].x, hv.x << 16), q1 = (uint32_t)split8_round(o[i].y, hv.x & 0xffff0000u);
                    const uint32_t q2 = (uint32_t)split8_round(o[i].z, hv.y << 16), q3 = (uint32_t)split8_round(o[i].w, hv.y & 0xffff0000u);
                    *(uint32_t *)((uint8_t *)Xlo + lo8_off(M_pad, row, c0)) =
                        __builtin_amdgcn_perm(q1, q0, 0x0c0c0501u) | (__builtin_amdgcn_perm(q3, q2, 0x0c0c0501u) << 16);
                } else if (Xlo)
                    *(uint2 *)(Xlo + xo) = make_uint2(pack2(o[i].x - bf_lo(hv.x), o[i].y - bf_hi(hv.x)),
                                                                         pack2(o[i].z - bf_lo(hv.y), o[i].w - bf_hi(hv.y)));
            }
        } else {
            o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (AB) {
        float s2 = 0.f, q2 = 0.f;
        for (int i = 0; i < 4; i++) {
            s2 += (o[i].x + o[i].y) + (o[i].z + o[i].w);
            q2 += fmaf(o[i].x, o[i].x, o[i].y * o[i].y) + fmaf(o[i].z, o[i].z, o[i].w * o[i].w);
        }
        s2 = wave_sum(s2);
        q2 = wave_sum(q2);
        const float m2 = s2 / (float)d, r2 = 1.0f / sqrtf(fmaxf(q2 / (float)d - m2 * m2, 0.f) + 1e-5f);
        if (lane == 0) AB[row] = make_float2(r2, -r2 * m2);
    }
    if (!list) return;
  }
}

// per row: the d/64 partial (sum, sum of squares) pairs an EPI_RESID_STATS_* GEMM wrote ([d/64][rows]) -> (rstd, -rstd*mean)
__global__ void k_rowstats(const float2 *__restrict__ part, uint32_t np, uint32_t rows, float inv_d,
                           float2 *__restrict__ AB)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    float s = 0.f, q = 0.f;
    for (uint32_t i = 0; i < np; i++) {
        const float2 p = part[(size_t)i * rows + row];
        s += p.x;
        q += p.y;
    }
    const float mu = s * inv_d, r = 1.0f / sqrtf(fmaxf(q * inv_d - mu * mu, 0.f) + 1e-5f);
    AB[row] = make_float2(r, -r * mu);
}

// ---- fp8 blocks (option "vit_fp8": BASELINE.json configs[4] names an "fp8 MFMA ViT") ----
//
// The four Linear products of a transformer block on the MX-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the bf16 rate):
//   activations  OCP e4m3, one E8M0 scale byte per (row, group of 64 columns): 2^(byte - 127) is the smallest power of two with
//                amax / scale <= 448 (q8_scale_byte), the element = round-to-nearest-even e4m3 of x / scale.  Bytes in planes
//                [cols / 64][M_pad][64]; scale bytes [cols / 256][M_pad][4] (the consumer fetches one dword per row and K-tile pair)
//   weights      e4m3 of w / s_w, s_w = ONE power of two per matrix (a floating-point format loses nothing to a coarse scale while the
//                values stay in its normal range: 2^-6 .. 448, a factor 28 672 under the matrix's largest weight); row-major [N][K]
//   accumulation fp32; the epilogue multiplies by s_w, adds the bias and goes on as the bf16 kernels do (the residual stream, the
//                LayerNorm statistics, q / k / v and the attention arithmetic stay as they are)
// Producers: k_ln_q8 (LayerNorm -> operand of QKV and fc1), k_attention_s<NW, true> (-> operand of the out-projection),
// EPI_F8_BIAS_GELU_Q8 (fc1 -> operand of fc2).  The quantisation is a bit-level specification (oracle/clip_fp8.py restates it): the
// parity tests hold the HIP path to that restatement at the bf16 path's bar and REPORT what the specification costs against fp32.

// LayerNorm of the residual's bf16 hi copy -> e4m3 operand: y = (x * rstd - rstd * mean) * gamma + beta with the row's (rstd, -rstd * mean)
// from k_rowstats.  A work item = 64 rows of one 64-column plane; a thread takes 8 columns of rows r and r + 32, the 8 lanes of a row
// agree on its scale.  Persistent grid (one item per workgroup would be two million dispatches per call at configs[4]).
__global__ __launch_bounds__(256) void k_ln_q8(const uint16_t *__restrict__ Xhi, const float2 *__restrict__ AB, const float *__restrict__ gamma,
                                               const float *__restrict__ beta, uint32_t M_pad, uint32_t d, uint8_t *__restrict__ A8,
                                               uint8_t *__restrict__ SA)
{
    const uint32_t planes = d >> 6, items = planes * (M_pad >> 6);
    const uint32_t c8 = (threadIdx.x & 7) * 8, r = threadIdx.x >> 3;
    for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint32_t p = it % planes, b = it / planes;
        const float4 g0 = *(const float4 *)(gamma + p * 64 + c8), g1 = *(const float4 *)(gamma + p * 64 + c8 + 4);
        const float4 e0 = *(const float4 *)(beta + p * 64 + c8), e1 = *(const float4 *)(beta + p * 64 + c8 + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        uint4 h[2];
        float2 ab[2];
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const size_t row = (size_t)b * 64 + r + 32 * ii;
            h[ii] = *(const uint4 *)(Xhi + ((size_t)p * M_pad + row) * 64 + c8);
            ab[ii] = AB[row];
        }
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const size_t row = (size_t)b * 64 + r + 32 * ii;
            const uint32_t hw[4] = {h[ii].x, h[ii].y, h[ii].z, h[ii].w};
            float f[8];
            float am = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float x = (e & 1) ? bf_hi(hw[e >> 1]) : bf_lo(hw[e >> 1]);
                f[e] = fmaf(fmaf(x, ab[ii].x, ab[ii].y), gm[e], bt[e]);
                am = fmaxf(am, fabsf(f[e]));
            }
            const uint32_t sb = q8_scale_byte(row8_max(am));
            *(uint2 *)(A8 + (((size_t)p * M_pad + row) << 6) + c8) = q8_pack8(f, q8_inv_scale(sb));
            if ((threadIdx.x & 7) == 0) SA[q8_scale_off(M_pad, row, p)] = (uint8_t)sb;
        }
    }
}

// weights: largest magnitude of a bf16 matrix (as the bits of its fp32 value), then e4m3 of w / 2^(scale byte - 127)
__global__ __launch_bounds__(256) void k_amax_bf16(const uint16_t *__restrict__ W, size_t n, uint32_t *__restrict__ amax_bits)
{
    uint32_t m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = max(m, (uint32_t)(W[i] & 0x7fffu) << 16);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, m);
}
__global__ __launch_bounds__(256) void k_quant_w8(const uint16_t *__restrict__ W, size_t n, const uint32_t *__restrict__ amax_bits,
                                                  uint8_t *__restrict__ W8)
{
    const float inv = q8_inv_scale(q8_scale_byte(__uint_as_float(*amax_bits)));
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        const uint2 w = *(const uint2 *)(W + i);
        uint32_t q = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(w.x) * inv, bf_hi(w.x) * inv, 0, false);
        q = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(w.y) * inv, bf_hi(w.y) * inv, q, true);
        *(uint32_t *)(W8 + i) = q;
    }
}
// parity hook (d2r_debug_gemm_fp8): fp32 row-major [M][K] -> the activation format above; rows >= M are written as zeros
__global__ __launch_bounds__(256) void k_quant_rows_q8(const float *__restrict__ A, uint32_t M, uint32_t K, uint32_t M_pad, uint8_t *__restrict__ A8,
                                                       uint8_t *__restrict__ SA)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;           // (row, 8-column group)
    const uint32_t per_row = K >> 3;
    const size_t row = t / per_row;
    if (row >= M_pad) return;
    const uint32_t col = (uint32_t)(t - row * per_row) * 8;
    float f[8];
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        f[e] = row < M ? A[row * K + col + e] : 0.f;
        am = fmaxf(am, fabsf(f[e]));
    }
    const uint32_t sb = q8_scale_byte(row8_max(am));
    *(uint2 *)(A8 + (((size_t)(col >> 6) * M_pad + row) << 6) + (col & 63)) = q8_pack8(f, q8_inv_scale(sb));
    if ((col & 63) == 0) SA[q8_scale_off(M_pad, row, col >> 6)] = (uint8_t)sb;
}

// ---- layer-0 reuse of background tokens (d2r_render_score) ----
//
// A composited candidate differs from the background frame only inside the rectangle its rays were generated in
// (k_raygen_rect).  A patch whose resampling footprint does not meet that rectangle is the background's own patch, so its
// patch embedding, its pre-LayerNorm residual row and its layer-0 q / k / v rows are the background's own rows — computed
// once per background (d2r_clip_layer0_background) and broadcast, while the patch-embedding and QKV products of layer 0 run on
// the TOUCHED tokens only (about a tenth of them for a 60-pixel object at 640x360), compacted into a list whose length stays in
// device memory.  Exact: every kernel involved computes a row from that row alone (tests hold the logits bit-identical).

// one block per image: the touched patches of the image -> list (image * g2 + patch), length added to *cnt
__global__ __launch_bounds__(256) void k_touch_list(const int4 *__restrict__ rects, uint32_t w, ResampleTables R, uint32_t S, uint32_t P,
                                                    uint32_t *__restrict__ cnt, uint32_t *__restrict__ list)
{
    __shared__ uint32_t ids[1024], n_ids, base;
    const uint32_t img = blockIdx.x, g = S / P, g2 = g * g;
    if (threadIdx.x == 0) n_ids = 0;
    __syncthreads();
    const int4 rc = rects[img];
    const bool empty = rc.x > rc.z || rc.y > rc.w;
    // the rectangle in the rotated source (rot90 maps frame column x to row w-1-x, frame row y to column y)
    const int rlo = (int)w - 1 - rc.z, rhi = (int)w - 1 - rc.x, clo = rc.y, chi = rc.w;
    for (uint32_t p = threadIdx.x; p < g2 && !empty; p += blockDim.x) {
        const uint32_t pr = p / g, pcol = p - pr * g;
        int ys, ye, xs, xe;                                   // source rows / columns the patch's pixels are resampled from: [s, e)
        if (R.need_v) {
            ys = R.bounds_v[2 * (R.top + pr * P)];
            const int last = R.top + pr * P + P - 1;
            ye = R.bounds_v[2 * last] + R.bounds_v[2 * last + 1];
        } else { ys = R.top + pr * P; ye = ys + P; }
        if (R.need_h) {
            xs = R.bounds_h[2 * (R.left + pcol * P)];
            const int last = R.left + pcol * P + P - 1;
            xe = R.bounds_h[2 * last] + R.bounds_h[2 * last + 1];
        } else { xs = R.left + pcol * P; xe = xs + P; }
        if (!(rhi < ys || rlo >= ye || chi < xs || clo >= xe)) ids[atomicAdd(&n_ids, 1u)] = img * g2 + p;
    }
    __syncthreads();
    if (threadIdx.x == 0) base = n_ids ? atomicAdd(cnt, n_ids) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_ids; i += blockDim.x) list[base + i] = ids[i];
}

// rows list[i] of a row-major bf16 matrix [..][row_elems] -> rows i of `out` (one wave per row, 16-byte pieces)
__global__ __launch_bounds__(256) void k_gather_rows(const uint16_t *__restrict__ src, uint32_t row_elems, const uint32_t *__restrict__ list,
                                                     const uint32_t *__restrict__ cnt, uint16_t *__restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63, n = *cnt, per = row_elems / 8;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        const uint4 *s = (const uint4 *)(src + (size_t)list[i] * row_elems);
        uint4 *o = (uint4 *)(out + (size_t)i * row_elems);
        for (uint32_t c = lane; c < per; c += 64) o[c] = s[c];
    }
}

// every token row r = image * T + t <- the background's row t: bf16 hi (tile-major), lo bytes (lo8_off layout), (rstd, -rstd mean)
__global__ __launch_bounds__(256) void k_bcast_x0(const uint16_t *__restrict__ bg_hi, const uint8_t *__restrict__ bg_lo, const float2 *__restrict__ bg_ab,
                                                  uint32_t bg_rows, uint16_t *__restrict__ Xhi, uint8_t *__restrict__ Xlo, float2 *__restrict__ AB,
                                                  uint32_t rows, uint32_t M_pad, uint32_t T, uint32_t d)
{
    const uint32_t lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint32_t t = row % T;
    for (uint32_t c8 = lane; c8 < d / 8; c8 += 64) {            // 8 columns: 16 bytes of hi, 8 bytes of lo
        const uint32_t c = c8 * 8;
        *(uint4 *)(Xhi + ((size_t)(c >> 6) * M_pad + row) * 64 + (c & 63)) = *(const uint4 *)(bg_hi + ((size_t)(c >> 6) * bg_rows + t) * 64 + (c & 63));
        *(uint2 *)(Xlo + lo8_off(M_pad, row, c)) = *(const uint2 *)(bg_lo + lo8_off(bg_rows, t, c));
    }
    if (lane == 0) AB[row] = bg_ab[t];
}

// every token's layer-0 q / k / v rows <- the background's: planes [3 d / 64][M_pad][64] bf16 (one thread per 16 bytes)
__global__ __launch_bounds__(256) void k_bcast_qkv(const uint16_t *__restrict__ bg_qkv, uint32_t bg_rows, uint16_t *__restrict__ QKV, uint32_t rows,
                                                   uint32_t M_pad, uint32_t T, uint32_t planes)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // (plane, row, piece of 8)
    const size_t total = (size_t)planes * rows * 8;
    if (i >= total) return;
    const uint32_t piece = (uint32_t)(i & 7);
    const size_t pr = i >> 3;
    const uint32_t row = (uint32_t)(pr % rows), plane = (uint32_t)(pr / rows);
    *(uint4 *)(QKV + ((size_t)plane * M_pad + row) * 64 + piece * 8) = *(const uint4 *)(bg_qkv + ((size_t)plane * bg_rows + row % T) * 64 + piece * 8);
}

// the touched tokens' operand rows (tile-major hi) and LayerNorm pairs -> compact tile-major [d / 64][cap_pad][64], [cap_pad]
__global__ __launch_bounds__(256) void k_gather_operand_rows(const uint16_t *__restrict__ Xhi, const float2 *__restrict__ AB, uint32_t M_pad, uint32_t T,
                                                             const uint32_t *__restrict__ list, const uint32_t *__restrict__ cnt,
                                                             uint16_t *__restrict__ A2, float2 *__restrict__ AB2, uint32_t cap_pad, uint32_t d)
{
    const uint32_t lane = threadIdx.x & 63, n = *cnt;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        const uint32_t e = list[i], bi = e / (T - 1), row = bi * T + 1 + (e - bi * (T - 1));
        for (uint32_t c8 = lane; c8 < d / 8; c8 += 64) {
            const uint32_t c = c8 * 8;
            *(uint4 *)(A2 + ((size_t)(c >> 6) * cap_pad + i) * 64 + (c & 63)) = *(const uint4 *)(Xhi + ((size_t)(c >> 6) * M_pad + row) * 64 + (c & 63));
        }
        if (lane == 0) AB2[i] = AB[row];
    }
}

// compact q / k / v rows [planes][cap_pad][64] -> the touched tokens' rows of QKV [planes][M_pad][64]
__global__ __launch_bounds__(256) void k_scatter_qkv(const uint16_t *__restrict__ Q2, uint32_t cap_pad, const uint32_t *__restrict__ list,
                                                     const uint32_t *__restrict__ cnt, uint16_t *__restrict__ QKV, uint32_t M_pad, uint32_t T, uint32_t planes)
{
    const uint32_t lane = threadIdx.x & 63, n = *cnt;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        const uint32_t e = list[i], bi = e / (T - 1), row = bi * T + 1 + (e - bi * (T - 1));
        for (uint32_t j = lane; j < planes * 8; j += 64) {
            const uint32_t plane = j >> 3, piece = j & 7;
            *(uint4 *)(QKV + ((size_t)plane * M_pad + row) * 64 + piece * 8) = *(const uint4 *)(Q2 + ((size_t)plane * cap_pad + i) * 64 + piece * 8);
        }
    }
}

// LayerNorm folding of a Linear that follows a LayerNorm (weights prepared once at create):
//   Wf[n][k] = bf16(W[n][k] * gamma[k]);   cs[n] = sum_k float(Wf[n][k]);   bf[n] = b[n] + sum_k W[n][k] * beta[k]
// `scale` multiplies the whole output row (weight row before its bf16 rounding, bias): the attention scale of the q rows (ATTN_Q_SCALE)
__global__ void k_fold_ln_weight(const float *__restrict__ W, const float *__restrict__ gamma, const float *__restrict__ beta,
                                 const float *__restrict__ b, uint16_t *__restrict__ Wf, float *__restrict__ cs,
                                 float *__restrict__ bf, uint32_t N, uint32_t K, float scale = 1.0f)
{
    const uint32_t n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float s = 0.f, t = 0.f;
    for (uint32_t k = lane; k < K; k += 64) {
        const float w = W[(size_t)n * K + k];
        const uint16_t h = f2bf(w * gamma[k] * scale);
        Wf[(size_t)n * K + k] = h;
        s += __uint_as_float((uint32_t)h << 16);
        t = fmaf(w, beta[k], t);
    }
    s = wave_sum(s);
    t = wave_sum(t);
    if (lane == 0) {
        cs[n] = s;
        bf[n] = (b[n] + t) * scale;
    }
}

// Y(bf16) = LayerNorm(X fp32), one wave per row
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ X, const float *__restrict__ w,
                                                   const float *__restrict__ b, uint16_t *__restrict__ Y,
                                                   uint32_t rows, uint32_t d)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[4];
    for (int i = 0; i < 4; i++) {
        uint32_t c0 = (lane + 64 * i) * 4;
        v[i] = c0 < d ? *(const float4 *)(X + (size_t)row * d + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int i = 0; i < 4; i++) {
        uint32_t c0 = (lane + 64 * i) * 4;
        if (c0 < d) {
            float a = v[i].x - mu, bb = v[i].y - mu, c = v[i].z - mu, e = v[i].w - mu;
            q += a * a + bb * bb + c * c + e * e;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    for (int i = 0; i < 4; i++) {
        uint32_t c0 = (lane + 64 * i) * 4;
        if (c0 < d) {
            float4 ww = *(const float4 *)(w + c0), bb = *(const float4 *)(b + c0);
            uint2 o;
            o.x = pack2((v[i].x - mu) * rstd * ww.x + bb.x, (v[i].y - mu) * rstd * ww.y + bb.y);
            o.y = pack2((v[i].z - mu) * rstd * ww.z + bb.z, (v[i].w - mu) * rstd * ww.w + bb.w);
            *(uint2 *)(Y + (size_t)row * d + c0) = o;
        }
    }
}

// ------------------------------------------------------------- attention

// combine the two lanes (l, l ^ 32) that hold the halves of one query's scores: v_permlane32_swap exchanges the
// upper half of one register with the lower half of the other on the VALU (a __shfl_xor goes through the LDS)
__device__ __forceinline__ float half_max(float x)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// QKV bf16, tile-major [3 d / 64][M_pad][64] (plane s*H + h = section s (q, k, v) of head h: what the QKV GEMM's
// epilogue writes), out AO bf16 tile-major [d / 64][M_pad][64] (the out-projection's A operand): every (image, head)
// reads and writes contiguous T x 128-byte blocks.
// grid (heads, images); block 256; dynamic LDS: K [T_pad][64] swizzled + Vt [64][T_pad+4].
#define ATTN_THREADS 512
// One 32-query tile of one (image, head) against all keys staged in LDS (Ks: K rows, swizzled 16-byte chunks;
// Vt: V^T, [64][vstride]): S^T = K Q^T so that a query's scores are lane-local, online softmax in the exp2
// domain, P fed back as the MFMA B operand, O^T accumulated; writes the tile's rows of AO.
template <bool CAUSAL>
__device__ __forceinline__ void attn_qtile(const uint8_t *__restrict__ Ks, const uint16_t *__restrict__ Vt, uint32_t vstride,
                                           const uint4 (&qf)[4], uint32_t qt, uint32_t T, uint32_t n_kt, uint32_t li,
                                           uint32_t hi, uint16_t *__restrict__ AO, size_t row_base, uint32_t M_pad, uint32_t head)
{
    const float sm_c = 0.125f * 1.4426950408889634f;   // head_dim^-0.5 * log2(e), head_dim = 64
    const uint32_t qrow = qt * 32 + li;
    {
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; r++) o0[r] = o1[r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        // causal: key tiles beyond the last query row of this tile are fully masked
        const uint32_t kt_end = CAUSAL ? min(n_kt, qt + 1) : n_kt;
        // LDS addresses of this lane's fragments advance by a constant per key tile (32 K rows = 4096 B
        // with an unchanged swizzle term, 32 keys = 64 B along a V^T row): two running pointers and
        // immediate offsets instead of a dozen address computations per tile (the kernel is VALU-bound)
        const uint8_t *kp[4];
#pragma unroll
        for (int s = 0; s < 4; s++) kp[s] = Ks + lds_off(li, 2 * s + hi);
        const uint16_t *vp0 = Vt + (size_t)li * vstride + 4 * hi, *vp1 = Vt + (size_t)(32 + li) * vstride + 4 * hi;
        // K fragments are software-pipelined: the reads of tile kt+1 are issued right after the S MFMAs of tile
        // kt (into the registers those MFMAs have just consumed), so their LDS latency hides under the
        // softmax and the PV MFMAs instead of standing in front of every S chain
        uint4 ka[4];
#pragma unroll
        for (int s = 0; s < 4; s++) ka[s] = *(const uint4 *)(kp[s]);
        for (uint32_t kt = 0; kt < kt_end; kt++) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; r++) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                union { uint4 u; bf16x8 v; } a, b;
                a.u = ka[s];
                b.u = qf[s];
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, sacc, 0, 0, 0);
            }
            if (kt + 1 < kt_end) {
#pragma unroll
                for (int s = 0; s < 4; s++) ka[s] = *(const uint4 *)(kp[s] + (kt + 1) * 4096u);
            }
            // lane (q, hi), reg r  <->  key kt*32 + (r&3) + 8*(r>>2) + 4*hi.  Softmax in the exp2 domain:
            // p = exp2(s*c - m) with c = head_dim^-0.5 * log2(e) and m the running maximum of s*c.  Only
            // the tile that holds keys >= T (and, for the causal text tower, the diagonal tile) needs
            // the mask; the select keeps garbage in K rows >= T out of the arithmetic.
            const bool need_mask = (kt + 1) * 32 > T || (CAUSAL && kt == qt);      // wave-uniform
            if (need_mask) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    sacc[r] = (key < T && (!CAUSAL || key <= qrow)) ? sacc[r] : -INFINITY;
                }
            }
            float tmax = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; r++) tmax = fmaxf(tmax, sacc[r]);
            tmax = half_max(tmax) * sm_c;
            const float m_new = fmaxf(m_run, tmax);
            // rescale the running sums only when some query's maximum moved (exact: alpha would be 1)
            if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // m_run = -inf on the first tile -> 0
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
                m_run = m_new;
            }
            // s*c - m on packed pairs (v_pk_fma_f32), 16 exp2, pairwise tree sum (v_pk_add_f32)
            {
                typedef float f32x8 __attribute__((ext_vector_type(8)));
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x16 cv = sm_c, mv = -m_new;
                const f32x16 t = __builtin_elementwise_fma(sacc, cv, mv);
#pragma unroll
                for (int r = 0; r < 16; r++) sacc[r] = __builtin_amdgcn_exp2f(t[r]);
                const f32x8 s8 = sacc.lo + sacc.hi;
                const f32x4 s4 = s8.lo + s8.hi;
                const f32x2 s2 = s4.lo + s4.hi;
                l_run += s2.x + s2.y;
            }
            // O^T[d][q] += V^T[d][key] P^T[key][q]; P regs 8s..8s+7 are the B fragment of k-step s
            const uint16_t *vq0 = vp0 + kt * 32, *vq1 = vp1 + kt * 32;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                union { uint4 u; bf16x8 v; } pb, va0, va1;
                pb.u.x = pack2(sacc[8 * s + 0], sacc[8 * s + 1]);
                pb.u.y = pack2(sacc[8 * s + 2], sacc[8 * s + 3]);
                pb.u.z = pack2(sacc[8 * s + 4], sacc[8 * s + 5]);
                pb.u.w = pack2(sacc[8 * s + 6], sacc[8 * s + 7]);
                // A slot (hi, j) <-> key kt*32 + 16s + 8(j>>2) + 4hi + (j&3)
                const uint2 a00 = *(const uint2 *)(vq0 + 16 * s);
                const uint2 a01 = *(const uint2 *)(vq0 + 16 * s + 8);
                const uint2 a10 = *(const uint2 *)(vq1 + 16 * s);
                const uint2 a11 = *(const uint2 *)(vq1 + 16 * s + 8);
                va0.u = make_uint4(a00.x, a00.y, a01.x, a01.y);
                va1.u = make_uint4(a10.x, a10.y, a11.x, a11.y);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0.v, pb.v, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1.v, pb.v, o1, 0, 0, 0);
            }
        }
        const float l_tot = half_sum(l_run);
        const float inv_l = 1.0f / l_tot;
        if (qrow < T) {
            uint16_t *dst = AO + ((size_t)head * M_pad + row_base + qrow) * 64;       // tile-major: head h is plane h of [d/64][M_pad][64]
            // lane (q, hi), reg r of o{0,1} <-> dim 32*{0,1} + (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                uint2 w0, w1;
                w0.x = pack2(o0[4 * g4 + 0] * inv_l, o0[4 * g4 + 1] * inv_l);
                w0.y = pack2(o0[4 * g4 + 2] * inv_l, o0[4 * g4 + 3] * inv_l);
                w1.x = pack2(o1[4 * g4 + 0] * inv_l, o1[4 * g4 + 1] * inv_l);
                w1.y = pack2(o1[4 * g4 + 2] * inv_l, o1[4 * g4 + 3] * inv_l);
                *(uint2 *)(dst + 8 * g4 + 4 * hi) = w0;
                *(uint2 *)(dst + 32 + 8 * g4 + 4 * hi) = w1;
            }
        }
    }
}

// The resident form: one workgroup per (sequence, head) stages the whole K (swizzled rows) and V^T of the head in LDS,
// then its waves take the 32-query tiles.  Serves the text tower (causal, <= 77 tokens, a handful of sequences per
// task); the vision tower streams K / V instead (k_attention_s below).
template <bool CAUSAL>
__global__ __launch_bounds__(ATTN_THREADS, 4) void k_attention(const uint16_t *__restrict__ QKV, uint16_t *__restrict__ AO, uint32_t T,
                                                              uint32_t T_pad, uint32_t d, uint32_t M_pad)
{
    constexpr uint32_t TH = ATTN_THREADS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *Ks = smem;                                  // T_pad * 128 B
    uint16_t *Vt = (uint16_t *)(smem + (size_t)T_pad * 128);
    const uint32_t vstride = T_pad + 4;
    const uint32_t head = blockIdx.x, img = blockIdx.y;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
    const size_t row_base = (size_t)img * T;
    const uint32_t ld = 64, H = d >> 6;
    const uint16_t *Qg = QKV + ((size_t)head * M_pad + row_base) * 64;
    const uint16_t *Kg = QKV + ((size_t)(H + head) * M_pad + row_base) * 64;
    const uint16_t *Vg = QKV + ((size_t)(2 * H + head) * M_pad + row_base) * 64;

    // this wave's first query tile is requested before the staging so that its latency hides under it
    // (B operand of S^T = K Q^T: lane (q, hi) holds Q[q][16s + 8hi .. +8))
    const uint32_t n_qt = (T + 31) / 32, n_kt = T_pad / 32;
    uint4 qf[4];
    {
        const uint32_t qrow0 = wave * 32 + li;
#pragma unroll
        for (int s = 0; s < 4; s++)
            qf[s] = qrow0 < T ? *(const uint4 *)(Qg + (size_t)qrow0 * ld + 16 * s + 8 * hi) : make_uint4(0, 0, 0, 0);
    }
    // K (row-major, swizzled 16-byte chunks) goes straight to LDS by LDS-DMA, 8 key rows per wave
    // instruction, issued before anything else so it overlaps the V transposes.  Rows >= T repeat row
    // T-1: their scores are masked by a select below, so any finite value does.
    {
        const uint32_t ks0 = __builtin_amdgcn_readfirstlane(lds_addr(Ks));
        for (uint32_t b = __builtin_amdgcn_readfirstlane(wave); b < T_pad / 8; b += TH / 64) {
            const uint32_t row = b * 8 + (lane >> 3), src_row = row < T ? row : T - 1;
            glds16(Kg + (size_t)src_row * ld + ((lane & 7) ^ ((row >> 1) & 7u)) * 8, ks0 + b * 1024);
        }
    }
    // stage V^T: a task takes 4 keys x 8 dims (four 16-byte loads), transposes the 4x8 block in
    // registers (v_perm_b32) and writes one 8-byte word of 4 consecutive keys per dim
    for (uint32_t i = tid; i < (T_pad / 4) * 8; i += TH) {
        const uint32_t kb = (i >> 3) * 4, c = i & 7;
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            v[k] = *(const uint4 *)(Vg + (size_t)(kb + k < T ? kb + k : T - 1) * ld + c * 8);   // keys >= T: finite filler, their P is 0
        const uint32_t *w0 = (const uint32_t *)&v[0], *w1 = (const uint32_t *)&v[1];
        const uint32_t *w2 = (const uint32_t *)&v[2], *w3 = (const uint32_t *)&v[3];
#pragma unroll
        for (int p = 0; p < 4; p++) {      // dims 2p, 2p+1 live in word p of every key's chunk
            uint2 lo, hi2;
            lo.x = __builtin_amdgcn_perm(w1[p], w0[p], 0x05040100);   // (key0.lo16, key1.lo16)
            lo.y = __builtin_amdgcn_perm(w3[p], w2[p], 0x05040100);
            hi2.x = __builtin_amdgcn_perm(w1[p], w0[p], 0x07060302);  // (key0.hi16, key1.hi16)
            hi2.y = __builtin_amdgcn_perm(w3[p], w2[p], 0x07060302);
            *(uint2 *)(Vt + (size_t)(c * 8 + 2 * p) * vstride + kb) = lo;
            *(uint2 *)(Vt + (size_t)(c * 8 + 2 * p + 1) * vstride + kb) = hi2;
        }
    }
    wait_vmcnt<0>();              // this wave's K copies have landed
    __syncthreads();
    for (uint32_t qt = wave; qt < n_qt; qt += TH / 64) {
        if (qt != wave) {
            const uint32_t qrow = qt * 32 + li;
#pragma unroll
            for (int s = 0; s < 4; s++)
                qf[s] = qrow < T ? *(const uint4 *)(Qg + (size_t)qrow * ld + 16 * s + 8 * hi) : make_uint4(0, 0, 0, 0);
        }
        attn_qtile<CAUSAL>(Ks, Vt, vstride, qf, qt, T, n_kt, li, hi, AO, row_base, M_pad, head);
    }
}

// ---- streamed attention (vision tower) ----
//
// One workgroup = up to eight 32-query tiles (one per wave) of one (image, head).  K and V are NOT held whole: they
// stream through a ring of ATS_STAGES slots of 32 keys (K 4 KiB + V 4 KiB each), filled by LDS-DMA three key tiles
// ahead of the arithmetic — every wave issues ONE 1-KiB request per key tile (waves 0-3 the K rows, 4-7 the V rows),
// waits for its own request of the tile about to be used with a counted vmcnt, and one barrier per key tile both
// publishes the tile and retires the slot read in the previous iteration.  40 KiB of LDS whatever the sequence length
// (the resident kernel above needs 57 KiB at 197 tokens and 156 KiB at 577, i.e. one workgroup per CU there), the
// first MFMA starts when the first 8 KiB have landed instead of after the whole head, and loads and arithmetic of one
// workgroup overlap instead of alternating.
//   K slot: rows of 128 B, 16-byte chunks XOR-swizzled with (row >> 1) & 7 (lds_off) for conflict-free ds_read_b128.
//   V slot: ROW-major as it lies in HBM (no register transposes, no V^T image); the A operand of O^T = V^T P^T is read
//           with ds_read_b64_tr_b16 (a 16-lane group reads a [4 keys][16 dims] block, lane i receives dim i of the four
//           keys).  Chunk c of key row r sits at chunk c ^ (((r >> 1) & 1) << 2), so the four rows a group reads cover
//           all 64 banks once.
// Softmax: exp2 domain, running maximum with a deferred rescale — the accumulators are rescaled only when some
// query's tile maximum exceeds its running maximum by more than ATS_DEFER (p <= 2^ATS_DEFER in between; bf16 keeps
// fp32's exponent, so the relative rounding of P is unchanged).  The last key tile of a CLIP sequence holds T mod 32
// = 5 (197 tokens) or 1 (257, 577) valid keys: when that is <= 8 only accumulator registers 0..3 (keys 0..7 of the
// tile) go through the softmax and only the first of the two PV k-steps runs.
#define ATTN_Q_SCALE (0.125f * 1.4426950408889634f)     /* head_dim^-0.5 * log2(e), head_dim = 64: folded into the vision tower's W_q, b_q at create */
#define ATS_THREADS 512
#define ATS_STAGES 5
#define ATS_SLOT 8192u
#define ATS_DEFER 8.0f
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const uint8_t *lds_cptr;       // pointer arithmetic on it folds constants into the DS offset field
__device__ __forceinline__ uint2 lds_read_tr16(lds_cptr p)
{
    union { s16x4 s; uint2 u; } c;
    c.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
    return c.u;
}

// softmax of one key tile's scores.  The scores arrive SHIFTED and in the exp2 domain: sacc = s c - m_run, because (round 6)
//   * c = head_dim^-0.5 * log2(e) is folded into W_q, b_q once at d2r_clip_create (the product is rounded to bf16 once either way), and
//   * the S^T accumulator starts at -m_run instead of 0, so the MFMA's own additions do the subtraction:
// no multiply and no scale-subtract between the MFMAs and the exponentials (ablation D2R_ATTN_ABLATE 1024 of round 5 priced it:
// -3.7 % of the kernel, profiles/r06_attn_price.txt).  Running maximum with deferred rescale as before: only when some query's tile
// maximum exceeds its running maximum by more than ATS_DEFER are the accumulators rescaled and the tile's scores shifted again
// (p <= 2^ATS_DEFER in between).  P = exp2(sacc) packed to bf16 as the B fragments of the two PV k-steps (P regs 8s..8s+7 -> pb[s]).
// NR = 16: all of the tile's keys; NR = 4: the tile's first 8 keys only (registers 0..3; the other keys' P is 0).
template <int NR>
__device__ __forceinline__ void ats_softmax(f32x16 &sacc, f32x16 &o0, f32x16 &o1, float &m_run, float &l_run, uint4 (&pb)[2])
{
#if D2R_ATTN_ABLATE & 32
    l_run += sacc[0];
    o0[0] += sacc[1];
#else
    float tmax;
    // v_max3 on the raw accumulator registers and a plain v_max across the lane halves: MFMA results are never signalling NaNs, the
    // canonicalising self-maxes hipcc puts in front of fmaxf (four per key tile) buy nothing
    if constexpr (NR == 16) {
        float a, b;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(sacc[0]), "v"(sacc[1]), "v"(sacc[2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(b) : "v"(sacc[3]), "v"(sacc[4]), "v"(sacc[5]));
        asm("v_max3_f32 %0, %1, %2, %0" : "+v"(a) : "v"(sacc[6]), "v"(sacc[7]));
        asm("v_max3_f32 %0, %1, %2, %0" : "+v"(b) : "v"(sacc[8]), "v"(sacc[9]));
        asm("v_max3_f32 %0, %1, %2, %0" : "+v"(a) : "v"(sacc[10]), "v"(sacc[11]));
        asm("v_max3_f32 %0, %1, %2, %0" : "+v"(b) : "v"(sacc[12]), "v"(sacc[13]));
        asm("v_max3_f32 %0, %1, %2, %0" : "+v"(a) : "v"(sacc[14]), "v"(sacc[15]));
        asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(a), "v"(b));
    } else {
        float a;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(sacc[0]), "v"(sacc[1]), "v"(sacc[2]));
        asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(a), "v"(sacc[3]));
    }
    {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(r[0]), "v"(r[1]));
    }
    if (__builtin_amdgcn_ballot_w64(tmax > ATS_DEFER) != 0) {
        const float dm = fmaxf(tmax, 0.f);                                   // m_new - m_run
        const float alpha = __builtin_amdgcn_exp2f(-dm);
        l_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < NR; r++) sacc[r] -= dm;
        m_run += dm;
    }
    if constexpr (NR == 16) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        // the exponentials land in aligned register PAIRS (the empty asm ties each pair to a 64-bit register) so that the row sum is seven
        // v_pk_add_f32 + two adds instead of thirteen
        f32x2 pr[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            pr[q] = f32x2{__builtin_amdgcn_exp2f(sacc[2 * q]), __builtin_amdgcn_exp2f(sacc[2 * q + 1])};
            asm volatile("" : "+v"(pr[q]));
            sacc[2 * q] = pr[q].x;
            sacc[2 * q + 1] = pr[q].y;
        }
        const f32x2 a0 = pr[0] + pr[4], a1 = pr[1] + pr[5], a2 = pr[2] + pr[6], a3 = pr[3] + pr[7];
        const f32x2 b0 = a0 + a2, b1 = a1 + a3;
        const f32x2 s2 = b0 + b1;
        l_run += s2.x + s2.y;
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) sacc[r] = __builtin_amdgcn_exp2f(sacc[r]);
        l_run += (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
    }
#endif
    pb[0].x = pack2(sacc[0], sacc[1]);
    pb[0].y = pack2(sacc[2], sacc[3]);
    if constexpr (NR == 16) {
        pb[0].z = pack2(sacc[4], sacc[5]);
        pb[0].w = pack2(sacc[6], sacc[7]);
        pb[1] = make_uint4(pack2(sacc[8], sacc[9]), pack2(sacc[10], sacc[11]), pack2(sacc[12], sacc[13]), pack2(sacc[14], sacc[15]));
    } else {
        pb[0].z = pb[0].w = 0u;
        pb[1] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// grid (heads, images, groups of NW query tiles starting at tile qt0); block NW * 64; dynamic LDS ATS_STAGES * ATS_SLOT.
// NW = 8: the main launch (full groups of eight query tiles).  NW = 1 / 2 / 4: the LEFTOVER launch for sequences whose tile
// count is 8g + r with a small r (257 tokens: r = 1; 577: r = 3) — the same per-wave arithmetic in a workgroup with as many
// waves as it has query tiles, every wave staging 8 / NW slices of each key tile, so that a CU holds four such workgroups
// (LDS-limited) with every wave computing, instead of two eight-wave workgroups with one active wave each.
// Q8: the output as the fp8 blocks' e4m3 operand (planes of 64 BYTES per row, AOS = the scale bytes; see "fp8 blocks")
template <int NW, bool Q8 = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) void k_attention_s(const uint16_t *__restrict__ QKV, uint16_t *__restrict__ AO, uint32_t T,
                                                                          uint32_t d, uint32_t M_pad, uint32_t qt0, uint8_t *__restrict__ AOS = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NS = 8 / NW;                            // key-tile slices (1 KiB each: eight K or eight V rows) a wave stages
    const uint32_t head = blockIdx.x, img = blockIdx.y;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, hi = lane >> 5;
    const size_t row_base = (size_t)img * T;
    const uint32_t H = d >> 6;
    const uint32_t n_kt = (T + 31) / 32;
    // (wave i of a workgroup does NOT always sit on SIMD i mod 4 — the dispatcher rotates the first SIMD from workgroup to workgroup
    // (tools/probes/wave_placement_probe.hip), so the idle wave and the 5-query tail wave of a 197-token sequence are spread over the
    // four SIMDs already; rotating the wave -> query-tile map per workgroup measured neutral, profiles/r06_ab_attention.md)
    const uint32_t qt = qt0 + blockIdx.z * NW + wave;
#if (D2R_ATTN_ABLATE & 512)
    const bool active = qt < n_kt && (qt + 1) * 32 <= T;  // ablation: the partial last query tile (5 of 32 queries at 197 tokens) computes nothing — the upper bound of what a cheaper tail tile can return
#else
    const bool active = __builtin_amdgcn_readfirstlane(qt < n_kt ? 1u : 0u) != 0u;     // query tiles = key tiles = ceil(T / 32); a SCALAR condition (hipcc kept it in a lane mask and rebuilt it with two VALU per key tile)
#endif
    const uint32_t qrow = qt * 32 + li;
    const uint16_t *Qg = QKV + ((size_t)head * M_pad + row_base) * 64;
    // slice c of a key tile (c = 0..3: eight K rows each, 4..7: eight V rows each) is staged by wave c % NW; a lane fetches
    // the 16-byte chunk whose swizzled position in the slot is (its row, lane & 7)
    const uint16_t *src_k = QKV + ((size_t)(H + head) * M_pad + row_base) * 64, *src_v = QKV + ((size_t)(2 * H + head) * M_pad + row_base) * 64;
    const uint32_t smem0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // ring positions as running byte offsets (scalar add + wrap; a modulo by five per use costs a multiply-high and a
    // vector add per LDS address): slot_req = where the next request lands, slot_k / slot_v = the slots the K
    // fragments are read from next and this tile's V is read from
    uint32_t slot_req = 0, slot_k = 0, slot_v = 0;
    auto advance = [](uint32_t &slot) { slot = slot + ATS_SLOT == ATS_STAGES * ATS_SLOT ? 0u : slot + ATS_SLOT; };
    uint32_t src_off[NS], dst_off[NS], r_loc[NS];         // per slice: element offset inside a row (the swizzled chunk), LDS offset, row in the tile
    bool is_v[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const uint32_t c = wave + (uint32_t)j * NW;
        r_loc[j] = (c & 3) * 8 + (lane >> 3);
        is_v[j] = c >= 4;
        src_off[j] = ((lane & 7) ^ (is_v[j] ? ((r_loc[j] >> 1) & 1u) << 2 : (r_loc[j] >> 1) & 7u)) * 8;
        dst_off[j] = (is_v[j] ? 4096u : 0u) + (c & 3) * 1024u;
    }
    auto request = [&](uint32_t kt) {                    // key tile kt -> the next ring slot (rows >= T repeat row T-1: masked / P = 0)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const uint32_t row = kt * 32 + r_loc[j];
            if (!(D2R_ATTN_ABLATE & 16)) glds16((is_v[j] ? src_v : src_k) + (size_t)(row < T ? row : T - 1) * 64 + src_off[j], smem0 + slot_req + dst_off[j]);
        }
        advance(slot_req);
    };
    // Q fragments (B operand of S^T = K Q^T: lane (q, hi) holds Q[q][16s + 8hi .. +8)) first: oldest in the vmcnt order.
    // Loaded from inline asm and awaited by hand: a load hipcc can see makes it put `s_waitcnt vmcnt(0)` in front of the
    // first use INSIDE the tile loop, which would drain the request ring every iteration.  Query rows >= T read row
    // T-1 (finite; they are never stored).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 qv[4];
    {
        // every wave loads (waves without a query tile read row T-1 too) and nothing branches between the loads and
        // their wait: a conditional here makes hipcc merge the destination registers with copies placed BEFORE the wait
        const uint32_t qr = active && qrow < T ? qrow : T - 1;
        const uint16_t *qp = Qg + (size_t)qr * 64 + 8 * hi;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                     "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                     : "=&v"(qv[0]), "=&v"(qv[1]), "=&v"(qv[2]), "=&v"(qv[3]) : "v"(qp) : "memory");
    }
    // ATS_STAGES - 1 requests whatever the sequence length (tiles >= n_kt re-read row T-1 into slots nobody reads), so
    // that the waits below are the same straight-line code for every launch
    static_assert(ATS_STAGES == 5, "the waits below are written for a five-slot ring");
#pragma unroll
    for (uint32_t c = 0; c < ATS_STAGES - 1; c++) request(c);
    // the Q loads are older than the four requests: done when only those remain outstanding; then tile 0 (the oldest
    // request), published by the first barrier
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3]) : "n"(4 * NS) : "memory");
    wait_vmcnt<3 * NS>();
    __syncthreads();
    uint4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; s++) qf[s] = make_uint4(qv[s][0], qv[s][1], qv[s][2], qv[s][3]);

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; r++) o0[r] = o1[r] = 0.f;
    float m_run = 0.f, l_run = 0.f;                       // m_run: set from the first key tile below
    // per-lane LDS offsets inside a slot.  K: row li, chunks 2s + hi.  V (transposing read): a 16-lane group g = li >> 4
    // with lane-in-group i reads key row 4hi + (i >> 2) (+ 16s + 8j per read), dims 32t + 16g + 4(i & 3)
    uint32_t koff[4];
#pragma unroll
    for (int s = 0; s < 4; s++) koff[s] = lds_off(li, 2 * s + hi);
    const uint32_t vi = li & 15, vg = li >> 4, vrow = 4 * hi + (vi >> 2), vsw = (vrow >> 1) & 1u;
    const uint32_t vlc = 2 * vg + ((vi & 3) >> 1);                               // logical 16-byte chunk of O tile 0 (tile 1: + 4)
    const uint32_t voff0 = 4096u + vrow * 128u + ((vlc ^ (vsw << 2)) << 4) + (vi & 1) * 8u;
    const uint32_t voff1 = 4096u + vrow * 128u + (((vlc + 4) ^ (vsw << 2)) << 4) + (vi & 1) * 8u;
    const uint32_t n_last = T - (n_kt - 1) * 32;                                 // valid keys of the last tile (1..32)

    // K fragments of the tile about to be used: read one tile ahead, under the previous tile's softmax
    uint4 ka[4];
    auto read_k = [&]() {                                 // the tile in slot_k, then on to the next slot
        // (every wave reads, a wave without a query tile too: the condition cost two VALU and a branch per key tile, the reads of one idle wave in eight cost nothing that is short)
#pragma unroll
        for (int s = 0; s < 4; s++) ka[s] = *(const uint4 *)(lds_cptr)(size_t)(smem0 + slot_k + koff[s]);
        advance(slot_k);
    };
    read_k();
    // S^T = K Q^T of the tile whose K fragments are in ka (4 dependent MFMAs)
    f32x16 sacc;
    auto s_mfma = [&](f32x16 &acc, int s) {
        union { uint4 u; bf16x8 v; } a, b;
        a.u = ka[s];
        b.u = qf[s];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
    };
    if (active) {
#pragma unroll
        for (int r = 0; r < 16; r++) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < ((D2R_ATTN_ABLATE & 128) ? 0 : 4); s++) s_mfma(sacc, s);
        // the first key tile's maximum is the first running maximum (key rows >= T repeat row T-1: a duplicate of a real score
        // changes no maximum); from here on the scores reach ats_softmax with m_run already subtracted
        float m0 = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; r++) m0 = fmaxf(m0, sacc[r]);
        m_run = half_max(m0);
#pragma unroll
        for (int r = 0; r < 16; r++) sacc[r] -= m_run;
    }
    // One key tile, whose scores are already in sacc.  Unless LAST, first the hand-over to the next tile: wait for
    // this wave's request of tile kt+1 (WAIT younger requests may stay in flight), barrier — tile kt+1 is complete and
    // every wave has finished tile kt-1, whose slot takes the request for tile kt + ATS_STAGES - 1 (REQ) — and the K
    // fragments of tile kt+1 are read, their latency under this tile's softmax.  Then O^T += V^T P^T (two accumulators,
    // two k-steps) and the NEXT tile's S^T = K Q^T.
    auto tile = [&](uint32_t kt, auto wait_tag, auto req_tag, auto last_tag) {
        constexpr int WAIT = decltype(wait_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value, REQ = decltype(req_tag)::value;
        if constexpr (!LAST) {
            wait_vmcnt<WAIT * NS>();
            if (!(D2R_ATTN_ABLATE & 256)) __syncthreads();
            if constexpr (REQ) request(kt + ATS_STAGES - 1);
            read_k();
        }
        const lds_cptr v0 = (lds_cptr)(size_t)(smem0 + slot_v + voff0), v1 = (lds_cptr)(size_t)(smem0 + slot_v + voff1);
        advance(slot_v);
        if (!active) return;
        // lane (q, hi), reg r  <->  key kt*32 + (r&3) + 8*(r>>2) + 4*hi
        uint4 pb[2];
        const bool short_tile = LAST && n_last <= 8;          // wave-uniform
        if (short_tile) {
#pragma unroll
            for (int r = 0; r < 4; r++) sacc[r] = (uint32_t)r + 4 * hi < n_last ? sacc[r] : -INFINITY;
            ats_softmax<4>(sacc, o0, o1, m_run, l_run, pb);
        } else {
            if (LAST && n_last < 32) {
#pragma unroll
                for (int r = 0; r < 16; r++) sacc[r] = (uint32_t)((r & 3) + 8 * (r >> 2)) + 4 * hi < n_last ? sacc[r] : -INFINITY;
            }
            ats_softmax<16>(sacc, o0, o1, m_run, l_run, pb);
        }
        // A slot (hi, j) <-> key 16s + 8(j>>2) + 4hi + (j&3): two transposing reads of 4 keys each per O tile and k-step
#pragma unroll
        for (int s = 0; s < ((D2R_ATTN_ABLATE & 64) ? 0 : 2); s++) {
            if (s == 1 && short_tile) break;                  // keys 16.. of a short last tile: P = 0
            union { uint4 u; bf16x8 v; } p, va0, va1;
            p.u = pb[s];
            const uint2 a00 = lds_read_tr16(v0 + 2048u * s), a01 = lds_read_tr16(v0 + 2048u * s + 1024u);
            const uint2 a10 = lds_read_tr16(v1 + 2048u * s), a11 = lds_read_tr16(v1 + 2048u * s + 1024u);
            va0.u = make_uint4(a00.x, a00.y, a01.x, a01.y);
            va1.u = make_uint4(a10.x, a10.y, a11.x, a11.y);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0.v, p.v, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1.v, p.v, o1, 0, 0, 0);
        }
        // the next tile's scores (interleaving these four with the four above, three independent chains, measured
        // no faster and costs ten registers)
        if constexpr (!LAST) {
#pragma unroll
            for (int r = 0; r < 16; r++) sacc[r] = (D2R_ATTN_ABLATE & 128) ? __uint_as_float(ka[r & 3].x) : 0.f;
            if (!(D2R_ATTN_ABLATE & 128)) {
                // the accumulator starts at -m_run (the MFMAs subtract the running maximum): eight 64-bit moves (hipcc emits sixteen v_mov_b32
                // for the plain assignment; volatile: identical statements must not be merged)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 nm = {-m_run, -m_run}, pr[8];
#pragma unroll
                for (int q = 0; q < 8; q++) asm volatile("v_mov_b64 %0, %1" : "=v"(pr[q]) : "v"(nm));
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    sacc[2 * q] = pr[q].x;
                    sacc[2 * q + 1] = pr[q].y;
                }
            }
#pragma unroll
            for (int s = 0; s < ((D2R_ATTN_ABLATE & 128) ? 0 : 4); s++) s_mfma(sacc, s);
        }
    };
    using std::integral_constant;
    const integral_constant<bool, false> no{};
    const integral_constant<bool, true> yes{};
    uint32_t kt = 0;
    for (; kt + ATS_STAGES - 1 < n_kt; kt++) tile(kt, integral_constant<int, ATS_STAGES - 3>{}, yes, no);
    // the last ATS_STAGES - 1 tiles request nothing; r tiles left -> r - 2 requests younger than tile kt+1's
    if (n_kt - kt == 4) { tile(kt, integral_constant<int, 2>{}, no, no); kt++; }
    if (n_kt - kt == 3) { tile(kt, integral_constant<int, 1>{}, no, no); kt++; }
    if (n_kt - kt == 2) { tile(kt, integral_constant<int, 0>{}, no, no); kt++; }
    tile(kt, integral_constant<int, 0>{}, no, yes);
    wait_vmcnt<0>();          // sequences shorter than the prologue's four tiles leave filler requests in flight: nothing may land after the exit
    if (!active) return;
    const float inv_l = 1.0f / half_sum(l_run);
    // lane (q, hi), reg r of o{0,1} <-> dim 32*{0,1} + (r&3) + 8*(r>>2) + 4*hi: after bf16 packing a lane owns 4 dims of
    // every 8-dim group; v_permlane32_swap pairs the groups (2g, 2g+1) so that lanes 0-31 hold dims 16g .. 16g+7 and
    // lanes 32-63 dims 16g+8 .. 16g+15 of their query: 16-byte stores, 4 per O tile instead of 8 of 8 bytes.
    // (the lane id is taken afresh so that the output address is not kept in registers across the tile loop)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t qrow_e = qt * 32 + (lane_e & 31);
    if constexpr (Q8) {
        // reg 4j + i of o_t <-> dim 32t + 8j + 4hi + i: a dword of four e4m3 per (t, j); the halves swap dwords so that a lane stores 8
        // consecutive dims (hi = 0: 16g .. 16g+7, hi = 1: 16g+8 .. 16g+15 of each 32-dim tile)
        float am = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) am = fmaxf(am, fmaxf(fabsf(o0[r]), fabsf(o1[r])));
        am *= inv_l;
        const u32x2 ams = __builtin_amdgcn_permlane32_swap(__float_as_uint(am), __float_as_uint(am), false, false);
        const uint32_t sb = q8_scale_byte(fmaxf(__uint_as_float(ams[0]), __uint_as_float(ams[1])));
        const float sc = inv_l * q8_inv_scale(sb);
        uint8_t *dst8 = (uint8_t *)AO + (((size_t)head * M_pad + row_base + qrow_e) << 6) + 8 * (lane_e >> 5);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const f32x16 &o = t ? o1 : o0;
#pragma unroll
            for (int g = 0; g < 2; g++) {
                uint32_t a = __builtin_amdgcn_cvt_pk_fp8_f32(o[8 * g + 0] * sc, o[8 * g + 1] * sc, 0, false);
                a = __builtin_amdgcn_cvt_pk_fp8_f32(o[8 * g + 2] * sc, o[8 * g + 3] * sc, a, true);
                uint32_t b = __builtin_amdgcn_cvt_pk_fp8_f32(o[8 * g + 4] * sc, o[8 * g + 5] * sc, 0, false);
                b = __builtin_amdgcn_cvt_pk_fp8_f32(o[8 * g + 6] * sc, o[8 * g + 7] * sc, b, true);
                const u32x2 x = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                if (qrow_e < T) *(uint2 *)(dst8 + 32 * t + 16 * g) = make_uint2(x[0], x[1]);
            }
        }
        if (qrow_e < T && lane_e < 32) AOS[q8_scale_off(M_pad, row_base + qrow_e, head)] = (uint8_t)sb;
        return;
    }
    uint16_t *dst = AO + ((size_t)head * M_pad + row_base + qrow_e) * 64 + 8 * (lane_e >> 5);   // tile-major: head h is plane h of [d/64][M_pad][64]
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const f32x16 &o = t ? o1 : o0;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            uint32_t a0 = pack2(o[8 * g + 0] * inv_l, o[8 * g + 1] * inv_l), a1 = pack2(o[8 * g + 2] * inv_l, o[8 * g + 3] * inv_l);
            uint32_t b0 = pack2(o[8 * g + 4] * inv_l, o[8 * g + 5] * inv_l), b1 = pack2(o[8 * g + 6] * inv_l, o[8 * g + 7] * inv_l);
            const u32x2 x = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const u32x2 y = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            if (qrow_e < T) *(uint4 *)(dst + 32 * t + 16 * g) = make_uint4(x[0], y[0], x[1], y[1]);
        }
    }
}

// ---- last transformer block: only the class token feeds the head ----
//
// post_layernorm and the projection read token 0 of every image and nothing else (reference clip_scoring.py:180-181:
// CLIPModel's pooled output), so in the LAST block only that row's attention output, out-projection and MLP are
// live: 1 query instead of T per (image, head), M = n rows instead of n*T for three of the four GEMMs.  The other
// rows of the last block are dead code; the result is the same function of the inputs (K and V still come from
// all tokens).  k_attention_cls: one wave per (image, head); fp32 softmax over the T keys.
__global__ __launch_bounds__(256) void k_attention_cls(const uint16_t *__restrict__ QKV, const uint16_t *__restrict__ Q_cls,
                                                       uint16_t *__restrict__ AO_cls, uint32_t T, uint32_t d, uint32_t M_pad,
                                                       uint32_t n_heads, uint32_t n_items)
{
    __shared__ float ps[4][1024];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t item = blockIdx.x * 4 + w;
    if (item >= n_items) return;
    const uint32_t img = item / n_heads, head = item - img * n_heads, H = d >> 6;
    const size_t row0 = (size_t)img * T;
    // q of token 0: from the compact [n][d] array when the block's q was only computed for the class tokens
    const uint16_t *Qg = Q_cls ? Q_cls + (size_t)img * d + head * 64 : QKV + ((size_t)head * M_pad + row0) * 64;     // tile-major planes (see k_attention)
    const uint16_t *Kg = QKV + ((size_t)(H + head) * M_pad + row0) * 64;
    const uint16_t *Vg = QKV + ((size_t)(2 * H + head) * M_pad + row0) * 64;
    // q (token 0) as 64 floats, same in every lane
    float q[64];
    {
        const uint4 *qp = (const uint4 *)Qg;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint4 v = qp[c];
            q[8 * c + 0] = bf_lo(v.x); q[8 * c + 1] = bf_hi(v.x); q[8 * c + 2] = bf_lo(v.y); q[8 * c + 3] = bf_hi(v.y);
            q[8 * c + 4] = bf_lo(v.z); q[8 * c + 5] = bf_hi(v.z); q[8 * c + 6] = bf_lo(v.w); q[8 * c + 7] = bf_hi(v.w);
        }
    }
    // scores: lane takes keys lane, lane + 64, ... (T <= 1024), kept in LDS
    float mx = -INFINITY;
    for (uint32_t key = lane; key < T; key += 64) {
        const uint4 *kp = (const uint4 *)(Kg + (size_t)key * 64);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint4 v = kp[c];
            a = fmaf(q[8 * c + 0], bf_lo(v.x), a); a = fmaf(q[8 * c + 1], bf_hi(v.x), a);
            a = fmaf(q[8 * c + 2], bf_lo(v.y), a); a = fmaf(q[8 * c + 3], bf_hi(v.y), a);
            a = fmaf(q[8 * c + 4], bf_lo(v.z), a); a = fmaf(q[8 * c + 5], bf_hi(v.z), a);
            a = fmaf(q[8 * c + 6], bf_lo(v.w), a); a = fmaf(q[8 * c + 7], bf_hi(v.w), a);
        }
        ps[w][key] = a;                                           // q carries head_dim^-0.5 * log2(e) (folded into W_q at create): exp2 domain
        mx = fmaxf(mx, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (uint32_t key = lane; key < T; key += 64) {
        const float p = __builtin_amdgcn_exp2f(ps[w][key] - mx);
        ps[w][key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    // o[dim = lane] = sum_key p[key] V[key][lane] / sum
    float o = 0.f;
    for (uint32_t key = 0; key < T; key++) o = fmaf(ps[w][key], __uint_as_float((uint32_t)Vg[(size_t)key * 64 + lane] << 16), o);
    AO_cls[(size_t)img * d + head * 64 + lane] = f2bf(o / sum);
}

// class-token rows of the residual stream -> compact fp32 [n][d] (bf16 hi (+ lo) tile-major planes, or fp32 row-major X)
__global__ void k_gather_cls(const float *__restrict__ X, const uint16_t *__restrict__ Xhi, const uint16_t *__restrict__ Xlo,
                             uint32_t M_pad, uint32_t T, uint32_t d, uint32_t n, float *__restrict__ X_cls, int lo8 = 0)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * d) return;
    const uint32_t img = i / d, c = i - img * d;
    const size_t row = (size_t)img * T;
    float v;
    if (Xhi) {
        const size_t xo = ((size_t)(c >> 6) * M_pad + row) * 64 + (c & 63);
        v = __uint_as_float((uint32_t)Xhi[xo] << 16);
        if (Xlo && lo8) v = split8_value(Xhi[xo], ((const int8_t *)Xlo)[lo8_off(M_pad, row, c)]);
        else if (Xlo) v += __uint_as_float((uint32_t)Xlo[xo] << 16);
    } else {
        v = X[row * d + c];
    }
    X_cls[i] = v;
}

// class-token rows of the bf16 operand copy (tile-major) and their LayerNorm pairs -> compact row-major [n][d], [n]:
// the A operand and row statistics of the last block's q projection, which only the class tokens need
__global__ void k_gather_cls_operand(const uint16_t *__restrict__ Xhi, const float2 *__restrict__ AB, uint32_t M_pad, uint32_t T,
                                     uint32_t d, uint32_t n, uint16_t *__restrict__ Xq, float2 *__restrict__ ABq)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, per = d / 8;
    if (i >= n * per) return;
    const uint32_t img = i / per, c = (i - img * per) * 8;
    const size_t row = (size_t)img * T;
    *(uint4 *)(Xq + (size_t)img * d + c) = *(const uint4 *)(Xhi + ((size_t)(c >> 6) * M_pad + row) * 64 + (c & 63));
    if (c == 0) ABq[img] = AB[row];
}

// ------------------------------------------------------------------ head

// one 1024-thread block per HEAD_NI images: post_layernorm(X[b*T]) -> proj -> L2 normalise -> logits.
// 16 waves and 4 output rows per wave iteration keep ~48 loads per lane in flight: the
// projection is a latency problem (1.5 MB of fp32 weights out of L2), not a flop one — and the images of a block share
// every weight load (round 5: four images per block, a quarter of the L2 reads; each image's arithmetic and summation order
// are those of the one-image kernel: the same bits).
#define HEAD_THREADS 1024
#define HEAD_NI 4
__global__ __launch_bounds__(HEAD_THREADS) void k_head(const float *__restrict__ X, const uint16_t *__restrict__ Xb,
                                                       const uint16_t *__restrict__ Xlo, uint32_t M_pad,
                                                       const uint32_t *__restrict__ pool_row, uint32_t T, uint32_t d,
                                                       const float *__restrict__ lw, const float *__restrict__ lb,
                                                       const float *__restrict__ proj, uint32_t D,
                                                       const float *__restrict__ text, uint32_t C, float logit_scale,
                                                       float *__restrict__ logits, float *__restrict__ embeds, uint32_t n_items, int lo8 = 0)
{
    __shared__ float xs[HEAD_NI][1024];
    __shared__ float es[HEAD_NI][1024];
    __shared__ float red[HEAD_NI][16];
    constexpr uint32_t NW = HEAD_THREADS / 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t item0 = blockIdx.x * HEAD_NI;
    const uint32_t ni = min((uint32_t)HEAD_NI, n_items - item0);              // block-uniform
    // vision: the class token (row 0 of the image); text: the EOS token's row
    // (the residual stream is fp32 X, or bf16 Xb when the tower runs with a bf16 residual stream)
    float xv[HEAD_NI];
#pragma unroll
    for (uint32_t m = 0; m < HEAD_NI; m++) {
        xv[m] = 0.f;
        if (m < ni && tid < d) {                                                                               // d <= 1024
            const size_t xrow = (size_t)(item0 + m) * T + (pool_row ? pool_row[item0 + m] : 0u);
            const size_t xo = ((size_t)(tid >> 6) * M_pad + xrow) * 64 + (tid & 63);                            // bf16 residual: tile-major
            xv[m] = Xb ? __uint_as_float((uint32_t)Xb[xo] << 16) : X[xrow * d + tid];
            if (Xb && Xlo && lo8) xv[m] = split8_value(Xb[xo], ((const int8_t *)Xlo)[lo8_off(M_pad, xrow, tid)]);
            else if (Xb && Xlo) xv[m] += __uint_as_float((uint32_t)Xlo[xo] << 16);
        }
        const float s = wave_sum(xv[m]);
        if (lane == 0) red[m][wave] = s;
    }
    __syncthreads();
    float mu[HEAD_NI], dv[HEAD_NI];
#pragma unroll
    for (uint32_t m = 0; m < HEAD_NI; m++) {
        float tot = 0.f;
#pragma unroll
        for (uint32_t i = 0; i < NW; i++) tot += red[m][i];
        mu[m] = tot / (float)d;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t m = 0; m < HEAD_NI; m++) {
        dv[m] = tid < d ? xv[m] - mu[m] : 0.f;
        const float q = wave_sum(dv[m] * dv[m]);
        if (lane == 0) red[m][wave] = q;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t m = 0; m < HEAD_NI; m++) {
        float tot = 0.f;
#pragma unroll
        for (uint32_t i = 0; i < NW; i++) tot += red[m][i];
        const float rstd = 1.0f / sqrtf(tot / (float)d + 1e-5f);
        if (tid < d) xs[m][tid] = dv[m] * rstd * lw[tid] + lb[tid];
    }
    __syncthreads();
    // projection: wave handles rows o, o+NW, ... four at a time, for every image of the block
    float nrm[HEAD_NI];
#pragma unroll
    for (uint32_t m = 0; m < HEAD_NI; m++) nrm[m] = 0.f;
    for (uint32_t o = wave * 4; o < D; o += NW * 4) {
        float a[HEAD_NI][4];
#pragma unroll
        for (uint32_t m = 0; m < HEAD_NI; m++) a[m][0] = a[m][1] = a[m][2] = a[m][3] = 0.f;
        const float *p0 = proj + (size_t)o * d;
        const bool v1 = o + 1 < D, v2 = o + 2 < D, v3 = o + 3 < D;
        for (uint32_t i = lane; i < d; i += 64) {
            const float w0 = p0[i], w1 = v1 ? p0[d + i] : 0.f, w2 = v2 ? p0[2 * d + i] : 0.f, w3 = v3 ? p0[3 * d + i] : 0.f;
#pragma unroll
            for (uint32_t m = 0; m < HEAD_NI; m++) {
                const float xi = xs[m][i];
                a[m][0] = fmaf(w0, xi, a[m][0]);
                if (v1) a[m][1] = fmaf(w1, xi, a[m][1]);
                if (v2) a[m][2] = fmaf(w2, xi, a[m][2]);
                if (v3) a[m][3] = fmaf(w3, xi, a[m][3]);
            }
        }
#pragma unroll
        for (uint32_t m = 0; m < HEAD_NI; m++) {
            const float a0 = wave_sum(a[m][0]), a1 = wave_sum(a[m][1]), a2 = wave_sum(a[m][2]), a3 = wave_sum(a[m][3]);
            if (lane == 0) {
                es[m][o] = a0;
                if (v1) es[m][o + 1] = a1;
                if (v2) es[m][o + 2] = a2;
                if (v3) es[m][o + 3] = a3;
            }
            nrm[m] += a0 * a0 + (v1 ? a1 * a1 : 0.f) + (v2 ? a2 * a2 : 0.f) + (v3 ? a3 * a3 : 0.f);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t m = 0; m < HEAD_NI; m++)
        if (lane == 0) red[m][wave] = nrm[m];
    __syncthreads();
#pragma unroll
    for (uint32_t m = 0; m < HEAD_NI; m++) {
        float tot = 0.f;
#pragma unroll
        for (uint32_t i = 0; i < NW; i++) tot += red[m][i];
        const float inv = 1.0f / sqrtf(tot);
        if (m < ni && tid < D) {
            es[m][tid] *= inv;
            if (embeds) embeds[(size_t)(item0 + m) * D + tid] = es[m][tid];
        }
    }
    __syncthreads();
    for (uint32_t m = 0; m < ni; m++)
        for (uint32_t c = wave; c < C; c += NW) {
            float a = 0.f;
            for (uint32_t i = lane; i < D; i += 64) a = fmaf(es[m][i], text[(size_t)c * D + i], a);
            a = wave_sum(a);
            if (lane == 0 && logits) logits[(size_t)(item0 + m) * C + c] = logit_scale * a;
        }
}

// X[c*T + t] = token_embedding[ids[c][t]] + position_embedding[t]; pool_row[c] = argmax_t ids[c][t]
// (EOS has the largest id: HF 4.27 pools at argmax, 5.x at the first EOS — identical here)
__global__ void k_text_embed(const int32_t *__restrict__ ids, const float *__restrict__ tok, const float *__restrict__ pos,
                             float *__restrict__ X, uint32_t *__restrict__ pool_row, uint32_t Cn, uint32_t T, uint32_t d,
                             uint32_t vocab)
{
    const uint32_t row = blockIdx.x, c = row / T, t = row % T;
    int32_t id = ids[row];
    id = id < 0 ? 0 : (id >= (int32_t)vocab ? (int32_t)vocab - 1 : id);
    for (uint32_t i = threadIdx.x; i < d; i += blockDim.x)
        X[(size_t)row * d + i] = tok[(size_t)id * d + i] + pos[(size_t)t * d + i];
    if (t == 0 && threadIdx.x == 0) {
        int32_t best = ids[(size_t)c * T];
        uint32_t bi = 0;
        for (uint32_t k = 1; k < T; k++) {
            int32_t v = ids[(size_t)c * T + k];
            if (v > best) { best = v; bi = k; }
        }
        pool_row[c] = bi;
    }
}

// fp32 -> bf16 weight conversion with optional K padding
__global__ void k_convert_bf16(const float *__restrict__ src, uint16_t *__restrict__ dst, uint32_t rows,
                               uint32_t K, uint32_t K_pad, float scale = 1.0f)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * K_pad) return;
    uint32_t r = i / K_pad, c = i % K_pad;
    dst[i] = c < K ? f2bf(src[(size_t)r * K + c] * scale) : (uint16_t)0;
}
__global__ void k_scale_f32(float *__restrict__ p, uint32_t n, float scale)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] *= scale;
}

// ---------------------------------------------------------------- host side

static inline uint32_t round_up(uint32_t a, uint32_t b) { return (a + b - 1) / b * b; }

static double bicubic_filter(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Pillow's antialiased bicubic coefficient tables (Resample.c precompute_coeffs +
// normalize_coeffs_8bpc), restated; oracle counterpart: d2r_oracle_resample_coeffs
static int resample_coeffs(int in_size, int out_size, std::vector<int> &bounds, std::vector<int> &kk)
{
    double scale = (double)in_size / out_size, filterscale = scale < 1.0 ? 1.0 : scale;
    double support = 2.0 * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; xx++) {
        double center = (xx + 0.5) * scale, ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; x++) {
            k[x] = bicubic_filter((x + xmin - center + 0.5) * ss);
            ww += k[x];
        }
        for (int x = 0; x < ksize; x++) {
            double v = x < xmax ? (ww != 0.0 ? k[x] / ww : k[x]) : 0.0;
            kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS))
                                               : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[xx * 2] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    return ksize;
}

// Resampling geometry + Pillow coefficient tables for one frame size, uploaded once.
static int prep_tables(d2r_ctx *ctx, d2r_clip *clip, uint32_t w, uint32_t h, int rot90, const PrepCache **out)
{
    std::lock_guard<std::mutex> lock(clip->prep_mu);
    for (const PrepCache &pc : clip->prep)
        if (pc.w == w && pc.h == h && pc.rot90 == rot90) {
            *out = &pc;
            return D2R_OK;
        }
    const uint32_t S = clip->desc.image_size, P = clip->desc.patch_size;
    ResampleTables R{};
    R.iw = rot90 ? (int)h : (int)w;
    R.ih = rot90 ? (int)w : (int)h;
    // HF get_resize_output_image_size(shortest_edge=S, default_to_square=False)
    int short_e = R.iw < R.ih ? R.iw : R.ih, long_e = R.iw < R.ih ? R.ih : R.iw;
    int new_long = (int)((double)S * long_e / short_e);
    R.rw = R.iw <= R.ih ? (int)S : new_long;
    R.rh = R.iw <= R.ih ? new_long : (int)S;
    R.need_h = R.rw != R.iw;
    R.need_v = R.rh != R.ih;
    R.left = (R.rw - (int)S) / 2;
    R.top = (R.rh - (int)S) / 2;
    std::vector<int> bh, kh, bv, kv;
    R.ks_h = resample_coeffs(R.iw, R.rw, bh, kh);
    R.ks_v = resample_coeffs(R.ih, R.rh, bv, kv);
    const uint32_t band = P;
    int max_rows = (int)band;
    if (R.need_v) {
        max_rows = 0;
        for (uint32_t r0 = 0; r0 < S; r0 += band) {
            uint32_t last = std::min(r0 + band, S) - 1;
            int y0 = bv[2 * (R.top + r0)], y1 = bv[2 * (R.top + last)] + bv[2 * (R.top + last) + 1];
            max_rows = std::max(max_rows, y1 - y0);
        }
    }
    R.max_rows = max_rows;
    std::vector<int> all;
    size_t o_bh = 0, o_kh = o_bh + bh.size(), o_bv = o_kh + kh.size(), o_kv = o_bv + bv.size();
    all.insert(all.end(), bh.begin(), bh.end());
    all.insert(all.end(), kh.begin(), kh.end());
    all.insert(all.end(), bv.begin(), bv.end());
    all.insert(all.end(), kv.begin(), kv.end());
    int *dev = nullptr;
    if (hipMalloc(&dev, all.size() * sizeof(int)) != hipSuccess) return d2r_fail(ctx, D2R_ERR_MEMORY, "hipMalloc failed");
    clip->allocs.push_back(dev);
    D2R_HIP(ctx, hipMemcpy(dev, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
    R.bounds_h = dev + o_bh;
    R.kk_h = dev + o_kh;
    R.bounds_v = dev + o_bv;
    R.kk_v = dev + o_kv;
    clip->prep.push_back(PrepCache{w, h, rot90, R});
    *out = &clip->prep.back();
    return D2R_OK;
}

// frames (device, [n][h][w][3] u8) -> patches (bf16 [n*g*g][Kp_pad]) and/or pixel_values
int d2r_launch_preprocess(d2r_ctx *ctx, d2r_clip *clip, const uint8_t *frames_dev, uint32_t n, uint32_t w,
                          uint32_t h, int rot90, uint16_t *patches_dev, float *pixel_values_dev, const void *rects_dev,
                          const uint16_t *bg_patches_dev, bool touched_only)
{
    const uint32_t S = clip->desc.image_size, P = clip->desc.patch_size;
    const PrepCache *pc = nullptr;
    int rc = prep_tables(ctx, clip, w, h, rot90, &pc);
    if (rc) return rc;
    const ResampleTables &R = pc->R;
    const uint32_t band = P;
    size_t lds = (size_t)R.max_rows * S * 3;
    if (lds > 150 * 1024) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "preprocess band does not fit in LDS");
    static PerDeviceOnce prep_attr;
    prep_attr.run(ctx->device, [] {
        (void)hipFuncSetAttribute((const void *)k_preprocess, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    hipLaunchKernelGGL(k_preprocess, dim3((S + band - 1) / band, n), dim3(256), lds, ctx->stream, frames_dev, w, h,
                       rot90, R, S, P, band, patches_dev, clip->Kp_pad, pixel_values_dev, (const int4 *)rects_dev, bg_patches_dev,
                       touched_only ? 1 : 0);
    D2R_HIP(ctx, hipGetLastError());
    return D2R_OK;
}

template <int EPI, int WGM, int WGN, int MT, int STAGES>
static int launch_gemm_cfg(d2r_ctx *ctx, const uint16_t *A, const uint16_t *W, const float *bias, void *C,
                           uint32_t M_real, uint32_t N, uint32_t K, const EpiAux &aux)
{
    constexpr uint32_t TBM = WGM * MT * 32, TBN = WGN * 64, LDS = STAGES * (TBM + TBN) * BK * 2;
    const uint32_t M_pad = round_up(M_real, BM);
    const uint32_t nwg = (M_pad / TBM) * (N / TBN);
    static PerDeviceOnce attr_set;                       // per device: a process may hold contexts on several GPUs
    attr_set.run(ctx->device, [] {
        (void)hipFuncSetAttribute((const void *)k_gemm<EPI, WGM, WGN, MT, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    // (a device-sized product runs on one workgroup per CU walking the live tiles)
    hipLaunchKernelGGL((k_gemm<EPI, WGM, WGN, MT, STAGES>), dim3(aux.m_dev ? std::min<uint32_t>(nwg, (uint32_t)ctx->n_cu) : nwg), dim3(WGM * WGN * 64), LDS,
                       ctx->stream, A, W, bias, C, M_pad, N, K, (uint32_t)ctx->n_xcd, aux);
    D2R_HIP(ctx, hipGetLastError());
    return D2R_OK;
}

// the persistent 256x256 kernel (bf16 operands: K % 128 == 0; fp8 operands, EPI_F8_*: K in bytes, K % 256 == 0); N % 256 == 0
template <int EPI>
static int launch_gemm8(d2r_ctx *ctx, const uint16_t *A, const uint16_t *W, const float *bias, void *C, uint32_t M_real, uint32_t N, uint32_t K,
                        const EpiAux &aux)
{
    const uint64_t tiles256 = (uint64_t)(round_up(M_real, BM) / 256) * (N / 256);
    constexpr uint32_t ELEM = EPI_IS_F8(EPI) ? 1u : 2u;          // operand bytes per element
    const uint32_t M_pad = round_up(M_real, BM);
    constexpr uint32_t LDS8 = 128 * 1024 + 8 * EP_WAVE_FLOATS * 4 + 8 * 1024;     // ring + transpose buffers + (rstd, -rstd*mean) strips
    static PerDeviceOnce attr8;
    attr8.run(ctx->device, [] {
        if constexpr (EPI_IS_F8(EPI)) (void)hipFuncSetAttribute((const void *)k_gemm8f<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
        else (void)hipFuncSetAttribute((const void *)k_gemm8<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
    });
    // persistent: one workgroup per CU (a multiple of the XCD count, so every XCD gets the same number)
    const uint32_t nwg8 = (uint32_t)(ctx->n_cu / ctx->n_xcd * ctx->n_xcd);
    // one eighth of a tile period in s_sleep(127) units (8 128 cycles each): a K-tile costs ~2.9 k cycles, an epilogue 5-18 k
    const uint32_t tiles_per_wg = (uint32_t)((tiles256 + nwg8 - 1) / nwg8);
    const uint32_t period = (K * ELEM / 128u) * 2900u + 9000u;
    const uint32_t sleeps = (ctx->gemm_stagger && tiles_per_wg >= 4) ? std::max(1u, period / 8u / 8128u) : 0u;
    // column tiles per group of the tile order: the choice that misses the L2 least by a simple model —
    // every group re-reads A once; a group whose W panels (gn * K * 512 B) fit beside the streaming A stays
    // resident, a wider one is re-read every round of workgroups
    uint32_t gn = (uint32_t)ctx->gemm_group;
    if (gn == 0) {
        const uint32_t tn = N / 256;
        const double a_bytes = (double)M_pad * K * ELEM, w_bytes = (double)N * K * ELEM;
        const double rounds = (double)tiles256 / (double)nwg8 * ctx->n_xcd, l2_budget = 2.0 * 1024 * 1024;
        double best = 1e300;
        for (uint32_t g = 1; g <= tn; g++) {
            const double cost = a_bytes * ((tn + g - 1) / g) + ((double)g * K * 256.0 * ELEM <= l2_budget ? w_bytes : w_bytes * rounds);
            if (cost < best) { best = cost; gn = g; }
        }
    }
    gn = std::max(1u, std::min(gn, N / 256));
    // column sections: only when they divide the XCDs and the column tiles evenly
    uint32_t ns = (uint32_t)ctx->gemm_nsplit;
    if (ns == 0) ns = 2;            // default: two sections where they fit (fc1: -0.2 ms per launch; four measure the same)
    if (ns < 1 || (uint32_t)ctx->n_xcd % ns || (N / 256) % ns) ns = 1;
    if constexpr (EPI_IS_F8(EPI))
        hipLaunchKernelGGL((k_gemm8f<EPI>), dim3(nwg8), dim3(512), LDS8, ctx->stream, (const uint8_t *)A, (const uint8_t *)W, bias, C, M_pad, N, K,
                           (uint32_t)ctx->n_xcd, aux, (ns << 16) | std::min(gn, 0xffffu));
    else
    hipLaunchKernelGGL((k_gemm8<EPI>), dim3(nwg8), dim3(512), LDS8, ctx->stream, A, W, bias, C, M_pad, N, K,
                       (uint32_t)ctx->n_xcd, aux, sleeps, (ns << 16) | std::min(gn, 0xffffu));
    D2R_HIP(ctx, hipGetLastError());
    return D2R_OK;
}

template <int EPI>
static int launch_gemm(d2r_ctx *ctx, const uint16_t *A, const uint16_t *W, const float *bias, void *C,
                       uint32_t M_real, uint32_t N, uint32_t K, const EpiAux &aux = EpiAux{})
{
    if (N % 128 || K % BK) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "GEMM N must be a multiple of 128 and K of 64");
    if ((uint64_t)round_up(M_real, BM) * N >= (1ull << 32))
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "GEMM output too large for 32-bit indexing");
    if (EPI_IS_STATS(EPI) && (uint64_t)round_up(M_real, BM) * N * (EPI == EPI_RESID_STATS_F32X ? 4 : 2) >= (1ull << 32))      // its epilogue indexes the residual arrays in bytes
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "residual array too large for 32-bit byte offsets");
    if (EPI_IS_STATS(EPI) && aux.hm_rows != round_up(M_real, BM))
        return d2r_fail(ctx, D2R_ERR_INVALID, "residual + statistics epilogues write tile-major planes of M_pad rows");
    // wide outputs: 256x256 tiles (more flops per byte staged); narrow ones keep 256x128 so the tile
    // count still covers the 256 CUs a few times
    if (ctx->gemm_cfg == 2) return launch_gemm_cfg<EPI, 4, 2, 2, 3>(ctx, A, W, bias, C, M_real, N, K, aux);   // force 256x128
    // 256x256 tiles whenever they still cover the 256 CUs at least ~4 times, else 256x128
    const uint64_t tiles256 = (uint64_t)(round_up(M_real, BM) / 256) * (N / 256);
    if (N % 256 == 0 && (N >= 2048 || tiles256 >= 1024)) {
        if (K % 128 == 0 && ctx->gemm_cfg != 1)          // gemm_cfg 1 = the two-stage K loop (kept for comparison)
            return launch_gemm8<EPI>(ctx, A, W, bias, C, M_real, N, K, aux);
        return launch_gemm_cfg<EPI, 2, 4, 4, 2>(ctx, A, W, bias, C, M_real, N, K, aux);
    }
    return launch_gemm_cfg<EPI, 4, 2, 2, 3>(ctx, A, W, bias, C, M_real, N, K, aux);
}

// fp8 operands ("fp8 blocks"): A8 = e4m3 planes [K / 64][M_pad][64] with scale dwords a_scale [K / 256][M_pad], W8 = e4m3 [N][K], one scale
template <int EPI>
static int launch_gemm_f8(d2r_ctx *ctx, const uint8_t *A8, const uint8_t *a_scale, const uint8_t *W8, float w_scale, const float *bias, void *C,
                          uint32_t M_real, uint32_t N, uint32_t K, EpiAux aux)
{
    static_assert(EPI_IS_F8(EPI), "fp8 epilogues only");
    const uint64_t M_pad = round_up(M_real, BM);
    if (N % 256 || K % 256) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "fp8 GEMM needs N and K to be multiples of 256");
    if (M_pad * N >= (1ull << 32) || M_pad * 64 * 4 >= (1ull << 32) || (uint64_t)N * K >= (1ull << 32))
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "fp8 GEMM operand too large for 32-bit indexing");
    if (EPI_IS_STATS(EPI) && (M_pad * N * 2 >= (1ull << 32) || aux.hm_rows != M_pad))
        return d2r_fail(ctx, D2R_ERR_INVALID, "residual + statistics epilogues write tile-major planes of M_pad rows (< 4 GiB)");
    if (EPI == EPI_F8_BIAS_GELU_Q8 && (aux.hm_rows != M_pad || !aux.q8 || !aux.q8_scale))
        return d2r_fail(ctx, D2R_ERR_INVALID, "fp8 output planes have M_pad rows");
    aux.a_scale = (const uint32_t *)a_scale;
    aux.w_scale = w_scale;
    return launch_gemm8<EPI>(ctx, (const uint16_t *)A8, (const uint16_t *)W8, bias, C, M_real, N, K, aux);
}

// vision tower: streamed attention, one workgroup per (image, head, group of eight query tiles)
static void launch_attention_vision(d2r_ctx *ctx, const uint16_t *QKV, uint16_t *AO, uint32_t T, uint32_t d, uint32_t M_pad,
                                    uint32_t n_heads, uint32_t n, uint8_t *q8_scales = nullptr)
{
    // n_qt = 8 g + r query tiles: g full groups on the eight-wave kernel; a short remainder (r <= 4) on workgroups of
    // r (rounded up to 1 / 2 / 4) waves, r >= 5 as one more eight-wave group
    const uint32_t n_qt = (T + 31) / 32, r = n_qt % 8, g = r == 0 || r > (uint32_t)ctx->attn_rem ? (n_qt + 7) / 8 : n_qt / 8;
    const uint32_t lds = ATS_STAGES * ATS_SLOT;
    uint8_t *const no_scales = nullptr;
    if (q8_scales) {              // "fp8 blocks": e4m3 output + scale bytes
        if (g) hipLaunchKernelGGL((k_attention_s<8, true>), dim3(n_heads, n, g), dim3(512), lds, ctx->stream, QKV, AO, T, d, M_pad, 0u, q8_scales);
        if (g * 8 < n_qt) {
            if (r == 1) hipLaunchKernelGGL((k_attention_s<1, true>), dim3(n_heads, n, 1), dim3(64), lds, ctx->stream, QKV, AO, T, d, M_pad, g * 8, q8_scales);
            else if (r == 2) hipLaunchKernelGGL((k_attention_s<2, true>), dim3(n_heads, n, 1), dim3(128), lds, ctx->stream, QKV, AO, T, d, M_pad, g * 8, q8_scales);
            else hipLaunchKernelGGL((k_attention_s<4, true>), dim3(n_heads, n, 1), dim3(256), lds, ctx->stream, QKV, AO, T, d, M_pad, g * 8, q8_scales);
        }
        return;
    }
    if (g) hipLaunchKernelGGL((k_attention_s<8>), dim3(n_heads, n, g), dim3(512), lds, ctx->stream, QKV, AO, T, d, M_pad, 0u, no_scales);
    if (g * 8 < n_qt) {
        if (r == 1) hipLaunchKernelGGL((k_attention_s<1>), dim3(n_heads, n, 1), dim3(64), lds, ctx->stream, QKV, AO, T, d, M_pad, g * 8, no_scales);
        else if (r == 2) hipLaunchKernelGGL((k_attention_s<2>), dim3(n_heads, n, 1), dim3(128), lds, ctx->stream, QKV, AO, T, d, M_pad, g * 8, no_scales);
        else hipLaunchKernelGGL((k_attention_s<4>), dim3(n_heads, n, 1), dim3(256), lds, ctx->stream, QKV, AO, T, d, M_pad, g * 8, no_scales);
    }
}

// LDS footprint of the resident k_attention (text tower) for a padded sequence length, and the one-time opt-in to
// more than 64 KiB of dynamic LDS
static int attention_setup(d2r_ctx *ctx, uint32_t T_pad, size_t *lds_out)
{
    const size_t attn_lds = (size_t)T_pad * 128 + (size_t)64 * (T_pad + 4) * 2;
    if (attn_lds > 160 * 1024) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "sequence too long for the resident attention LDS layout");
    static PerDeviceOnce attn_attr;
    attn_attr.run(ctx->device, [] {
        (void)hipFuncSetAttribute((const void *)k_attention<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    *lds_out = attn_lds;
    return D2R_OK;
}

// The last block on class-token rows only (see k_attention_cls).  QKV must already hold the block's k/v (tile-major) and
// its q, either there too or for the class tokens only in Qc (compact bf16 [n][d]);
// the residual stream is X (fp32 row-major) or Xhi (+ Xlo) (bf16 tile-major).  Leaves the block's output rows in
// Xc (compact fp32 [n][d]) for k_head with T = 1.  Workspaces: Xc, AOc (bf16 [n_pad][d]), Xnc (bf16 [n_pad][d]),
// Hc (bf16 [n_pad][mlp]) — all row-major, padded rows hold finite leftovers.
static int last_block_cls(d2r_ctx *ctx, const d2r_clip_desc &D, const ClipWeights::Layer &L, uint32_t n, uint32_t T, uint32_t rows_pad,
                          const uint16_t *QKV, const uint16_t *Qc, const float *X, const uint16_t *Xhi, const uint16_t *Xlo, float *Xc,
                          uint16_t *AOc, uint16_t *Xnc, uint16_t *Hc, int lo8 = 0)
{
    const uint32_t d = D.hidden_size, mlp = D.mlp_size, items = n * D.num_heads;
    int rc;
    hipLaunchKernelGGL(k_attention_cls, dim3((items + 3) / 4), dim3(256), 0, ctx->stream, QKV, Qc, AOc, T, d, rows_pad, D.num_heads, items);
    hipLaunchKernelGGL(k_gather_cls, dim3((n * d + 255) / 256), dim3(256), 0, ctx->stream, X, Xhi, Xlo, rows_pad, T, d, n, Xc, lo8);
    if ((rc = launch_gemm<EPI_BIAS_RESID_F32>(ctx, AOc, L.w_o, L.b_o, Xc, n, d, d))) return rc;
    hipLaunchKernelGGL(k_layernorm, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, Xc, L.ln2_w, L.ln2_b, Xnc, n, d);
    if ((rc = launch_gemm<EPI_BIAS_GELU_BF16>(ctx, Xnc, L.w_fc1, L.b_fc1, Hc, n, mlp, d))) return rc;
    if ((rc = launch_gemm<EPI_BIAS_RESID_F32>(ctx, Hc, L.w_fc2, L.b_fc2, Xc, n, d, mlp))) return rc;
    return D2R_OK;
}

// patches (bf16 [n*(T-1) padded to 128][Kp_pad]) -> logits/embeds.  Workspaces:
//  clipws[0] patch_out f32, [1] X f32, [2] Xn bf16, [3] QKV bf16, [4] AO bf16, [5] H bf16
// can d2r_clip_forward take a ClipL0Reuse for this model under the context's options?
bool d2r_clip_l0_supported(const d2r_ctx *ctx, const d2r_clip *clip)
{
    const d2r_clip_desc &D = clip->desc;
    const uint32_t g = D.image_size / D.patch_size;
    return ctx->ln_fold == 4 && D.num_layers >= 2 && g * g == clip->T - 1 && g * g <= 1024;
}

// The background's own layer-0 rows for ClipL0Reuse: patch embedding -> + position, pre-LayerNorm -> bf16 hi + lo bytes and LayerNorm
// pairs -> layer-0 q / k / v, of ONE image (the background frame's patches), with the kernels the batched forward uses (a row's
// result does not depend on the batch it is computed in).  Buffers live in ctx->bg_l0; rows padded to a multiple of 256.
int d2r_clip_layer0_background(d2r_ctx *ctx, const d2r_clip *clip, const uint16_t *bg_patches_dev, ClipL0Reuse *out)
{
    const d2r_clip_desc &D = clip->desc;
    const uint32_t d = D.hidden_size, T = clip->T, R = round_up(T, BM);
    if (T > 1025 || D.num_layers < 1) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "layer-0 reuse needs at most 1024 patches per image");
    const size_t b_hi = (size_t)R * d * 2, b_lo = (size_t)R * d, b_ab = (size_t)R * 8, b_qkv = (size_t)R * 3 * d * 2, b_po = (size_t)R * d * 4;
    int rc;
    if ((rc = d2r_reserve(ctx, ctx->bg_l0, b_hi + b_lo + b_ab + b_qkv + b_po + 1024))) return rc;
    uint8_t *base = (uint8_t *)ctx->bg_l0.p;
    uint16_t *hi = (uint16_t *)base;
    uint8_t *lo = base + b_hi;
    float2 *ab = (float2 *)(lo + b_lo);
    uint16_t *qkv = (uint16_t *)((uint8_t *)ab + b_ab);
    float *po = (float *)((uint8_t *)qkv + b_qkv);
    if ((rc = launch_gemm<EPI_F32>(ctx, bg_patches_dev, clip->w.w_patch, nullptr, po, T - 1, d, clip->Kp_pad))) return rc;
    hipLaunchKernelGGL(k_embed_ln, dim3((T + 3) / 4), dim3(256), 0, ctx->stream, po, clip->w.cls, clip->w.pos, clip->w.pre_w, clip->w.pre_b,
                       (float *)nullptr, T, T, d, hi, ab, (uint16_t *)lo, R, 1);
    const ClipWeights::Layer &L = clip->layers[0];
    EpiAux ln{};
    ln.hm_rows = R;
    ln.a_rs = 64;
    ln.a_ks = 64 * R;
    ln.ab = ab;
    ln.cs = L.cs_qkv;
    if ((rc = launch_gemm<EPI_LN_BIAS_BF16>(ctx, hi, L.wf_qkv, L.bf_qkv, qkv, T, 3 * d, d, ln))) return rc;
    D2R_HIP(ctx, hipGetLastError());
    out->bg_hi = hi;
    out->bg_lo = lo;
    out->bg_ab = ab;
    out->bg_qkv = qkv;
    out->bg_rows = R;
    return D2R_OK;
}

// "fp8 blocks": the e4m3 copies of the Linear weights (from the bf16 operand copies: what the bf16 path multiplies with), on first use
static int ensure_f8_weights(d2r_ctx *ctx, d2r_clip *clip)
{
    std::lock_guard<std::mutex> lock(clip->f8_mu);
    if (!clip->f8.empty()) return D2R_OK;
    const d2r_clip_desc &D = clip->desc;
    const size_t d = D.hidden_size, mlp = D.mlp_size, nl = D.num_layers;
    const size_t sizes[4] = {3 * d * d, d * d, mlp * d, d * mlp};
    const size_t per_layer = sizes[0] + sizes[1] + sizes[2] + sizes[3];
    uint8_t *w8 = nullptr;
    uint32_t *amax = nullptr;
    if (hipMalloc(&w8, per_layer * nl) != hipSuccess || hipMalloc(&amax, nl * 4 * 4) != hipSuccess) {
        if (w8) (void)hipFree(w8);
        return d2r_fail(ctx, D2R_ERR_MEMORY, "hipMalloc failed for the fp8 weight copies");
    }
    clip->allocs.push_back(w8);
    clip->allocs.push_back(amax);
    D2R_HIP(ctx, hipMemsetAsync(amax, 0, nl * 16, ctx->stream));
    std::vector<d2r_clip::F8Layer> f8(nl);
    for (size_t l = 0; l < nl; l++) {
        const ClipWeights::Layer &L = clip->layers[l];
        const uint16_t *src[4] = {L.w_qkv, L.w_o, L.w_fc1, L.w_fc2};
        uint8_t *dst = w8 + l * per_layer;
        const uint8_t *out[4];
        for (int i = 0; i < 4; i++) {
            hipLaunchKernelGGL(k_amax_bf16, dim3(1024), dim3(256), 0, ctx->stream, src[i], sizes[i], amax + l * 4 + i);
            hipLaunchKernelGGL(k_quant_w8, dim3(1024), dim3(256), 0, ctx->stream, src[i], sizes[i], (const uint32_t *)(amax + l * 4 + i), dst);
            out[i] = dst;
            dst += sizes[i];
        }
        f8[l].w_qkv = out[0]; f8[l].w_o = out[1]; f8[l].w_fc1 = out[2]; f8[l].w_fc2 = out[3];
    }
    std::vector<uint32_t> bits(nl * 4);
    D2R_HIP(ctx, hipMemcpyAsync(bits.data(), amax, nl * 16, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto scale_of = [](uint32_t b) {             // q8_scale_byte on the host
        const int e = (int)((b + 0x200000u) >> 23);
        const int byte = std::min(std::max(e - 8, 1), 253);
        return std::ldexp(1.0f, byte - 127);
    };
    for (size_t l = 0; l < nl; l++) {
        f8[l].s_qkv = scale_of(bits[l * 4 + 0]); f8[l].s_o = scale_of(bits[l * 4 + 1]);
        f8[l].s_fc1 = scale_of(bits[l * 4 + 2]); f8[l].s_fc2 = scale_of(bits[l * 4 + 3]);
    }
    clip->f8 = std::move(f8);
    return D2R_OK;
}

int d2r_clip_forward(d2r_ctx *ctx, const d2r_clip *clip, const uint16_t *patches_dev, uint32_t n,
                     const float *text_dev, uint32_t C, float logit_scale, float *logits_dev, float *embeds_dev,
                     const ClipL0Reuse *reuse)
{
    const d2r_clip_desc &D = clip->desc;
    const uint32_t d = D.hidden_size, T = clip->T, mlp = D.mlp_size;
    const uint32_t rows = n * T, rows_pad = round_up(rows, BM);
    const uint32_t prow = n * (T - 1);
    int rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[0], (size_t)round_up(prow, BM) * d * 4))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[1], (size_t)rows_pad * d * 4))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[2], (size_t)rows_pad * d * 2))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[3], (size_t)rows_pad * 3 * d * 2))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[4], (size_t)rows_pad * d * 2))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[5], (size_t)rows_pad * mlp * 2))) return rc;
    float *patch_out = (float *)ctx->clipws[0].p, *X = (float *)ctx->clipws[1].p;
    uint16_t *Xn = (uint16_t *)ctx->clipws[2].p, *QKV = (uint16_t *)ctx->clipws[3].p;
    uint16_t *AO = (uint16_t *)ctx->clipws[4].p, *H = (uint16_t *)ctx->clipws[5].p;

    const int fold = (int)ctx->ln_fold;
    // layer-0 reuse (see k_touch_list): needs the hi + lo-byte residual, square images cut into whole patches, a second layer
    const bool l0 = reuse && reuse->rects;
    if (l0 && (!d2r_clip_l0_supported(ctx, clip) || T > reuse->bg_rows))
        return d2r_fail(ctx, D2R_ERR_INVALID, "d2r_clip_forward: layer-0 reuse requested for a model / option set that does not support it");
    uint32_t *l0_cnt = nullptr, *l0_list = nullptr;
    const uint32_t cap_pad = round_up(prow, BM);
    if (l0) {
        const PrepCache *pc = nullptr;
        if ((rc = prep_tables(ctx, (d2r_clip *)clip, reuse->w, reuse->h, 1, &pc))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->l0_misc, 256 + (size_t)prow * 4 + (size_t)cap_pad * 8))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->l0_a1, (size_t)cap_pad * clip->Kp_pad * 2))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->l0_q2, (size_t)cap_pad * 3 * d * 2))) return rc;
        l0_cnt = (uint32_t *)ctx->l0_misc.p;
        l0_list = l0_cnt + 64;
        D2R_HIP(ctx, hipMemsetAsync(l0_cnt, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_touch_list, dim3(n), dim3(256), 0, ctx->stream, (const int4 *)reuse->rects, reuse->w, pc->R, D.image_size, D.patch_size, l0_cnt, l0_list);
        hipLaunchKernelGGL(k_gather_rows, dim3(2048), dim3(256), 0, ctx->stream, patches_dev, clip->Kp_pad, l0_list, l0_cnt, (uint16_t *)ctx->l0_a1.p);
        if (reuse->touched_out) D2R_HIP(ctx, hipMemcpyAsync(reuse->touched_out, l0_cnt, 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (getenv("D2R_L0_DEBUG")) {                          // development aid: how many tokens the chunk touched
            uint32_t c = 0;
            (void)hipMemcpyAsync(&c, l0_cnt, 4, hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            fprintf(stderr, "[d2r l0_reuse] %u of %u patch tokens touched (%.1f %%)\n", c, prow, 100.0 * c / prow);
        }
        EpiAux a1{};
        a1.m_dev = l0_cnt;
        if ((rc = launch_gemm_cfg<EPI_F32, 4, 2, 2, 3>(ctx, (const uint16_t *)ctx->l0_a1.p, clip->w.w_patch, nullptr, patch_out, prow, d, clip->Kp_pad, a1))) return rc;
    } else if ((rc = launch_gemm<EPI_F32>(ctx, patches_dev, clip->w.w_patch, nullptr, patch_out, prow, d, clip->Kp_pad)))
        return rc;
    // the last block runs on the class-token rows only (last_block_cls); needs its small workspaces to fit what exists
    const bool cls_last = ctx->cls_last && D.num_layers >= 1 && T <= 1024 && round_up(n, BM) <= round_up(prow, BM);
    // bf16 activations between kernels are tile-major ([cols/64][rows_pad][64], see EpiAux): QKV, AO, H and the bf16
    // residual arrays; only k_layernorm's output (fold 0, text tower) and the patch matrix are row-major
    EpiAux out_tm{}, a_tm{};
    out_tm.hm_rows = rows_pad;                    // GEMM whose A is row-major and whose bf16 output is tile-major
    a_tm.a_rs = 64;                               // GEMM whose A is tile-major (fp32 row-major output)
    a_tm.a_ks = (uint32_t)64 * rows_pad;
    if ((uint64_t)64 * rows_pad >= (1ull << 32)) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "activation plane too large for 32-bit indexing");
    if (fold == 0) {
        // separate LayerNorm kernels, fp32 residual stream (also what the text tower runs)
        hipLaunchKernelGGL(k_embed_ln, dim3((rows + 3) / 4), dim3(256), 0, ctx->stream, patch_out, clip->w.cls,
                           clip->w.pos, clip->w.pre_w, clip->w.pre_b, X, rows, T, d, (uint16_t *)nullptr, (float2 *)nullptr, (uint16_t *)nullptr, rows_pad);
        for (uint32_t l = 0; l < D.num_layers; l++) {
            const ClipWeights::Layer &L = clip->layers[l];
            hipLaunchKernelGGL(k_layernorm, dim3((rows + 3) / 4), dim3(256), 0, ctx->stream, X, L.ln1_w, L.ln1_b, Xn,
                               rows, d);
            if ((rc = launch_gemm<EPI_BIAS_BF16>(ctx, Xn, L.w_qkv, L.b_qkv, QKV, rows, 3 * d, d, out_tm))) return rc;
            if (cls_last && l + 1 == D.num_layers) {
                if ((rc = last_block_cls(ctx, D, L, n, T, rows_pad, QKV, nullptr, X, nullptr, nullptr, patch_out, AO, Xn, H))) return rc;
                break;
            }
            launch_attention_vision(ctx, QKV, AO, T, d, rows_pad, D.num_heads, n);
            if ((rc = launch_gemm<EPI_BIAS_RESID_F32>(ctx, AO, L.w_o, L.b_o, X, rows, d, d, a_tm))) return rc;
            hipLaunchKernelGGL(k_layernorm, dim3((rows + 3) / 4), dim3(256), 0, ctx->stream, X, L.ln2_w, L.ln2_b, Xn,
                               rows, d);
            if ((rc = launch_gemm<EPI_BIAS_GELU_BF16>(ctx, Xn, L.w_fc1, L.b_fc1, H, rows, mlp, d, out_tm))) return rc;
            if ((rc = launch_gemm<EPI_BIAS_RESID_F32>(ctx, H, L.w_fc2, L.b_fc2, X, rows, d, mlp, a_tm))) return rc;
        }
        hipLaunchKernelGGL(k_head, dim3((n + HEAD_NI - 1) / HEAD_NI), dim3(HEAD_THREADS), 0, ctx->stream, cls_last ? patch_out : X, (const uint16_t *)nullptr, (const uint16_t *)nullptr, rows_pad,
                           (const uint32_t *)nullptr, cls_last ? 1u : T, d,
                           clip->w.post_w, clip->w.post_b, clip->w.proj, D.proj_dim, text_dev, C, logit_scale, logits_dev, embeds_dev, n);
        D2R_HIP(ctx, hipGetLastError());
        return D2R_OK;
    }
    // LayerNorm folded into the GEMMs (EPI_LN_* / EPI_RESID_STATS_*): Xn holds the RAW residual rows in bf16 (the
    // A operand of QKV and fc1), the residual GEMMs emit per-row partial sums, k_rowstats turns them into
    // (rstd, -rstd*mean).
    const uint32_t np = d / 64;
    if ((rc = d2r_reserve(ctx, ctx->clipws[7], (size_t)rows_pad * np * 8 + (size_t)rows_pad * 8))) return rc;
    float2 *part = (float2 *)ctx->clipws[7].p, *AB = part + (size_t)rows_pad * np;
    // fold 1: residual = hi (Xn, the operand copy) + lo (bf16 array in X's workspace); fold 2: hi only; fold 3: fp32 X + hi;
    // fold 4: hi + one lo BYTE per element (EPI_RESID_STATS_SPLIT8, byte array in X's workspace)
    const bool xf32 = fold == 3, split8 = fold == 4, split = fold == 1 || split8;
    const int lo8 = split8 ? 1 : 0;
    uint16_t *Xlo = split ? (uint16_t *)X : (uint16_t *)nullptr;
    if (l0) {
        // every row <- the background's row of its token, then the touched tokens' own rows over them
        hipLaunchKernelGGL(k_bcast_x0, dim3((rows + 3) / 4), dim3(256), 0, ctx->stream, reuse->bg_hi, reuse->bg_lo, reuse->bg_ab, reuse->bg_rows, Xn,
                           (uint8_t *)Xlo, AB, rows, rows_pad, T, d);
        hipLaunchKernelGGL(k_embed_ln, dim3(std::min<uint32_t>((prow + 3) / 4, 4096u)), dim3(256), 0, ctx->stream, patch_out, clip->w.cls, clip->w.pos, clip->w.pre_w,
                           clip->w.pre_b, (float *)nullptr, rows, T, d, Xn, AB, Xlo, rows_pad, lo8, (const uint32_t *)l0_list, (const uint32_t *)l0_cnt);
    } else
    hipLaunchKernelGGL(k_embed_ln, dim3((rows + 3) / 4), dim3(256), 0, ctx->stream, patch_out, clip->w.cls, clip->w.pos,
                       clip->w.pre_w, clip->w.pre_b, xf32 ? X : (float *)nullptr, rows, T, d, Xn, AB, Xlo, rows_pad, lo8);
    EpiAux ln = out_tm, st = a_tm;                // LN-folded GEMMs: A = tile-major residual copy, output tile-major
    ln.a_rs = a_tm.a_rs;
    ln.a_ks = a_tm.a_ks;
    ln.ab = AB;
    st.hm_rows = rows_pad;                        // residual GEMMs: A tile-major (AO / H), bf16 residual arrays tile-major
    st.xb = Xn;
    st.part = part;
    st.xlo = Xlo;
    const float inv_d = 1.0f / (float)d;
    // "fp8 blocks" (option vit_fp8, hi + lo-byte residual only): every block but the first when its rows are reused from the background
    // (that reuse is exact only in the arithmetic the background's rows were computed in) and the class-token-only last one
    uint32_t f8_first = D.num_layers, f8_end = D.num_layers;
    uint8_t *A8 = nullptr, *SA8 = nullptr, *AOS8 = nullptr, *H8 = nullptr, *SH8 = nullptr;
    uint32_t q8_grid = 0;
    if (ctx->vit_fp8 && split8 && d % 256 == 0 && mlp % 256 == 0 && mlp >= 2 * d) {
        f8_first = l0 ? 1u : 0u;
        f8_end = cls_last ? D.num_layers - 1 : D.num_layers;
        if (f8_first < f8_end) {
            if ((rc = ensure_f8_weights(ctx, (d2r_clip *)clip))) return rc;
            // AO's workspace (2 d bytes per row) holds the attention output's bytes + scales; H's (2 mlp per row) fc1's bytes + scales, then
            // the LayerNorm operand's bytes + scales
            const size_t rp = rows_pad;
            AOS8 = (uint8_t *)AO + rp * d;
            H8 = (uint8_t *)H;
            SH8 = H8 + rp * mlp;
            A8 = SH8 + rp * (mlp / 64);
            SA8 = A8 + rp * d;
            q8_grid = (uint32_t)ctx->n_cu * 8u;
        }
    }
    for (uint32_t l = 0; l < D.num_layers; l++) {
        const ClipWeights::Layer &L = clip->layers[l];
        ln.cs = L.cs_qkv;
        if (cls_last && l + 1 == D.num_layers) {
            // k and v of every token (planes H .. 3H-1 of QKV: weight rows d .. 3d-1), q of the class tokens only: their
            // operand rows and LayerNorm pairs gathered into H's workspace (free until the block's fc1), a [n][d] product
            EpiAux kv = ln;
            kv.cs = L.cs_qkv + d;
            if ((rc = launch_gemm<EPI_LN_BIAS_BF16>(ctx, Xn, L.wf_qkv + (size_t)d * d, L.bf_qkv + d, QKV + (size_t)d * rows_pad, rows, 2 * d, d, kv)))
                return rc;
            const uint32_t n_pad = round_up(n, BM);
            uint16_t *Xq = H, *Qc = H + (size_t)n_pad * d;
            float2 *ABq = part;                       // the partial sums are consumed: their workspace is free
            hipLaunchKernelGGL(k_gather_cls_operand, dim3((n * (d / 8) + 255) / 256), dim3(256), 0, ctx->stream, Xn, AB, rows_pad, T, d, n, Xq, ABq);
            EpiAux lq{};
            lq.cs = L.cs_qkv;
            lq.ab = ABq;
            if ((rc = launch_gemm<EPI_LN_BIAS_BF16>(ctx, Xq, L.wf_qkv, L.bf_qkv, Qc, n, d, d, lq))) return rc;
            // (the gather reads the residual rows out of Xn / Xlo before Xn's first rows are reused for the LayerNorm output)
            if ((rc = last_block_cls(ctx, D, L, n, T, rows_pad, QKV, Qc, X, xf32 ? nullptr : Xn, Xlo, patch_out, AO, Xn, H, lo8))) return rc;
            break;
        }
        if (f8_first <= l && l < f8_end) {
            // "fp8 blocks": LayerNorm -> e4m3 operand; the four products on the fp8 MFMA; q / k / v, the residual stream and its statistics as ever
            const d2r_clip::F8Layer &F = clip->f8[l];
            hipLaunchKernelGGL(k_ln_q8, dim3(q8_grid), dim3(256), 0, ctx->stream, Xn, AB, L.ln1_w, L.ln1_b, rows_pad, d, A8, SA8);
            if ((rc = launch_gemm_f8<EPI_F8_BIAS_BF16>(ctx, A8, SA8, F.w_qkv, F.s_qkv, L.b_qkv, QKV, rows, 3 * d, d, out_tm))) return rc;
            launch_attention_vision(ctx, QKV, AO, T, d, rows_pad, D.num_heads, n, AOS8);
            if ((rc = launch_gemm_f8<EPI_F8_RESID_STATS_SPLIT8>(ctx, (const uint8_t *)AO, AOS8, F.w_o, F.s_o, L.b_o, Xn, rows, d, d, st))) return rc;
            hipLaunchKernelGGL(k_rowstats, dim3((rows_pad + 255) / 256), dim3(256), 0, ctx->stream, part, np, rows_pad, inv_d, AB);
            hipLaunchKernelGGL(k_ln_q8, dim3(q8_grid), dim3(256), 0, ctx->stream, Xn, AB, L.ln2_w, L.ln2_b, rows_pad, d, A8, SA8);
            EpiAux hq = out_tm;
            hq.q8 = H8;
            hq.q8_scale = SH8;
            if ((rc = launch_gemm_f8<EPI_F8_BIAS_GELU_Q8>(ctx, A8, SA8, F.w_fc1, F.s_fc1, L.b_fc1, nullptr, rows, mlp, d, hq))) return rc;
            if ((rc = launch_gemm_f8<EPI_F8_RESID_STATS_SPLIT8>(ctx, H8, SH8, F.w_fc2, F.s_fc2, L.b_fc2, Xn, rows, d, mlp, st))) return rc;
            if (l + 1 < D.num_layers)
                hipLaunchKernelGGL(k_rowstats, dim3((rows_pad + 255) / 256), dim3(256), 0, ctx->stream, part, np, rows_pad, inv_d, AB);
            continue;
        }
        if (l0 && l == 0) {
            // layer 0: the background's q / k / v rows everywhere, then the touched tokens' own (a product over the compact list)
            const uint32_t planes = 3 * d / 64;
            const size_t pieces = (size_t)planes * rows * 8;
            hipLaunchKernelGGL(k_bcast_qkv, dim3((uint32_t)((pieces + 255) / 256)), dim3(256), 0, ctx->stream, reuse->bg_qkv, reuse->bg_rows, QKV, rows, rows_pad, T, planes);
            float2 *AB2 = (float2 *)((uint8_t *)ctx->l0_misc.p + 256 + (size_t)prow * 4);
            hipLaunchKernelGGL(k_gather_operand_rows, dim3(2048), dim3(256), 0, ctx->stream, Xn, AB, rows_pad, T, (const uint32_t *)l0_list, (const uint32_t *)l0_cnt,
                               AO, AB2, cap_pad, d);
            EpiAux l2 = ln;
            l2.a_ks = 64 * cap_pad;
            l2.hm_rows = cap_pad;
            l2.ab = AB2;
            l2.m_dev = l0_cnt;
            if ((rc = launch_gemm_cfg<EPI_LN_BIAS_BF16, 4, 2, 2, 3>(ctx, AO, L.wf_qkv, L.bf_qkv, ctx->l0_q2.p, prow, 3 * d, d, l2))) return rc;
            hipLaunchKernelGGL(k_scatter_qkv, dim3(8192), dim3(256), 0, ctx->stream, (const uint16_t *)ctx->l0_q2.p, cap_pad, (const uint32_t *)l0_list,
                               (const uint32_t *)l0_cnt, QKV, rows_pad, T, planes);
        } else {
            const size_t tq = ctx->timing_begin(D2R_T_VIT_QKV);
            if ((rc = launch_gemm<EPI_LN_BIAS_BF16>(ctx, Xn, L.wf_qkv, L.bf_qkv, QKV, rows, 3 * d, d, ln))) return rc;
            ctx->timing_end(tq);
        }
        size_t tk = ctx->timing_begin(D2R_T_VIT_ATTN);
        launch_attention_vision(ctx, QKV, AO, T, d, rows_pad, D.num_heads, n);
        ctx->timing_end(tk);
        tk = ctx->timing_begin(D2R_T_VIT_OUT);
        if (xf32) rc = launch_gemm<EPI_RESID_STATS_F32X>(ctx, AO, L.w_o, L.b_o, X, rows, d, d, st);
        else if (split8) rc = launch_gemm<EPI_RESID_STATS_SPLIT8>(ctx, AO, L.w_o, L.b_o, Xn, rows, d, d, st);
        else if (split) rc = launch_gemm<EPI_RESID_STATS_SPLIT>(ctx, AO, L.w_o, L.b_o, Xn, rows, d, d, st);
        else rc = launch_gemm<EPI_RESID_STATS_BF16>(ctx, AO, L.w_o, L.b_o, Xn, rows, d, d, st);
        if (rc) return rc;
        ctx->timing_end(tk);
        hipLaunchKernelGGL(k_rowstats, dim3((rows_pad + 255) / 256), dim3(256), 0, ctx->stream, part, np, rows_pad, inv_d, AB);
        ln.cs = L.cs_fc1;
        tk = ctx->timing_begin(D2R_T_VIT_FC1);
        if ((rc = launch_gemm<EPI_LN_BIAS_GELU_BF16>(ctx, Xn, L.wf_fc1, L.bf_fc1, H, rows, mlp, d, ln))) return rc;
        ctx->timing_end(tk);
        tk = ctx->timing_begin(D2R_T_VIT_FC2);
        if (xf32) rc = launch_gemm<EPI_RESID_STATS_F32X>(ctx, H, L.w_fc2, L.b_fc2, X, rows, d, mlp, st);
        else if (split8) rc = launch_gemm<EPI_RESID_STATS_SPLIT8>(ctx, H, L.w_fc2, L.b_fc2, Xn, rows, d, mlp, st);
        else if (split) rc = launch_gemm<EPI_RESID_STATS_SPLIT>(ctx, H, L.w_fc2, L.b_fc2, Xn, rows, d, mlp, st);
        else rc = launch_gemm<EPI_RESID_STATS_BF16>(ctx, H, L.w_fc2, L.b_fc2, Xn, rows, d, mlp, st);
        if (rc) return rc;
        ctx->timing_end(tk);
        if (l + 1 < D.num_layers)
            hipLaunchKernelGGL(k_rowstats, dim3((rows_pad + 255) / 256), dim3(256), 0, ctx->stream, part, np, rows_pad, inv_d, AB);
    }
    if (cls_last)
        hipLaunchKernelGGL(k_head, dim3((n + HEAD_NI - 1) / HEAD_NI), dim3(HEAD_THREADS), 0, ctx->stream, patch_out, (const uint16_t *)nullptr, (const uint16_t *)nullptr, rows_pad,
                           (const uint32_t *)nullptr, 1u, d, clip->w.post_w, clip->w.post_b, clip->w.proj, D.proj_dim, text_dev, C, logit_scale, logits_dev,
                           embeds_dev, n);
    else
    hipLaunchKernelGGL(k_head, dim3((n + HEAD_NI - 1) / HEAD_NI), dim3(HEAD_THREADS), 0, ctx->stream, X, xf32 ? (const uint16_t *)nullptr : (const uint16_t *)Xn,
                       (const uint16_t *)Xlo, rows_pad, (const uint32_t *)nullptr, T, d, clip->w.post_w, clip->w.post_b, clip->w.proj, D.proj_dim, text_dev, C,
                       logit_scale, logits_dev, embeds_dev, n, lo8);
    D2R_HIP(ctx, hipGetLastError());
    return D2R_OK;
}

int d2r_launch_patchify(d2r_ctx *ctx, const d2r_clip *clip, const float *pv_dev, uint32_t n, uint16_t *patches_dev)
{
    size_t total = (size_t)n * 3 * clip->desc.image_size * clip->desc.image_size;
    hipLaunchKernelGGL(k_patchify, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, pv_dev, n,
                       clip->desc.image_size, clip->desc.patch_size, patches_dev, clip->Kp_pad);
    D2R_HIP(ctx, hipGetLastError());
    return D2R_OK;
}

uint32_t d2r_clip_tokens(const d2r_clip *clip) { return clip->T; }

// largest image count whose GEMM outputs (rows padded to BM x the widest layer) keep 32-bit indices
uint32_t d2r_clip_max_images(const d2r_clip *clip)
{
    const uint64_t widest = std::max<uint64_t>(3ull * clip->desc.hidden_size, clip->desc.mlp_size);
    const uint64_t rows = (0xffffffffull / widest) / BM * BM;
    return (uint32_t)std::max<uint64_t>(1, rows / clip->T);
}

size_t d2r_clip_patch_bytes(const d2r_clip *clip, uint32_t n)
{
    return (size_t)round_up(n * (clip->T - 1), BM) * clip->Kp_pad * 2;
}

// ------------------------------------------------------------ create/destroy

extern "C" int d2r_clip_create(d2r_ctx *ctx, const d2r_clip_desc *desc, const float *weights, size_t n_floats,
                               d2r_clip **out)
{
    if (!ctx || !desc || !weights || !out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    const uint32_t d = desc->hidden_size, P = desc->patch_size, S = desc->image_size, mlp = desc->mlp_size;
    if (d % 128 || mlp % 128 || d > 1024 || desc->proj_dim > 1024 || d / desc->num_heads != 64 || S % P)
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "unsupported CLIP geometry (need head_dim 64, d,mlp % 128 == 0, d <= 1024)");
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    d2r_clip *c = new d2r_clip();
    c->ctx = ctx;
    c->desc = *desc;
    c->T = (S / P) * (S / P) + 1;
    c->Kp = 3 * P * P;
    c->Kp_pad = round_up(c->Kp, BK);
    const uint32_t T = c->T, Dp = desc->proj_dim;
    size_t expect = (size_t)d * c->Kp + d + (size_t)T * d + 2 * d +
                    (size_t)desc->num_layers * (4 * (size_t)d + 4 * ((size_t)d * d + d) + (size_t)mlp * d + mlp + (size_t)d * mlp + d) +
                    2 * d + (size_t)Dp * d;
    if (n_floats != expect) {
        delete c;
        return d2r_fail(ctx, D2R_ERR_INVALID, "weight blob size does not match the descriptor");
    }
    // stage the whole blob on the device once, then carve fp32 views and bf16 copies
    float *blob = nullptr;
    if (hipMalloc(&blob, n_floats * 4) != hipSuccess) {
        delete c;
        return d2r_fail(ctx, D2R_ERR_MEMORY, "hipMalloc failed for CLIP weights");
    }
    c->allocs.push_back(blob);
    if (hipMemcpy(blob, weights, n_floats * 4, hipMemcpyHostToDevice) != hipSuccess) {
        d2r_clip_destroy(c);
        return d2r_fail(ctx, D2R_ERR_DEVICE, "weight upload failed");
    }
    size_t off = 0;
    auto f32 = [&](size_t n) { const float *p = blob + off; off += n; return p; };
    bool ok = true;
    auto bf16 = [&](const float *src, uint32_t rows, uint32_t K, uint32_t K_pad) -> const uint16_t * {
        uint16_t *p = nullptr;
        if (hipMalloc(&p, (size_t)rows * K_pad * 2) != hipSuccess) { ok = false; return nullptr; }
        c->allocs.push_back(p);
        size_t tot = (size_t)rows * K_pad;
        hipLaunchKernelGGL(k_convert_bf16, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, ctx->stream, src, p, rows, K, K_pad);
        return p;
    };
    c->w.w_patch = bf16(f32((size_t)d * c->Kp), d, c->Kp, c->Kp_pad);
    c->w.cls = f32(d);
    c->w.pos = f32((size_t)T * d);
    c->w.pre_w = f32(d);
    c->w.pre_b = f32(d);
    c->layers.resize(desc->num_layers);
    for (uint32_t l = 0; l < desc->num_layers && ok; l++) {
        ClipWeights::Layer &L = c->layers[l];
        L.ln1_w = f32(d);
        L.ln1_b = f32(d);
        // q, k, v are consecutive [d][d] + [d] pairs: fuse into one [3d][d] operand and one [3d] bias
        uint16_t *wqkv = nullptr;
        float *bqkv = nullptr;
        if (hipMalloc(&wqkv, (size_t)3 * d * d * 2) != hipSuccess || hipMalloc(&bqkv, (size_t)3 * d * 4) != hipSuccess) { ok = false; break; }
        c->allocs.push_back(wqkv);
        c->allocs.push_back(bqkv);
        // the q rows (weights, bias) carry the attention scale head_dim^-0.5 * log2(e): k_attention_s / k_attention_cls work in the exp2
        // domain on scores that need no further scaling (the scaled weight is rounded to bf16 once, like the unscaled one was)
        for (int j = 0; j < 3; j++) {
            const float *wj = f32((size_t)d * d), *bj = f32(d);
            size_t tot = (size_t)d * d;
            hipLaunchKernelGGL(k_convert_bf16, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, ctx->stream, wj,
                               wqkv + (size_t)j * d * d, d, d, d, j == 0 ? ATTN_Q_SCALE : 1.0f);
            hipMemcpyAsync(bqkv + (size_t)j * d, bj, (size_t)d * 4, hipMemcpyDeviceToDevice, ctx->stream);
            if (j == 0) hipLaunchKernelGGL(k_scale_f32, dim3((d + 255) / 256), dim3(256), 0, ctx->stream, bqkv, d, ATTN_Q_SCALE);
        }
        L.w_qkv = wqkv;
        L.b_qkv = bqkv;
        // layer_norm1 folded into q/k/v (EPI_LN_BIAS_BF16): the three [d][d] + [d] pairs sit 3 blocks back in the blob
        uint16_t *wfq = nullptr, *wf1 = nullptr;
        float *csq = nullptr, *bfq = nullptr, *cs1 = nullptr, *bf1 = nullptr;
        if (hipMalloc(&wfq, (size_t)3 * d * d * 2) != hipSuccess || hipMalloc(&csq, (size_t)3 * d * 4) != hipSuccess ||
            hipMalloc(&bfq, (size_t)3 * d * 4) != hipSuccess || hipMalloc(&wf1, (size_t)mlp * d * 2) != hipSuccess ||
            hipMalloc(&cs1, (size_t)mlp * 4) != hipSuccess || hipMalloc(&bf1, (size_t)mlp * 4) != hipSuccess) { ok = false; break; }
        for (void *pp : {(void *)wfq, (void *)csq, (void *)bfq, (void *)wf1, (void *)cs1, (void *)bf1}) c->allocs.push_back(pp);
        {
            const float *q0 = blob + off - 3 * ((size_t)d * d + d);
            for (int j = 0; j < 3; j++) {
                const float *wj = q0 + (size_t)j * ((size_t)d * d + d), *bj = wj + (size_t)d * d;
                hipLaunchKernelGGL(k_fold_ln_weight, dim3((d + 3) / 4), dim3(256), 0, ctx->stream, wj, L.ln1_w, L.ln1_b, bj,
                                   wfq + (size_t)j * d * d, csq + (size_t)j * d, bfq + (size_t)j * d, d, d, j == 0 ? ATTN_Q_SCALE : 1.0f);
            }
        }
        L.wf_qkv = wfq; L.cs_qkv = csq; L.bf_qkv = bfq;
        L.w_o = bf16(f32((size_t)d * d), d, d, d);
        L.b_o = f32(d);
        L.ln2_w = f32(d);
        L.ln2_b = f32(d);
        const float *w1 = f32((size_t)mlp * d);
        L.w_fc1 = bf16(w1, mlp, d, d);
        L.b_fc1 = f32(mlp);
        hipLaunchKernelGGL(k_fold_ln_weight, dim3((mlp + 3) / 4), dim3(256), 0, ctx->stream, w1, L.ln2_w, L.ln2_b, L.b_fc1, wf1, cs1, bf1, mlp, d);
        L.wf_fc1 = wf1; L.cs_fc1 = cs1; L.bf_fc1 = bf1;
        L.w_fc2 = bf16(f32((size_t)d * mlp), d, mlp, mlp);
        L.b_fc2 = f32(d);
    }
    c->w.post_w = f32(d);
    c->w.post_b = f32(d);
    c->w.proj = f32((size_t)Dp * d);
    if (!ok || hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        d2r_clip_destroy(c);
        return d2r_fail(ctx, D2R_ERR_DEVICE, "CLIP weight conversion failed");
    }
    *out = c;
    return D2R_OK;
}

extern "C" void d2r_clip_destroy(d2r_clip *c)
{
    if (!c) return;
    if (c->ctx && c->ctx->bg_patches_for == (const void *)c) c->ctx->bg_patches_for = nullptr;   // a later model may reuse the address
    if (c->ctx && c->ctx->bg_l0_for == (const void *)c) c->ctx->bg_l0_for = nullptr;
    for (void *p : c->allocs) hipFree(p);
    delete c;
}

// ---------------------------------------------------------------- text tower

struct d2r_text {
    d2r_ctx *ctx;
    d2r_text_desc desc;
    std::vector<void *> allocs;
    const float *tok, *pos, *fin_w, *fin_b, *proj;
    std::vector<ClipWeights::Layer> layers;
};

extern "C" int d2r_text_create(d2r_ctx *ctx, const d2r_text_desc *desc, const float *weights, size_t n_floats,
                               d2r_text **out)
{
    if (!ctx || !desc || !weights || !out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    const uint32_t d = desc->hidden_size, mlp = desc->mlp_size, V = desc->vocab_size, Tc = desc->context_length;
    if (d % 128 || mlp % 128 || d > 1024 || desc->proj_dim > 1024 || d / desc->num_heads != 64 || Tc == 0 || Tc > 512)
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "unsupported text tower geometry (need head_dim 64, d,mlp % 128 == 0)");
    (void)hipSetDevice(ctx->device);
    size_t expect = (size_t)V * d + (size_t)Tc * d +
                    (size_t)desc->num_layers * (4 * (size_t)d + 4 * ((size_t)d * d + d) + (size_t)mlp * d + mlp + (size_t)d * mlp + d) +
                    2 * d + (size_t)desc->proj_dim * d;
    if (n_floats != expect) return d2r_fail(ctx, D2R_ERR_INVALID, "text weight blob size does not match the descriptor");
    d2r_text *t = new d2r_text();
    t->ctx = ctx;
    t->desc = *desc;
    float *blob = nullptr;
    if (hipMalloc(&blob, n_floats * 4) != hipSuccess) { delete t; return d2r_fail(ctx, D2R_ERR_MEMORY, "hipMalloc failed for text weights"); }
    t->allocs.push_back(blob);
    if (hipMemcpy(blob, weights, n_floats * 4, hipMemcpyHostToDevice) != hipSuccess) { d2r_text_destroy(t); return d2r_fail(ctx, D2R_ERR_DEVICE, "weight upload failed"); }
    size_t off = 0;
    auto f32 = [&](size_t n) { const float *p = blob + off; off += n; return p; };
    bool ok = true;
    auto bf16 = [&](const float *src, uint32_t rows, uint32_t K) -> uint16_t * {
        uint16_t *p = nullptr;
        if (hipMalloc(&p, (size_t)rows * K * 2) != hipSuccess) { ok = false; return nullptr; }
        t->allocs.push_back(p);
        size_t tot = (size_t)rows * K;
        hipLaunchKernelGGL(k_convert_bf16, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, ctx->stream, src, p, rows, K, K);
        return p;
    };
    t->tok = f32((size_t)V * d);
    t->pos = f32((size_t)Tc * d);
    t->layers.resize(desc->num_layers);
    for (uint32_t l = 0; l < desc->num_layers && ok; l++) {
        ClipWeights::Layer &L = t->layers[l];
        L.ln1_w = f32(d);
        L.ln1_b = f32(d);
        uint16_t *wqkv = nullptr;
        float *bqkv = nullptr;
        if (hipMalloc(&wqkv, (size_t)3 * d * d * 2) != hipSuccess || hipMalloc(&bqkv, (size_t)3 * d * 4) != hipSuccess) { ok = false; break; }
        t->allocs.push_back(wqkv);
        t->allocs.push_back(bqkv);
        for (int j = 0; j < 3; j++) {
            const float *wj = f32((size_t)d * d), *bj = f32(d);
            size_t tot = (size_t)d * d;
            hipLaunchKernelGGL(k_convert_bf16, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, ctx->stream, wj,
                               wqkv + (size_t)j * d * d, d, d, d);
            (void)hipMemcpyAsync(bqkv + (size_t)j * d, bj, (size_t)d * 4, hipMemcpyDeviceToDevice, ctx->stream);
        }
        L.w_qkv = wqkv;
        L.b_qkv = bqkv;
        L.w_o = bf16(f32((size_t)d * d), d, d);
        L.b_o = f32(d);
        L.ln2_w = f32(d);
        L.ln2_b = f32(d);
        L.w_fc1 = bf16(f32((size_t)mlp * d), mlp, d);
        L.b_fc1 = f32(mlp);
        L.w_fc2 = bf16(f32((size_t)d * mlp), d, mlp);
        L.b_fc2 = f32(d);
    }
    t->fin_w = f32(d);
    t->fin_b = f32(d);
    t->proj = f32((size_t)desc->proj_dim * d);
    if (!ok || hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        d2r_text_destroy(t);
        return d2r_fail(ctx, D2R_ERR_DEVICE, "text weight conversion failed");
    }
    *out = t;
    return D2R_OK;
}

extern "C" void d2r_text_destroy(d2r_text *t)
{
    if (!t) return;
    for (void *p : t->allocs) (void)hipFree(p);
    delete t;
}

// CLIPModel text forward (causal pre-LN transformer, EOS pooling, text_projection, L2 norm):
// replaces the text half of reference clip_scoring.py:177-180, run ONCE per task instead of per batch.
extern "C" int d2r_text_encode(d2r_ctx *ctx, const d2r_text *tt, const int32_t *input_ids, uint32_t Cn, uint32_t T,
                               float *embeds_out)
{
    if (!ctx || !tt || !input_ids || !embeds_out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    const d2r_text_desc &D = tt->desc;
    if (Cn == 0 || T == 0 || T > D.context_length) return d2r_fail(ctx, D2R_ERR_INVALID, "bad caption batch shape");
    (void)hipSetDevice(ctx->device);
    const uint32_t d = D.hidden_size, mlp = D.mlp_size, rows = Cn * T, rows_pad = round_up(rows, BM);
    int rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[1], (size_t)rows_pad * d * 4))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[2], (size_t)rows_pad * d * 2))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[3], (size_t)rows_pad * 3 * d * 2))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[4], (size_t)rows_pad * d * 2))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[5], (size_t)rows_pad * mlp * 2))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->pix, (size_t)rows * 4 + (size_t)Cn * 4))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->logits, (size_t)Cn * D.proj_dim * 4))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->text, 16))) return rc;
    float *X = (float *)ctx->clipws[1].p;
    uint16_t *Xn = (uint16_t *)ctx->clipws[2].p, *QKV = (uint16_t *)ctx->clipws[3].p;
    uint16_t *AO = (uint16_t *)ctx->clipws[4].p, *H = (uint16_t *)ctx->clipws[5].p;
    int32_t *ids_dev = (int32_t *)ctx->pix.p;
    uint32_t *pool = (uint32_t *)(ids_dev + rows);
    D2R_HIP(ctx, hipMemcpyAsync(ids_dev, input_ids, (size_t)rows * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_text_embed, dim3(rows), dim3(128), 0, ctx->stream, ids_dev, tt->tok, tt->pos, X, pool, Cn, T, d,
                       D.vocab_size);
    const uint32_t T_pad = round_up(T, 32);
    size_t attn_lds = 0;
    if ((rc = attention_setup(ctx, T_pad, &attn_lds))) return rc;
    EpiAux out_tm{}, a_tm{};                      // tile-major bf16 activations, as in the vision tower
    out_tm.hm_rows = rows_pad;
    a_tm.a_rs = 64;
    a_tm.a_ks = (uint32_t)64 * rows_pad;
    for (uint32_t l = 0; l < D.num_layers; l++) {
        const ClipWeights::Layer &L = tt->layers[l];
        hipLaunchKernelGGL(k_layernorm, dim3((rows + 3) / 4), dim3(256), 0, ctx->stream, X, L.ln1_w, L.ln1_b, Xn, rows, d);
        if ((rc = launch_gemm<EPI_BIAS_BF16>(ctx, Xn, L.w_qkv, L.b_qkv, QKV, rows, 3 * d, d, out_tm))) return rc;
        hipLaunchKernelGGL(k_attention<true>, dim3(D.num_heads, Cn), dim3(ATTN_THREADS), attn_lds, ctx->stream, QKV, AO, T, T_pad, d, rows_pad);
        if ((rc = launch_gemm<EPI_BIAS_RESID_F32>(ctx, AO, L.w_o, L.b_o, X, rows, d, d, a_tm))) return rc;
        hipLaunchKernelGGL(k_layernorm, dim3((rows + 3) / 4), dim3(256), 0, ctx->stream, X, L.ln2_w, L.ln2_b, Xn, rows, d);
        if ((rc = launch_gemm<EPI_BIAS_GELU_BF16>(ctx, Xn, L.w_fc1, L.b_fc1, H, rows, mlp, d, out_tm))) return rc;
        if ((rc = launch_gemm<EPI_BIAS_RESID_F32>(ctx, H, L.w_fc2, L.b_fc2, X, rows, d, mlp, a_tm))) return rc;
    }
    hipLaunchKernelGGL(k_head, dim3((Cn + HEAD_NI - 1) / HEAD_NI), dim3(HEAD_THREADS), 0, ctx->stream, X, (const uint16_t *)nullptr, (const uint16_t *)nullptr, rows_pad, (const uint32_t *)pool, T, d, tt->fin_w,
                       tt->fin_b, tt->proj, D.proj_dim, (const float *)ctx->text.p, 0u, 1.0f, (float *)nullptr,
                       (float *)ctx->logits.p, Cn);
    D2R_HIP(ctx, hipGetLastError());
    D2R_HIP(ctx, hipMemcpyAsync(embeds_out, ctx->logits.p, (size_t)Cn * D.proj_dim * 4, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}

uint32_t d2r_clip_image_size(const d2r_clip *c) { return c->desc.image_size; }
uint32_t d2r_clip_proj_dim(const d2r_clip *c) { return c->desc.proj_dim; }

// ------------------------------------------------------------ parity hook: one fp8 product ("fp8 blocks")
//
// C = q8(A) q8(W)^T * s_w + bias through the product path's own kernels: k_quant_rows_q8 (the activation format), k_amax_bf16 +
// k_quant_w8 on the bf16-rounded weights (the weight format), k_gemm8f with EPI_F8_BIAS_BF16 (kind 0: bf16 result) or
// EPI_F8_BIAS_GELU_Q8 (kind 1: quick_gelu, e4m3 result returned dequantised).  Host arrays in and out; a_q8 / a_scales (optional)
// return the quantised A as [M][K] bytes and [M][K / 64] scale bytes, out_q8 / out_scales (kind 1, optional) the raw result.
extern "C" int d2r_debug_gemm_fp8(d2r_ctx *ctx, const float *A, const float *W, const float *bias, uint32_t M, uint32_t N, uint32_t K, int kind,
                                  float *out, uint8_t *a_q8, uint8_t *a_scales, float *w_scale_out, uint8_t *out_q8, uint8_t *out_scales)
{
    if (!ctx || !A || !W || !bias || !out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    if (!M || N % 256 || K % 256 || kind < 0 || kind > 1) return d2r_fail(ctx, D2R_ERR_INVALID, "d2r_debug_gemm_fp8: N and K must be multiples of 256, kind 0 or 1");
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    const size_t Mp = round_up(M, BM);
    const size_t b_a = (size_t)M * K * 4, b_w = (size_t)N * K * 4, b_wb = (size_t)N * K * 2, b_w8 = (size_t)N * K, b_a8 = Mp * K, b_sa = Mp * (K / 64),
                 b_c = Mp * N * 2, b_sc = Mp * (N / 64), b_bias = (size_t)N * 4;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t total = al(b_a) + al(b_w) + al(b_wb) + al(b_w8) + al(b_a8) + al(b_sa) + al(b_c) + al(b_sc) + al(b_bias) + 256;
    int rc;
    // (its own allocation: the context's workspaces carry invariants — zero padding columns of the patch matrix — that scratch data would break)
    uint8_t *p = nullptr;
    if (hipMalloc(&p, total) != hipSuccess) return d2r_fail(ctx, D2R_ERR_MEMORY, "hipMalloc failed in d2r_debug_gemm_fp8");
    struct Free { void *q; ~Free() { (void)hipFree(q); } } free_p{p};
    auto take = [&](size_t n) { uint8_t *q = p; p += al(n); return q; };
    float *dA = (float *)take(b_a), *dW = (float *)take(b_w);
    uint16_t *dWb = (uint16_t *)take(b_wb);
    uint8_t *dW8 = take(b_w8), *dA8 = take(b_a8), *dSA = take(b_sa), *dC = take(b_c), *dSC = take(b_sc);
    float *dBias = (float *)take(b_bias);
    uint32_t *dAmax = (uint32_t *)take(4);
    D2R_HIP(ctx, hipMemcpyAsync(dA, A, b_a, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(dW, W, b_w, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(dBias, bias, b_bias, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemsetAsync(dAmax, 0, 4, ctx->stream));
    D2R_HIP(ctx, hipMemsetAsync(dSC, 0, b_sc, ctx->stream));
    const size_t nw = (size_t)N * K;
    hipLaunchKernelGGL(k_convert_bf16, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, ctx->stream, dW, dWb, N, K, K);
    hipLaunchKernelGGL(k_amax_bf16, dim3(1024), dim3(256), 0, ctx->stream, dWb, nw, dAmax);
    hipLaunchKernelGGL(k_quant_w8, dim3(1024), dim3(256), 0, ctx->stream, dWb, nw, (const uint32_t *)dAmax, dW8);
    const size_t nt = Mp * (K / 8);
    hipLaunchKernelGGL(k_quant_rows_q8, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, ctx->stream, dA, M, K, (uint32_t)Mp, dA8, dSA);
    uint32_t bits = 0;
    D2R_HIP(ctx, hipMemcpyAsync(&bits, dAmax, 4, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int e = (int)((bits + 0x200000u) >> 23);
    const float ws = std::ldexp(1.0f, std::min(std::max(e - 8, 1), 253) - 127);
    if (w_scale_out) *w_scale_out = ws;
    EpiAux aux{};
    if (kind == 0) {
        if ((rc = launch_gemm_f8<EPI_F8_BIAS_BF16>(ctx, dA8, dSA, dW8, ws, dBias, dC, M, N, K, aux))) return rc;      // row-major bf16
    } else {
        aux.hm_rows = (uint32_t)Mp;
        aux.q8 = dC;
        aux.q8_scale = dSC;
        if ((rc = launch_gemm_f8<EPI_F8_BIAS_GELU_Q8>(ctx, dA8, dSA, dW8, ws, dBias, nullptr, M, N, K, aux))) return rc;
    }
    std::vector<uint8_t> hC(kind == 0 ? b_c : Mp * N), hS(b_sc), hA8(b_a8), hSA(b_sa);
    D2R_HIP(ctx, hipMemcpyAsync(hC.data(), dC, hC.size(), hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(hS.data(), dSC, b_sc, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(hA8.data(), dA8, b_a8, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(hSA.data(), dSA, b_sa, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto scale_at = [&](const std::vector<uint8_t> &S, size_t row, uint32_t g) {
        return S[((((size_t)(g >> 1) * (Mp >> 6) + (row >> 6)) * 32 + (row & 31)) << 2) + (((row >> 5) & 1) << 1) + (g & 1u)];
    };
    auto e4m3 = [](uint8_t b) {
        const int s = b >> 7, ex = (b >> 3) & 15, m = b & 7;
        const float v = ex == 0 ? std::ldexp((float)m, -9) : (ex == 15 && m == 7 ? NAN : std::ldexp(1.0f + m / 8.0f, ex - 7));
        return s ? -v : v;
    };
    for (size_t r = 0; r < M; r++)
        for (uint32_t c = 0; c < N; c++) {
            if (kind == 0) {
                const uint16_t h = ((const uint16_t *)hC.data())[r * N + c];
                union { uint32_t u; float f; } cv;
                cv.u = (uint32_t)h << 16;
                out[r * N + c] = cv.f;
            } else {
                const uint8_t q = hC[(((size_t)(c >> 6) * Mp + r) << 6) + (c & 63)], sb = scale_at(hS, r, c >> 6);
                out[r * N + c] = e4m3(q) * std::ldexp(1.0f, (int)sb - 127);
                if (out_q8) out_q8[r * N + c] = q;
                if (out_scales && (c & 63) == 0) out_scales[r * (N / 64) + (c >> 6)] = sb;
            }
        }
    if (a_q8 || a_scales)
        for (size_t r = 0; r < M; r++)
            for (uint32_t c = 0; c < K; c++) {
                if (a_q8) a_q8[r * K + c] = hA8[(((size_t)(c >> 6) * Mp + r) << 6) + (c & 63)];
                if (a_scales && (c & 63) == 0) a_scales[r * (K / 64) + (c >> 6)] = scale_at(hSA, r, c >> 6);
            }
    return D2R_OK;
}
