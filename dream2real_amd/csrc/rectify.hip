// Sensor-depth background of a render view on the GPU.
//
// reference reconstruction/combined_rendering.py:107-110 with rectify_depth (:166-187) and rectify_mask (:189-209):
// the 1280x720 depth map (fp16 metres, data_loader.py:43,58) and the movable-object mask of the view are centre-cropped
// to a square, resized to the render resolution with cv2.resize(..., interpolation=cv2.INTER_CUBIC), and the depth is
// pushed to "far" (100) wherever the resized mask is 0.  OpenCV's INTER_CUBIC as published: fx = (dx + 0.5) * scale -
// 0.5 with scale = 1 / (dsize / ssize) in double, four taps sx-1 .. sx+2 with replicated borders, Keys kernel with
// A = -0.75 evaluated in float; float images filter horizontally then vertically in float (left-to-right sums), uint8
// images use 11-bit fixed-point coefficients (saturate_cast<short>(c * 2048)), integer sums and a rounding shift by 22.
// The tap tables are built on the host in exactly that arithmetic; a thread produces one output pixel.
#include "d2r_internal.h"

#include <cmath>
#include <vector>

namespace {

struct AxisTable {
    std::vector<int> tap;        // [dst][4] source indices (border replicated)
    std::vector<float> cf;       // [dst][4] float coefficients
    std::vector<int> ci;         // [dst][4] fixed-point coefficients (x 2048, as a short)
};

// interpolateCubic(): every intermediate rounded to float, OpenCV's expression order
void cubic_coeffs(float x, float *c)
{
    const float A = -0.75f;
    const float x1 = x + 1.0f;
    c[0] = ((A * x1 - 5.0f * A) * x1 + 8.0f * A) * x1 - 4.0f * A;
    c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    const float xm = 1.0f - x;
    c[2] = ((A + 2.0f) * xm - (A + 3.0f)) * xm * xm + 1.0f;
    c[3] = 1.0f - c[0] - c[1] - c[2];
}

AxisTable axis_table(uint32_t src, uint32_t dst)
{
    AxisTable t;
    t.tap.resize((size_t)dst * 4);
    t.cf.resize((size_t)dst * 4);
    t.ci.resize((size_t)dst * 4);
    const double inv_scale = (double)dst / (double)src, scale = 1.0 / inv_scale;
    for (uint32_t d = 0; d < dst; d++) {
        float fx = (float)(((double)d + 0.5) * scale - 0.5);
        const int s = (int)std::floor((double)fx);
        fx = fx - (float)s;
        float c[4];
        cubic_coeffs(fx, c);
        for (int k = 0; k < 4; k++) {
            int idx = s - 1 + k;
            idx = idx < 0 ? 0 : (idx > (int)src - 1 ? (int)src - 1 : idx);
            t.tap[(size_t)d * 4 + k] = idx;
            t.cf[(size_t)d * 4 + k] = c[k];
            const float q = std::nearbyint(c[k] * 2048.0f);            // round half to even (default rounding mode)
            t.ci[(size_t)d * 4 + k] = (int)(q < -32768.f ? -32768.f : (q > 32767.f ? 32767.f : q));
        }
    }
    return t;
}

}   // namespace

// depth: fp32 or fp16 [src_h][src_w]; mask: uint8 [src_h][src_w] or null.  (x0, y0): origin of the centred square.
// xt / yt: int4 taps (relative to the square) per output column / row; xc / yc float4; xi / yi int4.
__global__ __launch_bounds__(256) void k_rectify(const void *__restrict__ depth, int is_fp16, const uint8_t *__restrict__ mask,
                                                 uint32_t src_w, uint32_t x0, uint32_t y0, uint32_t W, uint32_t H,
                                                 const int4 *__restrict__ xt, const float4 *__restrict__ xc,
                                                 const int4 *__restrict__ xi, const int4 *__restrict__ yt,
                                                 const float4 *__restrict__ yc, const int4 *__restrict__ yi,
                                                 float *__restrict__ depth_out, uint8_t *__restrict__ mask_out, float far_value)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const uint32_t e = i / W, d = i - e * W;
    const int4 tx = xt[d], ty = yt[e];
    const float4 cx = xc[d], cy = yc[e];
    const int txa[4] = {tx.x, tx.y, tx.z, tx.w}, tya[4] = {ty.x, ty.y, ty.z, ty.w};
    const float cxa[4] = {cx.x, cx.y, cx.z, cx.w}, cya[4] = {cy.x, cy.y, cy.z, cy.w};
    // float image: horizontal pass of the four source rows, then the vertical pass (products and sums each rounded
    // to float: the library is built with -ffp-contract=off)
    float row[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const size_t base = (size_t)(y0 + tya[r]) * src_w + x0;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t at = base + txa[k];
            const float s = is_fp16 ? (float)((const _Float16 *)depth)[at] : ((const float *)depth)[at];
            const float p = s * cxa[k];
            acc = k == 0 ? p : acc + p;
        }
        row[r] = acc;
    }
    float dv = row[0] * cya[0];
    dv = dv + row[1] * cya[1];
    dv = dv + row[2] * cya[2];
    dv = dv + row[3] * cya[3];
    if (mask) {
        const int4 ix = xi[d], iy = yi[e];
        const int ixa[4] = {ix.x, ix.y, ix.z, ix.w}, iya[4] = {iy.x, iy.y, iy.z, iy.w};
        int v = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t base = (size_t)(y0 + tya[r]) * src_w + x0;
            int h = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) h += (int)mask[base + txa[k]] * ixa[k];
            v += h * iya[r];
        }
        v = (v + (1 << 21)) >> 22;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        if (mask_out) mask_out[i] = (uint8_t)v;
        if (v == 0) dv = far_value;                  // reference combined_rendering.py:110  depth[mask == 0] = 100
    }
    depth_out[i] = dv;
}

extern "C" int d2r_rectify_background_depth(d2r_ctx *ctx, const void *depth, int depth_is_fp16, const uint8_t *mask,
                                            uint32_t src_w, uint32_t src_h, uint32_t W, uint32_t H, float *depth_out,
                                            uint8_t *mask_out)
{
    if (!ctx || !depth || !depth_out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    if (!src_w || !src_h || !W || !H || W > 16384 || H > 16384 || src_w > 65535 || src_h > 65535)
        return d2r_fail(ctx, D2R_ERR_INVALID, "bad image size");
    if (mask_out && !mask) return d2r_fail(ctx, D2R_ERR_INVALID, "mask_out without a mask");
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    // centre crop to a square (reference :176-184): the longer side loses (long - short) / 2 at its start
    const uint32_t S = std::min(src_w, src_h);
    const uint32_t x0 = src_w > src_h ? (src_w - src_h) / 2 : 0, y0 = src_h > src_w ? (src_h - src_w) / 2 : 0;
    const AxisTable tx = axis_table(S, W), ty = axis_table(S, H);
    const size_t esz = depth_is_fp16 ? 2 : 4, n_src = (size_t)src_w * src_h, n_dst = (size_t)W * H;
    // one staging allocation: source images, six tables, outputs
    const size_t o_depth = 0, o_mask = o_depth + ((n_src * esz + 15) & ~(size_t)15), o_tab = o_mask + ((n_src + 15) & ~(size_t)15);
    const size_t tab_x = (size_t)W * 16, tab_y = (size_t)H * 16;
    const size_t o_out = o_tab + 3 * tab_x + 3 * tab_y, o_mout = o_out + n_dst * 4, total = o_mout + ((n_dst + 15) & ~(size_t)15);
    int rc;
    if ((rc = d2r_reserve(ctx, ctx->rect_ws, total))) return rc;
    uint8_t *base = (uint8_t *)ctx->rect_ws.p;
    D2R_HIP(ctx, hipMemcpyAsync(base + o_depth, depth, n_src * esz, hipMemcpyHostToDevice, ctx->stream));
    if (mask) D2R_HIP(ctx, hipMemcpyAsync(base + o_mask, mask, n_src, hipMemcpyHostToDevice, ctx->stream));
    uint8_t *t = base + o_tab;
    D2R_HIP(ctx, hipMemcpyAsync(t, tx.tap.data(), tab_x, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(t + tab_x, tx.cf.data(), tab_x, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(t + 2 * tab_x, tx.ci.data(), tab_x, hipMemcpyHostToDevice, ctx->stream));
    uint8_t *u = t + 3 * tab_x;
    D2R_HIP(ctx, hipMemcpyAsync(u, ty.tap.data(), tab_y, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(u + tab_y, ty.cf.data(), tab_y, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(u + 2 * tab_y, ty.ci.data(), tab_y, hipMemcpyHostToDevice, ctx->stream));
    // the host vectors above must outlive the copies: pageable-memory copies are staged before the call returns, and
    // the stream is synchronised below before they go out of scope in any case
    hipLaunchKernelGGL(k_rectify, dim3((uint32_t)((n_dst + 255) / 256)), dim3(256), 0, ctx->stream, (const void *)(base + o_depth),
                       depth_is_fp16, mask ? (const uint8_t *)(base + o_mask) : (const uint8_t *)nullptr, src_w, x0, y0, W, H,
                       (const int4 *)t, (const float4 *)(t + tab_x), (const int4 *)(t + 2 * tab_x), (const int4 *)u,
                       (const float4 *)(u + tab_y), (const int4 *)(u + 2 * tab_y), (float *)(base + o_out),
                       mask_out ? base + o_mout : (uint8_t *)nullptr, 100.0f);
    D2R_HIP(ctx, hipGetLastError());
    D2R_HIP(ctx, hipMemcpyAsync(depth_out, base + o_out, n_dst * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out) D2R_HIP(ctx, hipMemcpyAsync(mask_out, base + o_mout, n_dst, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}
