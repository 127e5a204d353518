// api.hip — the C ABI of libd2r.so (see include/d2r.h).  Host-side glue only: argument
// checking, device memory, and the order in which the kernels of nerf.hip / clip.hip run.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>

#include "d2r_internal.h"
#include "pngio.h"

// implemented in clip.hip / nerf.hip
struct d2r_clip;
int d2r_launch_preprocess(d2r_ctx *, d2r_clip *, const uint8_t *frames_dev, uint32_t n, uint32_t w, uint32_t h,
                          int rot90, uint16_t *patches_dev, float *pixel_values_dev, const void *rects_dev = nullptr,
                          const uint16_t *bg_patches_dev = nullptr, bool touched_only = false);
bool d2r_clip_l0_supported(const d2r_ctx *, const d2r_clip *);
int d2r_clip_forward(d2r_ctx *, const d2r_clip *, const uint16_t *patches_dev, uint32_t n, const float *text_dev,
                     uint32_t C, float logit_scale, float *logits_dev, float *embeds_dev, const ClipL0Reuse *reuse = nullptr);
int d2r_clip_layer0_background(d2r_ctx *, const d2r_clip *, const uint16_t *bg_patches_dev, ClipL0Reuse *out);
int d2r_launch_patchify(d2r_ctx *, const d2r_clip *, const float *pv_dev, uint32_t n, uint16_t *patches_dev);
size_t d2r_clip_patch_bytes(const d2r_clip *, uint32_t n);
int d2r_launch_eval_points(d2r_ctx *, const d2r_nerf *, const float *xyz, const float *dirs, uint32_t n, float *out);
uint32_t d2r_clip_image_size(const d2r_clip *);
uint32_t d2r_clip_proj_dim(const d2r_clip *);
uint32_t d2r_clip_max_images(const d2r_clip *);
uint32_t d2r_clip_tokens(const d2r_clip *);

// Candidates (images) per pass: the "chunk" option, capped so that one pass stays inside the 32-bit
// indexing of the ray queue (rays) and of the GEMM outputs (rows x widest layer).
static uint32_t pass_size(const d2r_ctx *ctx, const d2r_clip *clip, size_t px)
{
    uint64_t per = (uint64_t)std::max<int64_t>(1, ctx->chunk);
    if (clip) per = std::min<uint64_t>(per, d2r_clip_max_images(clip));
    if (px) per = std::min<uint64_t>(per, std::max<uint64_t>(1, 0xffffffffull / px));
    return (uint32_t)per;
}

static thread_local std::string g_err;

int d2r_fail(d2r_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    g_err = msg;
    return code;
}

int d2r_reserve(d2r_ctx *ctx, d2r_ctx::Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return D2R_OK;
    if (b.p) {
        D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));     // an asynchronous fault of earlier work surfaces here: keep the old buffer, report it
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        return d2r_fail(ctx, D2R_ERR_MEMORY, "hipMalloc(" + std::to_string(want) + ") failed");
    }
    b.cap = want - 64;     // kernels may read a few bytes past the last element (k_preprocess: 4-byte pixel loads)
    D2R_HIP(ctx, hipMemsetAsync(b.p, 0, want, ctx->stream));   // padded rows/columns of GEMM operands must be finite
    return D2R_OK;
}

size_t d2r_ctx::timing_begin(int kind)
{
    if (!timing || (kind >= D2R_T_VIT_QKV && timing < 2)) return (size_t)-1;
    while (ev_pool.size() < ev_used + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return (size_t)-1;
        ev_pool.push_back(e);
    }
    size_t b = ev_used, e = ev_used + 1;
    ev_used += 2;
    ev_pairs.push_back({kind, {b, e}});
    (void)hipEventRecord(ev_pool[b], stream);
    return ev_pairs.size() - 1;
}

void d2r_ctx::timing_end(size_t pair)
{
    if (pair == (size_t)-1) return;
    (void)hipEventRecord(ev_pool[ev_pairs[pair].second.second], stream);
}

extern "C" {

int d2r_abi_version(void) { return D2R_ABI_VERSION; }

int d2r_get_timing(d2r_ctx *ctx, d2r_timing *out)
{
    if (!ctx || !out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double ms[D2R_T_KINDS] = {};
    uint64_t n[D2R_T_KINDS] = {};
    for (auto &p : ctx->ev_pairs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, ctx->ev_pool[p.second.first], ctx->ev_pool[p.second.second]) == hipSuccess) {
            ms[p.first] += t;
            n[p.first]++;
        }
    }
    out->march_ms = ms[D2R_T_MARCH];   out->march_launches = n[D2R_T_MARCH];
    out->raygen_ms = ms[D2R_T_RAYGEN]; out->raygen_launches = n[D2R_T_RAYGEN];
    out->prep_ms = ms[D2R_T_PREP];     out->prep_launches = n[D2R_T_PREP];
    out->clip_ms = ms[D2R_T_CLIP];     out->clip_launches = n[D2R_T_CLIP];
    out->sort_ms = ms[D2R_T_SORT];     out->sort_launches = n[D2R_T_SORT];
    out->vit_qkv_ms = ms[D2R_T_VIT_QKV];   out->vit_qkv_launches = n[D2R_T_VIT_QKV];
    out->vit_attn_ms = ms[D2R_T_VIT_ATTN]; out->vit_attn_launches = n[D2R_T_VIT_ATTN];
    out->vit_out_ms = ms[D2R_T_VIT_OUT];   out->vit_out_launches = n[D2R_T_VIT_OUT];
    out->vit_fc1_ms = ms[D2R_T_VIT_FC1];   out->vit_fc1_launches = n[D2R_T_VIT_FC1];
    out->vit_fc2_ms = ms[D2R_T_VIT_FC2];   out->vit_fc2_launches = n[D2R_T_VIT_FC2];
    ctx->ev_used = 0;
    ctx->ev_pairs.clear();
    return D2R_OK;
}

const char *d2r_last_error(d2r_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int d2r_ctx_create(int device, d2r_ctx **out)
{
    if (!out) return d2r_fail(nullptr, D2R_ERR_INVALID, "null out pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return d2r_fail(nullptr, D2R_ERR_DEVICE, "no HIP device: libd2r has no CPU path");
    if (device < 0 || device >= n) return d2r_fail(nullptr, D2R_ERR_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return d2r_fail(nullptr, D2R_ERR_DEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return d2r_fail(nullptr, D2R_ERR_DEVICE, std::string("libd2r is built for gfx950 only, found ") + prop.gcnArchName);
    if (hipSetDevice(device) != hipSuccess) return d2r_fail(nullptr, D2R_ERR_DEVICE, "hipSetDevice failed");
    d2r_ctx *c = new d2r_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int xcc = 0;
    if (hipDeviceGetAttribute(&xcc, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && xcc > 0 && c->n_cu % xcc == 0)
        c->n_xcd = xcc;
    else
        c->n_xcd = c->n_cu % 8 == 0 ? 8 : 1;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return d2r_fail(nullptr, D2R_ERR_DEVICE, "hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    *out = c;
    return D2R_OK;
}

void d2r_ctx_destroy(d2r_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    (void)d2r_comm_destroy(c);
    if (c->render_stream) hipStreamSynchronize(c->render_stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    delete c->pool;
    d2r_ctx::Buf *bufs[] = {&c->cams, &c->queue, &c->queue2, &c->sort_counts, &c->counters, &c->frames, &c->rgba, &c->depth, &c->poses,
                            &c->text, &c->logits, &c->pix, &c->bg_rgba, &c->bg_depth, &c->bg_u8, &c->rects, &c->bg_patches, &c->rect_ws, &c->lens_tab,
                            &c->patches2, &c->frames2, &c->bg_l0, &c->l0_a1, &c->l0_q2, &c->l0_misc};
    for (auto *b : bufs)
        if (b->p) hipFree(b->p);
    for (auto &b : c->clipws)
        if (b.p) hipFree(b.p);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; k++) {
        if (c->text_ev[k]) (void)hipEventDestroy(c->text_ev[k]);
        if (c->text_host[k]) (void)hipHostFree(c->text_host[k]);
    }
    hipEvent_t evs[] = {c->ev_fork, c->ev_prep[0], c->ev_prep[1], c->ev_clip[0], c->ev_clip[1], c->ev_march[0], c->ev_march[1],
                        c->ev_copy[0], c->ev_copy[1]};
    for (hipEvent_t e : evs)
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; k++)
        if (c->frame_host[k]) (void)hipHostFree(c->frame_host[k]);
    if (c->render_stream) (void)hipStreamDestroy(c->render_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int d2r_ctx_set_stream(d2r_ctx *ctx, void *s)
{
    if (!ctx) return d2r_fail(nullptr, D2R_ERR_INVALID, "null ctx");
    hipStreamSynchronize(ctx->stream);
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return D2R_OK;
}

int d2r_ctx_synchronize(d2r_ctx *ctx)
{
    if (!ctx) return d2r_fail(nullptr, D2R_ERR_INVALID, "null ctx");
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}

int d2r_ctx_set_option(d2r_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    if (!strcmp(key, "chunk")) {
        if (value < 1 || value > 16384) return d2r_fail(ctx, D2R_ERR_INVALID, "chunk must be in [1, 16384]");
        ctx->chunk = value;
    } else if (!strcmp(key, "march_compact")) {
        if (value < 0 || value > 1) return d2r_fail(ctx, D2R_ERR_INVALID, "march_compact must be 0 or 1");
        ctx->march_compact = value;
    } else if (!strcmp(key, "refill_min")) {
        if (value < 1 || value > 64) return d2r_fail(ctx, D2R_ERR_INVALID, "refill_min must be in [1, 64]");
        ctx->refill_min = value;
    } else if (!strcmp(key, "ln_fold")) {
        if (value < 0 || value > 4) return d2r_fail(ctx, D2R_ERR_INVALID, "ln_fold must be 0..4");
        ctx->ln_fold = value;
    } else if (!strcmp(key, "gemm_nsplit")) {
        ctx->gemm_nsplit = value;
    } else if (!strcmp(key, "prep_reuse")) {
        ctx->prep_reuse = value != 0;
    } else if (!strcmp(key, "cls_last")) {
        ctx->cls_last = value != 0;
    } else if (!strcmp(key, "vit_fp8")) {
        ctx->vit_fp8 = value != 0;
    } else if (!strcmp(key, "l0_reuse")) {
        ctx->l0_reuse = value != 0;
    } else if (!strcmp(key, "attn_rem")) {
        if (value < 0 || value > 4) return d2r_fail(ctx, D2R_ERR_INVALID, "attn_rem must be 0..4");
        ctx->attn_rem = value;
    } else if (!strcmp(key, "overlap")) {
        ctx->overlap = value != 0;
    } else if (!strcmp(key, "debug_fail_chunk")) {
        ctx->debug_fail_chunk = value;          // test hook, ONE-SHOT: the next d2r_render_score* fails in this chunk, then the hook disarms itself (-1 = off)
    } else if (!strcmp(key, "ray_sort")) {
        if (value < 0 || value > 1) return d2r_fail(ctx, D2R_ERR_INVALID, "ray_sort must be 0 or 1");
        ctx->ray_sort = value;
    } else if (!strcmp(key, "ray_sort_log2")) {
        if (value < 1 || value > 4) return d2r_fail(ctx, D2R_ERR_INVALID, "ray_sort_log2 must be 1 .. 4");
        ctx->ray_sort_log2 = value;
    } else if (!strcmp(key, "march_threads")) {
        if (value < 0 || value > D2R_MARCH_THREADS || value % 64)
            return d2r_fail(ctx, D2R_ERR_INVALID, "march_threads must be 0 (auto) or a multiple of 64 up to " + std::to_string(D2R_MARCH_THREADS) + " (the size the marcher was compiled for)");
        ctx->march_threads = value;
    } else if (!strcmp(key, "march_threads_auto_mib")) {
        if (value < 0 || value > 1 << 20) return d2r_fail(ctx, D2R_ERR_INVALID, "march_threads_auto_mib out of range");
        ctx->march_threads_auto_mib = value;
    } else if (!strcmp(key, "march_blocks")) {
        if (value < 0 || value > 65535) return d2r_fail(ctx, D2R_ERR_INVALID, "march_blocks out of range");
        ctx->march_blocks = value;
#ifdef D2R_DEV
    // experiment switches of development builds (make DEV=1): schedules that were measured no faster and tile
    // configurations kept for comparison; a product build does not know these keys
    } else if (!strcmp(key, "gemm_cfg")) {
        ctx->gemm_cfg = value;
    } else if (!strcmp(key, "gemm_group")) {
        if (value < 0 || value > 65535) return d2r_fail(ctx, D2R_ERR_INVALID, "gemm_group out of range");
        ctx->gemm_group = value;
    } else if (!strcmp(key, "gemm_stagger")) {
        ctx->gemm_stagger = value != 0;
#endif
    } else if (!strcmp(key, "gbrick_slots")) {
        if (value < 0 || value > 8) return d2r_fail(ctx, D2R_ERR_INVALID, "gbrick_slots must be in [0, 8]");
        ctx->gbrick_slots = value;
    } else if (!strcmp(key, "brick_slots_total")) {
        if (value < 0 || value > 8) return d2r_fail(ctx, D2R_ERR_INVALID, "brick_slots_total must be in [0, 8]");
        ctx->brick_slots_total = value;
    } else if (!strcmp(key, "lds_slots_max")) {          // read by d2r_nerf_create: set it before creating the model
        if (value < 0 || value > 5) return d2r_fail(ctx, D2R_ERR_INVALID, "lds_slots_max must be in [0, 5]");
        ctx->lds_slots_max = value;
    } else if (!strcmp(key, "gbrick_max_mib")) {         // read by d2r_nerf_create too
        if (value < 0 || value > 512) return d2r_fail(ctx, D2R_ERR_INVALID, "gbrick_max_mib must be in [0, 512]");
        ctx->gbrick_max_mib = value;
    } else if (!strcmp(key, "mlp_f16")) {
        ctx->mlp_f16 = value != 0;
    } else if (!strcmp(key, "bricks")) {
        ctx->use_bricks = value != 0;
    } else if (!strcmp(key, "raygen_rect")) {
        ctx->raygen_rect = value != 0;
    } else if (!strcmp(key, "timing")) {
        if (value < 0 || value > 2) return d2r_fail(ctx, D2R_ERR_INVALID, "timing must be 0, 1 or 2");
        ctx->timing = value;
        ctx->ev_used = 0;
        ctx->ev_pairs.clear();
    } else {
        return d2r_fail(ctx, D2R_ERR_INVALID, std::string("unknown option ") + key);
    }
    return D2R_OK;
}

int d2r_ctx_get_option(d2r_ctx *ctx, const char *key, int64_t *value)
{
    if (!ctx || !key || !value) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    const struct { const char *k; int64_t v; } tab[] = {
        {"chunk", ctx->chunk}, {"refill_min", ctx->refill_min}, {"march_compact", ctx->march_compact}, {"ln_fold", ctx->ln_fold}, {"gemm_nsplit", ctx->gemm_nsplit},
        {"prep_reuse", ctx->prep_reuse}, {"cls_last", ctx->cls_last}, {"vit_fp8", ctx->vit_fp8}, {"l0_reuse", ctx->l0_reuse},
        {"attn_rem", ctx->attn_rem}, {"overlap", ctx->overlap}, {"march_blocks", ctx->march_blocks}, {"march_threads", ctx->march_threads}, {"ray_sort", ctx->ray_sort}, {"ray_sort_log2", ctx->ray_sort_log2}, {"march_threads_auto_mib", ctx->march_threads_auto_mib}, {"gbrick_slots", ctx->gbrick_slots},
        {"brick_slots_total", ctx->brick_slots_total}, {"lds_slots_max", ctx->lds_slots_max}, {"gbrick_max_mib", ctx->gbrick_max_mib},
        {"bricks", ctx->use_bricks}, {"mlp_f16", ctx->mlp_f16}, {"raygen_rect", ctx->raygen_rect}, {"timing", ctx->timing}, {"debug_fail_chunk", ctx->debug_fail_chunk},
        {"march_lds_slots", (int64_t)ctx->last_march_nb}, {"march_hbm_brick_slots", (int64_t)ctx->last_march_ngb},
        {"march_hbm_brick_bytes", (int64_t)ctx->last_march_gbrick_bytes}, {"march_threads_used", (int64_t)ctx->last_march_threads}};
    for (const auto &e : tab)
        if (!strcmp(key, e.k)) {
            *value = e.v;
            return D2R_OK;
        }
    return d2r_fail(ctx, D2R_ERR_INVALID, std::string("unknown option ") + key);
}

int d2r_get_render_stats(d2r_ctx *ctx, d2r_render_stats *out)
{
    if (!ctx || !out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    *out = ctx->stats;
    return D2R_OK;
}

// ------------------------------------------------------------------- NeRF

static float half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static uint16_t float_to_bf16(float f)
{
    uint32_t b;
    memcpy(&b, &f, 4);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((b >> 16) | 0x40);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}

// MFMA A-operand fragments of the five weight matrices (nerf.hip mlp_tile).
//   kind 2: hash-grid features in slot order              col = 2*(2*(4*s + (j>>1)) + hi) + (j&1)
//   kind 0: k index is a network INPUT in natural order   col = 16*s + 8*hi + j
//   kind 1: k index comes from a previous layer's C layout col = 16*s + 8*(j>>2) + 4*hi + (j&3)
// f16: the fragment holds the snapshot's fp16 bits unrounded (option mlp_f16) instead of their bf16 rounding
static void build_frag(std::vector<uint16_t> &dst, bool f16, int frag, const uint16_t *w, int n_out, int n_in, int mtile,
                       int s, int kind, int col_base = 0)
{
    for (int lane = 0; lane < 64; lane++) {
        int i = lane & 31, hi = lane >> 5, row = mtile * 32 + i;
        for (int j = 0; j < 8; j++) {
            int col = col_base + (kind == 0   ? 16 * s + 8 * hi + j
                                  : kind == 1 ? 16 * s + 8 * (j >> 2) + 4 * hi + (j & 3)
                                              : 2 * (2 * (4 * s + (j >> 1)) + hi) + (j & 1));
            const bool in = row < n_out && col < n_in;
            float v = in ? half_to_float(w[row * n_in + col]) : 0.f;
            dst[((size_t)frag * 64 + lane) * 8 + j] = f16 ? (in ? w[row * n_in + col] : (uint16_t)0) : float_to_bf16(v);
        }
    }
}

int d2r_nerf_create(d2r_ctx *ctx, const d2r_nerf_desc *d, d2r_nerf **out)
{
    if (!ctx || !d || !out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    if (!((d->n_levels == 16 && d->n_features == 2) || (d->n_levels == 8 && d->n_features == 4)))
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "hash grid layout must be L=16,F=2 or L=8,F=4 (32 network inputs)");
    if (!d->level_scale || !d->level_res || !d->level_size || !d->level_offset || !d->grid_fp16 || !d->dw1_fp16 ||
        !d->dw2_fp16 || !d->cw1_fp16 || !d->cw2_fp16 || !d->cw3_fp16 || !d->occupancy_bits)
        return d2r_fail(ctx, D2R_ERR_INVALID, "null field in d2r_nerf_desc");
    const uint32_t aabb = d->aabb_scale ? d->aabb_scale : 1u;
    if ((aabb & (aabb - 1u)) || aabb > 128u) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "aabb_scale must be a power of two <= 128");
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    d2r_nerf *m = new d2r_nerf();
    m->ctx = ctx;
    NerfParams &P = m->P;
    P.n_levels = d->n_levels;
    LevelMeta lv[D2R_MAX_LEVELS];
    int n_dense = 0;
    bool prefix = true;
    uint32_t hash_size = 0;
    for (uint32_t l = 0; l < d->n_levels; l++) {
        LevelMeta &lm = lv[l];
        lm.scale = d->level_scale[l];
        lm.res = d->level_res[l];
        lm.size = d->level_size[l];
        lm.offset = d->level_offset[l];
        uint64_t r3 = (uint64_t)lm.res * lm.res * lm.res;
        lm.hashed = r3 > lm.size;
        if (lm.hashed && (lm.size & (lm.size - 1))) {
            delete m;
            return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "hashed levels must have a power-of-two size");
        }
        if (lm.hashed) {
            if (hash_size && hash_size != lm.size) {
                delete m;
                return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "hashed levels must share one table size");
            }
            hash_size = lm.size;
        }
        if ((uint64_t)lm.offset + lm.size > d->n_entries) {
            delete m;
            return d2r_fail(ctx, D2R_ERR_INVALID, "level table exceeds n_entries");
        }
        if (!lm.hashed && prefix) n_dense++;
        if (lm.hashed) prefix = false;
        if (!lm.hashed && !prefix && (int)l >= n_dense) n_dense = -1000;   // dense after hashed: irregular
    }
    // F = 4: a slot is ONE level and its two halves are feature pairs of the same entries, so the slot kinds
    // (dense / hashed / mixed, in half-levels) count every level twice
    const bool f4 = d->n_features == 4;
    const uint32_t n_slots = 8;
    if (f4 && n_dense >= 0) n_dense *= 2;
    P.n_dense = n_dense < 0 ? -1 : n_dense;
    // slot tables: F = 2: entry e of level 2i+h at word 2e+h of slot i (the two levels interleaved);
    //              F = 4: the level's own table (entry e = words 2e, 2e+1)
    std::vector<uint32_t> tab;
    const uint32_t *src = (const uint32_t *)d->grid_fp16;       // half2 words: one per entry (F = 2), two per entry (F = 4)
    // word of (level-half) vertex: F = 2 -> src[offset + idx]; F = 4 -> src[2 (offset + idx) + h]
    auto half_level = [&](uint32_t slot, int h) -> const LevelMeta & { return f4 ? lv[slot] : lv[2 * slot + h]; };
    auto src_word = [&](uint32_t slot, int h, uint32_t idx) -> uint32_t {
        const LevelMeta &L = half_level(slot, h);
        return f4 ? src[2 * ((size_t)L.offset + idx) + h] : src[(size_t)L.offset + idx];
    };
    for (uint32_t i = 0; i < n_slots; i++) {
        const LevelMeta &a = half_level(i, 0), &b = half_level(i, 1);
        SlotMeta &sm = P.slot[i];
        sm.scale[0] = a.scale; sm.scale[1] = b.scale;
        sm.res[0] = a.res; sm.res[1] = b.res;
        sm.size[0] = a.size; sm.size[1] = b.size;
        sm.hashed[0] = a.hashed; sm.hashed[1] = b.hashed;
        sm.mask8 = hash_size ? (((hash_size - 1u) << 3) | 4u) : 0u;
        sm.off = (uint32_t)(tab.size() * 4);
        const uint32_t n = std::max(a.size, b.size);
        const size_t base = tab.size();
        tab.resize(base + (size_t)n * 2, 0u);
        for (uint32_t e = 0; e < a.size; e++) tab[base + 2 * (size_t)e] = src_word(i, 0, e);
        for (uint32_t e = 0; e < b.size; e++) tab[base + 2 * (size_t)e + 1] = src_word(i, 1, e);
    }
    // occupancy -> 4x4x4 bricks per cascade + bounding box of occupied cells, in the unit cube of
    // the model's box (cascade c spans side 2^c / aabb_scale of it, centred)
    uint32_t n_casc = 1;
    while ((1u << (n_casc - 1)) < aabb) n_casc++;
    P.aabb_scale = aabb;
    P.n_casc = n_casc;
    P.side = (float)aabb;
    P.inv_side = 1.0f / (float)aabb;
    std::vector<uint64_t> bricks((size_t)n_casc * 32 * 32 * 32, 0);
    float blo[3] = {2.f, 2.f, 2.f}, bhi[3] = {-2.f, -2.f, -2.f};
    for (uint32_t cs = 0; cs < n_casc; cs++) {
        const uint8_t *bits = d->occupancy_bits + (size_t)cs * (D2R_GRID * D2R_GRID * D2R_GRID / 8);
        const float side = (float)(1u << cs) / (float)aabb, org = 0.5f - 0.5f * side;   // of the cascade, normalised
        for (int z = 0; z < D2R_GRID; z++)
            for (int y = 0; y < D2R_GRID; y++)
                for (int x = 0; x < D2R_GRID; x++) {
                    uint32_t idx = x + D2R_GRID * (y + D2R_GRID * z);
                    if ((bits[idx >> 3] >> (idx & 7)) & 1) {
                        bricks[(size_t)cs * 32768 + (x >> 2) + 32 * ((y >> 2) + 32 * (z >> 2))] |= 1ull << ((x & 3) + 4 * (y & 3) + 16 * (z & 3));
                        int c[3] = {x, y, z};
                        for (int a = 0; a < 3; a++) {
                            blo[a] = std::min(blo[a], org + side * (float)c[a] / D2R_GRID);
                            bhi[a] = std::max(bhi[a], org + side * (float)(c[a] + 1) / D2R_GRID);
                        }
                    }
                }
    }
    for (int a = 0; a < 3; a++) {
        if (bhi[a] < blo[a]) {          // nothing occupied: empty box
            P.bbox_lo[a] = 2.f;
            P.bbox_hi[a] = -2.f;
        } else {
            P.bbox_lo[a] = blo[a] - 1e-4f;
            P.bbox_hi[a] = bhi[a] + 1e-4f;
        }
    }
    {
        const float half = 0.5f * (float)aabb, box_lo = 0.5f - half, box_hi = 0.5f + half, inv_side = 1.0f / (2.0f * half);
        bool crop = false;
        for (int a = 0; a < 6; a++) crop = crop || d->render_aabb[a] != 0.f;
        for (int a = 0; a < 3; a++) {
            P.raabb_lo[a] = crop ? fmaxf(box_lo, d->render_aabb[a]) : box_lo;
            P.raabb_hi[a] = crop ? fminf(box_hi, d->render_aabb[3 + a]) : box_hi;
            P.rn_lo[a] = aabb >= 2u ? fmaf(P.raabb_lo[a] - 0.5f, inv_side, 0.5f) : P.raabb_lo[a];
            P.rn_hi[a] = aabb >= 2u ? fmaf(P.raabb_hi[a] - 0.5f, inv_side, 0.5f) : P.raabb_hi[a];
        }
    }
    // De-hashed, bounding-box-local dense bricks of the leading levels (small objects): the
    // vertices a sample inside an occupied cell can touch, with the table value (incl. tiny-cuda-nn's
    // index wrap) resolved here once.  Served from LDS by k_march; values identical to the tables.
    std::vector<uint32_t> brick_tab, gbrick_tab;
    P.n_brick_slots = 0;
    P.brick_words = 0;
    P.n_gbrick_slots = 0;
    // extent of the brick of level 2i+h over the occupied bounding box: first vertex and vertex counts per axis
    auto level_brick_extent = [&](uint32_t i, int h, int (&g0)[3], int (&n)[3]) -> size_t {
        const LevelMeta &L = half_level(i, h);
        for (int a = 0; a < 3; a++) {
            // same correctly-rounded fma the kernels use: monotone, so these bound every sample
            float plo = fmaf(L.scale, blo[a], 0.5f);
            float phi = fmaf(L.scale, bhi[a], 0.5f);
            g0[a] = (int)floorf(plo);
            n[a] = (int)floorf(phi) + 1 - g0[a] + 1;
        }
        return (size_t)n[0] * n[1] * n[2];
    };
    // words a slot's two bricks take (computed from the box and the levels' resolutions alone: the budget tests below run on
    // this BEFORE anything is built)
    auto slot_brick_words = [&](uint32_t i) -> size_t {
        int g0[3], n[3];
        return level_brick_extent(i, 0, g0, n) + level_brick_extent(i, 1, g0, n);
    };
    // appends the brick of level 2i+h to `words` (the caller has checked the budget)
    auto add_level_brick = [&](uint32_t i, int h, std::vector<uint32_t> &words, size_t word0) {
        SlotMeta &sm = P.slot[i];
        const LevelMeta &L = half_level(i, h);
        int g0[3], n[3];
        const size_t cnt = level_brick_extent(i, h, g0, n);
        words.reserve(words.size() + cnt);
        const size_t base = word0 + words.size();
        sm.bnx[h] = (uint32_t)n[0];
        sm.bnxy[h] = (uint32_t)(n[0] * n[1]);
        sm.bbase[h] = (int32_t)((int64_t)base - ((int64_t)g0[0] + (int64_t)n[0] * g0[1] + (int64_t)n[0] * n[1] * g0[2]));

        for (int z = 0; z < n[2]; z++)
            for (int y = 0; y < n[1]; y++)
                for (int x = 0; x < n[0]; x++) {
                    uint32_t gx = (uint32_t)(g0[0] + x), gy = (uint32_t)(g0[1] + y), gz = (uint32_t)(g0[2] + z);
                    uint64_t idx;
                    if (L.hashed)
                        idx = (uint32_t)(gx ^ (gy * 2654435761u) ^ (gz * 805459861u));
                    else
                        idx = (uint64_t)gx + (uint64_t)gy * L.res + (uint64_t)gz * L.res * L.res;
                    words.push_back(src_word(i, h, (uint32_t)(idx % L.size)));
                }
    };
    if (bhi[0] >= blo[0] && P.n_dense >= 0) {
        // LDS: the longest prefix of slots (at most 5: levels 0-9) whose bricks fit beside the MLP fragments
        const size_t budget_words = (160 * 1024 - (size_t)D2R_N_WFRAG * 64 * 16 - 512 /* the cone-stepping tables of the CONE kernels */) / 4;
        const uint32_t lds_max = (uint32_t)std::min<int64_t>(5, std::max<int64_t>(0, ctx->lds_slots_max));
        std::vector<uint32_t> words;
        for (uint32_t i = 0; i < n_slots && i < lds_max; i++) {
            if (words.size() + slot_brick_words(i) > budget_words) break;
            add_level_brick(i, 0, words, 0);
            add_level_brick(i, 1, words, 0);
            P.n_brick_slots = i + 1;
            P.brick_words = (uint32_t)words.size();
        }
        brick_tab.assign(words.begin(), words.begin() + P.brick_words);
        // HBM: the slots behind the LDS ones as dense bounding-box bricks too (neighbouring rays touch neighbouring entries,
        // one 8-byte gather returns both x-corners) while a slot's brick stays below `gbrick_max_mib` (a brick grows with the
        // cube of the object's size and of the level's resolution, the hashed table it replaces is 4 MiB per slot whatever
        // the object) and the total below 512 MiB.  Round 5: for ANY number of LDS slots — an object slightly too large for
        // five LDS slots used to fall to none and to no HBM bricks either.
        const size_t gbudget = ((size_t)512 << 20) >> 2;        // words
        const size_t slot_cap = ((size_t)std::max<int64_t>(0, ctx->gbrick_max_mib) << 20) >> 2;
        for (uint32_t i = P.n_brick_slots; i < n_slots; i++) {
            // size first: a slot whose brick would break the per-slot cap or the total is never built, and a slot
            // that fits is appended in place — no copy of the (up to 512 MiB) table per candidate slot
            const size_t need = slot_brick_words(i);
            if (need > slot_cap || gbrick_tab.size() + need > gbudget) break;
            add_level_brick(i, 0, gbrick_tab, 0);
            add_level_brick(i, 1, gbrick_tab, 0);
            P.n_gbrick_slots = i + 1 - P.n_brick_slots;
        }
    }
    // weight fragments
    // [0, 24): bf16 fragments; [24, 48): the same fragments in fp16 (exact)
    std::vector<uint16_t> wf((size_t)2 * D2R_N_WFRAG * 64 * 8);
    const int n_in = (int)(d->n_levels * d->n_features);
    for (int pass = 0; pass < 2; pass++) {
        const bool f16 = pass == 1;
        const int o = pass * D2R_N_WFRAG;
        for (int mt = 0; mt < 2; mt++)
            for (int s = 0; s < 2; s++) build_frag(wf, f16, o + 0 + mt * 2 + s, d->dw1_fp16, 64, n_in, mt, s, 2);
        for (int q = 0; q < 4; q++) build_frag(wf, f16, o + 4 + q, d->dw2_fp16, 16, 64, 0, q, 1);
        for (int mt = 0; mt < 2; mt++) {
            build_frag(wf, f16, o + 8 + mt * 2 + 0, d->cw1_fp16, 64, 32, mt, 0, 1);          // density outputs (C layout)
            build_frag(wf, f16, o + 8 + mt * 2 + 1, d->cw1_fp16, 64, 32, mt, 0, 0, 16);      // SH, natural order at col 16
        }
        for (int mt = 0; mt < 2; mt++)
            for (int q = 0; q < 4; q++) build_frag(wf, f16, o + 12 + mt * 4 + q, d->cw2_fp16, 64, 64, mt, q, 1);
        for (int q = 0; q < 4; q++) build_frag(wf, f16, o + 20 + q, d->cw3_fp16, 16, 64, 0, q, 1);
    }

    const size_t grid_bytes = tab.size() * 4;
    if (grid_bytes >= (1ull << 32)) {
        delete m;
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "grid larger than 4 GiB");
    }
    bool ok = hipMalloc(&m->d_grid, grid_bytes) == hipSuccess &&
              hipMalloc(&m->d_bricks, bricks.size() * 8) == hipSuccess &&
              hipMalloc(&m->d_wfrag, wf.size() * 2) == hipSuccess &&
              hipMalloc(&m->d_brick_tab, std::max<size_t>(brick_tab.size(), 1) * 4) == hipSuccess &&
              hipMalloc(&m->d_gbrick_tab, std::max<size_t>(gbrick_tab.size(), 1) * 4) == hipSuccess;
    ok = ok && hipMemcpy(m->d_grid, tab.data(), grid_bytes, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(m->d_bricks, bricks.data(), bricks.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(m->d_wfrag, wf.data(), wf.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
         (brick_tab.empty() || hipMemcpy(m->d_brick_tab, brick_tab.data(), brick_tab.size() * 4, hipMemcpyHostToDevice) == hipSuccess) &&
         (gbrick_tab.empty() || hipMemcpy(m->d_gbrick_tab, gbrick_tab.data(), gbrick_tab.size() * 4, hipMemcpyHostToDevice) == hipSuccess);
    if (!ok) {
        d2r_nerf_destroy(m);
        return d2r_fail(ctx, D2R_ERR_MEMORY, "device allocation/upload failed for the NeRF model");
    }
    P.grid = (const uint32_t *)m->d_grid;
    P.grid_bytes = (uint32_t)grid_bytes;
    P.bricks = (const uint64_t *)m->d_bricks;
    P.wfrag = (const uint4 *)m->d_wfrag;
    P.wfrag16 = P.wfrag + (size_t)D2R_N_WFRAG * 64;
    P.brick_tab = (const uint32_t *)m->d_brick_tab;
    P.gbrick_tab = (const uint32_t *)m->d_gbrick_tab;
    P.gbrick_bytes = (uint32_t)(std::max<size_t>(gbrick_tab.size(), 1) * 4);
    *out = m;
    return D2R_OK;
}

void d2r_nerf_destroy(d2r_nerf *m)
{
    if (!m) return;
    if (m->d_grid) hipFree(m->d_grid);
    if (m->d_bricks) hipFree(m->d_bricks);
    if (m->d_wfrag) hipFree(m->d_wfrag);
    if (m->d_brick_tab) hipFree(m->d_brick_tab);
    if (m->d_gbrick_tab) hipFree(m->d_gbrick_tab);
    delete m;
}

static int check_view(d2r_ctx *ctx, const d2r_view *v)
{
    if (!v) return d2r_fail(ctx, D2R_ERR_INVALID, "null view");
    if (v->width == 0 || v->height == 0 || v->width > 8192 || v->height > 8192)
        return d2r_fail(ctx, D2R_ERR_INVALID, "bad view size");
    if (!(v->scale > 0.f)) return d2r_fail(ctx, D2R_ERR_INVALID, "view.scale must be positive");
    if (v->lens_mode != D2R_LENS_PERSPECTIVE && v->lens_mode != D2R_LENS_OPENCV)
        return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, "view.lens_mode " + std::to_string(v->lens_mode) + " is not implemented (0 = perspective, 1 = OpenCV k1, k2, p1, p2)");
    if (v->lens_mode == D2R_LENS_OPENCV)
        for (int i = 0; i < 4; i++)
            if (!std::isfinite(v->lens_params[i]) || fabsf(v->lens_params[i]) > 16.f)
                return d2r_fail(ctx, D2R_ERR_INVALID, "view.lens_params must be finite OpenCV coefficients (k1, k2, p1, p2; magnitude at most 16)");
    return D2R_OK;
}

static int fetch_stats(d2r_ctx *ctx, uint64_t rays_total, bool accumulate)
{
    uint32_t c[8];
    D2R_HIP(ctx, hipMemcpyAsync(c, ctx->counters.p, 32, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t samples, iters;
    memcpy(&samples, &c[2], 8);
    memcpy(&iters, &c[4], 8);
    if (!accumulate) ctx->stats = d2r_render_stats{};
    ctx->stats.rays_total += rays_total;
    ctx->stats.rays_alive += c[0];
    ctx->stats.samples += samples;
    ctx->stats.wave_iters += iters;
    return D2R_OK;
}

int d2r_render(d2r_ctx *ctx, const d2r_nerf *model, const d2r_view *view, const float *cams_nerf, uint32_t n,
               float *rgba_out, float *depth_out, uint64_t *n_samples_out)
{
    if (!ctx || !model || !cams_nerf) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    int rc = check_view(ctx, view);
    if (rc) return rc;
    if (n == 0) return D2R_OK;
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    const ViewParams V = d2r_view_params(view);
    const size_t px = (size_t)V.W * V.H;
    ctx->stats = d2r_render_stats{};
    ctx->last_chunks = 0;
    // bound the pass size: 2^31 rays and ~1 GiB of fp32 frames
    uint32_t per = (uint32_t)std::max<size_t>(1, std::min<size_t>(n, (64u << 20) / px + 1));
    for (uint32_t c0 = 0; c0 < n; c0 += per) {
        uint32_t nc = std::min(per, n - c0);
        if ((rc = d2r_reserve(ctx, ctx->poses, (size_t)nc * 12 * 4))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->cams, (size_t)nc * 12 * 4))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->rgba, (size_t)nc * px * 16))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->depth, (size_t)nc * px * 4))) return rc;
        D2R_HIP(ctx, hipMemcpyAsync(ctx->poses.p, cams_nerf + (size_t)c0 * 12, (size_t)nc * 48, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = d2r_launch_cameras_direct(ctx, V, (const float *)ctx->poses.p, nc, (float *)ctx->cams.p))) return rc;
        if ((rc = d2r_launch_render(ctx, model, V, (const float *)ctx->cams.p, nc, false, (float *)ctx->rgba.p,
                                    (float *)ctx->depth.p, nullptr)))
            return rc;
        if (rgba_out)
            D2R_HIP(ctx, hipMemcpyAsync(rgba_out + (size_t)c0 * px * 4, ctx->rgba.p, (size_t)nc * px * 16, hipMemcpyDeviceToHost, ctx->stream));
        if (depth_out)
            D2R_HIP(ctx, hipMemcpyAsync(depth_out + (size_t)c0 * px, ctx->depth.p, (size_t)nc * px * 4, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = fetch_stats(ctx, (uint64_t)nc * px, true))) return rc;
    }
    if (n_samples_out) *n_samples_out = ctx->stats.samples;
    return D2R_OK;
}

int d2r_lens_undistort_view(d2r_ctx *ctx, const d2r_view *view, float *dirs_out)
{
    if (!ctx || !dirs_out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    int rc = check_view(ctx, view);
    if (rc) return rc;
    if (view->lens_mode != D2R_LENS_OPENCV) return d2r_fail(ctx, D2R_ERR_INVALID, "d2r_lens_undistort_view: the view has no lens (lens_mode must be D2R_LENS_OPENCV)");
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    ViewParams V = d2r_view_params(view);
    if ((rc = d2r_lens_table(ctx, V))) return rc;
    D2R_HIP(ctx, hipMemcpyAsync(dirs_out, V.lens_tab, (size_t)V.W * V.H * 8, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}

int d2r_nerf_eval_points(d2r_ctx *ctx, const d2r_nerf *model, const float *xyz, const float *dirs, uint32_t n,
                         float *sigma_rgb_out)
{
    if (!ctx || !model || !xyz || !dirs || !sigma_rgb_out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    if (n == 0) return D2R_OK;
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = d2r_reserve(ctx, ctx->rgba, (size_t)n * 16))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->depth, (size_t)n * 12))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->pix, (size_t)n * 12))) return rc;
    D2R_HIP(ctx, hipMemcpyAsync(ctx->depth.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(ctx->pix.p, dirs, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = d2r_launch_eval_points(ctx, model, (const float *)ctx->depth.p, (const float *)ctx->pix.p, n, (float *)ctx->rgba.p)))
        return rc;
    D2R_HIP(ctx, hipMemcpyAsync(sigma_rgb_out, ctx->rgba.p, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}

int d2r_set_background(d2r_ctx *ctx, const d2r_view *view, const float *bg_rgba, const float *bg_depth)
{
    if (!ctx || !bg_rgba || !bg_depth) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    int rc = check_view(ctx, view);
    if (rc) return rc;
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)view->width * view->height;
    if ((rc = d2r_reserve(ctx, ctx->bg_rgba, px * 16))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->bg_depth, px * 4))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->bg_u8, px * 3 + 16))) return rc;
    D2R_HIP(ctx, hipMemcpyAsync(ctx->bg_rgba.p, bg_rgba, px * 16, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(ctx->bg_depth.p, bg_depth, px * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = d2r_launch_bg_quantize(ctx, view->width, view->height))) return rc;
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->bg_w = view->width;
    ctx->bg_h = view->height;
    ctx->bg_patches_for = nullptr;          // the background's CLIP patches are recomputed on the next d2r_render_score
    ctx->png_base.reset();                  // ... and so is the frame files' shared background coding
    return D2R_OK;
}

int d2r_render_composite(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_view *view, const float *obj_pose_now,
                         const float *cam_pose, const float *obj_poses, uint32_t K, uint8_t *frames_out)
{
    if (!ctx || !fg || !obj_pose_now || !cam_pose || !obj_poses || !frames_out)
        return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    int rc = check_view(ctx, view);
    if (rc) return rc;
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    const ViewParams V = d2r_view_params(view);
    const size_t px = (size_t)V.W * V.H;
    ctx->stats = d2r_render_stats{};
    ctx->last_chunks = 0;
    const uint32_t per = pass_size(ctx, nullptr, px);
    for (uint32_t c0 = 0; c0 < K; c0 += per) {
        uint32_t nc = std::min(per, K - c0);
        if ((rc = d2r_reserve(ctx, ctx->poses, (size_t)nc * 64))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->cams, (size_t)nc * 48))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->frames, (size_t)nc * px * 3))) return rc;
        D2R_HIP(ctx, hipMemcpyAsync(ctx->poses.p, obj_poses + (size_t)c0 * 16, (size_t)nc * 64, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = d2r_launch_cameras_virtual(ctx, V, obj_pose_now, cam_pose, (const float *)ctx->poses.p, nc, (float *)ctx->cams.p)))
            return rc;
        if ((rc = d2r_launch_render(ctx, fg, V, (const float *)ctx->cams.p, nc, true, nullptr, nullptr, (uint8_t *)ctx->frames.p)))
            return rc;
        D2R_HIP(ctx, hipMemcpyAsync(frames_out + (size_t)c0 * px * 3, ctx->frames.p, (size_t)nc * px * 3, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = fetch_stats(ctx, (uint64_t)nc * px, true))) return rc;
    }
    return D2R_OK;
}

// -------------------------------------------------------------------- CLIP

static int upload_text(d2r_ctx *ctx, const d2r_clip *clip, const float *text, uint32_t C)
{
    if (!text || C == 0 || C > 1024) return d2r_fail(ctx, D2R_ERR_INVALID, "bad text embeddings");
    size_t bytes = (size_t)C * d2r_clip_proj_dim(clip) * 4;
    int rc = d2r_reserve(ctx, ctx->text, bytes);
    if (rc) return rc;
    // staged through a pinned slot: the caller's buffer is fully read before this returns (d2r.h: "host pointers are
    // read before the call returns"), and the copy itself stays asynchronous on the context's stream
    const uint32_t slot = ctx->text_turn++ & 1u;
    if (ctx->text_ev[slot]) D2R_HIP(ctx, hipEventSynchronize(ctx->text_ev[slot]));        // the slot's previous copy has left it
    else D2R_HIP(ctx, hipEventCreateWithFlags(&ctx->text_ev[slot], hipEventDisableTiming));
    if (ctx->text_host_cap[slot] < bytes) {
        if (ctx->text_host[slot]) (void)hipHostFree(ctx->text_host[slot]);
        ctx->text_host[slot] = nullptr;
        ctx->text_host_cap[slot] = 0;
        if (hipHostMalloc(&ctx->text_host[slot], bytes + 4096, hipHostMallocDefault) != hipSuccess)
            return d2r_fail(ctx, D2R_ERR_MEMORY, "hipHostMalloc failed for the text staging buffer");
        ctx->text_host_cap[slot] = bytes + 4096;
    }
    memcpy(ctx->text_host[slot], text, bytes);
    D2R_HIP(ctx, hipMemcpyAsync(ctx->text.p, ctx->text_host[slot], bytes, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipEventRecord(ctx->text_ev[slot], ctx->stream));
    return D2R_OK;
}

int d2r_clip_score_frames(d2r_ctx *ctx, const d2r_clip *clip, const uint8_t *frames, uint32_t n, uint32_t w,
                          uint32_t h, int rot90, const float *text_embeds, uint32_t C, float logit_scale,
                          float *logits_out, float *embeds_out)
{
    if (!ctx || !clip || !frames || !logits_out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    int rc = upload_text(ctx, clip, text_embeds, C);
    if (rc) return rc;
    const size_t px = (size_t)w * h;
    const uint32_t D = d2r_clip_proj_dim(clip);
    const uint32_t per = pass_size(ctx, clip, 0);
    for (uint32_t c0 = 0; c0 < n; c0 += per) {
        uint32_t nc = std::min(per, n - c0);
        if ((rc = d2r_reserve(ctx, ctx->frames, (size_t)nc * px * 3))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->clipws[6], d2r_clip_patch_bytes(clip, nc)))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->logits, (size_t)nc * (C + D) * 4))) return rc;
        float *lg = (float *)ctx->logits.p, *em = lg + (size_t)nc * C;
        D2R_HIP(ctx, hipMemcpyAsync(ctx->frames.p, frames + (size_t)c0 * px * 3, (size_t)nc * px * 3, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = d2r_launch_preprocess(ctx, (d2r_clip *)clip, (const uint8_t *)ctx->frames.p, nc, w, h, rot90,
                                        (uint16_t *)ctx->clipws[6].p, nullptr)))
            return rc;
        if ((rc = d2r_clip_forward(ctx, clip, (const uint16_t *)ctx->clipws[6].p, nc, (const float *)ctx->text.p, C,
                                   logit_scale, lg, em)))
            return rc;
        D2R_HIP(ctx, hipMemcpyAsync(logits_out + (size_t)c0 * C, lg, (size_t)nc * C * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (embeds_out)
            D2R_HIP(ctx, hipMemcpyAsync(embeds_out + (size_t)c0 * D, em, (size_t)nc * D * 4, hipMemcpyDeviceToHost, ctx->stream));
        D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return D2R_OK;
}

int d2r_clip_preprocess(d2r_ctx *ctx, const d2r_clip *clip, const uint8_t *frames, uint32_t n, uint32_t w,
                        uint32_t h, int rot90, float *pixel_values_out)
{
    if (!ctx || !clip || !frames || !pixel_values_out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)w * h, S = d2r_clip_image_size(clip);
    int rc;
    if ((rc = d2r_reserve(ctx, ctx->frames, (size_t)n * px * 3))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->pix, (size_t)n * 3 * S * S * 4))) return rc;
    D2R_HIP(ctx, hipMemcpyAsync(ctx->frames.p, frames, (size_t)n * px * 3, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = d2r_launch_preprocess(ctx, (d2r_clip *)clip, (const uint8_t *)ctx->frames.p, n, w, h, rot90, nullptr,
                                    (float *)ctx->pix.p)))
        return rc;
    D2R_HIP(ctx, hipMemcpyAsync(pixel_values_out, ctx->pix.p, (size_t)n * 3 * S * S * 4, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}

int d2r_clip_embed_pixels(d2r_ctx *ctx, const d2r_clip *clip, const float *pixel_values, uint32_t n,
                          float *embeds_out)
{
    if (!ctx || !clip || !pixel_values || !embeds_out) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    const size_t S = d2r_clip_image_size(clip);
    const uint32_t D = d2r_clip_proj_dim(clip);
    int rc;
    const uint32_t per = pass_size(ctx, clip, 0);
    for (uint32_t c0 = 0; c0 < n; c0 += per) {
        uint32_t nc = std::min(per, n - c0);
        if ((rc = d2r_reserve(ctx, ctx->pix, (size_t)nc * 3 * S * S * 4))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->clipws[6], d2r_clip_patch_bytes(clip, nc)))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->logits, (size_t)nc * D * 4))) return rc;
        if ((rc = d2r_reserve(ctx, ctx->text, (size_t)D * 4))) return rc;
        D2R_HIP(ctx, hipMemcpyAsync(ctx->pix.p, pixel_values + (size_t)c0 * 3 * S * S, (size_t)nc * 3 * S * S * 4, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = d2r_launch_patchify(ctx, clip, (const float *)ctx->pix.p, nc, (uint16_t *)ctx->clipws[6].p))) return rc;
        if ((rc = d2r_clip_forward(ctx, clip, (const uint16_t *)ctx->clipws[6].p, nc, (const float *)ctx->text.p, 0, 1.0f,
                                   nullptr, (float *)ctx->logits.p)))
            return rc;
        D2R_HIP(ctx, hipMemcpyAsync(embeds_out + (size_t)c0 * D, ctx->logits.p, (size_t)nc * D * 4, hipMemcpyDeviceToHost, ctx->stream));
        D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return D2R_OK;
}

// ------------------------------------------------------------ fused hot path

// Runs launches on another stream of the context for the lifetime of the guard (the launchers of nerf.hip / clip.hip
// and the timing events all go to ctx->stream).
struct StreamSwap {
    d2r_ctx *c;
    hipStream_t saved;
    StreamSwap(d2r_ctx *ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { ctx->stream = s; }
    ~StreamSwap() { c->stream = saved; }
};

static int ensure_pipeline(d2r_ctx *ctx, bool need_copy)
{
    if (!ctx->render_stream) D2R_HIP(ctx, hipStreamCreateWithFlags(&ctx->render_stream, hipStreamNonBlocking));
    if (need_copy && !ctx->copy_stream) D2R_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    hipEvent_t *all[] = {&ctx->ev_fork, &ctx->ev_prep[0], &ctx->ev_prep[1], &ctx->ev_clip[0], &ctx->ev_clip[1],
                         &ctx->ev_march[0], &ctx->ev_march[1], &ctx->ev_copy[0], &ctx->ev_copy[1]};
    for (hipEvent_t *e : all)
        if (!*e) D2R_HIP(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    return D2R_OK;
}

// Candidates per pass when frames leave the GPU: two pinned staging buffers of at most 1 GiB each.
static uint32_t frame_pass_size(uint32_t per, size_t px)
{
    const uint64_t fit = std::max<uint64_t>(64, (1ull << 30) / (px * 3));
    uint64_t p = std::min<uint64_t>(per, fit);
    if (p >= 256) p -= p % 256;
    return (uint32_t)std::max<uint64_t>(1, p);
}

// Hands the frames of one finished chunk (in pinned buffer b) to the worker pool: PNG files and / or the copy into
// the caller's array.  The buffer is free again when the pool has finished group b.
static void dispatch_frames(d2r_ctx *ctx, int b, uint32_t c0, uint32_t nc, uint32_t W, uint32_t H, uint8_t *frames_out,
                            const d2r_frame_sink *sink)
{
    const uint8_t *src = (const uint8_t *)ctx->frame_host[b];
    const size_t fb = (size_t)W * H * 3;
    if (sink && sink->png_dir) {
        const std::string dir(sink->png_dir);
        const uint32_t first = sink->png_first_index + c0;
        const int level = sink->png_level;
        // default encoding: scanlines a candidate's object did not touch are the background's, coded once per background (pngio.cpp)
        const std::shared_ptr<const D2rPngBase> base = level < 0 ? ctx->png_base : nullptr;
        for (uint32_t i = 0; i < nc; i++) {
            if (base) ctx->pool->submit(b, [=](std::string &err) { return d2r_png_write_file_delta(*base, src + fb * i, d2r_png_name(dir, first + i), err); });
            else ctx->pool->submit(b, [=](std::string &err) { return d2r_png_write_file(src + fb * i, W, H, level, d2r_png_name(dir, first + i), err); });
        }
    }
    if (frames_out) {
        const uint32_t step = std::max<uint32_t>(1, (uint32_t)((32u << 20) / fb));
        for (uint32_t i = 0; i < nc; i += step) {
            const uint32_t m = std::min(step, nc - i);
            uint8_t *dst = frames_out + ((size_t)c0 + i) * fb;
            ctx->pool->submit(b, [=](std::string &) { memcpy(dst, src + fb * i, fb * m); return 0; });
        }
    }
}

// The pass over K candidates: per chunk  cameras -> ray generation -> march + composite -> (frames to the host) ->
// rot90 + CLIP preprocess  on the RENDER stream, ViT forward + logits on the context's stream; the patch buffer is
// double-buffered between the two, so chunk i+1 renders while chunk i is scored ("overlap" option; with 0 both
// halves run on the context's stream in program order, as before round 4).  poses_dev / logits_dev are device
// pointers ordered on the context's stream.  With frames_out / sink the frames of every chunk are copied to one of two
// pinned host buffers on a copy stream and handed to the worker pool; the call then returns after the last file is
// written, otherwise it is asynchronous.
static int render_score_body(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_clip *clip, const d2r_view *view,
                             const float *obj_pose_now, const float *cam_pose, const float *poses_dev, uint32_t K,
                             uint32_t C, float logit_scale, float *logits_dev, uint8_t *frames_out, const d2r_frame_sink *sink)
{
    int rc;
    const ViewParams V = d2r_view_params(view);
    const size_t px = (size_t)V.W * V.H;
    const bool to_host = frames_out || (sink && sink->png_dir);
    uint32_t per = pass_size(ctx, clip, px);
    if (to_host) per = frame_pass_size(per, px);
    const uint32_t nchunks = (K + per - 1) / per;
    ctx->last_pass = per;
    ctx->last_chunks = nchunks;
    const uint32_t cap = std::min(per, K);
    const bool two = ctx->overlap != 0 && nchunks > 1;
    if ((rc = ensure_pipeline(ctx, to_host))) return rc;
    // workspaces are sized once for a full chunk so that no allocation happens inside the loop
    const size_t patch_bytes = d2r_clip_patch_bytes(clip, cap);
    if ((rc = d2r_reserve(ctx, ctx->cams, (size_t)cap * 48))) return rc;
    if ((rc = d2r_reserve_render(ctx, (size_t)cap * px))) return rc;          // queue, and the ray sort's second queue + bin counts
    if ((rc = d2r_reserve(ctx, ctx->frames, (size_t)cap * px * 3))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->clipws[6], patch_bytes))) return rc;
    if (two && (rc = d2r_reserve(ctx, ctx->patches2, patch_bytes))) return rc;
    if (to_host && nchunks > 1 && (rc = d2r_reserve(ctx, ctx->frames2, (size_t)cap * px * 3))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->counters, 64 + 32 * (size_t)nchunks))) return rc;
    if (to_host) {
        const size_t hb = (size_t)cap * px * 3;
        for (int b = 0; b < (nchunks > 1 ? 2 : 1); b++)
            if (ctx->frame_host_cap[b] < hb) {
                if (ctx->frame_host[b]) (void)hipHostFree(ctx->frame_host[b]);
                ctx->frame_host[b] = nullptr;
                ctx->frame_host_cap[b] = 0;
                if (hipHostMalloc(&ctx->frame_host[b], hb, hipHostMallocDefault) != hipSuccess)
                    return d2r_fail(ctx, D2R_ERR_MEMORY, "hipHostMalloc(" + std::to_string(hb) + ") failed for the frame staging buffer");
                ctx->frame_host_cap[b] = hb;
            }
        // the background frame as the candidates' frames hold it (k_frames_init copies bg_u8), for the PNG files' shared scanline coding
        if (sink && sink->png_dir && sink->png_level < 0 && !ctx->png_base && ctx->bg_u8.p && ctx->bg_w == V.W && ctx->bg_h == V.H) {
            std::vector<uint8_t> bg(px * 3);
            D2R_HIP(ctx, hipMemcpy(bg.data(), ctx->bg_u8.p, px * 3, hipMemcpyDeviceToHost));
            ctx->png_base = d2r_png_base_build(bg.data(), V.W, V.H);
        }
        const int want = sink && sink->png_threads > 0 ? sink->png_threads : d2r_default_io_threads();
        if (!ctx->pool || ctx->pool->size() != want) {
            delete ctx->pool;
            ctx->pool = new D2rJobPool(want);
        }
    }
    // the background frame's own patches, once per (background, CLIP model): bands of a candidate that its rays cannot
    // have touched are copies of these (k_preprocess)
    const bool reuse_bg = ctx->prep_reuse && ctx->raygen_rect && ctx->bg_w == V.W && ctx->bg_h == V.H && ctx->bg_u8.p;
    if (reuse_bg) {
        if ((rc = d2r_reserve(ctx, ctx->rects, (size_t)cap * 16 * 2))) return rc;      // two halves: the ViT of chunk i reads its rectangles while chunk i+1 is rendered ("overlap")
        if (ctx->bg_patches_for != (const void *)clip) {
            if ((rc = d2r_reserve(ctx, ctx->bg_patches, d2r_clip_patch_bytes(clip, 1)))) return rc;
            if ((rc = d2r_launch_preprocess(ctx, (d2r_clip *)clip, (const uint8_t *)ctx->bg_u8.p, 1, V.W, V.H, 1,
                                            (uint16_t *)ctx->bg_patches.p, nullptr)))
                return rc;
            ctx->bg_patches_for = (const void *)clip;
            ctx->bg_l0_for = nullptr;
        }
    }
    // layer-0 reuse: the background's own pre-LayerNorm rows and q / k / v rows, once per (background, CLIP model)
    ClipL0Reuse l0{};
    const bool use_l0 = reuse_bg && ctx->l0_reuse && d2r_clip_l0_supported(ctx, clip);
    if (use_l0) {
        if (ctx->bg_l0_for != (const void *)clip) {
            rc = d2r_clip_layer0_background(ctx, clip, (const uint16_t *)ctx->bg_patches.p, &ctx->bg_l0_desc);
            if (rc == D2R_OK) ctx->bg_l0_for = (const void *)clip;
            else if (rc != D2R_ERR_UNSUPPORTED) return rc;     // (more than 1024 patches per image: plain forward)
        }
        if (ctx->bg_l0_for == (const void *)clip) {
            l0 = ctx->bg_l0_desc;
            l0.w = V.W;
            l0.h = V.H;
            l0.rects = ctx->rects.p;
        }
    }
    ctx->stats = d2r_render_stats{};
    hipStream_t main = ctx->stream, rs = two ? ctx->render_stream : main, xs = ctx->copy_stream;
    if (rs != main) {       // the render stream starts after whatever the caller queued before this call (pose upload, text)
        D2R_HIP(ctx, hipEventRecord(ctx->ev_fork, main));
        D2R_HIP(ctx, hipStreamWaitEvent(rs, ctx->ev_fork, 0));
    }
    uint32_t prev_c0 = 0, prev_nc = 0;
    int prev_b = -1;
    for (uint32_t c0 = 0, ci = 0; c0 < K; c0 += per, ci++) {
        const uint32_t nc = std::min(per, K - c0);
        const int b = (int)(ci & 1u), pb = two ? b : 0, fbuf = (to_host && nchunks > 1) ? b : 0;
        uint8_t *frames_dev = (uint8_t *)(fbuf ? ctx->frames2.p : ctx->frames.p);
        uint16_t *patches = (uint16_t *)(pb ? ctx->patches2.p : ctx->clipws[6].p);
        void *rects = reuse_bg ? (void *)((uint8_t *)ctx->rects.p + (size_t)pb * cap * 16) : nullptr;
        if (to_host && ci >= 2) ctx->pool->wait(b);                 // pinned buffer b: chunk ci-2's files are written
        {
            StreamSwap sw(ctx, rs);
            if (to_host && ci >= 2) D2R_HIP(ctx, hipStreamWaitEvent(rs, ctx->ev_copy[b], 0));   // frames buffer b has left the GPU
            // half pb of the patch AND rectangle buffers: chunk ci-2 is scored.  The wait stands BEFORE the render because ray
            // generation overwrites the rectangles that chunk's ViT (k_touch_list on the main stream) reads.
            if (two && ci >= 2) D2R_HIP(ctx, hipStreamWaitEvent(rs, ctx->ev_clip[pb], 0));
            if ((rc = d2r_launch_cameras_virtual(ctx, V, obj_pose_now, cam_pose, poses_dev + (size_t)c0 * 16, nc, (float *)ctx->cams.p)))
                return rc;
            if ((rc = d2r_launch_render(ctx, fg, V, (const float *)ctx->cams.p, nc, true, nullptr, nullptr, frames_dev, rects)))
                return rc;
            // keep this chunk's counters for the stats read-back at the end
            D2R_HIP(ctx, hipMemcpyAsync((uint8_t *)ctx->counters.p + 64 + 32 * (size_t)ci, ctx->counters.p, 32,
                                        hipMemcpyDeviceToDevice, rs));
            if (ctx->debug_fail_chunk >= 0 && (int64_t)ci == ctx->debug_fail_chunk) {        // fault injection (tests): as if a launch of this chunk had failed
                ctx->debug_fail_chunk = -1;                                                // one-shot: the context is not left failing every later call
                return d2r_fail(ctx, D2R_ERR_DEVICE, "injected fault in chunk " + std::to_string(ci) + " (option debug_fail_chunk)");
            }
            size_t tp = ctx->timing_begin(D2R_T_PREP);
            // (with layer-0 reuse only the touched patches are produced: nothing downstream reads the others)
            if ((rc = d2r_launch_preprocess(ctx, (d2r_clip *)clip, frames_dev, nc, V.W, V.H, 1, patches, nullptr, rects,
                                            reuse_bg ? (const uint16_t *)ctx->bg_patches.p : nullptr, l0.rects != nullptr)))
                return rc;
            ctx->timing_end(tp);
            if (to_host) {
                // the frames leave the GPU under the ViT, AFTER the preprocess has read them: the runtime copies device -> pinned host with a
                // blit kernel that holds wave slots at PCIe speed, and a preprocess launched beside it took as long as the copy (10.9 ms
                // instead of 0.4 for 3 170 frames of 336 x 336, profiles/r04c_api_kernel_stats.md before this order)
                D2R_HIP(ctx, hipEventRecord(ctx->ev_march[b], rs));
                D2R_HIP(ctx, hipStreamWaitEvent(xs, ctx->ev_march[b], 0));
                D2R_HIP(ctx, hipMemcpyAsync(ctx->frame_host[b], frames_dev, (size_t)nc * px * 3, hipMemcpyDeviceToHost, xs));
                D2R_HIP(ctx, hipEventRecord(ctx->ev_copy[b], xs));
            }
            if (two) D2R_HIP(ctx, hipEventRecord(ctx->ev_prep[pb], rs));
        }
        if (two) D2R_HIP(ctx, hipStreamWaitEvent(main, ctx->ev_prep[pb], 0));
        size_t tc = ctx->timing_begin(D2R_T_CLIP);
        if (l0.rects) {
            l0.rects = rects;                              // this chunk's half of the rectangle buffer
            l0.touched_out = (uint32_t *)((uint8_t *)ctx->counters.p + 64 + 32 * (size_t)ci) + 6;      // word 6 of the chunk's counter record
        }
        if ((rc = d2r_clip_forward(ctx, clip, patches, nc, (const float *)ctx->text.p, C, logit_scale,
                                   logits_dev + (size_t)c0 * C, nullptr, l0.rects ? &l0 : nullptr)))
            return rc;
        ctx->timing_end(tc);
        if (two) D2R_HIP(ctx, hipEventRecord(ctx->ev_clip[pb], main));
        if (to_host) {
            if (prev_b >= 0) {                 // the previous chunk's frames: on the host by now, or soon
                D2R_HIP(ctx, hipEventSynchronize(ctx->ev_copy[prev_b]));
                dispatch_frames(ctx, prev_b, prev_c0, prev_nc, V.W, V.H, frames_out, sink);
            }
            prev_b = b;
            prev_c0 = c0;
            prev_nc = nc;
        }
    }
    if (to_host) {
        const bool io_debug = getenv("D2R_IO_DEBUG") != nullptr;                  // development aid: where the frame files' time goes
        const auto t0 = std::chrono::steady_clock::now();
        auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        if (prev_b >= 0) {
            D2R_HIP(ctx, hipEventSynchronize(ctx->ev_copy[prev_b]));
            if (io_debug) fprintf(stderr, "[d2r io] last chunk's frames on the host after %.1f ms\n", ms());
            dispatch_frames(ctx, prev_b, prev_c0, prev_nc, V.W, V.H, frames_out, sink);
        }
        if (io_debug) {
            D2R_HIP(ctx, hipStreamSynchronize(main));
            fprintf(stderr, "[d2r io] device work done after %.1f ms\n", ms());
        }
        ctx->pool->wait(-1);
        if (io_debug) fprintf(stderr, "[d2r io] %u frames in %u chunks: files written %.1f ms after the last launch (%d threads)\n", K, nchunks, ms(), ctx->pool->size());
        std::string err;
        int prc = ctx->pool->take_error(err);
        D2R_HIP(ctx, hipStreamSynchronize(main));
        if (prc) return d2r_fail(ctx, prc, err);
    }
    ctx->stats.rays_total = (uint64_t)K * px;
    ctx->stats.l0_tokens = l0.rects ? (uint64_t)K * (d2r_clip_tokens(clip) - 1) : 0;
    return D2R_OK;
}

// render_score_body leaves at the first failing call, possibly with the render / copy streams forked off the context's stream
// and frame jobs in the worker pool.  Whatever happened, the context is handed back quiescent: every stream joined, the pool
// drained (its own error, if any, dropped in favour of the first one), the pipeline events reusable.
static int render_score_core(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_clip *clip, const d2r_view *view,
                             const float *obj_pose_now, const float *cam_pose, const float *poses_dev, uint32_t K,
                             uint32_t C, float logit_scale, float *logits_dev, uint8_t *frames_out, const d2r_frame_sink *sink)
{
    const int rc = render_score_body(ctx, fg, clip, view, obj_pose_now, cam_pose, poses_dev, K, C, logit_scale, logits_dev, frames_out, sink);
    if (rc == D2R_OK) return rc;
    const std::string first = ctx->err;
    if (ctx->render_stream) (void)hipStreamSynchronize(ctx->render_stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->pool) {
        ctx->pool->wait(-1);
        std::string dropped;
        (void)ctx->pool->take_error(dropped);
    }
    (void)hipGetLastError();          // a sticky launch error must not fail the next, unrelated call
    ctx->last_chunks = 0;             // the per-chunk counters are incomplete: d2r_collect_render_stats refuses them
    ctx->err = first;
    return rc;
}

static int check_render_score_args(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_clip *clip, const d2r_view *view,
                                   const float *obj_pose_now, const float *cam_pose, const void *poses, const void *logits,
                                   const d2r_frame_sink *sink)
{
    if (!ctx || !fg || !clip || !obj_pose_now || !cam_pose || !poses || !logits)
        return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    if (sink && sink->png_dir && (sink->png_level > 9 || sink->png_threads > 1024))
        return d2r_fail(ctx, D2R_ERR_INVALID, "bad d2r_frame_sink (png_level 0..9 or negative for the default, png_threads <= 1024)");
    return check_view(ctx, view);
}

int d2r_render_score(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_clip *clip, const d2r_view *view,
                     const float *obj_pose_now, const float *cam_pose, const float *obj_poses_dev, uint32_t K,
                     const float *text_embeds, uint32_t C, float logit_scale, float *logits_dev, uint8_t *frames_out)
{
    int rc = check_render_score_args(ctx, fg, clip, view, obj_pose_now, cam_pose, obj_poses_dev, logits_dev, nullptr);
    if (rc) return rc;
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = upload_text(ctx, clip, text_embeds, C))) return rc;
    return render_score_core(ctx, fg, clip, view, obj_pose_now, cam_pose, obj_poses_dev, K, C, logit_scale, logits_dev,
                             frames_out, nullptr);
}

int d2r_render_score_host(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_clip *clip, const d2r_view *view,
                          const float *obj_pose_now, const float *cam_pose, const float *obj_poses, uint32_t K,
                          const float *text_embeds, uint32_t C, float logit_scale, float *logits_out,
                          uint8_t *frames_out, const d2r_frame_sink *sink)
{
    int rc = check_render_score_args(ctx, fg, clip, view, obj_pose_now, cam_pose, obj_poses, logits_out, sink);
    if (rc) return rc;
    if (K == 0) return D2R_OK;
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = upload_text(ctx, clip, text_embeds, C))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->poses, (size_t)K * 64))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->logits, (size_t)K * C * 4))) return rc;
    D2R_HIP(ctx, hipMemcpyAsync(ctx->poses.p, obj_poses, (size_t)K * 64, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = render_score_core(ctx, fg, clip, view, obj_pose_now, cam_pose, (const float *)ctx->poses.p, K, C, logit_scale,
                                (float *)ctx->logits.p, frames_out, sink)))
        return rc;
    D2R_HIP(ctx, hipMemcpyAsync(logits_out, ctx->logits.p, (size_t)K * C * 4, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}

// Reads back the per-chunk counters of the last d2r_render_score (synchronises the stream).
int d2r_collect_render_stats(d2r_ctx *ctx, uint32_t K)
{
    if (!ctx) return d2r_fail(nullptr, D2R_ERR_INVALID, "null ctx");
    // K is kept in the signature for ABI stability; the chunk count is the one the last d2r_render_score stored
    (void)K;
    const uint32_t nchunks = ctx->last_chunks;
    if (nchunks == 0 || !ctx->counters.p || 64 + 32 * (size_t)nchunks > ctx->counters.cap)
        return d2r_fail(ctx, D2R_ERR_INVALID, "d2r_collect_render_stats needs a preceding d2r_render_score on this context");
    std::vector<uint32_t> c((size_t)nchunks * 8);
    D2R_HIP(ctx, hipMemcpyAsync(c.data(), (uint8_t *)ctx->counters.p + 64, c.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.rays_alive = 0;
    ctx->stats.samples = 0;
    ctx->stats.wave_iters = 0;
    ctx->stats.l0_touched = 0;
    for (uint32_t i = 0; i < nchunks; i++) {
        if (ctx->stats.l0_tokens) ctx->stats.l0_touched += c[i * 8 + 6];
        uint64_t s, it;
        memcpy(&s, &c[i * 8 + 2], 8);
        memcpy(&it, &c[i * 8 + 4], 8);
        ctx->stats.rays_alive += c[i * 8];
        ctx->stats.samples += s;
        ctx->stats.wave_iters += it;
    }
    return D2R_OK;
}

}  // extern "C"
