// Internal declarations shared by the HIP translation units of libd2r.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/d2r.h"

class D2rJobPool;   // pngio.h
struct D2rPngBase;  // pngio.h

#define D2R_MAX_LEVELS 16
#define D2R_MAX_DEVICES 64              // per-device "kernel attribute already set" flags
#define D2R_GRID 128
#define D2R_DT 0.0016914558f           // sqrt(3)/1024
#define D2R_INV_DT (1.0f / D2R_DT)
// aabb_scale 2 (oracle/d2r_oracle.c "cone stepping"): t_{k+1} = t_k + max(dt, t_k/256) in closed form
#define D2R_CONE 0.00390625f
// D2R_T_LINEAR: the distance at which steps start to grow, dt_min / cone as in the per-step rule dt = max(dt_min, t * cone).  Newer
// instant-ngp revisions step in an analytic log space and switch at dt_min / log1p(cone) (~256.5 dt_min; lattices agree to ~0.2 % of a
// step).  The reference's pinned commit is unknown: if a real snapshot ever disagrees, change this line and oracle/d2r_oracle.c's twin.
#define D2R_T_LINEAR (D2R_DT * 256.0f)
#define D2R_N_WFRAG 24                 // MLP weight fragments (see nerf.hip)

struct LevelMeta {
    float scale;
    uint32_t res;
    uint32_t size;
    uint32_t offset;
    uint32_t hashed;
};

// One "slot" = the pair of hash-grid levels (2i, 2i+1) that the two lanes (l, l^32) of a lane
// pair evaluate.  Their tables are stored INTERLEAVED in HBM: entry e of level 2i+h lives at
// byte  slot_off + 8*e + 4*h, so a corner address is (index << 3) | (h << 2) plus an SGPR offset.
struct SlotMeta {
    float scale[2];
    uint32_t res[2];
    uint32_t size[2];
    uint32_t hashed[2];
    uint32_t mask8;      // (hash table size - 1) << 3 | 4   (hashed levels share one size)
    uint32_t off;        // byte offset of the slot's interleaved table
    // bounding-box-local dense brick of each level (served from LDS when the slot is bricked):
    // vertex (gx,gy,gz) at LDS word  bbase + gx + bnx*gy + bnxy*gz
    uint32_t bnx[2], bnxy[2];
    int32_t bbase[2];
};

struct NerfParams {
    const uint32_t *grid;      // interleaved slot tables, half2 entries
    uint32_t grid_bytes;
    uint32_t n_levels;         // 16 (F = 2: a slot is the level pair 2i, 2i+1) or 8 (F = 4: a slot is level i, its two
                               // "halves" the feature pairs (0,1) and (2,3) of the 8-byte entries) — either way entry e
                               // of half h lives at byte 8e + 4h of the slot table and network input 4*slot + 2h + f
    int32_t n_dense;           // leading dense HALF-levels (2 per level when F = 4; slot kinds follow from it), -1 = irregular
    SlotMeta slot[D2R_MAX_LEVELS / 2];
    uint32_t refill_min;       // free lanes in a wave before it pulls new rays from the queue
    uint32_t compact;          // 1: a wave with <= 32 rays left moves them to lanes 0..31 (tile 1 then costs nothing)
    uint32_t sort_log2;        // ray sort: 0 off, else log2 of the bins per axis of the occupied box (2: 4x4x4 ... 4: 16x16x16); the bin of a ray's first sample rides in bits 20.. of its queue entry's k
    uint32_t n_brick_slots;    // leading slots whose levels are de-hashed into LDS bricks (0 .. 5)
    uint32_t brick_words;      // total words of those bricks
    const uint32_t *brick_tab; // [brick_words] half2 entries, copied to LDS by every workgroup
    uint32_t n_gbrick_slots;   // further slots de-hashed into dense bricks kept in HBM (spatially coherent)
    uint32_t gbrick_bytes;
    const uint32_t *gbrick_tab;
    const uint64_t *bricks;    // [n_cascades][32^3] 4x4x4-cell occupancy bricks
    uint32_t aabb_scale;       // 1, or a power of two up to 128: log2 + 1 cascades, cone stepping, positions normalised to the box (k_*<.., CONE>)
    uint32_t n_casc;
    float side, inv_side;      // aabb_scale and its reciprocal as floats
    const uint4 *wfrag;        // [24][64] MFMA A-operand fragments of the MLPs, bf16
    const uint4 *wfrag16;      // the same fragments holding the snapshot's fp16 weights unrounded (option mlp_f16)
    float bbox_lo[3], bbox_hi[3];  // bounding box of occupied cells (+margin), unit-cube units
    // Testbed.render_aabb clipped to the model's box: in ngp coordinates (ray slab test) and in the unit cube of
    // the box (per-sample containment test); the whole box when none is set
    float raabb_lo[3], raabb_hi[3], rn_lo[3], rn_hi[3];
};

struct ViewParams {
    uint32_t W, H;
    float focal[2], center[2];
    float scale, inv_scale;
    float offset[3];
    float background[4];
    float min_transmittance, near_distance;
    uint32_t lens_mode;        // D2R_LENS_*: 1 = every ray's camera-space direction is undistorted (lens_undistort, nerf.hip)
    float lens[4];             // OpenCV k1, k2, p1, p2
    const float2 *lens_tab;    // [H][W] undistorted (x, y) of the camera-space direction per pixel (k_lens_table), nullptr = perspective
};

struct ClipParams;   // clip.hip

// Layer-0 reuse of background tokens inside d2r_render_score (clip.hip "layer-0 reuse"): the frame rectangles the candidates
// of the chunk touched, and the background's own pre-LayerNorm residual rows and layer-0 q / k / v rows (d2r_clip_layer0_background)
struct ClipL0Reuse {
    const void *rects;           // device int4 [n]: x0, y0, x1, y1 in frame pixels, inclusive (k_raygen_rect)
    uint32_t w, h;               // frame size
    const uint16_t *bg_hi;       // [d / 64][bg_rows][64] bf16
    const uint8_t *bg_lo;        // lo bytes, lo8_off(bg_rows, ..) layout
    const float2 *bg_ab;         // [bg_rows] (rstd, -rstd * mean)
    const uint16_t *bg_qkv;      // [3 d / 64][bg_rows][64] bf16
    uint32_t bg_rows;            // tokens per image, rounded up to a multiple of 256
    uint32_t *touched_out;       // device, optional: receives the chunk's touched-token count (render statistics)
};

struct d2r_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    // growable device workspaces (one per role so sizes are independent)
    struct Buf { void *p = nullptr; size_t cap = 0; };
    Buf queue2, sort_counts;   // ray sort: the sorted queue, per-chunk bin counts / offsets
    Buf cams, queue, counters, frames, rgba, depth, poses, clipws[8], text, logits, pix;
    // host -> device uploads of small caller buffers on asynchronous entry points (text embeddings of d2r_render_score)
    // go through two library-owned pinned slots, so the caller's memory is consumed before the call returns whatever
    // the runtime does with pageable copies; an event per slot guards its reuse
    void *text_host[2] = {nullptr, nullptr};
    size_t text_host_cap[2] = {0, 0};
    hipEvent_t text_ev[2] = {nullptr, nullptr};
    uint32_t text_turn = 0;
    // background of the current view
    Buf bg_rgba, bg_depth, bg_u8;
    Buf lens_tab;                // undistorted camera-space directions of the current view's pixels (nerf.hip lens_table)
    float lens_key[10] = {};     // ... and the view (size, intrinsics, coefficients) it was computed for
    bool lens_key_valid = false;
    Buf rect_ws;                 // d2r_rectify_background_depth: source images, tap tables, outputs
    Buf rects;                   // per candidate of a pass: frame rectangle (x0, y0, x1, y1) its rays were generated in
    Buf bg_patches;              // CLIP patches of the background frame itself (one image)
    const void *bg_patches_for = nullptr;   // the d2r_clip they were computed with (nullptr = stale)
    Buf bg_l0, l0_a1, l0_q2, l0_misc;       // layer-0 reuse: the background's rows; gathered patch rows; compact q / k / v; count + list + pairs
    const void *bg_l0_for = nullptr;        // the d2r_clip bg_l0 was computed with (nullptr = stale)
    ClipL0Reuse bg_l0_desc{};               // pointers into bg_l0
    int64_t vit_fp8 = 0;         // vision tower: the Linear products of the blocks between the first and the last on the MX-scaled fp8 MFMA (clip.hip, "fp8 blocks")
    int64_t l0_reuse = 1;        // d2r_render_score: patch embedding + layer-0 QKV on the touched tokens only
    int64_t prep_reuse = 1;      // k_preprocess copies the background's patch rows for bands a candidate cannot have touched
    uint32_t bg_w = 0, bg_h = 0;
    d2r_render_stats stats{};
    // device geometry (hipDeviceProp_t / hipDeviceAttributeNumberOfXccs): persistent kernels launch one
    // workgroup per CU and map tiles to XCDs (one L2 each)
    int n_cu = 256, n_xcd = 8;
    // pose-shard communicator (comm.hip): an ncclComm_t, or null at world size 1
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    // the render-and-score pipeline (api.hip render_score_core): the render half of chunk i+1 runs on render_stream while
    // the ViT scores chunk i on `stream`; frames that leave the GPU go through copy_stream into two pinned buffers and
    // from there to the worker pool (PNG files, the caller's array)
    hipStream_t render_stream = nullptr, copy_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_prep[2] = {nullptr, nullptr}, ev_clip[2] = {nullptr, nullptr},
               ev_march[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
    Buf patches2, frames2;
    void *frame_host[2] = {nullptr, nullptr};
    size_t frame_host_cap[2] = {0, 0};
    D2rJobPool *pool = nullptr;
    std::shared_ptr<const D2rPngBase> png_base;     // the current background's scanlines, entropy-coded once for the frame files (reset by d2r_set_background)
    int64_t overlap = 0;        // 1: render half of chunk i+1 on render_stream under the ViT of chunk i (measured neutral: both sides fill whole CUs); 0: program order on `stream`
    int64_t debug_fail_chunk = -1;   // fault injection for the error path of render_score_core (tests/test_api_path.py)
    uint32_t last_chunks = 0;   // chunks of the last d2r_render_score (its per-chunk counters are behind counters+64)
    int64_t chunk = 4096;       // candidates per pass (capped per model/view by pass_size() in api.hip)
    uint32_t last_pass = 0;     // pass size the last d2r_render_score used (for its stats read-back)
    int64_t march_blocks = 0;  // 0 = auto
    int64_t ray_sort = 1;      // 1: the ray queue is sorted by the object region (Morton cell of the occupied box) a ray's first sample lies in before it is marched; 0: marched in generation order
    int64_t ray_sort_log2 = 4; // cells per axis = 2^this (1..4); measured 8^3 -> 16^3: another 6 %
    int64_t march_threads = 0; // threads per marcher workgroup: 0 = auto (768, or 512 when the HBM bricks exceed march_threads_auto_mib MiB), else 64 .. 768 in steps of 64
    int64_t march_threads_auto_mib = 64;   // apple 36 MB (issue-bound: 768 threads are 8 % faster), shelf 112 MB, 2.2x / 5x apple 394 / 89 MB (L2-miss-bound: 512 are 1-5 % faster)
    uint32_t last_march_threads = 0, last_march_gbrick_bytes = 0;   // and the workgroup size / HBM-brick bytes of that launch
    uint32_t last_march_nb = 0, last_march_ngb = 0;   // brick configuration the last march launch ran with (d2r_get_render_stats)
    int64_t refill_min = 64;   // measured on MI355X: a refill (queue + camera loads, ray setup, SH) costs several iterations,
                               // so a wave runs its 64 rays to the end (lane utilisation 0.79) rather than topping up at 16 free lanes (0.90)
    int64_t march_compact = 1; // marcher: compact a wave's last <= 32 rays into one tile (option "march_compact")
    int64_t ln_fold = 4;       // vision tower: 0 LayerNorm kernels + fp32 residual; LayerNorm folded into the GEMMs with 1 a split (hi + lo) bf16 residual, 2 a bf16 residual, 3 an fp32 residual + bf16 copy, 4 bf16 hi + one lo byte
    int64_t gemm_stagger = 0;      // persistent GEMM: stagger the workgroups' first tile over a tile period (epilogues spread in time)
    int64_t gemm_group = 65535;    // persistent GEMM: column tiles per group of the tile order (large = the plain column-fastest order, the default; 0 = the width a simple L2 model picks: a third fewer L2 misses, same time)
    int64_t gemm_nsplit = 0;     // column sections of the persistent GEMM's tile order (XCD sets own column ranges): 0 = two where the XCDs and the column tiles divide evenly, 1 = none
    int64_t attn_rem = 1;          // attention: a remainder of at most this many query tiles (sequence = 8 g + r tiles) runs on workgroups of r waves instead of one more eight-wave group
    int64_t cls_last = 1;          // vision tower: run the last block on the class-token rows only (the head reads nothing else)
    int64_t gemm_cfg = 0;      // experiment switch for the GEMM tile configuration (0 = default)
    int64_t mlp_f16 = 1;       // operand type of the NeRF MLPs' MFMAs: 1 (default since round 6) fp16 — the reference's operand type (tiny-cuda-nn), the snapshot's weights unrounded; 0 bf16 (north_star's wording).  Same MFMA rate; chosen by the distance table of tests/test_gpu_parity.py::test_render_distances_to_the_fp16_accumulation_emulation
    int64_t use_bricks = 1;
    int64_t raygen_rect = 1;   // composite mode: generate rays only inside the projected occupied bbox
    int64_t gbrick_slots = 8;  // at most this many slots use HBM bricks
    int64_t brick_slots_total = 7;   // ... and none at or beyond this slot index: bricking the finest slot (levels 14-15) measured slower (cache misses)
    int64_t lds_slots_max = 5;       // d2r_nerf_create: at most this many leading slots as LDS bricks (experiments: the marcher's regimes)
    int64_t gbrick_max_mib = 512;    // d2r_nerf_create: a slot is HBM-bricked only while its brick stays below this size (512 = only the 512 MiB total bounds them: measured, a 5x apple renders 1.45x faster on its 100+ MiB bricks than on the 4 MiB hashed tables, profiles/r05_march_regimes.md)
    // optional per-kernel timing (HIP events on the launch stream), see d2r_get_timing
    int64_t timing = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<int, std::pair<size_t, size_t>>> ev_pairs;   // (kind, (begin, end))
    size_t timing_begin(int kind);          // returns pair index, records the begin event (kinds >= D2R_T_VIT_QKV only with "timing" 2)
    void timing_end(size_t pair);
};
enum { D2R_T_MARCH = 0, D2R_T_RAYGEN = 1, D2R_T_CLIP = 2, D2R_T_PREP = 3, D2R_T_SORT = 4,
       // "timing" 2: the vision tower's products, one pair of events per launch (full-size launches of the default schedule only:
       // layer 0's compact QKV under l0_reuse and the class-token-only last block stay inside D2R_T_CLIP's remainder)
       D2R_T_VIT_QKV = 5, D2R_T_VIT_ATTN = 6, D2R_T_VIT_OUT = 7, D2R_T_VIT_FC1 = 8, D2R_T_VIT_FC2 = 9, D2R_T_KINDS = 10 };

struct d2r_nerf {
    d2r_ctx *ctx;
    NerfParams P{};
    void *d_grid = nullptr, *d_bricks = nullptr, *d_wfrag = nullptr, *d_brick_tab = nullptr, *d_gbrick_tab = nullptr;
};

// hipFuncSetAttribute once per (kernel instantiation, device), safe when different contexts are driven from
// different threads: one static instance per call site
struct PerDeviceOnce {
    std::once_flag flag[D2R_MAX_DEVICES];
    template <class F>
    void run(int device, F &&fn) { std::call_once(flag[device % D2R_MAX_DEVICES], fn); }
};

int d2r_fail(d2r_ctx *ctx, int code, const std::string &msg);
int d2r_reserve(d2r_ctx *ctx, d2r_ctx::Buf &b, size_t bytes);

#define D2R_HIP(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return d2r_fail(ctx, D2R_ERR_DEVICE,                                        \
                            std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

// nerf.hip launchers
int d2r_launch_cameras_direct(d2r_ctx *, const ViewParams &, const float *cams_nerf_dev, uint32_t n,
                              float *cams_out);
int d2r_launch_cameras_virtual(d2r_ctx *, const ViewParams &, const float *obj_now16,
                               const float *cam16, const float *obj_poses_dev, uint32_t n,
                               float *cams_out);
int d2r_launch_render(d2r_ctx *, const d2r_nerf *, const ViewParams &, const float *cams_dev,
                      uint32_t n, bool composite, float *rgba_dev, float *depth_dev,
                      uint8_t *frames_dev, void *rects_dev = nullptr);
int d2r_launch_bg_quantize(d2r_ctx *, uint32_t w, uint32_t h);
ViewParams d2r_view_params(const d2r_view *v);
int d2r_reserve_render(d2r_ctx *, size_t rays);    // ray queue (+ sorted copy and bin counts with the ray sort) for a pass of `rays` rays
int d2r_lens_table(d2r_ctx *, ViewParams &V);      // attaches the view's undistorted-direction table (built on first use)
