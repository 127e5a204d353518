/*
 * d2r.h — C ABI of libd2r.so: the MI355X (gfx950) render-and-score path of Dream2Real.
 *
 * The reference has no C ABI in tree; its two native seams on this path are Python
 * extension surfaces:
 *   (1) pyngp.Testbed  — load_snapshot / set_camera_to_training_view /
 *       set_nerf_camera_matrix / background_color / render_mode / render(w,h,spp,linear)
 *       used at reference reconstruction/combined_rendering.py:98-105,112-113,116,123-130
 *       and reconstruction/ngp_visual_model.py:24-28;
 *   (2) transformers.CLIPModel / CLIPProcessor used at reference clip_scoring.py:150-151,
 *       177-181.
 * Each entry point below names the reference interface it replaces.  The Python host
 * (dream2real_amd/) binds these with ctypes; INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative d2r_status; the message is
 *     available from d2r_last_error(ctx) (or d2r_last_error(NULL) for create failures);
 *   - no C++ exception crosses the ABI, nothing calls abort();
 *   - "host" pointers are read/written before the call returns; "dev" pointers are HIP
 *     device pointers on the context's device, used on the context's stream;
 *   - the library owns its handles; parameter blobs are copied to the device at create;
 *   - one context per GPU; calls on one context must be serialised by the caller;
 *   - there is NO CPU fallback: every compute entry point needs a gfx950 device.
 */
#ifndef D2R_H
#define D2R_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2R_API __attribute__((visibility("default")))
#define D2R_ABI_VERSION 9

typedef enum {
    D2R_OK = 0,
    D2R_ERR_INVALID = -1,   /* bad argument */
    D2R_ERR_DEVICE = -2,    /* HIP error, or no gfx950 device */
    D2R_ERR_MEMORY = -3,    /* allocation failed */
    D2R_ERR_UNSUPPORTED = -4
} d2r_status;

typedef struct d2r_ctx d2r_ctx;
typedef struct d2r_nerf d2r_nerf;
typedef struct d2r_clip d2r_clip;

/* ------------------------------------------------------------------ context */

D2R_API int d2r_abi_version(void);
/* One context per GPU.  Owns a HIP stream and a workspace that grows on demand. */
D2R_API int d2r_ctx_create(int device, d2r_ctx **out);
D2R_API void d2r_ctx_destroy(d2r_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the
 * context's own stream. */
D2R_API int d2r_ctx_set_stream(d2r_ctx *ctx, void *hip_stream);
D2R_API int d2r_ctx_synchronize(d2r_ctx *ctx);
D2R_API const char *d2r_last_error(d2r_ctx *ctx);

/* --------------------------------------------------------------- NeRF model */

/* State a pyngp.Testbed holds after load_snapshot (reference ngp_visual_model.py:24-28):
 * tiny-cuda-nn hash grid + density/colour MLPs + 128^3 occupancy bitfield(s). */
typedef struct {
    uint32_t n_levels;          /* L: 16, or 8 */
    uint32_t n_features;        /* F: 2 with L = 16, 4 with L = 8 (the two layouts with L*F = 32 inputs) */
    const float *level_scale;   /* host [L]: exp2(l*log2(b))*N_min - 1 */
    const uint32_t *level_res;  /* host [L]: ceil(scale)+1 */
    const uint32_t *level_size; /* host [L]: entries per level */
    const uint32_t *level_offset; /* host [L]: first entry of each level */
    uint32_t n_entries;         /* sum(level_size) */
    const uint16_t *grid_fp16;  /* host [n_entries][F] */
    const uint16_t *dw1_fp16;   /* host [64][L*F]  density layer 1, row-major [out][in] */
    const uint16_t *dw2_fp16;   /* host [16][64]   density layer 2 */
    const uint16_t *cw1_fp16;   /* host [64][32]   colour layer 1, in = [density out 16 | SH 16] */
    const uint16_t *cw2_fp16;   /* host [64][64] */
    const uint16_t *cw3_fp16;   /* host [16][64]   rows 0..2 = rgb */
    const uint8_t *occupancy_bits; /* host [n_cascades][128^3/8], bit x+128*(y+128*z), LSB first */
    uint32_t aabb_scale;        /* a power of two, 1..128 (0 is read as 1): the model's box is the cube of that side
                                 * centred at 0.5; n_cascades = log2(aabb_scale)+1 occupancy grids, cascade c covering
                                 * side 2^c; aabb_scale >= 2 marches with cone-angle 1/256 steps */
    float render_aabb[6];       /* Testbed.render_aabb: lo xyz, hi xyz in ngp coordinates — rays start where they
                                 * enter it and stop where they leave it; all zeros = the model's whole box */
} d2r_nerf_desc;

/* replaces Testbed(mode=Nerf) + load_snapshot */
D2R_API int d2r_nerf_create(d2r_ctx *ctx, const d2r_nerf_desc *desc, d2r_nerf **out);
D2R_API void d2r_nerf_destroy(d2r_nerf *m);

/*
 * replaces Testbed(mode=Nerf) + load_snapshot(path) on the bytes of an instant-ngp `.ingp` file (reference
 * reconstruction/ngp_visual_model.py:24-28): zlib/gzip msgpack -> level table, fp16 tables and MLPs, occupancy
 * bitfield (instant-ngp's threshold rule and cascade max-pool), render_aabb -> d2r_nerf_create.  `info` / `views`
 * (optional) receive what a Testbed keeps beside the model: dataset scale/offset for nerf_matrix_to_ngp, the saved
 * background colour, per-training-view intrinsics AND lens for set_camera_to_training_view.
 * The layout is the BELIEVED one (no instant-ngp file or source offline), so nothing is defaulted: every key the loader
 * needs must be there with the right msgpack kind and size (params_binary: fp16, density MLP | colour MLP | hash tables;
 * density_grid_binary: fp16 128^3 per cascade), every key that changes the rendered function must carry the value the
 * kernels implement (activations, interpolation, SH degree, exposure 0, no envmap / extra dims / rotated crop box, a
 * perspective or OpenCV lens per view, cone angle tied to aabb_scale), unknown keys inside encoding / network / rgb_network / dir_encoding
 * are refused; each failure returns D2R_ERR_INVALID / D2R_ERR_UNSUPPORTED with a d2r_last_error message naming the key.
 */
/* Lens of a training view (instant-ngp's Lens {mode, params}) as this path meets it.  The reference renders every frame after
 * set_camera_to_training_view (reconstruction/combined_rendering.py:98,116), which switches
 * nerf.render_with_lens_distortion on and makes the view's lens the render lens; its configs carry OpenCV coefficients
 * (configs/shopping_demo.json:51-56 -> the transforms the NeRFs are trained from, reconstruction/train_ngp.py:171-180,
 * utils/accio2ngp.py:47-56; train_ngp.py:70 sets the flag as well).  Other lens models (fisheye, f-theta, lat-long,
 * equirectangular, orthographic) are refused by the loader with D2R_ERR_UNSUPPORTED naming the view. */
enum { D2R_LENS_PERSPECTIVE = 0, D2R_LENS_OPENCV = 1 };
typedef struct {
    double fx, fy, cx, cy;   /* pixels at the training resolution */
    uint32_t w, h;
    uint32_t lens_mode;      /* D2R_LENS_*: metadata[].lens (no lens / an empty one = perspective) */
    float lens_params[4];    /* OpenCV k1, k2, p1, p2 */
} d2r_ingp_view;
typedef struct {
    uint32_t n_levels, n_features, aabb_scale;
    int32_t has_background;
    double dataset_scale, dataset_offset[3];
    float background_color[4];
    uint32_t n_views;            /* training views in the snapshot */
    uint32_t n_views_written;    /* how many of them went into `views` (at most views_cap) */
    uint32_t n_unknown_keys;     /* keys of the snapshot the loader neither reads, checks nor knows to be irrelevant to
                                  * rendering (d2r_ingp_inspect lists them with a '?') */
    int32_t render_with_lens_distortion;   /* snapshot.nerf.render_with_lens_distortion when the file carries it (else 0): the
                                  * Testbed's flag BEFORE the first set_camera_to_training_view, which sets it to 1 (ABI 9) */
} d2r_ingp_info;
D2R_API int d2r_nerf_load_ingp(d2r_ctx *ctx, const void *bytes, size_t len, d2r_nerf **out, d2r_ingp_info *info,
                               d2r_ingp_view *views, uint32_t views_cap);
/*
 * What a snapshot holds next to what d2r_nerf_load_ingp reads of it — the first check to run on a real
 * `method_out/<scene>/{fg,bg}_base.ingp` (reference install.sh:38-50), since the loader is written against the believed
 * layout.  HOST ONLY: needs no device and no context (errors go to d2r_last_error(NULL)).  Writes NUL-terminated text
 * into out[cap] (truncated if short; *needed = bytes for all of it): one line per leaf of the msgpack tree,
 * "<R|C|-|?> <kind> <elements or bytes> <path> [= value]" with R = read by the loader, C = checked (changes the rendered
 * function; one value implemented, anything else is a load error), - = known not to affect rendering (training state,
 * GUI camera, ...), ? = unknown to the loader; then "# unknown_keys N" and "# derived:" lines with the
 * parameter / density-grid counts and the level table the loader computes from the config, beside the sizes of
 * params_binary / density_grid_binary.  Malformed or truncated input returns D2R_ERR_INVALID.
 */
D2R_API int d2r_ingp_inspect(const void *bytes, size_t len, char *out, size_t cap, size_t *needed);
/* HOST ONLY: every check d2r_nerf_load_ingp makes on the bytes, without a device and without creating a model — 0 when the
 * loader would take the snapshot, else the loader's error (message in d2r_last_error(NULL), naming the key).  info
 * (optional) is filled as the loader would fill it. */
D2R_API int d2r_ingp_validate(const void *bytes, size_t len, d2r_ingp_info *info);

/* Camera state set on a Testbed before render(): set_camera_to_training_view (intrinsics),
 * background_color, nerf.render_min_transmittance, dataset scale/offset used by
 * set_nerf_camera_matrix (reference combined_rendering.py:98-105,116,123-127). */
typedef struct {
    uint32_t width, height;
    float focal[2];          /* pixels at this render size (rel. focal length * height) */
    float center[2];         /* principal point, relative (cx/w, cy/h) */
    float scale;             /* dataset scale */
    float offset[3];         /* dataset offset */
    float background[4];     /* Testbed.background_color RGBA */
    float min_transmittance; /* 0.01 */
    float near_distance;     /* 0 */
    /* ABI 9 — the render lens: what set_camera_to_training_view leaves in nerf.render_lens when
     * nerf.render_with_lens_distortion is on (D2R_LENS_PERSPECTIVE otherwise).  With D2R_LENS_OPENCV every ray's camera-space
     * direction (x, y, 1) goes through instant-ngp's iterative OpenCV undistortion (Newton steps on a central-difference
     * Jacobian, at most 100, |step|^2 < 1e-10) before the camera rotation; the rectangle cull of the composite passes stays
     * conservative under it (frames are bit-identical with "raygen_rect" 0 and 1). */
    uint32_t lens_mode;
    float lens_params[4];    /* k1, k2, p1, p2 */
} d2r_view;

/*
 * replaces n x { set_nerf_camera_matrix(cam); render_mode=Shade; render(w,h,1,True);
 *                render_mode=Depth; render(w,h,1,True) }   (combined_rendering.py:123-130)
 * in one pass.  cams_nerf: host [n][12], row-major 3x4, the matrix handed to
 * set_nerf_camera_matrix.  rgba_out: host [n][h][w][4] Shade frames; depth_out: host
 * [n][h][w] channel 0 of the Depth frames.  Either output may be NULL.
 * n_samples_out (optional): network evaluations performed.
 */
D2R_API int d2r_render(d2r_ctx *ctx, const d2r_nerf *model, const d2r_view *view,
                       const float *cams_nerf, uint32_t n, float *rgba_out, float *depth_out,
                       uint64_t *n_samples_out);

/* Parity hook for the lens: the camera-space direction (x, y) — z = 1 — of every pixel centre of `view` after the iterative
 * OpenCV undistortion, i.e. what the ray generators rotate by the camera matrix.  dirs_out: host [h][w][2].  The view must carry
 * a lens (lens_mode D2R_LENS_OPENCV).  (ABI 9) */
D2R_API int d2r_lens_undistort_view(d2r_ctx *ctx, const d2r_view *view, float *dirs_out);

/* Field evaluation at arbitrary unit-cube points (parity hook for the hash-grid encode and
 * both MLPs): xyz, dirs host [n][3] (dirs unit length); out host [n][4] = sigma, r, g, b
 * with rgb the network's sRGB-space prediction. */
D2R_API int d2r_nerf_eval_points(d2r_ctx *ctx, const d2r_nerf *model, const float *xyz,
                                 const float *dirs, uint32_t n, float *sigma_rgb_out);

/* ------------------------------------------------- background + compositing */

/* Fixes the per-view background the candidates are composited over: bg_image / bg_depth
 * of reference combined_rendering.py:105-113 (host [h][w][4] and [h][w]; depth is
 * channel 0, already rectified/masked if it came from depths_gt). */
D2R_API int d2r_set_background(d2r_ctx *ctx, const d2r_view *view, const float *bg_rgba,
                               const float *bg_depth);

/*
 * Sensor-depth background of a render view (reference combined_rendering.py:107-110 with rectify_depth :166-187 and
 * rectify_mask :189-209): depth (host, [src_h][src_w] metres, fp32 or fp16 as data_loader.py:43,58 stores it) and
 * the movable-object mask (host uint8 [src_h][src_w], 0 / 1; NULL = no masking) are centre-cropped to a square and
 * resized to w x h as cv2.resize(..., interpolation=cv2.INTER_CUBIC) does (float path for the depth, 8-bit
 * fixed-point path for the mask); where the resized mask is 0 the depth becomes 100.  depth_out: host [h][w] fp32,
 * what d2r_set_background takes as bg_depth; mask_out (optional): host [h][w] uint8, the resized mask.
 */
D2R_API int d2r_rectify_background_depth(d2r_ctx *ctx, const void *depth, int depth_is_fp16, const uint8_t *mask,
                                         uint32_t src_w, uint32_t src_h, uint32_t w, uint32_t h, float *depth_out,
                                         uint8_t *mask_out);

/*
 * replaces the body of renderer.render's loop over valid poses
 * (reference combined_rendering.py:118-155 with convert_virtual_pose :250-263):
 *   obj_pose_now  host [16]    T_WO_1, row-major 4x4, NGP convention (after converter)
 *   cam_pose      host [16]    T_WC_1 of the render view, NGP convention
 *   obj_poses     host [K][16] candidate poses T_WO_2, NGP convention
 *   frames_out    host [K][h][w][3] uint8 composited sRGB frames
 */
D2R_API int d2r_render_composite(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_view *view,
                                 const float *obj_pose_now, const float *cam_pose,
                                 const float *obj_poses, uint32_t K, uint8_t *frames_out);

/* -------------------------------------------------------------------- CLIP */

typedef struct {
    uint32_t image_size;   /* S: 224 / 336 */
    uint32_t patch_size;   /* P: 16 / 14 */
    uint32_t hidden_size;  /* d: 768 / 1024 */
    uint32_t num_layers;   /* 12 / 24 */
    uint32_t num_heads;    /* 12 / 16 (head dim must be 64) */
    uint32_t mlp_size;     /* 3072 / 4096 */
    uint32_t proj_dim;     /* D: 512 / 768 */
} d2r_clip_desc;

/*
 * replaces CLIPModel.from_pretrained(...).to(device) (vision tower + visual projection,
 * reference clip_scoring.py:150).  weights: host fp32 blob, tensors concatenated in this
 * order (Hugging Face names; Linear weights row-major [out][in]):
 *   patch_embedding.weight [d][3*P*P], class_embedding [d], position_embedding [T][d],
 *   pre_layrnorm.{weight,bias},
 *   per layer: layer_norm1.{weight,bias}, q_proj.{weight,bias}, k_proj.{weight,bias},
 *              v_proj.{weight,bias}, out_proj.{weight,bias}, layer_norm2.{weight,bias},
 *              fc1.{weight,bias}, fc2.{weight,bias},
 *   post_layernorm.{weight,bias}, visual_projection.weight [D][d]
 * n_floats must equal the size this order implies.
 */
D2R_API int d2r_clip_create(d2r_ctx *ctx, const d2r_clip_desc *desc, const float *weights,
                            size_t n_floats, d2r_clip **out);
D2R_API void d2r_clip_destroy(d2r_clip *clip);

/*
 * replaces np.rot90 (clip_scoring.py:145) + CLIPProcessor(images=...) (:177) +
 * CLIPModel vision forward and logits_per_image (:180-181) for frames already on the host:
 *   frames       host [n][h][w][3] uint8
 *   rot90        non-zero: rotate each frame 90 degrees CCW first
 *   text_embeds  host [C][D], L2-normalised text embeddings (cached, computed once)
 *   logit_scale  exp(logit_scale) of the checkpoint (100 for OpenAI CLIP)
 *   logits_out   host [n][C]   logits_per_image
 *   embeds_out   host [n][D]   optional, L2-normalised image embeddings
 */
D2R_API int d2r_clip_score_frames(d2r_ctx *ctx, const d2r_clip *clip, const uint8_t *frames,
                                  uint32_t n, uint32_t w, uint32_t h, int rot90,
                                  const float *text_embeds, uint32_t C, float logit_scale,
                                  float *logits_out, float *embeds_out);

/* CLIPProcessor(images=...) alone: pixel_values host [n][3][S][S] fp32 (parity hook). */
D2R_API int d2r_clip_preprocess(d2r_ctx *ctx, const d2r_clip *clip, const uint8_t *frames,
                                uint32_t n, uint32_t w, uint32_t h, int rot90,
                                float *pixel_values_out);

/* CLIPModel vision forward on given pixel_values host [n][3][S][S] -> embeds [n][D]
 * (parity hook for the ViT alone). */
D2R_API int d2r_clip_embed_pixels(d2r_ctx *ctx, const d2r_clip *clip, const float *pixel_values,
                                  uint32_t n, float *embeds_out);

/*
 * Parity hook for the vision tower's fp8 mode (option "vit_fp8", below): ONE product C = q(A) q(W)^T s_w + bias through the kernels
 * the mode uses — activations as OCP e4m3 with one E8M0 scale byte per (row, 64 columns), the bf16-rounded weights as e4m3 with one
 * power-of-two scale per matrix, fp32 accumulation on v_mfma_scale_f32_32x32x64_f8f6f4 (oracle/clip_fp8.py restates the format).
 *   A host [M][K], W host [N][K], bias host [N] fp32; N and K multiples of 256
 *   kind 0: out [M][N] = the bf16 result; kind 1: out = quick_gelu(...) quantised to e4m3 per (row, 64 columns), dequantised
 *   a_q8 [M][K] / a_scales [M][K/64] (optional): the quantised A;  w_scale_out (optional): s_w
 *   out_q8 [M][N] / out_scales [M][N/64] (kind 1, optional): the raw bytes of the result
 */
D2R_API int d2r_debug_gemm_fp8(d2r_ctx *ctx, const float *A, const float *W, const float *bias, uint32_t M, uint32_t N,
                               uint32_t K, int kind, float *out, uint8_t *a_q8, uint8_t *a_scales, float *w_scale_out,
                               uint8_t *out_q8, uint8_t *out_scales);

/* ------------------------------------------------------------- text tower */

typedef struct {
    uint32_t vocab_size;      /* 49408 */
    uint32_t context_length;  /* 77 */
    uint32_t hidden_size;     /* 512 / 768 (head dim must be 64) */
    uint32_t num_layers;      /* 12 */
    uint32_t num_heads;       /* 8 / 12 */
    uint32_t mlp_size;        /* 2048 / 3072 */
    uint32_t proj_dim;        /* D */
} d2r_text_desc;
typedef struct d2r_text d2r_text;

/*
 * replaces the text tower of CLIPModel (reference clip_scoring.py:150,180).  weights: host fp32
 * blob in this order (Hugging Face names): token_embedding.weight [V][d],
 * position_embedding.weight [ctx][d], per layer the same 16 tensors as d2r_clip_create,
 * final_layer_norm.{weight,bias}, text_projection.weight [D][d].
 */
D2R_API int d2r_text_create(d2r_ctx *ctx, const d2r_text_desc *desc, const float *weights,
                            size_t n_floats, d2r_text **out);
D2R_API void d2r_text_destroy(d2r_text *text);
/*
 * input_ids host [C][T] int32 (tokenised captions, EOS = largest id) -> embeds_out host [C][D],
 * L2-normalised: the cached text embeddings d2r_clip_score_frames / d2r_render_score take.
 * The reference re-encodes the captions for every image batch (clip_scoring.py:176-180); here the
 * caller does it once per task.
 */
D2R_API int d2r_text_encode(d2r_ctx *ctx, const d2r_text *text, const int32_t *input_ids, uint32_t C,
                            uint32_t T, float *embeds_out);

/* --------------------------------------------------- the fused hot path */

/*
 * Pose batch in, logits out: HOT LOOP A + HOT LOOP B of reference clip_scoring.py:136-185
 * with nothing but poses and logits crossing PCIe.  Needs d2r_set_background first.
 *   obj_poses_dev  DEVICE [K][16] fp32 candidate poses T_WO_2 (NGP convention)
 *   text_embeds    host [C][D]
 *   logits_dev     DEVICE [K][C] fp32
 *   frames_out     host [K][h][w][3] uint8, optional (NULL in benchmark mode)
 * Asynchronous on the context's stream when frames_out is NULL; call
 * d2r_ctx_synchronize (or synchronise the caller-owned stream) before reading logits_dev.
 * K is processed in chunks ("chunk" option); the render half of a chunk runs on a second, library-owned stream that
 * forks from and joins the context's stream inside the call ("overlap" option).
 */
D2R_API int d2r_render_score(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_clip *clip,
                             const d2r_view *view, const float *obj_pose_now,
                             const float *cam_pose, const float *obj_poses_dev, uint32_t K,
                             const float *text_embeds, uint32_t C, float logit_scale,
                             float *logits_dev, uint8_t *frames_out);

/*
 * Where the composited frames of a pass go besides (or instead of) the caller's array: the files the reference's
 * renderer writes inside its loop, `cb_render/cb_rgb_%04d.png` (reference reconstruction/combined_rendering.py:157-159),
 * which a later run re-scores with use_cache_renders (clip_scoring.py:89-104).  The library encodes them on a pool of
 * host threads while the GPU works on the next chunk.
 */
typedef struct {
    const char *png_dir;        /* existing directory for cb_rgb_%04d.png; NULL = write no files */
    uint32_t png_first_index;   /* file index of candidate 0 (a pose shard passes its first global render index) */
    int32_t png_threads;        /* encoder threads; 0 = the CPUs the process may use (hardware threads capped by the container's CPU quota), at most 64 */
    int32_t png_level;          /* negative = the default: Sub-filtered scanlines (cv2.imwrite's filter), Huffman-only deflate;
                                   0..9 = unfiltered scanlines at that zlib level (PNG is lossless: this only trades time for size) */
} d2r_frame_sink;

/*
 * The same pass for a caller that holds HOST memory — what the Python host's optimise_pose_grid / renderer call
 * (reference clip_scoring.py:136-185: `renderer.render(...)`, then the CLIP batches): candidate poses in, logits out,
 * frames kept on the GPU unless asked for.  Synchronous.
 *   obj_poses    host [K][16] fp32 candidate poses T_WO_2 (NGP convention)
 *   logits_out   host [K][C]
 *   frames_out   host [K][h][w][3] uint8, optional
 *   sink         optional: PNG files of the frames (streamed chunk by chunk; the host never holds more than two chunks
 *                of at most 1 GiB each in the library's pinned staging buffers)
 */
D2R_API int d2r_render_score_host(d2r_ctx *ctx, const d2r_nerf *fg, const d2r_clip *clip,
                                  const d2r_view *view, const float *obj_pose_now, const float *cam_pose,
                                  const float *obj_poses, uint32_t K, const float *text_embeds, uint32_t C,
                                  float logit_scale, float *logits_out, uint8_t *frames_out,
                                  const d2r_frame_sink *sink);

/* ------------------------------------------------- frame files (host only: no device, no context) */

/* One uint8 RGB image [h][w][3] -> an 8-bit RGB PNG file (best_render.png, reference clip_scoring.py:222-223).  level: see
 * d2r_frame_sink.png_level (negative = the default encoding). */
D2R_API int d2r_png_write(const uint8_t *rgb, uint32_t w, uint32_t h, const char *path, int level);
/* frames host [n][h][w][3] -> <dir>/cb_rgb_%04d.png for indices first_index .. first_index+n-1, encoded on `threads`
 * host threads (0 = auto): what renderer.render(save=True) leaves behind (combined_rendering.py:157-159). */
D2R_API int d2r_png_write_batch(const uint8_t *frames, uint32_t n, uint32_t w, uint32_t h, const char *dir,
                                uint32_t first_index, int threads, int level);
/* The same for frames that equal `background` ([h][w][3]) in most scanlines — the frames of a render-and-score pass, where a candidate's
 * object covers a band of the frame: the background's scanlines are entropy-coded once and a frame re-codes only the scanlines that differ
 * (default encoding only; identical pixels, files a few per cent larger).  d2r_render_score_host's frame sink writes its files this way. */
D2R_API int d2r_png_write_batch_bg(const uint8_t *frames, uint32_t n, uint32_t w, uint32_t h, const uint8_t *background, const char *dir,
                                   uint32_t first_index, int threads);
/* The reverse, for use_cache_renders (clip_scoring.py:95-104): n files of w x h -> frames_out host [n][h][w][3]; file i is
 * cb_rgb_%04d.png of indices[i], or of first_index + i when indices is NULL.  Reads 8-bit grey / RGB PNGs with or without
 * alpha, every scanline filter (what cv2.imwrite and PIL write); a file of another size, a missing file or an
 * unsupported format is an error naming the file. */
D2R_API int d2r_png_read_batch(const char *dir, const uint32_t *indices, uint32_t first_index, uint32_t n, uint32_t w,
                               uint32_t h, uint8_t *frames_out, int threads);
D2R_API int d2r_png_size(const char *path, uint32_t *w, uint32_t *h);
/* np.savetxt(path, data) with numpy's defaults ('%.18e', ' ', '\n'), byte for byte: the format of goal_pose.txt,
 * pose_batch.txt and pose_scores.txt (reference dream2real.py:356-358).  data host [rows][cols] fp64 (a 1-D array is
 * rows x 1: one number per line); formatted in row blocks on `threads` host threads (0 = auto). */
D2R_API int d2r_savetxt(const char *path, const double *data, uint64_t rows, uint64_t cols, int threads);

/* Counters of the last d2r_render / d2r_render_composite / d2r_render_score call, for
 * roofline accounting (bench.py): rays generated, rays that reached occupied space,
 * network evaluations (hash-grid samples). */
typedef struct {
    uint64_t rays_total;
    uint64_t rays_alive;
    uint64_t samples;
    uint64_t wave_iters;   /* marcher wave-iterations; lane utilisation = samples / (64 * wave_iters) */
    uint64_t l0_tokens;    /* d2r_render_score with "l0_reuse": patch tokens of the pass (0 when the reuse did not run) ... */
    uint64_t l0_touched;   /* ... and how many of them were recomputed (the others took the background's layer-0 rows) */
} d2r_render_stats;
D2R_API int d2r_get_render_stats(d2r_ctx *ctx, d2r_render_stats *out);
/* After an asynchronous d2r_render_score over K poses: synchronises the stream and gathers
 * the per-chunk device counters into the stats d2r_get_render_stats returns. */
D2R_API int d2r_collect_render_stats(d2r_ctx *ctx, uint32_t K);

/* Per-kernel device time of the calls made since timing was switched on with
 * d2r_ctx_set_option(ctx, "timing", 1): HIP events recorded on the launch stream around the
 * ray-march kernel, the ray-generation kernel, the CLIP preprocess kernel and the CLIP
 * forward (all its kernels).  d2r_get_timing synchronises the stream, sums the elapsed
 * times and resets the event list. */
typedef struct {
    double march_ms;   uint64_t march_launches;
    double raygen_ms;  uint64_t raygen_launches;
    double prep_ms;    uint64_t prep_launches;
    double clip_ms;    uint64_t clip_launches;   /* one "launch" = one forward over a chunk */
    /* ABI 9 */
    double sort_ms;    uint64_t sort_launches;   /* the ray sort's passes (k_sort_*), one "launch" = one sort; NOT part of raygen_ms */
    /* "timing" 2 only (an event pair per kernel launch: a few tenths of a per cent of a step, so not for timed regions): the
     * vision tower's full-size products of the default schedule — LayerNorm-folded QKV, streamed attention, out-projection,
     * fc1 (+ quick_gelu), fc2; layer 0's compact QKV under "l0_reuse", the class-token-only last block, row statistics, patch
     * embedding and head are clip_ms minus these five */
    double vit_qkv_ms;  uint64_t vit_qkv_launches;
    double vit_attn_ms; uint64_t vit_attn_launches;
    double vit_out_ms;  uint64_t vit_out_launches;
    double vit_fc1_ms;  uint64_t vit_fc1_launches;
    double vit_fc2_ms;  uint64_t vit_fc2_launches;
} d2r_timing;
D2R_API int d2r_get_timing(d2r_ctx *ctx, d2r_timing *out);

/* Tunables; unknown keys return D2R_ERR_INVALID.
 * "chunk" (default 4096, 1..16384): candidates / images per pass; the library lowers it per model and
 *     view so that one pass stays inside the 32-bit indexing of the ray queue and the GEMM outputs.
 * "refill_min" (default 64, 1..64): free lanes a marcher wave accumulates before it takes new rays
 *     from the queue (64 = a wave runs its 64 rays to the end).
 * "ray_sort" (default 1): before the march the ray queue is sorted (three small counting-sort passes) by the cell of
 *     the object's occupied box a ray's first sample lies in — 2^"ray_sort_log2" (default 4, 1..4) cells per axis,
 *     Morton order: every candidate renders the same object, so the waves running at one time then read the same few
 *     regions of the level tables / bricks.  Same pixels; 0 marches the rays in generation order.
 * "march_threads" (default 0 = auto; else a multiple of 64 up to the compiled 768 — larger values are refused): threads per marcher
 *     workgroup (one workgroup per CU).  Auto: 768 (three waves per SIMD); with "ray_sort" 0: 512 where the HBM
 *     bricks exceed "march_threads_auto_mib" (default 64) MiB — the unsorted marcher is bound by the L2-miss path
 *     there and fewer waves thrash less.  Read-only: "march_threads_used", "march_hbm_brick_bytes" (of the last march launch).
 * "march_compact" (default 1): once a marcher wave has no more than 32 rays left it moves them to its
 *     lanes 0..31, so that the second 32-sample tile of its iterations costs nothing (same pixels).
 * "bricks" (default 1): serve the de-hashed coarse levels of small models from LDS; 0 forces every
 *     level through the global tables.  Behind the LDS slots, further slots are served from de-hashed dense bricks in HBM:
 *     "gbrick_slots" (default 8, 0..8) caps how many, "brick_slots_total" (default 7, 0..8) the first slot that is never
 *     bricked (the finest one measured slower).  "raygen_rect" (default 1): composite mode generates rays only inside the
 *     projected occupied bounding box.  Results are bit-identical whatever these are set to.
 *     Read at d2r_nerf_create / d2r_nerf_load_ingp time (set them BEFORE creating the model): "lds_slots_max" (default 5,
 *     0..5): at most this many leading slots as LDS bricks; "gbrick_max_mib" (default 512, 0..512): a slot gets an HBM brick
 *     only while that brick stays below this size (the bricks of a model total at most 512 MiB).
 * "mlp_f16" (default 1 since round 6): operand type of the NeRF MLPs' MFMAs, fp32 accumulation either way, same MFMA rate.  1 = fp16: the
 *     reference's operand type (tiny-cuda-nn's fully fused MLPs are __half; BASELINE.json configs[4] names an "fp16 render") — the
 *     snapshot's fp16 weights enter the MFMA unrounded, features and activations keep 11 significant bits; activations must stay below
 *     65504, as in the reference.  0 = bf16 (north_star's wording: 8 significant bits).  The default follows the measurement: against an
 *     emulation of tiny-cuda-nn's half-accumulating arithmetic on a trained-like field the fp16-operand marcher sits where the fp32
 *     specification sits (|dlog sigma| 0.032 vs 0.030, no object pixel off by more than one LSB), the bf16-operand one twice as far
 *     (0.064, 9 % of the object's pixels) — DESIGN.md section 5.
 * "ln_fold" (default 4): schedule of the vision tower.  0: LayerNorm kernels between the GEMMs, fp32 residual
 *     stream.  1-3: LayerNorm folded into the QKV / fc1 GEMMs (LN(x) W^T + b = rstd (x (gamma o W)^T - mean
 *     colsum) + b'), row statistics emitted by the residual GEMMs' epilogues, which also write the bf16 operand
 *     copy of the residual row; the residual stream itself is kept as 1: two bf16 arrays hi + lo (16 mantissa
 *     bits), 2: one bf16 array (fastest; its accumulated rounding puts the logit error past 1e-3 of the logit
 *     scale in the tail, so it is not the default), 3: fp32 next to the bf16 copy, 4 (default): bf16 hi + ONE lo byte per
 *     element — the next 8 bits of the fp32 bit pattern, rounded (16 significant bits like 1, a quarter less residual
 *     traffic; measured logit error 4.2e-4 of the scale on 300 images against 5.1e-4 for 1, CLIP time -1.3 %).
 * "cls_last" (default 1): the last transformer block of the vision tower runs on the class-token rows only (the head
 *     reads nothing else; same result, ~6 % less ViT work).  "gemm_nsplit" (default 0 = 2 where the column tiles and
 *     XCDs divide evenly; 1 = off): XCD sets own column sections of the persistent GEMM's outputs so that a section's
 *     weight panels stay in their L2s; results do not depend on it.
 * "prep_reuse" (default 1): in d2r_render_score, the rows of CLIP patches of a candidate frame that its object cannot have
 *     touched (outside the rectangle its rays are generated in) are copied from the background frame's own patches,
 *     computed once per d2r_set_background; bit-identical to resampling them.
 * "attn_rem" (default 1, 0..4): vision-tower attention on a sequence of 8 g + r query tiles of 32: a remainder of at most this
 *     many tiles runs on workgroups of r waves (each staging whole key tiles itself) instead of one more eight-wave
 *     workgroup with 8 - r idle waves; 257 tokens (r = 1): -7 % kernel time, 577 tokens (r = 3): neutral.  Bit-identical.
 * "overlap" (default 0): 1 = d2r_render_score / d2r_render_score_host run the render half of chunk i+1 (cameras, ray
 *     generation, march, preprocess) on a second stream while the ViT scores chunk i; 0 = one stream, in program order.
 *     Results do not depend on it; measured neutral on MI355X (the marcher and the persistent GEMMs each fill whole CUs —
 *     LDS and registers — so the two streams time-share the CUs: DESIGN.md section 4).  Frames that leave the GPU are
 *     copied on their own stream under the ViT either way.  "march_blocks" (default 0 = one per CU): workgroups of the persistent marcher.
 * "l0_reuse" (default 1): in d2r_render_score / d2r_render_score_host, a candidate's patches whose resampling footprint lies outside the
 *     rectangle its rays were generated in are the background's own patches, so their patch embedding, pre-LayerNorm residual row
 *     and layer-0 q / k / v rows are broadcast from rows computed once per background, and the patch-embedding and layer-0 QKV
 *     products run on the touched tokens only (needs "ln_fold" 4, "prep_reuse" and "raygen_rect" on, at most 1024 patches
 *     per image).  Bit-identical logits; configs[1]: 17 % of the tokens touched, CLIP -2.1 ms per 4096 candidates.
 * "vit_fp8" (default 0): 1 = the four Linear products of every transformer block except the first (when its rows are reused, "l0_reuse")
 *     and the class-token-only last one run on the MX-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the bf16 MFMA rate):
 *     activations quantised to OCP e4m3 with one power-of-two scale per (row, 64 columns), weights to e4m3 with one power-of-two scale
 *     per matrix, fp32 accumulation; residual stream, LayerNorm statistics, q / k / v and attention arithmetic unchanged.  This is the
 *     "fp8 MFMA ViT" BASELINE.json configs[4] names.  It is NOT within north_star's 1e-3 cosine of the fp32 reference (measured
 *     in tests/test_fp8.py and DESIGN.md section 7) — which is why it is off unless asked for.  Runs for models whose hidden and MLP sizes are
 *     multiples of 256 under "ln_fold" 4 (ViT-B/16, ViT-L/14: yes); any other model or residual mode stays in bf16.
 * "debug_fail_chunk" (default -1 = off): fault injection for the error path of a chunked pass — the NEXT d2r_render_score /
 *     d2r_render_score_host fails with D2R_ERR_DEVICE when it reaches this chunk index, exactly as if a launch of that chunk had
 *     failed (streams joined, worker pool drained, context usable afterwards), and the hook disarms itself (one-shot).
 * "timing" (0/1/2): record HIP events per kernel group for d2r_get_timing; 2 adds an event pair around every product of the vision tower.
 * Development builds of the library (make DEV=1) also know experiment switches — schedules that were measured no faster
 * and tile configurations kept for comparison; they are not part of this interface. */
D2R_API int d2r_ctx_set_option(d2r_ctx *ctx, const char *key, int64_t value);
/* Reads a tunable back (the keys of d2r_ctx_set_option), plus read-only facts about the last ray-march launch on this
 * context: "march_lds_slots" / "march_hbm_brick_slots" = the brick configuration it ran with (ABI 8). */
D2R_API int d2r_ctx_get_option(d2r_ctx *ctx, const char *key, int64_t *value);

/* ------------------------------------------------------ multi-GPU (one process per GPU) */

/*
 * The path's only collective: candidate poses are sharded in contiguous
 * blocks over the GPUs of a node and every rank needs ALL logits before spatially_smooth_heatmap
 * (reference vision_3d/geometry_utils.py:252-269) and the argmax (clip_scoring.py:218).  The reference
 * itself is single-GPU (README.md:27) and has no counterpart.  Implemented on RCCL (ncclAllGather
 * over xGMI), bound at run time; bootstrap is the caller's: rank 0 obtains an id blob and hands it to
 * the other ranks by any means (torch.distributed, MPI, a file).
 */
#define D2R_COMM_ID_BYTES 128
/* rank 0 only: fills id_out[D2R_COMM_ID_BYTES] (ncclGetUniqueId) */
D2R_API int d2r_comm_get_unique_id(void *id_out);
/* every rank, collectively: joins the communicator on the context's device (ncclCommInitRank).
 * world == 1 with a NULL id needs no RCCL: the gather degenerates to a device copy (with an id, a
 * one-rank RCCL communicator is created all the same). */
D2R_API int d2r_comm_init(d2r_ctx *ctx, const void *id_blob, int rank, int world);
D2R_API int d2r_comm_destroy(d2r_ctx *ctx);
/*
 *   local_dev   DEVICE [n_local] fp32: this rank's logits (K_local * C values; ranks with a shorter
 *               shard pad to the common n_local)
 *   global_dev  DEVICE [world][n_local] fp32, rank-major
 * Asynchronous on the context's stream, ordered after the d2r_render_score that produced local_dev.
 */
D2R_API int d2r_allgather_scores(d2r_ctx *ctx, const float *local_dev, size_t n_local, float *global_dev);

/* ------------------------------------------ batched physics pre-filter */

/*
 * replaces the per-pose PyBullet loop of unsupcol_check (reference vision_3d/physics_utils.py:248-375), the
 * phys_check the path calls before rendering (clip_scoring.py:108-113, dream2real.py:304-326): duplicate
 * orientations (:260-278), optional regrasp rule (:281-301), then per pose collision (:314-321), support
 * (:329-340) and stability (:349-365).  Shapes are convex hulls given as vertex sets — PyBullet's GEOM_MESH
 * without the concave flag (:239) turns every shape (`o` / `g` group) of the mesh file into the convex hull of its
 * vertices, so a VHACD-decomposed object is a compound of convex parts; two bodies "collide / touch" when any pair
 * of their parts is in contact (GJK, one wavefront per pose), contact meaning the hulls come closer than the sum of
 * the two shapes' collision margins (d2r_phys_params.margin; 0 = plain hull intersection).
 *   movable_verts    host [movable_offsets[n_movable]][3]  the movable object's convex parts, concatenated, world
 *                    frame, at the object's initial pose (the reference's mesh files are written in world coordinates)
 *   movable_offsets  host [n_movable + 1]  first vertex of each movable part (movable_offsets[0] = 0)
 *   static_verts     host [static_offsets[n_static]][3]  convex parts of all static objects, concatenated
 *   static_offsets   host [n_static + 1]  first vertex of each static part
 */
typedef struct d2r_phys d2r_phys;
D2R_API int d2r_phys_create(d2r_ctx *ctx, const float *movable_verts, const uint32_t *movable_offsets, uint32_t n_movable,
                            const float *static_verts, const uint32_t *static_offsets, uint32_t n_static, d2r_phys **out);
D2R_API void d2r_phys_destroy(d2r_phys *phys);
typedef struct {
    uint32_t sample_res[6];   /* the pose grid's resolution: orientations per position = res[3] * res[4] * res[5] */
    float init_pose[16];      /* task_model.movable_obj.pose, row-major 4x4 (world) */
    float table_z;            /* scene_centre[2]: a pose below it counts as supported (:332-334) */
    float unsup_thresh;       /* 0.02: how far the object is lowered for the support test */
    float gravity[3];         /* GRAVITY_DIRECTION (0, 0, -1) */
    float perturb;            /* 0.04: sideways offset of the four stability probes */
    int32_t stability_check;  /* non-zero: run the stability probes */
    int32_t disallow_regrasp; /* non-zero: keep only orientations whose object z axis faces +z or -y (:281-301) */
    float margin;             /* collision margin of every convex part, metres: two parts are in contact when their hulls
                               * are closer than 2 * margin.  PyBullet loads GEOM_MESH convex hulls with a margin of 0.001
                               * (believed: its default collision margin for file-loaded convex shapes; PyBullet is not
                               * available offline, so this is unpinned); 0 = exact hull intersection */
} d2r_phys_params;
/*
 *   pose_batch  host [N][16] sampled poses (world, sample_poses_grid order: orientations fastest)
 *   valid_io    host [N] uint8: valid_so_far in, is_valid out
 */
D2R_API int d2r_phys_check(d2r_ctx *ctx, const d2r_phys *phys, const d2r_phys_params *params, const float *pose_batch,
                           uint32_t N, uint8_t *valid_io);

#ifdef __cplusplus
}
#endif
#endif /* D2R_H */
