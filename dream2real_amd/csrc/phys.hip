// phys.hip — batched physics pre-filter of candidate poses on gfx950.
//
// Replaces the serial PyBullet loop of the reference's unsupcol_check (vision_3d/physics_utils.py:248-375):
// for every sampled pose, (1) duplicate orientations are dropped (:260-278), (2) optionally orientations the
// robot could not regrasp (:281-301), then per surviving pose (3) collision of the movable object with the
// static objects (:314-321), (4) support: lowered by unsup_thresh along gravity it must touch a static object
// or lie below the table (:329-340), (5) stability: four sideways-perturbed lowered poses must all still
// touch (:349-365).  Shapes are convex hulls (PyBullet's GEOM_MESH without the concave flag is the convex
// hull of the mesh, :239), given as vertex sets; "touch / collide" is hull intersection, decided by GJK.
//
// One wave per pose.  The support function (arg max over the hull's vertices of a dot product) is what costs:
// its 64 lanes take a vertex each per step and the wave reduces with DPP/swizzle shuffles; the simplex logic
// is wave-uniform (every lane carries the same simplex).  Poses share nothing, so the kernel needs no LDS and
// no atomics, and the validity mask never leaves the GPU between the checks.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "d2r_internal.h"

struct d2r_phys {
    d2r_ctx *ctx;
    float *d_mov = nullptr;        // movable object's hull vertices (one or several convex parts, concatenated), world frame at the object's initial pose
    uint32_t *d_moff = nullptr;    // [n_mov + 1] first vertex of each movable hull
    uint32_t n_mov = 0, n_mov_verts = 0;
    float *d_stat = nullptr;       // static hull vertices, concatenated
    uint32_t *d_off = nullptr;     // [n_stat + 1] first vertex of each static hull
    uint32_t n_stat = 0, n_stat_verts = 0;
};

struct PhysKernelParams {
    float inv_init[16];            // inverse of the movable object's initial pose (row-major 4x4)
    float table_z, unsup_thresh, perturb;
    float gravity[3];
    int stability_check;
    uint32_t oris_per_pos;
    float margin2;                 // sum of the two shapes' collision margins: hulls closer than this "touch"
};

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// arg max over the n vertices of v . d, all 64 lanes cooperating; returns the vertex (same in every lane)
__device__ __forceinline__ V3 hull_support(const float *__restrict__ verts, uint32_t n, V3 d, uint32_t lane)
{
    float best = -INFINITY;
    uint32_t bi = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const float s = fmaf(verts[3 * i], d.x, fmaf(verts[3 * i + 1], d.y, verts[3 * i + 2] * d.z));
        if (s > best) { best = s; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }      // ties: lowest index, lane-order independent
    }
    return {verts[3 * bi], verts[3 * bi + 1], verts[3 * bi + 2]};
}

// Do hull A (vertices a, moved by x -> R x + t) and hull B come within `margin2` of each other (margin2 = 0: do they
// intersect)?  Boolean GJK on the Minkowski difference A - B, inflated by a sphere of radius margin2 through its support
// function (Bullet gives every convex shape a collision margin; two shapes are in contact when their cores are closer
// than the sum of the margins).
__device__ bool gjk_intersect(const float *__restrict__ a, uint32_t na, const float R[9], V3 t,
                              const float *__restrict__ b, uint32_t nb, uint32_t lane, float margin2)
{
    auto support = [&](V3 d) -> V3 {
        // arg max over A of (R v + t) . d = arg max of v . (R^T d)
        const V3 dl = {fmaf(R[0], d.x, fmaf(R[3], d.y, R[6] * d.z)), fmaf(R[1], d.x, fmaf(R[4], d.y, R[7] * d.z)),
                       fmaf(R[2], d.x, fmaf(R[5], d.y, R[8] * d.z))};
        const V3 va = hull_support(a, na, dl, lane);
        const V3 wa = {fmaf(R[0], va.x, fmaf(R[1], va.y, fmaf(R[2], va.z, t.x))),
                       fmaf(R[3], va.x, fmaf(R[4], va.y, fmaf(R[5], va.z, t.y))),
                       fmaf(R[6], va.x, fmaf(R[7], va.y, fmaf(R[8], va.z, t.z)))};
        const V3 vb = hull_support(b, nb, neg(d), lane);
        V3 p = wa - vb;
        if (margin2 > 0.f) {
            const float k = margin2 * rsqrtf(dot(d, d));
            p = {fmaf(k, d.x, p.x), fmaf(k, d.y, p.y), fmaf(k, d.z, p.z)};
        }
        return p;
    };
    V3 s[4];
    int n = 1;
    V3 d = {1.f, 0.f, 0.f};
    s[0] = support(d);
    d = neg(s[0]);
    for (int it = 0; it < 64; it++) {
        const float dd = dot(d, d);
        if (dd < 1e-20f) return true;                         // the origin lies on the simplex
        const V3 p = support(d);
        const float pd = dot(p, d);
        if (pd < 0.f) return false;                           // a separating direction
        // Progress test: d points from the simplex's closest feature towards the origin.  Were the origin strictly
        // inside, the support point along d would lie beyond the origin, hence strictly beyond every simplex point;
        // a support point that is no further along d than the simplex already reaches (to 1e-7 m) means the simplex
        // holds the closest feature and the origin is outside, or on the boundary to within rounding: no contact.
        // Without this test near-touching pairs cycle until the iteration cap.
        float reach = dot(s[0], d);
        for (int i = 1; i < n; i++) reach = fmaxf(reach, dot(s[i], d));
        if (pd - reach <= 1e-7f * sqrtf(dd)) return false;
        s[n++] = p;
        const V3 A = s[n - 1], AO = neg(A);
        if (n == 4) {
            // faces through the newest point, normals turned away from the opposite vertex
            const V3 B = s[2], C = s[1], D = s[0];
            const V3 f[3][3] = {{B, C, D}, {C, D, B}, {D, B, C}};
            int out = -1;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                V3 nrm = cross(f[k][0] - A, f[k][1] - A);
                if (dot(nrm, f[k][2] - A) > 0.f) nrm = neg(nrm);
                if (out < 0 && dot(nrm, AO) > 0.f) out = k;
            }
            if (out < 0) return true;                         // inside the tetrahedron
            s[0] = f[out][1];
            s[1] = f[out][0];
            s[2] = A;
            n = 3;
        }
        if (n == 3) {
            const V3 B = s[1], C = s[0], AB = B - A, AC = C - A, ABC = cross(AB, AC);
            bool edge_ab = false;
            if (dot(cross(ABC, AC), AO) > 0.f) {
                if (dot(AC, AO) > 0.f) {
                    s[0] = C; s[1] = A; n = 2;
                    d = cross(cross(AC, AO), AC);
                } else edge_ab = true;
            } else if (dot(cross(AB, ABC), AO) > 0.f) {
                edge_ab = true;
            } else {
                d = dot(ABC, AO) > 0.f ? ABC : neg(ABC);      // above or below the triangle
            }
            if (edge_ab) {
                if (dot(AB, AO) > 0.f) {
                    s[0] = B; s[1] = A; n = 2;
                    d = cross(cross(AB, AO), AB);
                } else {
                    s[0] = A; n = 1;
                    d = AO;
                }
            }
        } else if (n == 2) {
            const V3 B = s[0], AB = B - A;
            if (dot(AB, AO) > 0.f) {
                d = cross(cross(AB, AO), AB);
            } else {
                s[0] = A; n = 1;
                d = AO;
            }
        }
    }
    // the cap is only reached by an inflated (curved) difference whose boundary passes within rounding of the origin:
    // a touch at exactly the margin distance, reported like the stalled case above
    return false;
}

// block = 256 threads = 4 waves, one pose per wave
__global__ __launch_bounds__(256) void k_phys_check(PhysKernelParams P, const float *__restrict__ poses, uint32_t n_poses,
                                                    const uint8_t *__restrict__ ori_mask,
                                                    const float *__restrict__ mov, const uint32_t *__restrict__ moff, uint32_t n_mov,
                                                    const float *__restrict__ stat, const uint32_t *__restrict__ off,
                                                    uint32_t n_stat, uint8_t *__restrict__ valid)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t pose = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pose >= n_poses) return;
    if (!valid[pose]) return;
    if (!ori_mask[pose % P.oris_per_pos]) {                  // duplicate orientation / not regraspable
        if (lane == 0) valid[pose] = 0;
        return;
    }
    // transform = pose @ inv(init_pose): where the mesh (given at the initial pose) goes
    float T[12];
    const float *M = poses + (size_t)pose * 16;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            T[i * 4 + j] = fmaf(M[i * 4 + 0], P.inv_init[0 * 4 + j],
                                fmaf(M[i * 4 + 1], P.inv_init[1 * 4 + j],
                                     fmaf(M[i * 4 + 2], P.inv_init[2 * 4 + j], M[i * 4 + 3] * P.inv_init[3 * 4 + j])));
    const float R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const V3 pos = {T[3], T[7], T[11]};
    // PyBullet's pairwise test between two bodies is true when ANY pair of their convex parts is in contact
    auto touches_any = [&](V3 t) -> bool {
        for (uint32_t m = 0; m < n_mov; m++)
            for (uint32_t h = 0; h < n_stat; h++)
                if (gjk_intersect(mov + 3 * (size_t)moff[m], moff[m + 1] - moff[m], R, t, stat + 3 * (size_t)off[h], off[h + 1] - off[h],
                                  lane, P.margin2))
                    return true;
        return false;
    };
    bool ok = !touches_any(pos);                              // in collision -> invalid
    if (ok) {
        const V3 lower = {fmaf(P.unsup_thresh, P.gravity[0], pos.x), fmaf(P.unsup_thresh, P.gravity[1], pos.y),
                          fmaf(P.unsup_thresh, P.gravity[2], pos.z)};
        const bool below_table = M[11] < P.table_z;           // the sampled pose's own z (reference :332-333)
        if (!below_table) {
            ok = touches_any(lower);                          // unsupported unless something is right underneath
            if (ok && P.stability_check) {
                const float px[4] = {P.perturb, -P.perturb, 0.f, 0.f}, py[4] = {0.f, 0.f, P.perturb, -P.perturb};
                for (int k = 0; k < 4 && ok; k++) ok = touches_any({lower.x + px[k], lower.y + py[k], lower.z});
            }
        }
    }
    if (lane == 0) valid[pose] = ok ? 1 : 0;
}

extern "C" {

int d2r_phys_create(d2r_ctx *ctx, const float *movable_verts, const uint32_t *movable_offsets, uint32_t n_movable,
                    const float *static_verts, const uint32_t *static_offsets, uint32_t n_static, d2r_phys **out)
{
    if (!ctx || !movable_verts || !movable_offsets || !out || n_movable == 0 || (n_static && (!static_verts || !static_offsets)))
        return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    if (movable_offsets[0] != 0 || (n_static && static_offsets[0] != 0)) return d2r_fail(ctx, D2R_ERR_INVALID, "hull offsets must start at 0");
    for (uint32_t h = 0; h < n_movable; h++)
        if (movable_offsets[h + 1] <= movable_offsets[h]) return d2r_fail(ctx, D2R_ERR_INVALID, "movable hull offsets must increase");
    for (uint32_t h = 0; h < n_static; h++)
        if (static_offsets[h + 1] <= static_offsets[h]) return d2r_fail(ctx, D2R_ERR_INVALID, "static hull offsets must increase");
    (void)hipSetDevice(ctx->device);
    d2r_phys *p = new (std::nothrow) d2r_phys();
    if (!p) return d2r_fail(ctx, D2R_ERR_MEMORY, "out of host memory");
    p->ctx = ctx;
    p->n_mov = n_movable;
    p->n_mov_verts = movable_offsets[n_movable];
    p->n_stat = n_static;
    p->n_stat_verts = n_static ? static_offsets[n_static] : 0;
    const uint32_t zero = 0;
    bool ok = hipMalloc(&p->d_mov, (size_t)p->n_mov_verts * 12) == hipSuccess &&
              hipMalloc(&p->d_moff, ((size_t)n_movable + 1) * 4) == hipSuccess &&
              hipMalloc(&p->d_stat, std::max<size_t>(1, (size_t)p->n_stat_verts * 12)) == hipSuccess &&
              hipMalloc(&p->d_off, ((size_t)n_static + 1) * 4) == hipSuccess;
    ok = ok && hipMemcpy(p->d_mov, movable_verts, (size_t)p->n_mov_verts * 12, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_moff, movable_offsets, ((size_t)n_movable + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         (p->n_stat_verts == 0 || hipMemcpy(p->d_stat, static_verts, (size_t)p->n_stat_verts * 12, hipMemcpyHostToDevice) == hipSuccess) &&
         hipMemcpy(p->d_off, n_static ? static_offsets : &zero, ((size_t)n_static + 1) * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        d2r_phys_destroy(p);
        return d2r_fail(ctx, D2R_ERR_MEMORY, "device allocation/upload failed for the physics shapes");
    }
    *out = p;
    return D2R_OK;
}

void d2r_phys_destroy(d2r_phys *p)
{
    if (!p) return;
    if (p->d_mov) (void)hipFree(p->d_mov);
    if (p->d_moff) (void)hipFree(p->d_moff);
    if (p->d_stat) (void)hipFree(p->d_stat);
    if (p->d_off) (void)hipFree(p->d_off);
    delete p;
}

// general 4x4 inverse (Gauss-Jordan with partial pivoting, double); false when singular
static bool invert4(const float *m, float *out)
{
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            a[i][j] = m[i * 4 + j];
            a[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        for (int r = c + 1; r < 4; r++)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (fabs(a[piv][c]) < 1e-30) return false;
        for (int j = 0; j < 8; j++) std::swap(a[c][j], a[piv][j]);
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] *= inv;
        for (int r = 0; r < 4; r++)
            if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
            }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[i * 4 + j] = (float)a[i][4 + j];
    return true;
}

int d2r_phys_check(d2r_ctx *ctx, const d2r_phys *phys, const d2r_phys_params *prm, const float *pose_batch, uint32_t N,
                   uint8_t *valid_io)
{
    if (!ctx || !phys || !prm || !pose_batch || !valid_io) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    uint64_t expect = 1, oris = 1;
    for (int i = 0; i < 6; i++) {
        if (prm->sample_res[i] == 0) return d2r_fail(ctx, D2R_ERR_INVALID, "sample_res entries must be positive");
        expect *= prm->sample_res[i];
        if (i >= 3) oris *= prm->sample_res[i];
    }
    if (expect != N) return d2r_fail(ctx, D2R_ERR_INVALID, "pose count does not match sample_res");
    if (N == 0) return D2R_OK;
    PhysKernelParams P;
    if (!invert4(prm->init_pose, P.inv_init)) return d2r_fail(ctx, D2R_ERR_INVALID, "init_pose is singular");
    P.table_z = prm->table_z;
    P.unsup_thresh = prm->unsup_thresh;
    P.perturb = prm->perturb;
    for (int i = 0; i < 3; i++) P.gravity[i] = prm->gravity[i];
    P.stability_check = prm->stability_check;
    P.oris_per_pos = (uint32_t)oris;
    if (!(prm->margin >= 0.f) || !(prm->margin < 1.f)) return d2r_fail(ctx, D2R_ERR_INVALID, "collision margin must be in [0, 1) metres");
    P.margin2 = 2.f * prm->margin;
    // orientation uniqueness (reference :260-278: greedy over the orientations of the FIRST position, then tiled)
    // and the regrasp rule (:281-301) are a few thousand 3x3 compares at most: host side, uploaded as one mask
    std::vector<uint8_t> mask(oris, 1);
    {
        std::vector<uint32_t> kept;
        for (uint32_t i = 0; i < oris; i++) {
            const float *a = pose_batch + (size_t)i * 16;
            bool seen = false;
            for (uint32_t k : kept) {
                const float *b = pose_batch + (size_t)k * 16;
                bool close = true;
                for (int r = 0; r < 3 && close; r++)
                    for (int c = 0; c < 3; c++)
                        if (!(fabsf(a[r * 4 + c] - b[r * 4 + c]) <= 0.01f + 1e-5f * fabsf(b[r * 4 + c]))) {   // torch.isclose(atol=0.01)
                            close = false;
                            break;
                        }
                if (close) { seen = true; break; }
            }
            if (seen) mask[i] = 0;
            else kept.push_back(i);
        }
        if (prm->disallow_regrasp)
            for (uint32_t i = 0; i < oris; i++) {
                if (!mask[i] || !valid_io[i]) { mask[i] = 0; continue; }     // reference :285-287 reads valid_so_far of the first position
                const float *a = pose_batch + (size_t)i * 16;
                const float zx = a[2], zy = a[6], zz = a[10];                // the object's z axis: column 2 of the rotation
                (void)zx;
                if (!(zz > 0.9f || -zy > 0.9f)) mask[i] = 0;
            }
    }
    (void)hipSetDevice(ctx->device);
    int rc;
    if ((rc = d2r_reserve(ctx, ctx->poses, (size_t)N * 64))) return rc;
    if ((rc = d2r_reserve(ctx, ctx->pix, (size_t)N + oris + 64))) return rc;
    uint8_t *d_valid = (uint8_t *)ctx->pix.p, *d_mask = d_valid + ((N + 63) / 64) * 64;
    D2R_HIP(ctx, hipMemcpyAsync(ctx->poses.p, pose_batch, (size_t)N * 64, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(d_valid, valid_io, N, hipMemcpyHostToDevice, ctx->stream));
    D2R_HIP(ctx, hipMemcpyAsync(d_mask, mask.data(), oris, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_phys_check, dim3((N + 3) / 4), dim3(256), 0, ctx->stream, P, (const float *)ctx->poses.p, N,
                       (const uint8_t *)d_mask, (const float *)phys->d_mov, (const uint32_t *)phys->d_moff, phys->n_mov, (const float *)phys->d_stat,
                       (const uint32_t *)phys->d_off, phys->n_stat, d_valid);
    D2R_HIP(ctx, hipGetLastError());
    D2R_HIP(ctx, hipMemcpyAsync(valid_io, d_valid, N, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}

}  // extern "C"
