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
    float *d_stat_box = nullptr;   // [n_stat][6] axis-aligned box of each static hull (lo xyz, hi xyz)
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

// ---- closest point to the origin on a simplex of 2 / 3 / 4 points (Ericson, Real-Time Collision Detection 5.1.2, 5.1.5,
// 5.1.6: Voronoi-region tests).  The simplex is reduced in place to the sub-simplex that carries the closest point; the
// functions return that point.  Wave-uniform: every lane runs the same scalar arithmetic.

__device__ V3 closest_segment(V3 *s, int &n)
{
    const V3 a = s[0], b = s[1], ab = b - a;
    const float t = dot(neg(a), ab), den = dot(ab, ab);
    if (t <= 0.f || den <= 0.f) { n = 1; return a; }
    if (t >= den) { s[0] = b; n = 1; return b; }
    const float u = t / den;
    return {fmaf(u, ab.x, a.x), fmaf(u, ab.y, a.y), fmaf(u, ab.z, a.z)};
}

__device__ V3 closest_triangle(V3 *s, int &n)
{
    const V3 a = s[0], b = s[1], c = s[2], ab = b - a, ac = c - a;
    const float d1 = dot(ab, neg(a)), d2 = dot(ac, neg(a));
    if (d1 <= 0.f && d2 <= 0.f) { n = 1; return a; }
    const float d3 = dot(ab, neg(b)), d4 = dot(ac, neg(b));
    if (d3 >= 0.f && d4 <= d3) { s[0] = b; n = 1; return b; }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        const float v = d1 / (d1 - d3);
        n = 2;                                                     // edge ab: s[0], s[1] stay
        return {fmaf(v, ab.x, a.x), fmaf(v, ab.y, a.y), fmaf(v, ab.z, a.z)};
    }
    const float d5 = dot(ab, neg(c)), d6 = dot(ac, neg(c));
    if (d6 >= 0.f && d5 <= d6) { s[0] = c; n = 1; return c; }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        const float w = d2 / (d2 - d6);
        s[1] = c; n = 2;                                           // edge ac
        return {fmaf(w, ac.x, a.x), fmaf(w, ac.y, a.y), fmaf(w, ac.z, a.z)};
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        const V3 bc = c - b;
        s[0] = b; s[1] = c; n = 2;                                 // edge bc
        return {fmaf(w, bc.x, b.x), fmaf(w, bc.y, b.y), fmaf(w, bc.z, b.z)};
    }
    const float sum = va + vb + vc;
    if (!(sum > 0.f) || !(sum < INFINITY)) {
        // degenerate (collinear / repeated points: flat hulls, duplicate vertices) — the interior formula would divide by zero and a NaN
        // direction would make every later comparison false.  The closest point then lies on one of the three edges.
        V3 best_s[2] = {a, b};
        int best_n = 2;
        V3 e[2] = {a, b};
        int en = 2;
        V3 q = closest_segment(e, en), best = q;
        best_s[0] = e[0]; best_s[1] = e[1]; best_n = en;
        float bq = dot(q, q);
        const V3 cand[2][2] = {{a, c}, {b, c}};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            e[0] = cand[k][0]; e[1] = cand[k][1]; en = 2;
            q = closest_segment(e, en);
            const float qq = dot(q, q);
            if (qq < bq) { bq = qq; best = q; best_s[0] = e[0]; best_s[1] = e[1]; best_n = en; }
        }
        s[0] = best_s[0]; s[1] = best_s[1]; n = best_n;
        return best;
    }
    const float den = 1.f / sum, v = vb * den, w = vc * den;
    return {fmaf(w, ac.x, fmaf(v, ab.x, a.x)), fmaf(w, ac.y, fmaf(v, ab.y, a.y)), fmaf(w, ac.z, fmaf(v, ab.z, a.z))};
}

// false: the origin is inside the tetrahedron (the caller reports an intersection)
__device__ bool closest_tetrahedron(V3 *s, int &n, V3 &out)
{
    const V3 p[4] = {s[0], s[1], s[2], s[3]};
    const int face[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};      // three face vertices, then the opposite one
    float best = INFINITY;
    bool outside_any = false;
    V3 bs[3];
    int bn = 0;
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const V3 a = p[face[f][0]], b = p[face[f][1]], c = p[face[f][2]], d = p[face[f][3]];
        const V3 nrm = cross(b - a, c - a);
        const float sp = dot(neg(a), nrm), sd = dot(d - a, nrm);
        if (sp * sd < 0.f || sd == 0.f) {                          // the origin lies beyond this face (a flat tetrahedron counts as outside)
            outside_any = true;
            V3 t[3] = {a, b, c};
            int tn = 3;
            const V3 q = closest_triangle(t, tn);
            const float qq = dot(q, q);
            if (qq < best) {
                best = qq;
                out = q;
                bn = tn;
                bs[0] = t[0]; bs[1] = t[1]; bs[2] = t[2];
            }
        }
    }
    if (!outside_any) return false;
    n = bn;
    for (int i = 0; i < bn; i++) s[i] = bs[i];
    return true;
}

// Do hull A (vertices a, moved by x -> R x + t) and hull B come within `margin2` of each other (margin2 = 0: do they
// intersect)?  Distance GJK on the Minkowski difference of the two CORES, as Bullet does it: every convex shape carries a
// collision margin and two shapes are in contact when their cores are closer than the sum of the margins.  The
// difference of two polytopes is a polytope, so the iteration ends after finitely many support points; it also ends
// early as soon as the two bounds it carries decide the question — |v| (an upper bound of the distance: v is a point of
// the difference) at or below margin2: contact; v.w / |v| (a lower bound: w is the support point along -v) above
// margin2: none.  Pairs whose distance lies within rounding of margin2 (float32 on metre-sized coordinates: ~1e-6 m) may
// fall either way; everything else is exact.
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
        return wa - vb;
    };
    const float m2 = margin2 * margin2;
    V3 s[4];
    int n = 1;
    s[0] = support({1.f, 0.f, 0.f});
    V3 v = s[0];
    for (int it = 0; it < 48; it++) {
        const float vv = dot(v, v);
        if (!(vv == vv)) return true;                         // a non-finite direction (should not happen): contact, the conservative answer
        if (vv <= m2 || vv < 1e-18f) return true;             // a point of the difference within margin2 of the origin
        const V3 w = support(neg(v));
        const float vw = dot(v, w);
        if (vw > 0.f && vw * vw > m2 * vv) return false;      // the whole difference lies beyond the plane through w: distance > margin2
        // converged: w reaches no further towards the origin than the simplex already does -> |v| is the distance (> margin2 here)
        if (vv - vw <= 1e-6f * vv) return false;
        for (int i = 0; i < n; i++)
            if (s[i].x == w.x && s[i].y == w.y && s[i].z == w.z) return false;       // the same vertex again: no further progress possible
        s[n++] = w;
        if (n == 2) v = closest_segment(s, n);
        else if (n == 3) v = closest_triangle(s, n);
        else if (!closest_tetrahedron(s, n, v)) return true;  // the origin is inside the simplex: the cores intersect
    }
    return dot(v, v) <= m2;
}

// axis-aligned box of a rotated hull R v (no translation), all 64 lanes cooperating; the same six values in every lane
__device__ __forceinline__ void hull_aabb_rotated(const float *__restrict__ verts, uint32_t n, const float R[9], uint32_t lane, float lo[3], float hi[3])
{
    float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = lane; i < n; i += 64) {
        const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float c = fmaf(R[3 * k], x, fmaf(R[3 * k + 1], y, R[3 * k + 2] * z));
            l[k] = fminf(l[k], c);
            h[k] = fmaxf(h[k], c);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l[k] = fminf(l[k], __shfl_xor(l[k], o));
            h[k] = fmaxf(h[k], __shfl_xor(h[k], o));
        }
        lo[k] = l[k];
        hi[k] = h[k];
    }
}

// block = 256 threads = 4 waves, one pose per wave
__global__ __launch_bounds__(256) void k_phys_check(PhysKernelParams P, const float *__restrict__ poses, uint32_t n_poses,
                                                    const uint8_t *__restrict__ ori_mask,
                                                    const float *__restrict__ mov, const uint32_t *__restrict__ moff, uint32_t n_mov,
                                                    const float *__restrict__ stat, const uint32_t *__restrict__ off,
                                                    uint32_t n_stat, const float *__restrict__ stat_box, uint8_t *__restrict__ valid)
{
    // per wave: the boxes of the rotated movable parts (relative to the pose's translation), for the pair early-out
    constexpr uint32_t BOX_PARTS = 16;
    __shared__ float mbox[4][BOX_PARTS][6];
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
    float (*mb)[6] = mbox[threadIdx.x >> 6];
    for (uint32_t m = 0; m < n_mov && m < BOX_PARTS; m++) {
        float lo[3], hi[3];
        hull_aabb_rotated(mov + 3 * (size_t)moff[m], moff[m + 1] - moff[m], R, lane, lo, hi);
        if (lane == 0)
            for (int k = 0; k < 3; k++) {
                mb[m][k] = lo[k];
                mb[m][3 + k] = hi[k];
            }
    }
    __builtin_amdgcn_wave_barrier();
    // PyBullet's pairwise test between two bodies is true when ANY pair of their convex parts is in contact
    auto touches_any = [&](V3 t) -> bool {
        const float tt[3] = {t.x, t.y, t.z};
        for (uint32_t m = 0; m < n_mov; m++)
            for (uint32_t h = 0; h < n_stat; h++) {
                if (m < BOX_PARTS) {               // boxes (widened by a hair beyond the contact distance) apart: no GJK needed
                    const float *sb = stat_box + 6 * (size_t)h;
                    const float slack = P.margin2 + 1e-5f;
                    bool apart = false;
#pragma unroll
                    for (int k = 0; k < 3; k++) apart = apart || mb[m][k] + tt[k] > sb[3 + k] + slack || sb[k] > mb[m][3 + k] + tt[k] + slack;
                    if (apart) continue;
                }
                if (gjk_intersect(mov + 3 * (size_t)moff[m], moff[m + 1] - moff[m], R, t, stat + 3 * (size_t)off[h], off[h + 1] - off[h],
                                  lane, P.margin2))
                    return true;
            }
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
              hipMalloc(&p->d_off, ((size_t)n_static + 1) * 4) == hipSuccess &&
              hipMalloc(&p->d_stat_box, std::max<size_t>(1, (size_t)n_static) * 24) == hipSuccess;
    std::vector<float> boxes((size_t)std::max<uint32_t>(1, n_static) * 6, 0.f);
    for (uint32_t h = 0; h < n_static; h++)
        for (int k = 0; k < 3; k++) {
            float lo = INFINITY, hi = -INFINITY;
            for (uint32_t i = static_offsets[h]; i < static_offsets[h + 1]; i++) {
                lo = std::min(lo, static_verts[3 * (size_t)i + k]);
                hi = std::max(hi, static_verts[3 * (size_t)i + k]);
            }
            boxes[6 * (size_t)h + k] = lo;
            boxes[6 * (size_t)h + 3 + k] = hi;
        }
    ok = ok && hipMemcpy(p->d_mov, movable_verts, (size_t)p->n_mov_verts * 12, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_moff, movable_offsets, ((size_t)n_movable + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         (p->n_stat_verts == 0 || hipMemcpy(p->d_stat, static_verts, (size_t)p->n_stat_verts * 12, hipMemcpyHostToDevice) == hipSuccess) &&
         hipMemcpy(p->d_off, n_static ? static_offsets : &zero, ((size_t)n_static + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_stat_box, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
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
    if (p->d_stat_box) (void)hipFree(p->d_stat_box);
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
                       (const uint32_t *)phys->d_off, phys->n_stat, (const float *)phys->d_stat_box, d_valid);
    D2R_HIP(ctx, hipGetLastError());
    D2R_HIP(ctx, hipMemcpyAsync(valid_io, d_valid, N, hipMemcpyDeviceToHost, ctx->stream));
    D2R_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return D2R_OK;
}

}  // extern "C"
