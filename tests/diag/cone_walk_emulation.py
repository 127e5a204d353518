"""Development diagnostic (CPU): walk every ray of the shelf background the way k_march's next_sample does (brick /
cell skipping along the cone-step lattice, fp32, fma emulated through float64) and compare the set of lattice points it
visits in occupied cells with the oracle's exhaustive walk over every lattice point.  Tells algorithmic differences
(reproduced here) from hardware ones (approximate rcp / log2 on the GPU).  Uses the oracle: lives under tests/."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.scenes import make_scene
from oracle import host_ref, render_ref

f32 = np.float32
def fma(a, b, c): return f32(np.float64(f32(a)) * np.float64(f32(b)) + np.float64(f32(c)))
DT = f32(0.0016914558); INV_DT = f32(1.0) / DT; CONE = f32(0.00390625); T_LIN = DT * f32(256.0)
P2 = [f32(x) for x in (1.00390625, 1.0078277587890625, 1.015716791152954, 1.0316805839538574, 1.0643649101257324, 1.1328725814819336,
                       1.2834001779556274, 1.6471161842346191, 2.712991714477539, 7.360323429107666, 54.17436218261719, 2934.861572265625)]
def cone_pow(n):
    r = f32(1.0)
    for i in range(12):
        if (n >> i) & 1: r = f32(r * P2[i])
    return r

def main(kind="shelf", W=128, H=72):
    scene = make_scene(kind)
    m = scene.bg
    aabb = m.aabb_scale
    n_casc = int(aabb).bit_length()
    occ = m.occupancy_bool()                     # [c, z, y, x]
    side, inv_side = f32(aabb), f32(1.0) / f32(aabb)
    view = scene.view(W, H)
    cam = render_ref.nerf_matrix_to_ngp(host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0][:3], view.scale, view.offset).reshape(12)
    half = f32(0.5) * side
    blo, bhi = f32(0.5) - half, f32(0.5) + half
    n_bad = n_rays = 0
    for py in range(0, H, 1):
        for px in range(0, W, 1):
            u = (f32(px) + f32(0.5)) / f32(W); v = (f32(py) + f32(0.5)) / f32(H)
            dcx = f32(f32(u - f32(view.center[0])) * f32(W)) / f32(view.focal[0]); dcy = f32(f32(v - f32(view.center[1])) * f32(H)) / f32(view.focal[1])
            d = [fma(cam[i * 4 + 2], 1.0, fma(cam[i * 4 + 1], dcy, f32(cam[i * 4] * dcx))) for i in range(3)]
            o = [f32(cam[i * 4 + 3]) for i in range(3)]
            inv_len = f32(1.0) / f32(np.sqrt(fma(d[2], d[2], fma(d[1], d[1], f32(d[0] * d[0])))))
            d = [f32(x * inv_len) for x in d]
            tmin, tmax = f32(-np.inf), f32(np.inf)
            with np.errstate(divide="ignore"):
                for i in range(3):
                    inv = f32(1.0) / d[i]
                    a, b = f32(f32(blo - o[i]) * inv), f32(f32(bhi - o[i]) * inv)
                    tmin, tmax = max(tmin, min(a, b)), min(tmax, max(a, b))
            if not (tmax >= tmin and tmax > 0): continue
            t0 = f32(max(tmin, f32(0)) + f32(1e-6))
            k1 = 0 if t0 >= T_LIN else int(np.ceil(f32(f32(T_LIN - t0) * INV_DT)))
            t1 = fma(f32(k1), DT, t0)
            on = [fma(f32(o[i] - f32(0.5)), inv_side, 0.5) for i in range(3)]
            dn = [f32(d[i] * inv_side) for i in range(3)]
            lat = lambda k: fma(f32(k), DT, t0) if k <= k1 else f32(t1 * cone_pow(k - k1))
            def classify(k):
                t = lat(k)
                p = [fma(t, dn[i], on[i]) for i in range(3)]
                if any(p[i] < 0 or p[i] > 1 for i in range(3)): return None
                dt = max(DT, f32(t * CONE))
                mxw = f32(max(abs(f32(p[i] - f32(0.5))) for i in range(3)) * side)
                dt256 = f32(dt * f32(256)); hw, st, mip = f32(0.5), f32(1.0), 0
                for c in range(1, n_casc):
                    if mxw >= hw or dt256 >= st: mip = c
                    hw, st = f32(hw * 2), f32(st * 2)
                q = p if mip == n_casc - 1 else [fma(f32(p[i] - f32(0.5)), f32(side / f32(1 << mip)), 0.5) for i in range(3)]
                c = [min(max(int(f32(q[i] * f32(128))), 0), 127) for i in range(3)]
                return t, p, mip, c
            # oracle walk: every lattice point
            want, k = [], 0
            while k < 8192:
                r = classify(k)
                if r is None: break
                t, p, mip, c = r
                if occ[mip, c[2], c[1], c[0]]: want.append(k)
                k += 1
            k_end = k
            # kernel walk: skip empty bricks / cells (exact reciprocals here)
            got, k = [], 0
            with np.errstate(divide="ignore"):
                inv = [f32(1.0) / dn[i] for i in range(3)]
            while k < k_end:
                r = classify(k)
                if r is None: break
                t, p, mip, c = r
                if occ[mip, c[2], c[1], c[0]]:
                    got.append(k); k += 1; continue
                brick_empty = not occ[mip, (c[2] >> 2) * 4:(c[2] >> 2) * 4 + 4, (c[1] >> 2) * 4:(c[1] >> 2) * 4 + 4, (c[0] >> 2) * 4:(c[0] >> 2) * 4 + 4].any()
                sh = 2 if brick_empty else 0
                cell, corg = f32(1.0) / f32(128), f32(0)
                if mip != n_casc - 1:
                    sc = f32(side / f32(1 << mip)); cell = f32(1.0) / f32(f32(128) * sc); corg = f32(f32(0.5) - f32(0.5) / sc)
                cs = f32(f32(1 << sh) * cell)
                dist = f32(np.inf)
                for i in range(3):
                    l = fma(f32((c[i] >> sh) << sh), cell, corg)
                    dist = min(dist, f32(f32((f32(l + cs) if dn[i] > 0 else l) - p[i]) * inv[i]))
                if mip != n_casc - 1:
                    tb = f32(1.0)
                    while tb <= t: tb = f32(tb * 2)
                    dist = min(dist, f32(tb - t))
                n = int(np.floor(f32(dist / max(DT, f32(f32(t + max(dist, f32(0))) * CONE)))))
                k += max(n, 1)
            n_rays += 1
            if got != want:
                n_bad += 1
                if n_bad <= 5:
                    missing = sorted(set(want) - set(got))
                    print(f"pixel ({px},{py}): oracle {len(want)} occupied lattice points, walk {len(got)}; missed {missing[:8]}")
    print(f"{kind}: {n_rays} rays through the box, {n_bad} with a different set of occupied lattice points")

if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["shelf"]))
