"""Soak test of the closest-hit parity: many random scenes (ordinary magnitudes, a mix of
triangles and spheres, clustered so that near-ties and grazing hits occur) against the oracle.
   python tools/soak_parity.py [seeds] [rays per seed]
Prints the number of mismatching rays; exits non-zero if there is any."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
torch.cuda.init()
import numpy as np  # noqa: E402
import pt_amd as pt  # noqa: E402
import pt_oracle as orc  # noqa: E402

# Constants of the hierarchy's boxes (csrc/pt_bvh.hpp, k_bvh_prepare and bvh_begin; the membership
# rule is build_hierarchy's in csrc/pt_amd.hip), restated to measure how much of each box's padding
# the reference's hits actually use.
BVH_ORIGIN_ROUNDING, BVH_GRAZING, BVH_SPHERE_ROUNDING = 8e-6, 0.05, 4e-6


def pad_use(surf, o, d, want):
    """For every ray whose hit (by the oracle) is a surface the hierarchy holds: how far outside
    the surface's own bounding box the reported hit point lies, as a fraction of the padding the
    hierarchy gives that box (for a sphere: padding plus what the ray itself adds).  A hit the
    scan reports is found through the hierarchy as long as that fraction stays below 1.
    Returns {kind: (largest fraction, the padding's terms at that ray)}."""
    idx, _, point = want
    tri = surf["kind"] == pt.TRIANGLE
    a, b, c = (surf[k].astype(np.float64) for k in ("a", "b", "c"))
    centre = np.where(tri[:, None], (a + b + c) / 3.0, a)
    finite = np.isfinite(centre).all(1)
    spread = float((centre[finite].max(0) - centre[finite].min(0)).sum())
    lo = np.where(tri[:, None], np.minimum(np.minimum(a, b), c), a)
    hi = np.where(tri[:, None], np.maximum(np.maximum(a, b), c), a)
    size = (hi - lo).sum(1)
    radius = b[:, 0]
    reach_c = np.abs(a).sum(1)
    fits = ~tri & (radius > 0) & (6 * radius <= 0.25 * spread) & (radius >= 1e-3 * spread) & (reach_c <= 1e15)
    inside = (tri & (size <= 0.1 * spread)) | (fits & (fits.sum() >= 32))
    extent = np.maximum(np.abs(lo), np.abs(hi))
    extent[~tri] = np.abs(a[~tri]) + radius[~tri, None]
    ray_reach = min(4.0 * float(extent.max(0).sum()), 1e30)
    out = {}
    hit = (idx >= 0) & np.isfinite(point).all(1) & (np.abs(o).sum(1) <= ray_reach)
    hit[hit] &= inside[idx[hit]]
    for name, kind_mask in (("triangle", tri), ("sphere", ~tri)):
        rays = np.flatnonzero(hit & kind_mask[np.maximum(idx, 0)])
        if not len(rays):
            continue
        i = idx[rays]
        p = point[rays].astype(np.float64)
        if name == "triangle":
            outside = np.maximum(np.maximum(lo[i] - p, p - hi[i]), 0).max(1)
            l1, l2, l3 = (np.linalg.norm(v, axis=1) for v in (b[i] - a[i], c[i] - a[i], c[i] - b[i]))
            reach = np.maximum(np.abs(lo[i]), np.abs(hi[i])).max(1)
            terms = np.stack([1e-3 * size[i], 1e-5 * (1 + reach), np.full(len(i), BVH_ORIGIN_ROUNDING * ray_reach),
                              BVH_GRAZING * (ray_reach + reach) * l1 * l2 * np.maximum(np.maximum(l1, l2), l3)], 1)
            labels = ("edge tolerance 1e-3 size", "hit-point rounding", "origin rounding", "grazing")
        else:
            outside = np.maximum(np.abs(p - a[i]) - radius[i, None], 0).max(1)
            reach = np.abs(a[i]).max(1) + radius[i]
            in_r = radius[inside & ~tri]
            dd = d[rays].astype(np.float64)
            off_unit = np.abs((dd * dd).sum(1) - 1.0)
            far = np.abs(o[rays].astype(np.float64)).sum(1) + reach_c[inside & ~tri].max()
            terms = np.stack([1e-5 * (1 + reach) + 1e-5 * radius[i], np.full(len(i), BVH_ORIGIN_ROUNDING * ray_reach),
                              (BVH_SPHERE_ROUNDING + 2 * off_unit) / in_r.min() * far * far], 1)
            labels = ("hit-point rounding", "origin rounding", "the ray's own allowance for d2")
        use = outside / terms.sum(1)
        k = int(use.argmax())
        out[name] = (float(use[k]), float(outside[k]), dict(zip(labels, terms[k].tolist())), len(rays))
    return out


seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
m = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
ctx = pt.Context(0)
bad_total = redone = rays = 0
worst_use, pad_rays = {}, {}
t0 = time.time()
first_seed = int(os.environ.get("SOAK_FIRST_SEED", "0"))
for seed in range(first_seed, first_seed + seeds):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(4, 400))
    surf = np.zeros(n, dtype=pt.SURFACE)
    surf["kind"] = rng.random(n) < 0.3
    spread = float(rng.choice([0.5, 2.0, 8.0]))
    size = float(rng.choice([0.02, 0.3, 2.0]))
    surf["a"] = rng.uniform(-spread, spread, (n, 3))
    surf["b"] = surf["a"] + rng.uniform(-size, size, (n, 3))
    surf["c"] = surf["a"] + rng.uniform(-size, size, (n, 3))
    sph = surf["kind"] == pt.SPHERE
    surf["b"][sph, 0] = rng.uniform(0.05, 1.0, sph.sum()) * size
    dup = rng.integers(0, n, n // 8)                     # duplicates: exact ties
    surf[dup] = surf[rng.integers(0, n, n // 8)]
    mats = np.zeros(1, dtype=pt.MATERIAL)
    osc = orc.Scene(surf.view(orc.SURFACE), mats.view(orc.MATERIAL), [0, 0, 0],
                    orc.camera_init([0, 0, 0], 0, 0, 8, 0, 8, 8))
    o = rng.uniform(-1.5 * spread, 1.5 * spread, (m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # a third of the rays aim at points near surface edges and vertices (grazing the thresholds)
    k = rng.integers(0, n, m // 3)
    w = rng.choice([0.0, 1e-6, 1e-5, 0.5, 1.0 - 1e-6, 1.0], (m // 3, 2))
    target = surf["a"][k] + w[:, :1] * (surf["b"][k] - surf["a"][k]) + (1 - w[:, :1]) * w[:, 1:] * (surf["c"][k] - surf["a"][k])
    d[:m // 3] = (target - o[:m // 3]).astype(np.float32)
    # secondary rays: origins on surfaces
    i0, _, p0 = orc.intersect_rays(osc, o[m // 3:2 * m // 3], d[m // 3:2 * m // 3])
    hit = i0 >= 0
    o[m // 3:2 * m // 3][hit] = p0[hit]
    # a sixth of the rays run nearly parallel to a triangle's plane from far away, so that the
    # reference's determinant lies just above its 1e-5 threshold (1e-5 .. 1e-3), where its
    # fp32 u, v and t are least accurate and a hit can lie well off the triangle
    g = m // 6
    tri = np.flatnonzero(surf["kind"] == pt.TRIANGLE)
    if len(tri):
        k = rng.choice(tri, g)
        e1, e2 = (surf["b"][k] - surf["a"][k]).astype(np.float64), (surf["c"][k] - surf["a"][k]).astype(np.float64)
        nrm = np.cross(e1, e2)
        area2 = np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-30
        nrm /= area2
        w = rng.uniform(-0.1, 1.1, (g, 2))
        point = surf["a"][k] + w[:, :1] * e1 + w[:, 1:] * (1 - w[:, :1]) * e2
        phi = rng.uniform(0, 2 * np.pi, (g, 1))
        e1n = e1 / (np.linalg.norm(e1, axis=1, keepdims=True) + 1e-30)
        in_plane = np.cos(phi) * e1n + np.sin(phi) * np.cross(nrm, e1n)
        det = 10.0 ** rng.uniform(-5.3, -3.0, (g, 1)) * rng.choice([-1.0, 1.0], (g, 1))      # |d . N| wanted
        dirs = in_plane + (det / area2) * nrm
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        far = rng.choice([2.0, 5.0, 10.0, 20.0], (g, 1)) * spread
        o[-g:] = (point - far * dirs).astype(np.float32)
        d[-g:] = dirs.astype(np.float32)
    # a twelfth of the rays graze a sphere's silhouette from far away (unit directions, passing the
    # centre at radius * (1 +- 1e-6 .. 1e-3)): where the reference's d2 = L.L - tca^2 cancels most
    sph_idx = np.flatnonzero(surf["kind"] == pt.SPHERE)
    if len(sph_idx):
        h = m // 12
        lo = 2 * m // 3
        k = rng.choice(sph_idx, h)
        dirs = rng.normal(size=(h, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        perp = np.cross(dirs, rng.normal(size=(h, 3)))
        perp /= np.linalg.norm(perp, axis=1, keepdims=True) + 1e-30
        off = 1.0 + rng.choice([-1e-3, -1e-5, -1e-6, 0.0, 1e-6, 1e-5, 1e-3], (h, 1))
        far = rng.choice([1.0, 3.0, 10.0, 30.0], (h, 1)) * spread
        centre = surf["a"][k].astype(np.float64)
        radius = surf["b"][k, :1].astype(np.float64)
        o[lo:lo + h] = (centre + perp * radius * off - far * dirs).astype(np.float32)
        dn = dirs.astype(np.float32)
        d[lo:lo + h] = dn / np.linalg.norm(dn, axis=1, keepdims=True).astype(np.float32)
    want = orc.intersect_rays(osc, o, d)
    for kind, (use, outside, terms, n_rays) in pad_use(surf, o, d, want).items():
        if kind not in worst_use or use > worst_use[kind][0]:
            worst_use[kind] = (use, outside, terms, seed)
        pad_rays[kind] = pad_rays.get(kind, 0) + n_rays
    for accel in (pt.ACCEL_NONE, pt.ACCEL_BVH, pt.ACCEL_LDS_TILES):
        ctx.set_accel(accel)
        ctx.set_scene(surf, mats, [0, 0, 0])
        got = ctx.intersect_rays(o, d)
        bad = int((got[0] != want[0]).sum()) + int((~((got[1] == want[1]) | (np.isnan(got[1]) & np.isnan(want[1])))).sum())
        if bad:
            b = np.flatnonzero(got[0] != want[0])[:3]
            print("seed %d accel %d: %d mismatches, e.g. rays %s got %s want %s" % (
                seed, accel, bad, b.tolist(), got[0][b].tolist(), want[0][b].tolist()), flush=True)
            if os.environ.get("SOAK_DETAILS"):
                for r in b:
                    print("  ray %d o %s d %s |d|^2-1 %.3g" % (r, o[r].tolist(), d[r].tolist(),
                                                             float(np.dot(d[r].astype(np.float64), d[r].astype(np.float64)) - 1)))
                    for tag, res in (("got", got), ("want", want)):
                        i = int(res[0][r])
                        desc = "none" if i < 0 else "%s a %s b %s c %s" % (
                            "sphere" if surf["kind"][i] == pt.SPHERE else "triangle", surf["a"][i].tolist(),
                            surf["b"][i].tolist(), surf["c"][i].tolist())
                        print("   %s %d dist %r: %s" % (tag, i, float(res[1][r]), desc))
        bad_total += bad
    rays += m
    if seed % 10 == 9:
        print("%d seeds, %d rays x 3 modes, %d mismatches, %.0f s" % (seed + 1, rays, bad_total, time.time() - t0), flush=True)
print("total: %d rays x 3 modes, %d mismatches" % (rays, bad_total))
# The margin of the hierarchy's boxes as a number: the largest share of a box's padding that a hit
# of the reference's scan needed, over every ray that hit a surface inside a hierarchy.
thin = False
for kind, (use, outside, terms, seed) in sorted(worst_use.items()):
    print("padding of %s boxes: the reference's hit points lie at most %.3g outside the surface's own box = %.3f of the "
          "padding (seed %d, %d rays); its terms there: %s" % (
              kind, outside, use, seed, pad_rays[kind], ", ".join("%s %.3g" % kv for kv in terms.items())))
    thin |= use > 0.5
if thin:
    print("a hit used more than half of its box's padding: the margin is below 2x")
sys.exit(1 if bad_total or thin else 0)
