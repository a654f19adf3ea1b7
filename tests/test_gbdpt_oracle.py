"""CPU tests of oracle/gbdpt_oracle.hpp, the restatement of the reference's G-BDPT sampler (GBDPTRenderer::process / evaluate over libbidir,
src/integrators/gbdpt/gbdpt_proc.cpp:86-662, src/libbidir/{path,vertex,edge,mut_manifold,manifold}.cpp), including -- round 4 -- paths with
specular chains (propagatePerturbation, the manifold walk, generalized geometry terms).  PARITY UNPINNED (the reference cannot be built here): what holds the restatement is
  * closed forms: a directly seen emitter (the only strategy of maxDepth 1 without the light image) returns exactly its radiance; the
    "T0" gradient of an unshiftable path is -2 w f (gbdpt_proc.cpp:494-499,517-521);
  * the estimator's expectation: the primal image converges to -throughput + -direct of the G-PT oracle (an independent restatement of
    another integrator, itself checked against a plain path tracer), with and without the light image -- the MIS weights of
    miWeightBaseNoSweep_GBDPT sum to one over the strategies; the merged gradients (gbdpt.cpp:211-212) converge to the finite differences
    of that image -- the Jacobians and the balance-heuristic weights of miWeightGradNoSweep_GBDPT are consistent."""
import numpy as np
import pytest

from gradientdomain_mitsuba_amd import scenes
from oracle import gpt_oracle as go


def merged_gradients(img):
    """GBDPTIntegrator::prepareDataForSolver's merge (gbdpt.cpp:211-212,264-280) on developed buffers: dx = (posX - negX[x + 1]) / 2."""
    gx = 0.5 * img[3].copy(); gx[:, :-1] -= 0.5 * img[2][:, 1:]
    gy = 0.5 * img[4].copy(); gy[:-1] -= 0.5 * img[1][1:]
    return gx, gy


@pytest.fixture(scope="module")
def converged():
    W, H, spp, md = 20, 15, 384, 4
    sc = scenes.cornell_box(W, H, "diffuse")
    O = go.Scene(sc)
    out = {}
    for li in (True, False):
        b, l, c = O.gbdpt_render(go.gbdpt_config(maxDepth=md, spp=spp, lightImage=li))
        assert c["unsupported"] == 0 and c["invalidPuts"] == 0
        out[li] = go.gbdpt_develop(b, l, spp)
    acc, _ = O.render(go.config(maxDepth=md, spp=spp))
    dev = go.develop(acc)
    out["pt"] = dev[1] + dev[4]
    return out


def test_primal_converges_to_the_gpt_oracle(converged):
    pt = converged["pt"]
    for li in (True, False):
        img = converged[li][0]
        assert abs(img.mean() - pt.mean()) <= 0.02 * pt.mean(), (li, img.mean(), pt.mean())
        # per colour channel and per image quadrant (a wrong strategy weight shows up where that strategy dominates)
        for c in range(3):
            assert abs(img[..., c].mean() - pt[..., c].mean()) <= 0.03 * pt[..., c].mean()
        h, w = pt.shape[:2]
        for ys in (slice(0, h // 2), slice(h // 2, h)):
            for xs in (slice(0, w // 2), slice(w // 2, w)):
                assert abs(img[ys, xs].mean() - pt[ys, xs].mean()) <= 0.06 * pt[ys, xs].mean()


def test_merged_gradients_converge_to_finite_differences(converged):
    pt = converged["pt"]
    for li in (True, False):
        gx, gy = merged_gradients(converged[li])
        fdx, fdy = pt[:, 1:] - pt[:, :-1], pt[1:] - pt[:-1]
        scale = np.abs(fdx).mean() + np.abs(fdy).mean()
        # the mean error of the gradient estimate against finite differences of the (noisy) converged image is a small fraction of the
        # mean gradient magnitude, and the estimate is far better than "no gradient"
        ex, ey = np.abs(gx[:, :-1] - fdx).mean(), np.abs(gy[:-1] - fdy).mean()
        assert ex + ey <= 0.25 * scale, (li, ex, ey, scale)
        assert abs((gx[:, :-1] - fdx).mean()) <= 0.02 * scale and abs((gy[:-1] - fdy).mean()) <= 0.02 * scale      # unbiased


def test_directly_seen_emitter_known_answer():
    """maxDepth 1 without the light image: the one strategy is s = 0, t = 2 (the sensor subpath hits the emitter), weight 1, value = Le;
    such a path cannot be shifted (gbdpt_proc.cpp:200), so all four gradients take the T0 form 2 w (0 - f) = -2 Le."""
    W, H = 32, 32
    sc = scenes.cornell_box(W, H, "diffuse")
    O = go.Scene(sc)
    cfg = go.gbdpt_config(maxDepth=1, spp=1, lightImage=False)
    Le = np.array([17.0, 12.0, 4.0])
    seen = 0
    for py in range(0, 6):
        for px in range(8, 24):
            r = O.gbdpt_sample(cfg, px, py, 0)
            assert r["unsupported"] == 0 and len(r["light"]) == 0
            if r["primal"].any():
                seen += 1
                assert np.allclose(r["primal"], Le, rtol=1e-12)
                assert np.allclose(r["gradients"], -2 * Le[None, :], rtol=1e-12)
            else:
                assert not r["gradients"].any()
    assert seen >= 4


def test_light_image_strategies_split_the_directly_seen_emitter():
    """With the light image the directly seen emitter has two strategies (s = 0, t = 2 and s = 1, t = 1); in expectation their weighted
    sum is Le again."""
    W, H, spp = 16, 16, 2048
    sc = scenes.cornell_box(W, H, "diffuse")
    O = go.Scene(sc)
    rect = (6, 1, 10, 2)                                              # four pixels that look at the light
    b, l, c = O.gbdpt_render(go.gbdpt_config(maxDepth=1, spp=spp, lightImage=True), rect)
    img = go.gbdpt_develop(b, l, spp)[0]
    b0, l0, _ = O.gbdpt_render(go.gbdpt_config(maxDepth=1, spp=spp, lightImage=False), rect)
    ref = go.gbdpt_develop(b0, l0, spp)[0]
    # the light-tracing strategy splats over the whole light: compare the total energy on the film
    whole, _l, _c = O.gbdpt_render(go.gbdpt_config(maxDepth=1, spp=64, lightImage=True))
    whole0, _l0, _c0 = O.gbdpt_render(go.gbdpt_config(maxDepth=1, spp=64, lightImage=False))
    a = go.gbdpt_develop(whole, _l, 64)[0].sum(axis=(0, 1)); a0 = go.gbdpt_develop(whole0, _l0, 64)[0].sum(axis=(0, 1))
    assert np.allclose(a, a0, rtol=0.05), (a, a0)
    assert ref[1, 6:10].min() > 0 and img[1, 6:10].min() >= 0


def test_samples_are_reproducible_and_depend_on_the_seed():
    sc = scenes.cornell_box(24, 18, "twosided")
    O = go.Scene(sc)
    cfg = go.gbdpt_config(maxDepth=6, spp=4)
    a, b = O.gbdpt_sample(cfg, 11, 9, 2), O.gbdpt_sample(cfg, 11, 9, 2)
    assert np.array_equal(a["primal"], b["primal"]) and np.array_equal(a["gradients"], b["gradients"]) and np.array_equal(a["light"], b["light"])
    c = O.gbdpt_sample(go.gbdpt_config(maxDepth=6, spp=4, seed=7), 11, 9, 2)
    assert not np.array_equal(a["primal"], c["primal"]) or not np.array_equal(a["light"], c["light"])
    assert a["unsupported"] == 0 and a["raysTraced"] > 0


@pytest.mark.parametrize("variant", ["glass", "glossy", "nearspecular"])
def test_specular_chains_estimator_expectation(variant):
    """Stage C (round 4): offset paths through SPECULAR CHAINS -- ManifoldPerturbation::propagatePerturbation on the camera side, the manifold
    walk of SpecularManifold::{init, computeTangents, project, move, update} between b and c, the generalized geometry terms and determinants
    of SpecularManifold::{G, multiG, det} in the Jacobians and MIS weights (mut_manifold.cpp:989-1227, manifold.cpp:59-951, path.cpp:380-454).
    Scenes: a solid glass block + an aluminium mirror block ("glass": refraction chains, a non-symmetric BSDF), a mirror back wall ("glossy"),
    a rough conductor BELOW shiftThreshold ("nearspecular": glossy vertices inside a chain, perturbed with their half vector kept).
    What holds it: the primal image still converges to the G-PT oracle's path tracer and the merged gradients to its finite differences --
    a wrong Jacobian, determinant or generalized G biases exactly these -- and the walks are really taken (thousands per frame, most converge)."""
    W, H, spp, md = 20, 15, 256, 5
    sc = scenes.cornell_box(W, H, variant)
    O = go.Scene(sc)
    acc, _ = O.render(go.config(maxDepth=md, spp=4 * spp))
    dev = go.develop(acc)
    pt = dev[1] + dev[4]
    fdx, fdy = pt[:, 1:] - pt[:, :-1], pt[1:] - pt[:-1]
    scale = np.abs(fdx).mean() + np.abs(fdy).mean()
    for li in (True, False):
        b, l, c = O.gbdpt_render(go.gbdpt_config(maxDepth=md, spp=spp, lightImage=li))
        assert c["unsupported"] == 0 and c["invalidPuts"] == 0
        assert c["manifoldWalks"] > 1000 and c["manifoldWalksConverged"] > 0.5 * c["manifoldWalks"] and c["propagatedVertices"] > 100, c
        img = go.gbdpt_develop(b, l, spp)
        assert abs(img[0].mean() - pt.mean()) <= 0.03 * pt.mean(), (variant, li, img[0].mean(), pt.mean())
        h, w = pt.shape[:2]
        for ys in (slice(0, h // 2), slice(h // 2, h)):
            for xs in (slice(0, w // 2), slice(w // 2, w)):
                assert abs(img[0][ys, xs].mean() - pt[ys, xs].mean()) <= 0.08 * pt[ys, xs].mean(), (variant, li)
        gx, gy = merged_gradients(img)
        ex, ey = np.abs(gx[:, :-1] - fdx).mean(), np.abs(gy[:-1] - fdy).mean()
        assert ex + ey <= 0.3 * scale, (variant, li, ex, ey, scale)
        assert abs((gx[:, :-1] - fdx).mean()) <= 0.03 * scale and abs((gy[:-1] - fdy).mean()) <= 0.03 * scale, (variant, li)
    O.close()


@pytest.mark.parametrize("what", ["thinlens", "thinlens_wide", "point_beside_area", "points_only", "environment", "environment_only_glass", "envmap", "envmap_only_rotated"])
def test_thinlens_sensor_and_point_emitters_estimator_expectation(what):
    """Round 5: the thinlens sensor (thinlens.cpp: aperture position sample, importance through the pixel's focus-plane point, one more emitter step) and `point`
    emitters (point.cpp: discrete position measure, uniform directions, no cosine; no extra sensor step when every emitter is one) in the G-BDPT oracle.  What
    holds them: the primal image converges to the G-PT oracle's path tracer -- an estimator that shares the scene and BSDF code but none of the bidirectional
    layer (it samples the lens in sampleRay and the point light in sampleEmitterDirect) -- and the merged gradients to its finite differences.  A point light is
    also SEEN by the light image (s = 1, t = 1: one bright pixel a path tracer cannot produce): that pixel is left out of the comparison."""
    W, H, spp, md = 20, 15, 256, 4
    sc = scenes.cornell_box(W, H, "glass" if what == "environment_only_glass" else "diffuse")
    # the `constant` environment: to libbidir a sphere-shaped area light with a black BSDF that every escaping ray hits (scene.cpp:397-408, constant.cpp:67-160); the
    # path tracer evaluates it on escaping rays and samples it by direction (constant.cpp:163-205) -- no sphere, no surface vertex
    if what == "environment": sc.environment = ((0.6, 0.7, 0.9), len(sc.emitters))
    if what == "environment_only_glass": sc.emitters = []; sc.environment = ((0.6, 0.7, 0.9), 0)
    # the `envmap` environment (round 5, the last endpoint kind): the same sphere, positions uniform on it, but directions importance-sampled FROM THE MAP whatever the
    # position (envmap.cpp:412-498: a stated compromise) and a coloured evalDirection -- against the path tracer's evalEnvironment / sampleDirect (envmap.cpp:378-410,509-549);
    # a smooth map (no sun patch): the emitter subpaths of this strategy mostly miss the scene, and a 40x texel would leave the comparison to a handful of paths
    if what == "envmap": sc.environment_map = dict(rgb=scenes.sky_map(16, 8, sun=1.5), scale=1.3, index=len(sc.emitters))
    if what == "envmap_only_rotated":
        sc.emitters = []
        a = 0.7; rot = [[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]]
        sc.environment_map = dict(rgb=scenes.sky_map(16, 8, sun=1.5), scale=2.0, index=0, toWorld=rot)
    if what == "thinlens": sc.thinlens = (40.0, 1100.0)
    if what == "thinlens_wide": sc.thinlens = (120.0, 600.0)
    pl = ("point", (300.0, 400.0, 279.5), (4e4, 3e4, 2e4))        # (off the film's centre line: a splat exactly between two pixels lands in both, imageblock.h)
    if what == "point_beside_area": sc.emitters = sc.emitters + [pl]
    if what == "points_only": sc.emitters = [pl, ("point", (120.0, 90.0, 140.0), (1e4, 2e4, 3e4))]
    O = go.Scene(sc)
    acc, _ = O.render(go.config(maxDepth=md, spp=4 * spp))
    dev = go.develop(acc)
    pt = dev[1] + dev[4]
    fdx, fdy = pt[:, 1:] - pt[:, :-1], pt[1:] - pt[:-1]
    scale = np.abs(fdx).mean() + np.abs(fdy).mean()
    # (a point light BESIDE an area light, lightImage off: a sensor path that hits the area light while the emitter subpath drew the point light asks
    #  miWeightBaseNoSweep_GBDPT for strategy s = 0 with the emitter supernode's measure EDiscrete -- "not connectable", path.cpp:81,192 -- so the sum of
    #  strategy pdfs is 0 and the weight infinite: the reference drops such a sample as an invalid put (imageblock.h:160-175), and so do oracle and device.
    #  With the light image on the t = 1 strategy keeps the sum positive.  That combination is therefore compared with the light image only.)
    for li in ((True,) if what == "point_beside_area" else (True, False)):
        b, l, c = O.gbdpt_render(go.gbdpt_config(maxDepth=md, spp=spp, lightImage=li))
        assert c["unsupported"] == 0 and c["invalidPuts"] == 0
        img = go.gbdpt_develop(b, l, spp)
        keep = np.ones((H, W), bool)
        if li and what.startswith("point"):
            d = (img[0] - pt).sum(-1)
            for _ in range(2 if what == "points_only" else 1):
                y, x = np.unravel_index(np.argmax(np.where(keep, d, -np.inf)), d.shape)
                if d[y, x] > 5 * pt.mean(): keep[y, x] = False   # the light's own image: far brighter than anything it lights (the second light sits behind the short block)
            assert not keep.all()
        assert abs(img[0][keep].mean() - pt[keep].mean()) <= 0.03 * pt[keep].mean(), (what, li, img[0][keep].mean(), pt[keep].mean())
        h, w = pt.shape[:2]
        for ys in (slice(0, h // 2), slice(h // 2, h)):
            for xs in (slice(0, w // 2), slice(w // 2, w)):
                k = keep[ys, xs]
                assert abs(img[0][ys, xs][k].mean() - pt[ys, xs][k].mean()) <= 0.08 * pt[ys, xs][k].mean(), (what, li)
        gx, gy = merged_gradients(img)
        kx, ky = keep[:, :-1] & keep[:, 1:], keep[:-1] & keep[1:]
        ex, ey = np.abs(gx[:, :-1] - fdx)[kx].mean(), np.abs(gy[:-1] - fdy)[ky].mean()
        # (the wide aperture blurs the image: its gradients are small against the noise of the path-traced finite differences they are compared with)
        assert ex + ey <= (0.5 if what == "thinlens_wide" else 0.3) * scale, (what, li, ex, ey, scale)
    O.close()


def test_planar_mirror_known_answers_of_the_manifold():
    """Closed forms for a chain "diffuse a -- planar mirror m -- diffuse b" (the aluminium back wall of the "glossy" box):
      * SpecularManifold::G(a, b) (computeTangents + the tangent map of vertex 1, manifold.cpp:900-951) is the PLAIN geometry term between a
        and the mirror image b' of b: |cos_a| |cos_b'| / |a - b'|^2 (the mirror patch a sees subtends the solid angle of the image of b's patch);
      * SpecularManifold::move: the constraint is linear for a planar mirror, so the Newton walk lands on the exact chain vertex -- the
        intersection of the segment a -> b' with the mirror plane -- at its FIRST step (the second iteration only finds it has arrived)."""
    sc = scenes.cornell_box(32, 24, "glossy")
    O = go.Scene(sc)
    cfg = go.gbdpt_config(maxDepth=6, spp=4)
    seen = 0
    for py in range(0, 24, 2):
        for px in range(0, 32, 2):
            r = O.manifold_probe(cfg, px, py, 0)
            if r is None or r["material"] != 1:          # (conductor)
                continue
            nm = r["nm"]
            b_img = r["b"] - 2 * np.dot(r["b"] - r["m"], nm) * nm
            d = b_img - r["a"]; dist = np.linalg.norm(d); d /= dist
            nb_img = r["nb"] - 2 * np.dot(r["nb"], nm) * nm
            closed = abs(np.dot(d, r["na"])) * abs(np.dot(d, nb_img)) / dist ** 2
            assert np.isclose(r["G"], closed, rtol=1e-7), (px, py, r["G"], closed)
            t = np.cross(r["nb"], [0.3, 0.5, 0.8]); t /= np.linalg.norm(t)
            delta = 3.0 * t                                # b moves 3 units (of a 550-unit box) within its tangent plane
            w = O.manifold_probe(cfg, px, py, 0, delta)
            bt = r["b"] + delta
            bimg = bt - 2 * np.dot(bt - r["m"], nm) * nm
            dd = bimg - r["a"]
            mexp = r["a"] + (np.dot(r["m"] - r["a"], nm) / np.dot(dd, nm)) * dd
            if 0 < mexp[0] < 550 and 0 < mexp[1] < 540:  # (the exact vertex is still on the wall)
                assert w["converged"] and w["iterations"] <= 2, (px, py, w)
                assert np.abs(w["m_moved"] - mexp).max() <= 1e-4 and np.abs(w["b_moved"] - bt).max() <= 1e-4, (px, py, w["m_moved"], mexp)
                seen += 1
    assert seen >= 20
    O.close()


def _hand_bsphere_radius(lo, hi, extra_points):
    """Scene::getBSphere().radius derived by hand from the reference, independent of the oracle's code: GenericKDTree::buildInternal enlarges the
    bounds of the geometry by MTS_KD_AABB_EPSILON = 1e-3f -- `m_aabb.min -= (m_aabb.max - m_aabb.min) * eps + Vector(eps)` and then the same for
    max with the ALREADY MOVED min (gkdtree.h:50,1213-1219); Scene::initializeBidirectional expands that box by the sensor's and every emitter's
    AABB (scene.cpp:386-413); AABB::getBSphere is centre (max + min) / 2, radius |centre - max| (aabb.h:132-134, aabb.cpp:44-47)."""
    eps = float(np.float32(1e-3))
    lo = np.asarray(lo, np.float64); hi = np.asarray(hi, np.float64)
    lo = lo - ((hi - lo) * eps + eps)
    hi = hi + ((hi - lo) * eps + eps)
    for p in extra_points:
        lo = np.minimum(lo, p); hi = np.maximum(hi, p)
    ctr = (hi + lo) * 0.5
    return float(np.sqrt(((ctr - hi) ** 2).sum()))


def test_scene_bounding_sphere_includes_the_sensor_and_the_emitters():
    """ManifoldPerturbation::manifoldWalk's reversibility test divides by m_scene->getBSphere().radius (mut_manifold.cpp:1219), the sphere of the
    box Scene::initializeBidirectional builds: kd-tree bounds + sensor + emitters.  The Cornell box's camera stands at z = -800, outside the
    geometry (z in [0, 559.2]): the radius is 1.63x the kd-tree's alone -- rounds 3-4 took the kd-tree's."""
    lo, hi = (0.0, 0.0, 0.0), (556.0, 548.8, 559.2)           # the room: floor / ceiling / walls of scenes.cornell_box
    cam = np.array([278.0, 273.0, -800.0])
    O = go.Scene(scenes.cornell_box(16, 16, "glass"))
    want = _hand_bsphere_radius(lo, hi, [cam])
    assert abs(want - 784.4928504358834) < 1e-9                # (the number itself: half extents 278.6, 275.0, 680.0 with the enlargement)
    assert abs(O.bsphere_radius() - want) <= 1e-12 * want
    assert O.bsphere_radius() > 1.6 * _hand_bsphere_radius(lo, hi, [])
    O.close()
    # a point emitter outside the box (point.cpp:153-155: its AABB is its position) moves the sphere too; an area emitter's AABB is its shape's
    pl = np.array([900.0, 100.0, 200.0])
    O = go.Scene(scenes.cornell_box(16, 16, "diffuse", point_light=(tuple(pl), (1.0, 1.0, 1.0))))
    assert abs(O.bsphere_radius() - _hand_bsphere_radius(lo, hi, [cam, pl])) <= 1e-12 * want
    O.close()
    # thinlens: the sensor's AABB is the spatial bounds of the aperture box (-r, -r, 0) .. (r, r, 0) under its toWorld (thinlens.cpp:516-520);
    # lookAt along +z with up +y: the aperture's axes are world x and y
    sc = scenes.cornell_box(16, 16, "diffuse")
    sc.thinlens = (25.0, 900.0)
    O = go.Scene(sc)
    corners = [cam + np.array([sx * 25.0, sy * 25.0, 0.0]) for sx in (-1, 1) for sy in (-1, 1)]
    assert abs(O.bsphere_radius() - _hand_bsphere_radius(lo, hi, corners)) <= 1e-12 * want
    O.close()
    # a constant environment adds the centre of its own sphere (constant.cpp:234-240): inside the box, nothing changes
    O = go.Scene(scenes.cornell_box(16, 16, "diffuse", environment=(0.5, 0.5, 0.5)))
    assert abs(O.bsphere_radius() - want) <= 1e-12 * want
    O.close()


def _reflect_point(p, q, n):
    return p - 2 * np.dot(p - q, n) * n


def test_two_facing_mirrors_known_answers_of_the_manifold():
    """A chain "diffuse a -- mirror m1 -- mirror m2 -- diffuse b" (the aluminium back wall and the aluminium tall block of the "mirrors" box), closed
    forms that share no code with the oracle:
      * seen from a, b stands at its DOUBLE mirror image b'' = R_m1(R_m2(b)): SpecularManifold::G(a, b) (computeTangents over a block-tridiagonal
        system with two interior vertices, manifold.cpp:172-400,900-951) and Path multiG (one chain between two connectable vertices) are the plain
        geometry term |cos_a| |cos_b''| / |a - b''|^2;
      * both constraints are linear: the Newton walk arrives at the exact chain -- m1' = the segment a -> b'' cut by the plane of mirror 1, m2' = the
        segment m1' -> R_m2(b) cut by the plane of mirror 2 -- in its first step."""
    sc = scenes.cornell_box(40, 30, "mirrors")
    O = go.Scene(sc)
    cfg = go.gbdpt_config(maxDepth=8, spp=8)
    seen = walked = multi = 0
    for py in range(0, 30):
        for px in range(0, 40):
            for smp in range(2):
                r = O.manifold_probe2(cfg, px, py, smp)
                if r is None or r["materials"] != (1, 1):
                    continue
                a, m1, m2, b = r["p"]; na, n1, n2, nb = r["n"]
                if abs(np.dot(n1, n2)) > 0.999:              # (the same plane twice cannot be: a flat mirror does not see itself)
                    continue
                b1 = _reflect_point(b, m2, n2); b2 = _reflect_point(b1, m1, n1)
                nb2 = nb - 2 * np.dot(nb, n2) * n2; nb2 = nb2 - 2 * np.dot(nb2, n1) * n1
                d = b2 - a; dist = np.linalg.norm(d); d /= dist
                closed = abs(np.dot(d, na)) * abs(np.dot(d, nb2)) / dist ** 2
                assert np.isclose(r["G"], closed, rtol=1e-7), (px, py, smp, r["G"], closed)
                assert r["multiG"] == -1.0 or np.isclose(r["multiG"], closed, rtol=1e-7), (px, py, smp, r["multiG"], closed)
                seen += 1; multi += r["multiG"] != -1.0
                t = np.cross(nb, [0.3, 0.5, 0.8]); t /= np.linalg.norm(t)
                delta = 2.0 * t
                w = O.manifold_probe2(cfg, px, py, smp, delta)
                bt = b + delta
                bt1 = _reflect_point(bt, m2, n2); bt2 = _reflect_point(bt1, m1, n1)
                dd = bt2 - a
                m1e = a + (np.dot(m1 - a, n1) / np.dot(dd, n1)) * dd
                d2 = bt1 - m1e
                m2e = m1e + (np.dot(m2 - m1e, n2) / np.dot(d2, n2)) * d2
                # (the exact vertices must still lie on the faces the chain started on: a couple of units away from where they were, well inside)
                if np.abs(m1e - m1).max() < 8 and np.abs(m2e - m2).max() < 8 and w["converged"]:
                    assert w["iterations"] <= 2, (px, py, smp, w["iterations"])
                    assert np.abs(w["moved"][0] - m1e).max() <= 1e-4 and np.abs(w["moved"][1] - m2e).max() <= 1e-4 and np.abs(w["moved"][2] - bt).max() <= 1e-4, (px, py, smp)
                    walked += 1
    assert seen >= 20 and walked >= 10 and multi >= 5, (seen, walked, multi)
    O.close()


def test_glass_slab_known_answers_of_the_manifold():
    """A chain "diffuse a -- refraction r1 -- refraction r2 -- diffuse b" through two PARALLEL faces of the rectangular glass block of the "slab" box (a slab
    of thickness d, relative index eta), in closed form from Snell's law alone: a ray leaving a at angle t1 to the slab normal runs at t2 inside
    (sin t1 = eta sin t2) and leaves parallel to itself; in a plane parallel to the slab it lands at radius r(t1) = (h1 + h2) tan t1 + d tan t2 from
    the foot of a (h1, h2: the distances a -> slab, slab -> that plane).  The beam of solid angle dW = sin t1 dt1 dphi covers r r' dt1 dphi of that plane,
    cos t1 of it across the beam, so on b's surface dA_b = r r' cos t1 / (sin t1 |n_b . w|) dW and SpecularManifold::G(a, b) = |n_a . w| dW / dA_b
    (manifold.cpp:900-951: the plain term a <-> r1 times the area ratio of the tangent map).  The walk: b moved within its plane, the chain's new
    entry point is where the ray of the angle that solves r(t1) = R (bisection here) meets the first face."""
    sc = scenes.cornell_box(40, 30, "slab")
    O = go.Scene(sc)
    cfg = go.gbdpt_config(maxDepth=8, spp=8)
    seen = walked = multi = 0
    for py in range(0, 30):
        for px in range(0, 40):
            for smp in range(2):
                r = O.manifold_probe2(cfg, px, py, smp)
                if r is None or r["materials"] != (3, 3):
                    continue
                a, r1, r2, b = r["p"]; na, n1, n2, nb = r["n"]
                if abs(np.dot(n1, n2)) < 0.999999:           # (entered through one face, left through an adjacent one: a prism, not a slab)
                    continue
                eta = r["eta"]
                N = n1 if np.dot(n1, r1 - a) > 0 else -n1    # slab normal along the direction of travel
                w = r1 - a; w /= np.linalg.norm(w)
                h1 = np.dot(r1 - a, N); d = np.dot(r2 - r1, N); h2 = np.dot(b - r2, N)
                if not (h1 > 0 and d > 0 and h2 > 0):       # (reflected inside at the second face and left through the block's open bottom: not the slab's chain)
                    continue
                c1 = np.dot(w, N); s1 = np.sqrt(1 - c1 * c1); s2 = s1 / eta; c2 = np.sqrt(1 - s2 * s2)
                rr = (h1 + h2) * s1 / c1 + d * s2 / c2
                drr = (h1 + h2) / c1 ** 2 + d * (c1 / (eta * c2)) / c2 ** 2
                # the radius formula itself, against the chain the oracle traced: b's offset from the foot of a
                foot = b - a - np.dot(b - a, N) * N
                assert np.isclose(np.linalg.norm(foot), rr, rtol=1e-9), (px, py, smp)
                jac = rr * drr / s1 if s1 > 1e-6 else (h1 + h2 + d / eta) ** 2
                closed = abs(np.dot(na, w)) * abs(np.dot(nb, w)) / (c1 * jac)
                assert np.isclose(r["G"], closed, rtol=1e-6), (px, py, smp, r["G"], closed)
                assert r["multiG"] == -1.0 or np.isclose(r["multiG"], closed, rtol=1e-6), (px, py, smp)
                seen += 1; multi += r["multiG"] != -1.0
                # the walk: b moves within ITS tangent plane; the new chain by bisection on r(t1) in the plane of incidence through a, N and the new b
                t = np.cross(nb, [0.3, 0.5, 0.8]); t /= np.linalg.norm(t)
                delta = 1.5 * t
                wk = O.manifold_probe2(cfg, px, py, smp, delta)
                bt = b + delta
                H2 = np.dot(bt - r2, N)                      # (b's plane need not be parallel to the slab: its distance changes with the move)
                ft = bt - a - np.dot(bt - a, N) * N
                R = np.linalg.norm(ft); e = ft / R
                f = lambda th: (h1 + H2) * np.tan(th) + d * np.tan(np.arcsin(np.sin(th) / eta)) - R
                lo, hi = 0.0, np.pi / 2 - 1e-9
                for _ in range(200):
                    mid = 0.5 * (lo + hi)
                    lo, hi = (mid, hi) if f(mid) < 0 else (lo, mid)
                th = 0.5 * (lo + hi)
                r1e = a + h1 * N + h1 * np.tan(th) * e
                th2 = np.arcsin(np.sin(th) / eta)
                r2e = r1e + d * N + d * np.tan(th2) * e
                if wk["converged"] and np.abs(r1e - r1).max() < 6 and np.abs(r2e - r2).max() < 6:
                    assert np.abs(wk["moved"][0] - r1e).max() <= 1e-4 and np.abs(wk["moved"][1] - r2e).max() <= 1e-4 and np.abs(wk["moved"][2] - bt).max() <= 1e-4, (px, py, smp, wk["moved"], r1e, r2e)
                    walked += 1
    assert seen >= 20 and walked >= 10 and multi >= 5, (seen, walked, multi)
    O.close()
