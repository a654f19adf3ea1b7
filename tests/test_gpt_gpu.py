"""GPU parity tests of the gfx950 G-PT sampler, called through the C-ABI (include/gdpt_tracer.h).

Checker: oracle/gpt_oracle.cpp (PARITY UNPINNED -- a line-cited fp64 restatement).  Both sides draw the
same counter-based random numbers, so they walk IDENTICAL paths; what differs is libm (ocml vs glibc sin/cos/exp/log/
acos/atan2/pow, <= a few ulp) and the order of fp64 sums in the film.  Bar: fp64, relative 1e-9 of the buffer scale
per pixel (observed ~1e-15), identical ray counts.
"""
import numpy as np
import pytest

from gradientdomain_mitsuba_amd import scenes
from oracle import gpt_oracle as go
from oracle import poisson_oracle as po

pytestmark = pytest.mark.gpu
REL = 1e-9


@pytest.fixture(scope="module")
def G(gpu_required):
    import gradientdomain_mitsuba_amd.gpt as G
    return G


def close(a, b, rel=REL):
    scale = np.abs(b).max() + 1e-300
    return np.abs(a - b).max() <= rel * scale


def test_node_layout_follows_residency_and_both_layouts_find_the_same_hits(G, monkeypatch):
    """Scenes staged into LDS get 128-byte nodes with fp32 boxes, scenes in HBM 64-byte nodes with 8-bit boxes on the node's grid and the early exit from
    the inner-node loop ("BVH nodes"); both are conservative, so a ray meets the same triangle at the same distance, bit for bit, either way."""
    box = scenes.cornell_box(32, 32, "glossy")
    c = G.Scene(box); lc = c.layout()
    assert lc["lds_resident"] and lc["node_bytes"] == 128 and lc["leaf_exit"] == 0 and 0 < lc["table_bytes"] <= 40 * 1024
    a = G.Scene(scenes.atrium(64, 36)); la = a.layout()
    assert not la["lds_resident"] and la["node_bytes"] == 64 and la["leaf_exit"] == 1 and la["nodes"] > 10000 and 4 < la["stack_entries"] < 40
    a.close()
    monkeypatch.setenv("GDPT_SCENE_IN_HBM", "1")
    h = G.Scene(box); lh = h.layout()
    monkeypatch.delenv("GDPT_SCENE_IN_HBM")
    assert not lh["lds_resident"] and lh["node_bytes"] == 64 and lh["nodes"] == lc["nodes"] and lh["stack_entries"] == lc["stack_entries"]
    rng = np.random.default_rng(2)
    n = 20000
    o = rng.uniform(50, 500, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pc, tc, _ = c.intersect(o, d); ph, th, _ = h.intersect(o, d)
    assert (pc >= 0).mean() > 0.5 and np.array_equal(pc, ph) and np.array_equal(tc, th)
    sc, sh = c.trace_stats(o, d), h.trace_stats(o, d)
    assert sh["nodes_closest"] <= 1.15 * sc["nodes_closest"] and sh["tris_closest"] <= 1.15 * sc["tris_closest"]       # looser boxes: a few visits more, not many
    c.close(); h.close()


@pytest.mark.parametrize("builder", [lambda: scenes.cornell_box(64, 64, "glossy"), lambda: scenes.atrium(64, 36, columns=8, segments=12)])
def test_bvh_closest_hit_matches_brute_force(G, builder):
    sc = builder()
    S, O = G.Scene(sc), go.Scene(sc)
    rng = np.random.default_rng(5)
    lo, hi = sc.verts.reshape(-1, 3).min(0), sc.verts.reshape(-1, 3).max(0)
    n = 3000
    o = lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))
    d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    prim, t, p = S.intersect(o, d)
    mism = 0
    for i in range(n):
        op, ot, opos, _ = O.intersect(o[i], d[i])
        if op != prim[i]:
            # a different triangle is acceptable only for an exact tie in t (shared edge): count and bound
            assert op >= 0 and prim[i] >= 0 and abs(ot - t[i]) <= 1e-12 * abs(ot)
            mism += 1
        elif op >= 0:
            assert abs(ot - t[i]) <= 1e-13 * abs(ot) and np.allclose(opos, p[i], rtol=0, atol=1e-9)
    assert mism <= 3
    assert (prim >= 0).mean() > 0.5


@pytest.mark.parametrize("variant,md", [("diffuse", -1), ("diffuse", 3), ("glossy", 10), ("nearspecular", 10), ("twosided", 9), ("glass", 12)])
def test_samples_match_oracle(G, variant, md):
    sc = scenes.cornell_box(40, 32, variant)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md)
    rng = np.random.default_rng(11)
    for _ in range(40):
        px, py, s = int(rng.integers(0, 40)), int(rng.integers(0, 32)), int(rng.integers(0, 64))
        g = S.evaluate_point(integ.config(64), px, py, s)
        o = O.evaluate_point(go.config(maxDepth=md, spp=64), px, py, s)
        for k in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[k], o[k], rtol=1e-10, atol=1e-14), (variant, px, py, s, k)


@pytest.mark.parametrize("variant,md,strict", [("diffuse", -1, False), ("diffuse", 4, True), ("glossy", 9, False), ("glass", 10, False), ("twosided", 8, True)])
def test_environment_emitter_samples_and_film_match_oracle(G, variant, md, strict):
    """`<emitter type="constant">`: environment hits of base and offset paths, environmentShift, the environment in light
    sampling (gpt.cpp:96-114,348-369,786-804,1052-1074; constant.cpp).  The wide film sees the environment past the box."""
    W, H, spp = 44, 30, 5
    sc = scenes.cornell_box(W, H, variant, environment=(0.6, 0.8, 1.1))
    S = G.Scene(sc); O = go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp, strictNormals=strict)
    rng = np.random.default_rng(5)
    hit_env = 0
    for _ in range(120):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g = S.evaluate_point(cfg, px, py, s); o = O.evaluate_point(ocfg, px, py, s)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[key], o[key], rtol=1e-9, atol=1e-13), (variant, px, py, s, key, g[key], o[key])
        hit_env += int(np.allclose(o["veryDirect"], (0.6, 0.8, 1.1)))
    assert hit_env > 0                                   # some primaries do leave the scene
    F = G.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = O.render(ocfg)
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
    for b in range(5):
        assert close(acc[b], oacc[b]), (variant, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    F.close(); S.close(); O.close()


@pytest.mark.parametrize("variant,md,strict,env", [("smooth", -1, False, None), ("smooth", 6, True, (0.3, 0.4, 0.6)), ("bent", 8, False, None), ("bent", 7, True, None)])
def test_vertex_normals_samples_and_film_match_oracle(G, variant, md, strict, env):
    """Per-vertex normals (fillIntersectionRecord, skdtree.h:382-397): interpolated shading normal, geometric normal flipped to
    its side, frame from dpdu.  "bent" tilts the normals up to 0.6 rad so shading and geometric normals disagree and the
    strictNormals tests bite."""
    W, H, spp = 44, 30, 4
    sc = scenes.cornell_box(W, H, variant, environment=env)
    S = G.Scene(sc); O = go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp, strictNormals=strict)
    rng = np.random.default_rng(9)
    for _ in range(150):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g = S.evaluate_point(cfg, px, py, s); o = O.evaluate_point(ocfg, px, py, s)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[key], o[key], rtol=1e-9, atol=1e-13), (variant, px, py, s, key, g[key], o[key])
    F = G.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = O.render(ocfg)
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
    for b in range(5):
        assert close(acc[b], oacc[b]), (variant, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    # the normals matter: the flat rendering of the same geometry differs
    flat = scenes.cornell_box(W, H, variant, environment=env); flat.normals = None
    assert not np.allclose(go.Scene(flat).render(ocfg)[0][1], oacc[1])
    F.close(); S.close(); O.close()


@pytest.mark.parametrize("kind", [1, 2, 3, 4, 5])
def test_reconstruction_filters_match_oracle(G, kind):
    """tent / gaussian / mitchell / catmullrom / lanczos (src/rfilters, discretised by ReconstructionFilter::configure): every
    put spreads over its footprint through the exact generic path; the five buffers against the oracle's ImageBlock::put."""
    from gradientdomain_mitsuba_amd._lib import GdptError
    W, H, spp = 36, 26, 3
    sc = scenes.cornell_box(W, H, "glossy")
    sc.rfilter = scenes.RFILTER_DEFAULTS[kind]
    S = G.Scene(sc); F = G.Film(S)
    integ = G.GradientPathIntegrator(maxDepth=5)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = go.Scene(sc).render(go.config(maxDepth=5, spp=spp))
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
    for b in range(5):
        scale = np.abs(oacc[b]).max()
        assert np.abs(acc[b] - oacc[b]).max() <= 1e-10 * scale, (kind, G.BUFFER_NAMES[b])     # atomics: order of the fp64 sums is free
    box = scenes.cornell_box(W, H, "glossy")
    assert not np.allclose(go.Scene(box).render(go.config(maxDepth=5, spp=spp))[0][1], oacc[1])
    # strips: each renders the rows within the filter's reach itself, so three strips side by side ARE the whole film, bit for bit,
    # with no exchange (samples depend on seed, pixel and sample index only)
    parts = []
    for y0, y1 in ((0, 7), (7, 9), (9, H)):
        Fs = G.Film(S, y0, y1)
        integ.renderBlock(S, Fs, integ.config(spp), (0, y0, W, y1))
        parts.append(Fs.accum())
        Fs.close()
    assert np.array_equal(np.concatenate(parts, axis=1), acc)
    # blocks (round 5): a sub-rectangle renders the samples of ITS pixels and puts them within the filter's reach around it, as GPTBlockRenderer's unit
    # does into GPTWorkResult's bordered blocks (gpt_proc.cpp:52-56,74-91; gpt_wr.cpp:31-44) -- four ragged tiles into one film add up to the one-call
    # film (the sums of a pixel arrive tile by tile: their order, hence the last bits, differ), same rays; so do tiles rendered into films of their
    # own, copied back with their border (gdpt_film_accum_rect) and merged by addition as MultiFilm::putMulti merges them (gpt_proc.cpp:137-149)
    Ft = G.Film(S)
    tiles = [(0, 0, 17, 11), (17, 0, W, 11), (0, 11, 17, H), (17, 11, W, H)]
    for t in tiles:
        integ.renderBlock(S, Ft, integ.config(spp), t)
    acct = Ft.accum(); stt = Ft.stats()
    assert stt == st
    for b in range(5):
        assert np.abs(acct[b] - acc[b]).max() <= 1e-12 * np.abs(acc[b]).max(), (kind, "tiles", G.BUFFER_NAMES[b])
    reach = 4                                                 # ceil(radius) + 1 of the widest of the five (lanczos, 3 lobes); the others reach less
    merged = np.zeros_like(acc)
    for (x0, y0, x1, y1) in tiles:
        fy0, fy1 = max(0, y0 - reach), min(H, y1 + reach)
        Fb = G.Film(S, fy0, fy1)
        integ.renderBlock(S, Fb, integ.config(spp), (x0, y0, x1, y1))
        bx0, bx1 = max(0, x0 - reach), min(W, x1 + reach)
        part = Fb.accum_rect(bx0, fy0, bx1, fy1)
        assert np.array_equal(part, Fb.accum()[:, :, bx0:bx1])
        outside = Fb.accum().copy(); outside[:, :, bx0:bx1] = 0
        assert not outside.any()                              # nothing lands beyond the reach
        merged[:, fy0:fy1, bx0:bx1] += part
        Fb.close()
    for b in range(5):
        assert np.abs(merged[b] - acc[b]).max() <= 1e-12 * np.abs(acc[b]).max(), (kind, "blocks", G.BUFFER_NAMES[b])
    Ft.close(); F.close(); S.close()


def test_filter_strips_over_several_log_chunks(G):
    W, H, spp = 40, 30, 19                                  # 19 spp = two chunks of the sample log (16 + 3)
    sc = scenes.cornell_box(W, H, "diffuse")
    sc.rfilter = scenes.RFILTER_DEFAULTS[scenes.RFILTER_GAUSSIAN]
    S = G.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=4)
    F = G.Film(S)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    whole = F.accum()
    parts = []
    for y0, y1 in ((0, 16), (16, H)):
        Fs = G.Film(S, y0, y1)
        integ.renderBlock(S, Fs, integ.config(spp), (0, y0, W, y1))
        parts.append(Fs.accum()); Fs.close()
    assert np.array_equal(np.concatenate(parts, axis=1), whole)
    F.close(); S.close()


def test_vertex_normals_on_emitters_are_refused(G):
    from gradientdomain_mitsuba_amd._lib import GdptError
    sc = scenes.cornell_box(32, 24, "smooth")
    sc.normals[sc.emitters[0][0]] = [0, -1, 0] * 3
    with pytest.raises(GdptError, match="emitter"):
        G.Scene(sc)
    with pytest.raises(ValueError):
        go.Scene(sc)


@pytest.mark.parametrize("variant,md,strict,first,only", [("diffuse", 6, False, False, False), ("glossy", 8, False, True, False), ("glass", 9, True, False, False), ("diffuse", 5, False, False, True)])
def test_point_emitter_samples_and_film_match_oracle(G, variant, md, strict, first, only):
    """`<emitter type="point">` (point.cpp): EDiscrete light samples, no BSDF-sampling pdf against them (not EOnSurface), and the
    mainAtPointLight branch of the unconnected light-sample shift (gpt.cpp:667-672) also at glossy vertices."""
    W, H, spp = 40, 28, 4
    sc = scenes.cornell_box(W, H, variant, point_light=((278, 420, 279.5), (3e5, 2.5e5, 2e5), first))
    if only:
        sc.emitters = [e for e in sc.emitters if e[0] == "point"]
    S = G.Scene(sc); O = go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp, strictNormals=strict)
    rng = np.random.default_rng(3)
    for _ in range(120):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g = S.evaluate_point(cfg, px, py, s); o = O.evaluate_point(ocfg, px, py, s)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[key], o[key], rtol=1e-9, atol=1e-13), (variant, px, py, s, key, g[key], o[key])
    F = G.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = O.render(ocfg)
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
    for b in range(5):
        assert close(acc[b], oacc[b]), (variant, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    assert np.abs(oacc[1][..., :3]).max() > 0
    F.close(); S.close(); O.close()


def test_environment_only_scene_and_emitter_order(G):
    """No area light at all (the environment is the only emitter), and the environment first in the emitter list."""
    W, H, spp = 36, 24, 4
    sc = scenes.cornell_box(W, H, "diffuse", environment=(1.0, 0.9, 0.7))
    sc_first = scenes.cornell_box(W, H, "diffuse", environment=(1.0, 0.9, 0.7)); sc_first.environment = (sc_first.environment[0], 0)
    only = scenes.cornell_box(W, H, "diffuse", environment=(1.0, 0.9, 0.7)); only.emitters = []
    for sc_ in (sc_first, only):
        S = G.Scene(sc_); O = go.Scene(sc_); F = G.Film(S)
        integ = G.GradientPathIntegrator(maxDepth=6)
        integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
        oacc, orays = O.render(go.config(maxDepth=6, spp=spp))
        st = F.stats(); acc = F.accum()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b])
        F.close(); S.close(); O.close()
    a = go.Scene(sc).render(go.config(maxDepth=6, spp=spp))[0]
    b = go.Scene(sc_first).render(go.config(maxDepth=6, spp=spp))[0]
    assert not np.allclose(a[1], b[1])                   # the order decides which light samples pick the environment


@pytest.mark.parametrize("seed", range(24))
def test_fuzzed_materials_and_settings_match_oracle(G, seed):
    """Random materials (all carried BSDFs, both distributions, anisotropic and near-specular roughness, visible-normal
    sampling on/off, two-sided wrappers, dielectrics of random IOR) on the Cornell geometry, random integrator settings,
    random (pixel, sample) probes: every output of evaluatePoint and both ray counters against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    W, H = 40, 32
    sc = scenes.cornell_box(W, H, "random", seed=seed, environment=(0.5, 0.7, 0.9) if seed % 3 == 0 else None)
    md = int(rng.choice([-1, 2, 3, 5, 9]))
    rr = int(rng.choice([1, 3, 5]))
    strict = bool(rng.random() < 0.35)
    thr = float(rng.choice([0.001, 0.02, 0.0]))
    S = G.Scene(sc); O = go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, rrDepth=rr, strictNormals=strict, shiftThreshold=thr)
    cfg = integ.config(8)
    ocfg = go.config(maxDepth=md, rrDepth=rr, strictNormals=strict, spp=8, shiftThreshold=thr)
    for _ in range(60):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 8))
        g = S.evaluate_point(cfg, px, py, s); o = O.evaluate_point(ocfg, px, py, s)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[key], o[key], rtol=1e-9, atol=1e-13), (seed, px, py, s, key, g[key], o[key])
    S.close(); O.close()


@pytest.mark.parametrize("variant,W,H,spp,md,strict", [("diffuse", 48, 40, 6, -1, False), ("glossy", 40, 40, 6, 9, False),
                                                        ("nearspecular", 32, 32, 5, 8, True), ("diffuse", 35, 21, 3, 2, False),
                                                        ("twosided", 40, 36, 6, 9, False), ("twosided", 24, 24, 4, 6, True),
                                                        ("glass", 40, 40, 8, 12, False), ("glass", 24, 24, 4, -1, True)])
def test_film_matches_oracle(G, variant, W, H, spp, md, strict):
    sc = scenes.cornell_box(W, H, variant)
    S = G.Scene(sc); F = G.Film(S)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = go.Scene(sc).render(go.config(maxDepth=md, spp=spp, strictNormals=strict))
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays           # the same rays, one for one
    assert st["paths"] == W * H * spp
    for b in range(5):
        assert close(acc[b], oacc[b]), (G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())


@pytest.mark.parametrize("seed", range(100, 106))
def test_fuzzed_films_match_oracle(G, seed):
    """Whole small films of fuzzed scenes: the five buffers and both ray counters, with odd sizes, odd spp, forced slices and
    regeneration thresholds (none of which may change a result beyond the association of the per-pixel sums)."""
    rng = np.random.default_rng(seed)
    W, H, spp = int(rng.integers(17, 40)), int(rng.integers(9, 30)), int(rng.integers(1, 7))
    sc = scenes.cornell_box(W, H, "random", seed=seed, environment=(0.4, 0.5, 0.6) if seed % 2 == 0 else None)
    md, strict = int(rng.choice([-1, 3, 7])), bool(rng.random() < 0.3)
    S = G.Scene(sc); F = G.Film(S)
    F.set_slices(int(rng.integers(0, spp + 1))); F.set_regeneration(int(rng.choice([1, 24, 56, 64])))
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = go.Scene(sc).render(go.config(maxDepth=md, spp=spp, strictNormals=strict))
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays and st["paths"] == W * H * spp
    for b in range(5):
        assert close(acc[b], oacc[b]), (seed, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    F.close(); S.close()


@pytest.mark.parametrize("name,W,H,spp,md", [("cornell", 1280, 720, 2, -1), ("atrium", 1920, 1080, 1, 9)])
def test_whole_frame_at_configuration_size_matches_oracle(G, name, W, H, spp, md):
    """WHOLE frames at BASELINE's resolutions (config 2: the Cornell box at 1280x720; config 3: the atrium at 1920x1080, the scene
    of the config with its 113 k triangles) against the oracle: every pixel of the five buffers and both ray counters.  Few samples
    per pixel -- spp only lengthens the per-pixel loop (the 64 spp frame is held to the oracle by sample spot checks and by the
    size-independent properties of test_full_size_properties_1280x720x64) -- but every tile, every wave position, the sample queue
    with hundreds of thousands of slots and the continuation kernel's refills are in it."""
    sc = scenes.cornell_box(W, H, "diffuse") if name == "cornell" else scenes.atrium(W, H)
    S = G.Scene(sc); F = G.Film(S)
    integ = G.GradientPathIntegrator(maxDepth=md)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = go.Scene(sc).render(go.config(maxDepth=md, spp=spp))
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays and st["paths"] == W * H * spp
    for b in range(5):
        assert close(acc[b], oacc[b]), (name, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    F.close(); S.close()


def test_config1_cornell_512x512_frame_samples_and_l2_reconstruction(G):
    """BASELINE config 1 (Cornell box, 512x512, 64 spp, L2 reconstruct): the whole frame against the oracle at 2 spp (every pixel of the
    five buffers, both ray counters); the 64 spp frame through the integrator (render + develop + L2D) with its size-independent
    properties, single samples spot-checked against the oracle at this geometry, and the L2D reconstruction of the HIP path's own
    developed images against the oracle's solver."""
    W = H = 512
    sc = scenes.cornell_box(W, H, "diffuse")
    S, O = G.Scene(sc), go.Scene(sc)
    F = G.Film(S)
    integ = G.GradientPathIntegrator(maxDepth=-1, reconstructL1=False, reconstructL2=True)
    integ.renderBlock(S, F, integ.config(2), (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = O.render(go.config(maxDepth=-1, spp=2))
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays and st["paths"] == W * H * 2
    for b in range(5):
        assert close(acc[b], oacc[b]), (G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    F.close()
    spp = 64
    out = integ.render(S, spp)
    assert integ.stats["paths"] == W * H * spp and all(np.isfinite(v).all() for v in out.values())
    assert (out["-direct"] >= 0).all() and (out["-throughput"] >= -1e-6).all()
    cfg, ocfg = integ.config(spp), go.config(maxDepth=-1, spp=spp)
    rng = np.random.default_rng(21)
    for _ in range(25):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = S.evaluate_point(cfg, px, py, s), O.evaluate_point(ocfg, px, py, s)
        for k in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[k], o[k], rtol=1e-10, atol=1e-14), (px, py, s, k)
    ref = po.solve(po.preset("L2D"), out["-dx"].ravel(), out["-dy"].ravel(), out["-throughput"].ravel(), out["-direct"].ravel(), W, H).reshape(H, W, 3)
    assert np.abs(out["-final"] - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    S.close(); O.close()


def test_atrium_hbm_bvh_film_matches_oracle(G):
    """Sponza-class stand-in (20k triangles): the BVH and triangle tables do not fit the LDS budget, so traversal reads
    node packets from HBM/L2 and the shading tables through the global path; diffuse + rough-conductor materials."""
    W, H, spp = 64, 36, 4
    sc = scenes.atrium(W, H, columns=12, segments=16)
    assert sc.ntri > 20000
    S = G.Scene(sc); F = G.Film(S)
    integ = G.GradientPathIntegrator(maxDepth=7)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    oacc, orays = go.Scene(sc).render(go.config(maxDepth=7, spp=spp))
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
    for b in range(5):
        assert close(acc[b], oacc[b]), (G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())


def test_large_film_including_filter_edge_samples(G):
    """590k samples: ~24 of them fall within 1e-5 of a pixel edge, where the box filter's footprint is two pixels wide
    (box.cpp:38, imageblock.h:172-176) and the HIP path switches from per-pixel sums to exact atomic puts."""
    W = H = 192
    sc = scenes.cornell_box(W, H, "diffuse")
    S = G.Scene(sc); F = G.Film(S)
    integ = G.GradientPathIntegrator(maxDepth=4)
    integ.renderBlock(S, F, integ.config(16), (0, 0, W, H))
    acc = F.accum()
    oacc, _ = go.Scene(sc).render(go.config(maxDepth=4, spp=16))
    for b in range(5):
        assert close(acc[b], oacc[b]), G.BUFFER_NAMES[b]
    c2 = (1 / (2 * (0.5 + float(np.float32(1e-5))))) ** 2
    w = acc[1][1:-1, 1:-1, 3]
    assert (np.abs(w - 8 * 16 * c2) > 1e-9).sum() > 0          # some pixel did receive an edge sample's double footprint
    assert np.allclose(w, oacc[1][1:-1, 1:-1, 3], rtol=1e-12)


def test_two_strips_with_halo_exchange_equal_one_film(G):
    import torch
    W, H, spp = 56, 40, 5
    sc = scenes.cornell_box(W, H, "glossy")
    S = G.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=7)
    cfg = integ.config(spp)
    F = G.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    whole = F.accum()
    top, bot = G.Film(S, 0, 17), G.Film(S, 17, H)
    integ.renderBlock(S, top, cfg, (0, 0, W, 17))
    integ.renderBlock(S, bot, cfg, (0, 17, W, H))
    n = top.halo_bytes() // 8
    a, b = torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
    top.pack_halo(1, a); bot.pack_halo(0, b)           # each strip's boundary payload for the other
    top.unpack_halo(1, b); bot.unpack_halo(0, a)
    both = np.concatenate([top.accum(), bot.accum()], axis=1)
    for k in range(5):
        assert close(both[k], whole[k], 1e-12), G.BUFFER_NAMES[k]


def test_integrator_end_to_end_matches_oracle_pipeline(G):
    W, H, spp = 64, 48, 8
    sc = scenes.cornell_box(W, H, "diffuse")
    S = G.Scene(sc)
    for kw, preset in ((dict(reconstructL1=True), "L1D"), (dict(reconstructL1=False, reconstructL2=True), "L2D")):
        integ = G.GradientPathIntegrator(maxDepth=6, **kw)
        out = integ.render(S, spp)
        oacc, _ = go.Scene(sc).render(go.config(maxDepth=6, spp=spp))
        img = go.develop(oacc).astype(np.float32)             # gpt.cpp:1439-1442 casts to float
        for i, name in enumerate(G.BUFFER_NAMES[1:], 1):
            assert np.abs(out[name] - img[i]).max() <= 1e-6 * max(1.0, np.abs(img[i]).max()), name
        ref = po.solve(po.preset(preset), img[2].ravel(), img[3].ravel(), img[1].ravel(), img[4].ravel(), W, H).reshape(H, W, 3)
        tol = 2e-4 if preset == "L1D" else 5e-5
        assert np.abs(out["-final"] - ref).max() <= tol * max(1.0, np.abs(ref).max()), preset
        assert integ.stats["paths"] == W * H * spp


def test_gradient_domain_reconstruction_beats_the_primal_image(G):
    """What G-PT is for (Kettunen et al. 2015, the reference's method): at equal sample count the screened-Poisson
    reconstruction from the sampled gradients is closer to the converged image than the primal (-throughput + -direct) image.
    Reference image: the primal estimate at 64x the samples with another seed."""
    W, H, spp = 160, 120, 8
    S = G.Scene(scenes.cornell_box(W, H, "diffuse"))
    ref_run = G.GradientPathIntegrator(maxDepth=8, reconstructL1=False, reconstructL2=False).render(S, 64 * spp, seed=99)
    ref = ref_run["-throughput"] + ref_run["-direct"]
    err = {}
    for name, kw in (("L1", dict(reconstructL1=True)), ("L2", dict(reconstructL1=False, reconstructL2=True))):
        out = G.GradientPathIntegrator(maxDepth=8, **kw).render(S, spp)
        primal = out["-throughput"] + out["-direct"]
        rel = lambda img: float(np.mean((img - ref) ** 2 / (ref ** 2 + 1e-2)))
        err[name] = (rel(primal), rel(out["-final"]))
        assert np.isfinite(out["-final"]).all()
    # both reconstructions cut the relative MSE of the primal image substantially (measured: L1 0.22x, L2 0.25x; tools/gpu_gd_benefit.py)
    assert err["L1"][1] < 0.6 * err["L1"][0] and err["L2"][1] < 0.7 * err["L2"][0], err
    S.close()


def test_integrator_property_errors(G):
    with pytest.raises(RuntimeError, match="two reconstructions"):
        G.GradientPathIntegrator(reconstructL1=True, reconstructL2=True)
    with pytest.raises(RuntimeError, match="reconstructAlpha"):
        G.GradientPathIntegrator(reconstructAlpha=0.0)
    with pytest.raises(RuntimeError, match="maxDepth"):
        G.GradientPathIntegrator(maxDepth=0)
    with pytest.raises(RuntimeError, match="hideEmitters"):
        G.GradientPathIntegrator(hideEmitters=True)
    assert G.GradientPathIntegrator(minDepth=7).minDepth == 1          # gpt.cpp:1369
    from gradientdomain_mitsuba_amd._lib import GdptError
    sc = scenes.cornell_box(16, 16)
    bad = scenes.Scene(sc.verts, sc.tri_material, [dict(type=7)] * len(sc.materials), sc.emitters, sc.to_world, sc.fov_x, width=16, height=16)
    with pytest.raises(GdptError, match="not carried"):
        G.Scene(bad)
    S = G.Scene(sc); F = G.Film(S)
    with pytest.raises(GdptError):
        G.GradientPathIntegrator().renderBlock(S, F, G.GradientPathIntegrator().config(1), (0, 0, 17, 16))


def test_full_size_properties_1280x720x64(G):
    """BASELINE config 2 geometry and sample count: size-independent properties + spot checks against the oracle."""
    W, H, spp = 1280, 720, 64
    sc = scenes.cornell_box(W, H, "diffuse")
    S = G.Scene(sc); F = G.Film(S)
    integ = G.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(spp)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    assert st["paths"] == W * H * spp and st["raysTraced"] >= 5 * st["paths"] and np.isfinite(acc).all()
    c2 = (1 / (2 * (0.5 + float(np.float32(1e-5))))) ** 2
    inner = acc[1][1:-1, 1:-1, 3]
    assert np.median(inner) == pytest.approx(8 * spp * c2, rel=1e-12) and (np.abs(inner - 8 * spp * c2) < 10 * c2).all()
    assert (acc[4][..., :3] >= 0).all() and (acc[1][..., :3] >= -1e-12).all()
    # tiles rendered in pieces into one film == the one-call film: same samples, same ray counts; a pixel's samples are summed per
    # chunk of the sample queue (gdpt_film_set_pipeline) and the chunk length follows the launch's pixel count, so the fp64 sums of
    # differently shaped launches agree to rounding -- and bit for bit once the chunking is the same (everything in one kernel)
    F2 = G.Film(S); F2.set_slices(1)
    for (x0, y0, x1, y1) in ((0, 0, 640, 360), (640, 0, 1280, 360), (0, 360, 1280, 720)):
        integ.renderBlock(S, F2, cfg, (x0, y0, x1, y1))
    acc2 = F2.accum()
    assert np.allclose(acc2, acc, rtol=1e-13, atol=1e-13) and F2.stats() == st
    F2.close()
    Fa, Fb = G.Film(S), G.Film(S)
    for f in (Fa, Fb):
        f.set_pipeline(0); f.set_slices(1)
    integ.renderBlock(S, Fa, cfg, (0, 0, W, H))
    for (x0, y0, x1, y1) in ((0, 0, 640, 360), (640, 0, 1280, 360), (0, 360, 1280, 720)):
        integ.renderBlock(S, Fb, cfg, (x0, y0, x1, y1))
    a0, b0 = Fa.accum(), Fb.accum()
    assert (a0 == b0).mean() > 0.999 and np.allclose(a0, acc, rtol=1e-13, atol=1e-13) and Fa.stats() == st       # (the edge-sample atomics are the 0.1 %)
    Fa.close(); Fb.close()
    # the same with the launch's own choice of slices, and with a forced 5 (64 spp does not divide evenly): same samples, same
    # ray counts, sums equal to rounding; and reproducible bit for bit run to run
    for slices in (0, 5):
        runs = []
        for rep in range(2):
            F3 = G.Film(S); F3.set_slices(slices)
            for (x0, y0, x1, y1) in ((0, 0, 640, 360), (640, 0, 1280, 360), (0, 360, 1280, 720)):
                integ.renderBlock(S, F3, cfg, (x0, y0, x1, y1))
            runs.append((F3.accum(), F3.stats()))
            F3.close()
        assert runs[0][1] == st and np.allclose(runs[0][0], acc, rtol=1e-12, atol=1e-12)
        # (the rare filter-edge samples go through fp64 atomics whose order is free)
        assert (runs[0][0] == runs[1][0]).mean() > 0.999
    # spot checks of individual samples against the oracle at this geometry
    O = go.Scene(sc)
    rng = np.random.default_rng(2)
    for _ in range(25):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g = S.evaluate_point(cfg, px, py, s); o = O.evaluate_point(go.config(maxDepth=-1, spp=spp), px, py, s)
        assert np.allclose(g["throughput"], o["throughput"], rtol=1e-10, atol=1e-14) and np.allclose(g["gradients"], o["gradients"], rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("kind", [0, 2])
def test_strip_renderer_single_rank_equals_integrator_render(G, kind):
    """parallel.StripRenderer (the multi-GPU entry, here with one rank) == GradientPathIntegrator.render, box and gaussian film."""
    import torch
    from gradientdomain_mitsuba_amd import parallel
    W, H, spp = 48, 32, 4
    sc = scenes.cornell_box(W, H, "glossy")
    sc.rfilter = scenes.RFILTER_DEFAULTS[kind]
    S = G.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=6, reconstructL1=False, reconstructL2=True)
    ref = integ.render(S, spp)
    sr = parallel.StripRenderer(S, integ, 0, 1, torch.device("cuda", 0))
    out = sr.render(spp).cpu().numpy()
    assert sr.film.renders_own_border == (kind != 0) and sr.last["halo_bytes"] == 0
    assert np.array_equal(out, ref["-final"])
    imgs = sr.last["images"].cpu().numpy()
    for i, name in enumerate(("-throughput", "-dx", "-dy", "-direct")):
        assert np.array_equal(imgs[i], ref[name])
    assert sr.last["rays"] == integ.stats["raysTraced"] + integ.stats["shadowRaysTraced"]
    sr.close(); S.close()


def test_config3_geometry_properties_atrium_1920x1080(G):
    """BASELINE config 3's scene and resolution (HBM-resident BVH, 4-wave build) at 16 of its 256 spp: run-to-run reproducibility, two
    strips + halo exchange == one film, samples spot-checked against the oracle at this geometry."""
    import torch
    W, H, spp = 1920, 1080, 16
    sc = scenes.atrium(W, H)
    S = G.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(spp)
    F = G.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    F.clear(); integ.renderBlock(S, F, cfg, (0, 0, W, H))
    assert F.stats() == st and (F.accum() == acc).mean() > 0.999          # (filter-edge samples go through fp64 atomics)
    assert st["paths"] == W * H * spp and np.isfinite(acc).all()
    top, bot = G.Film(S, 0, 500), G.Film(S, 500, H)
    integ.renderBlock(S, top, cfg, (0, 0, W, 500)); integ.renderBlock(S, bot, cfg, (0, 500, W, H))
    n = top.halo_bytes() // 8
    a, b = torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
    top.pack_halo(1, a); bot.pack_halo(0, b)
    top.unpack_halo(1, b); bot.unpack_halo(0, a)
    both = np.concatenate([top.accum(), bot.accum()], axis=1)
    for k in range(5):
        assert close(both[k], acc[k], 1e-12), G.BUFFER_NAMES[k]
    ts, bs = top.stats(), bot.stats()
    assert ts["raysTraced"] + bs["raysTraced"] == st["raysTraced"] and ts["shadowRaysTraced"] + bs["shadowRaysTraced"] == st["shadowRaysTraced"]
    O = go.Scene(sc)
    rng = np.random.default_rng(5)
    ocfg = go.config(maxDepth=-1, spp=spp)
    for _ in range(12):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = S.evaluate_point(cfg, px, py, s), O.evaluate_point(ocfg, px, py, s)
        for k in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[k], o[k], rtol=1e-10, atol=1e-14), (px, py, s, k)
    for f in (F, top, bot):
        f.close()
    S.close(); O.close()


@pytest.mark.parametrize("heights", [[404, 56, 35, 793, 36, 127, 582, 127], [54, 123, 22, 67, 14, 137, 657, 1086], [1380, 136, 46, 20, 274, 150, 152, 2]])
def test_strips_with_arbitrary_boundaries_equal_one_film(G, heights):
    """Strips of arbitrary heights -- the partitions bench.py's rebalance step produces are timing-dependent -- against the one-film frame at config 4's size, 1 spp.
    The box filter's radius is 0.5 + 1e-5 (box.cpp:38): a sample within 1e-5 of a pixel edge lands in BOTH pixels (ImageBlock::put), so its neighbour puts reach
    two rows from its own -- the border of the reference's blocks is the filter's 1 + extraBorder 1 (gpt_wr.cpp:31-44).  Until round 5 a strip film kept (and
    shipped) one halo row of such puts: about one sample in 50 000 next to a boundary lost a put (found as a flaky 8-rank frame: tools/gpu_strips_random.py).  The
    first two partitions are ones that lost a put then; the third has a two-row strip."""
    import torch
    W, H, N = 3840, 2160, 8
    assert sum(heights) == H
    S = G.Scene(scenes.atrium(W, H))
    integ = G.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(1)
    F = G.Film(S); integ.renderBlock(S, F, cfg, (0, 0, W, H)); acc = F.accum(); st = F.stats(); F.close()
    b = np.concatenate([[0], np.cumsum(heights)]).tolist()
    strips = [(int(b[i]), int(b[i + 1])) for i in range(N)]
    films = [G.Film(S, y0, y1) for (y0, y1) in strips]
    for f, (y0, y1) in zip(films, strips):
        integ.renderBlock(S, f, cfg, (0, y0, W, y1))
    n = films[0].halo_bytes() // 8
    down = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(N)]; up = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(N)]
    for r, f in enumerate(films):                                                      # every rank packs before anyone unpacks (parallel.exchange_halos)
        if r + 1 < N: f.pack_halo(1, down[r])
        if r > 0: f.pack_halo(0, up[r])
    for r, f in enumerate(films):
        if r > 0: f.unpack_halo(0, down[r - 1])
        if r + 1 < N: f.unpack_halo(1, up[r + 1])
    rays = [0, 0]
    for f, (y0, y1) in zip(films, strips):
        a = f.accum(); s2 = f.stats(); rays[0] += s2["raysTraced"]; rays[1] += s2["shadowRaysTraced"]
        for k in range(5):
            assert close(a[k], acc[k][y0:y1], 1e-12), (G.BUFFER_NAMES[k], y0, y1)
        f.close()
    assert rays == [st["raysTraced"], st["shadowRaysTraced"]]
    S.close()


def test_config4_atrium_3840x2160_one_film_equals_eight_strips(G):
    """BASELINE config 4's tracer half (3840x2160 atrium, the 8-GPU headline), reduced spp: the frame rendered as ONE film equals the
    frame rendered as EIGHT strip films (the strips 8 GPUs would own, here one after the other on one device) whose one-pixel
    halos are packed / unpacked between neighbours exactly as parallel.exchange_halos ships them (gpt_proc.cpp:52-56,137-149: a
    block's border is merged into its neighbour by addition).  Equality to 1e-12 of each buffer's scale, identical ray sums;
    samples spot-checked against the oracle at this geometry."""
    import torch
    from gradientdomain_mitsuba_amd import parallel
    W, H, spp, N = 3840, 2160, 2, 8
    sc = scenes.atrium(W, H)
    S = G.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(spp)
    F = G.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    acc = F.accum(); st = F.stats()
    assert st["paths"] == W * H * spp and np.isfinite(acc).all() and F.invalid_puts() == 0
    F.close()
    strips = parallel.row_strips(H, N)
    films = [G.Film(S, y0, y1) for (y0, y1) in strips]
    for f, (y0, y1) in zip(films, strips):
        integ.renderBlock(S, f, cfg, (0, y0, W, y1))
    n = films[0].halo_bytes() // 8
    down = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(N)]      # payload of strip r for strip r+1
    up = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(N)]        # payload of strip r for strip r-1
    for r, f in enumerate(films):                                                      # every rank packs before anyone unpacks
        if r + 1 < N: f.pack_halo(1, down[r])
        if r > 0: f.pack_halo(0, up[r])
    for r, f in enumerate(films):
        if r > 0: f.unpack_halo(0, down[r - 1])
        if r + 1 < N: f.unpack_halo(1, up[r + 1])
    rays = [0, 0]
    y = 0
    for f, (y0, y1) in zip(films, strips):
        a = f.accum()
        for k in range(5):
            assert close(a[k], acc[k][y0:y1], 1e-12), (G.BUFFER_NAMES[k], y0, y1)
        s2 = f.stats(); rays[0] += s2["raysTraced"]; rays[1] += s2["shadowRaysTraced"]
        f.close()
    assert (rays[0], rays[1]) == (st["raysTraced"], st["shadowRaysTraced"])
    O = go.Scene(sc)
    rng = np.random.default_rng(9)
    ocfg = go.config(maxDepth=-1, spp=spp)
    for _ in range(10):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = S.evaluate_point(cfg, px, py, s), O.evaluate_point(ocfg, px, py, s)
        for k in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[k], o[k], rtol=1e-10, atol=1e-14), (px, py, s, k)
    S.close(); O.close()


def test_cancel_stops_a_running_frame(G):
    """Integrator::cancel: gdpt_render_rect is asynchronous; a cancel while it runs ends the frame early with whole samples only, and
    the film works again after clear()."""
    W, H, spp = 1280, 720, 256                       # ~0.7 s uncancelled
    sc = scenes.cornell_box(W, H, "diffuse")
    S = G.Scene(sc); F = G.Film(S)
    integ = G.GradientPathIntegrator(maxDepth=-1)
    import time
    integ.renderBlock(S, F, integ.config(1), (0, 0, W, H)); F.sync(); F.clear()      # (first launch loads the code object)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    assert not F.cancelled()
    time.sleep(0.05)
    F.cancel()
    F.sync()
    st = F.stats()
    assert F.cancelled() and 0 < st["paths"] < W * H * spp // 2 and F.render_ms() < 400.0
    acc = F.accum()
    assert np.isfinite(acc).all()
    c2 = (1 / (2 * (0.5 + float(np.float32(1e-5))))) ** 2
    assert acc[1][1:-1, 1:-1, 3].mean() < 0.5 * 8 * c2 * spp                  # (weight 8 c^2 per sample and pixel in the interior)
    integ.renderBlock(S, F, integ.config(4), (0, 0, W, H)); F.sync()
    assert F.stats() == st                                     # a cancelled frame takes no more samples
    F.clear()
    assert not F.cancelled()
    integ.renderBlock(S, F, integ.config(4), (0, 0, W, H)); F.sync()
    assert F.stats()["paths"] == W * H * 4
    # with a filter wider than box the interrupted chunk is dropped, earlier chunks stay
    sc.rfilter = scenes.RFILTER_DEFAULTS[scenes.RFILTER_GAUSSIAN]
    S2 = G.Scene(sc); F2 = G.Film(S2)
    integ.renderBlock(S2, F2, integ.config(128), (0, 0, W, H))
    F2.cancel(); F2.sync()
    assert np.isfinite(F2.accum()).all()
    for f in (F, F2):
        f.close()
    S.close(); S2.close()


@pytest.mark.parametrize("radiance", [(-17.0, 12.0, 4.0), (float("inf"), 12.0, 4.0), (float("nan"), 1.0, 1.0)])
@pytest.mark.parametrize("rfilter", [None, scenes.RFILTER_DEFAULTS[scenes.RFILTER_GAUSSIAN]])
def test_invalid_puts_are_dropped_like_imageblock_put(G, radiance, rfilter):
    """one NaN sample must not poison its pixel (and, through the CG's global dot products, the whole reconstruction).
    ImageBlock::put drops a put with a non-finite channel -- or a negative one outside dx / dy -- value and weight; the HIP film
    (per-pixel sums fast path, generic spill path, filter gather) must leave exactly what the oracle's 15 checked puts leave."""
    sc = scenes.cornell_box(40, 24, "diffuse")
    sc.emitters = [(sc.emitters[0][0], sc.emitters[0][1], radiance)]
    sc.rfilter = rfilter
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=5)
    spp = 3
    F = G.Film(S)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, 40, 24))
    acc = F.accum()
    oacc, orays = O.render(go.config(maxDepth=5, spp=spp))
    st = F.stats()
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
    assert all(np.isfinite(acc[b]).all() for b in range(5))
    assert F.invalid_puts() == O.invalid_puts() > 0
    for b in range(5):
        assert close(acc[b], oacc[b]), G.BUFFER_NAMES[b]
    out = integ.render(S, spp)                                   # the reconstruction stays finite
    assert all(np.isfinite(v).all() for v in out.values())


def _pipeline_cases():
    env = lambda: scenes.cornell_box(48, 40, "glossy", environment=(0.6, 0.7, 0.9))
    gauss = lambda: _with_rfilter(scenes.cornell_box(48, 40, "diffuse"), scenes.RFILTER_DEFAULTS[scenes.RFILTER_GAUSSIAN])
    return [("diffuse", lambda: scenes.cornell_box(48, 40, "diffuse"), dict(maxDepth=-1)),
            ("glossy", lambda: scenes.cornell_box(48, 40, "glossy"), dict(maxDepth=12)),
            ("glass", lambda: scenes.cornell_box(48, 40, "glass"), dict(maxDepth=14)),          # specular chains: offsets stay unconnected for many bounces
            ("nearspecular-strict", lambda: scenes.cornell_box(48, 40, "nearspecular"), dict(maxDepth=10, strictNormals=True)),
            ("bent-normals-strict", lambda: scenes.cornell_box(48, 40, "bent"), dict(maxDepth=9, strictNormals=True)),
            ("environment", env, dict(maxDepth=8)),
            ("gaussian-film", gauss, dict(maxDepth=6)),
            ("atrium", lambda: scenes.atrium(64, 36, columns=8, segments=12), dict(maxDepth=-1))]


def _with_rfilter(sc, rf):
    sc.rfilter = rf
    return sc


@pytest.mark.parametrize("name,builder,kw", _pipeline_cases(), ids=[c[0] for c in _pipeline_cases()])
def test_staged_pipeline_equals_the_single_kernel_and_the_oracle(G, name, builder, kw):
    """gdpt_film_set_pipeline: primary pass + general kernel + continuation kernel + per-chunk fold (the default) against everything in
    one kernel (the round-1 form, with its own builds of the render kernel) and against the oracle: identical ray counts and path statistics, films equal to rounding of the
    per-pixel sums; forced small queue chunks (several chunks per launch) and sample slices give the same; reproducible run to run."""
    import os
    sc = builder()
    W, H, spp = sc.width, sc.height, 5
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(**kw)
    cfg = integ.config(spp)
    out = {}
    for stages in (0, 2):
        F = G.Film(S); F.set_pipeline(stages)
        integ.renderBlock(S, F, cfg, (0, 0, W, H))
        out[stages] = (F.accum(), F.stats(), F.invalid_puts())
        F.close()
    oacc, orays = O.render(go.config(spp=spp, **kw))
    for stages in (0, 2):
        acc, st, inv = out[stages]
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays, stages
        assert st == out[0][1] and inv == out[0][2] == O.invalid_puts()
        for b in range(5):
            assert close(acc[b], oacc[b]), (stages, G.BUFFER_NAMES[b])
            assert close(acc[b], out[0][0][b], 1e-12), (stages, G.BUFFER_NAMES[b])
    # several queue chunks per launch (1 MB budget -> one sample per chunk), odd refill threshold, sample slices
    os.environ["GDPT_QUEUE_MB"] = "1"
    try:
        runs = []
        for rep in range(2):
            F = G.Film(S); F.set_pipeline(2, 7); F.set_slices(2)
            integ.renderBlock(S, F, cfg, (0, 0, W, H))
            runs.append((F.accum(), F.stats()))
            F.close()
    finally:
        del os.environ["GDPT_QUEUE_MB"]
    assert runs[0][1] == out[0][1] and (runs[0][0] == runs[1][0]).mean() > 0.999       # (filter-edge samples go through fp64 atomics)
    for b in range(5):
        assert close(runs[0][0][b], out[2][0][b], 1e-12), G.BUFFER_NAMES[b]
    S.close(); O.close()


@pytest.mark.parametrize("name,builder,kw", _pipeline_cases(), ids=[c[0] for c in _pipeline_cases()])
def test_wavefront_continuation_is_bit_identical_to_the_staged_pipeline(G, name, builder, kw):
    """gdpt_film_set_pipeline(3): the continuation phase in wavefront form (csrc/gpt_wavefront.hip.h: ray queues, traversal-only kernels,
    shading passes that replay bounce() with the traced results) against the staged pipeline: the replay runs the same source, so films,
    ray counts and path statistics must be IDENTICAL, bit for bit -- for 1 traced bounce, for the default, and for more bounces than any
    path has (k_continue left with nothing); through the LDS-scene and the HBM-scene builds; with several queue chunks per launch."""
    import os
    from gradientdomain_mitsuba_amd import _build
    from gradientdomain_mitsuba_amd._lib import GdptError
    sc = builder()
    W, H, spp = sc.width, sc.height, 5
    if not _build.WITH_WAVEFRONT:
        # round 5: the wavefront unit is a development build (GDPT_WITH_WAVEFRONT=1 python -c "import __graft_entry__ as g; g.build()"): measured slower than
        # the staged pipeline, so the product library and this suite do not pay for it; the product says so instead of running something else
        S = G.Scene(sc); F = G.Film(S)
        with pytest.raises(GdptError, match="development build"):
            F.set_pipeline(3)
        F.close(); S.close()
        pytest.skip("product build: pipeline 3 lives in the GDPT_WITH_WAVEFRONT=1 development build (tools/gpu_wf_check.py holds it bit-identical there)")
    integ = G.GradientPathIntegrator(**kw)
    cfg = integ.config(spp)
    for hbm in (False, True):
        if hbm:
            os.environ["GDPT_SCENE_IN_HBM"] = "1"
        try:
            S = G.Scene(sc)
            F = G.Film(S); F.set_pipeline(2)
            integ.renderBlock(S, F, cfg, (0, 0, W, H))
            ref = (F.accum(), F.stats(), F.invalid_puts())
            F.close()
            for iters, queue_mb in ((1, None), (6, None), (40, None), (3, "1")):
                os.environ["GDPT_WF_ITERS"] = str(iters)
                if queue_mb:
                    os.environ["GDPT_QUEUE_MB"] = queue_mb
                try:
                    F = G.Film(S); F.set_pipeline(3)
                    integ.renderBlock(S, F, cfg, (0, 0, W, H))
                    got = (F.accum(), F.stats(), F.invalid_puts())
                    F.close()
                finally:
                    os.environ.pop("GDPT_WF_ITERS", None); os.environ.pop("GDPT_QUEUE_MB", None)
                assert got[1] == ref[1] and got[2] == ref[2], (hbm, iters)
                if name == "gaussian-film" or queue_mb:      # (the log's gather / filter-edge samples of other chunkings: fp64 atomics in another order)
                    for b in range(5):
                        assert close(got[0][b], ref[0][b], 1e-12), (hbm, iters, G.BUFFER_NAMES[b])
                else:
                    for b in range(5):
                        assert np.array_equal(got[0][b], ref[0][b]), (hbm, iters, G.BUFFER_NAMES[b])
            S.close()
        finally:
            os.environ.pop("GDPT_SCENE_IN_HBM", None)


@pytest.mark.parametrize("flt,wrap", [(scenes.TEXFILTER_BILINEAR, scenes.TEXWRAP_REPEAT), (scenes.TEXFILTER_NEAREST, scenes.TEXWRAP_MIRROR), (scenes.TEXFILTER_BILINEAR, scenes.TEXWRAP_ZERO),
                                      (scenes.TEXFILTER_EWA, scenes.TEXWRAP_REPEAT), (scenes.TEXFILTER_TRILINEAR, scenes.TEXWRAP_CLAMP)])
def test_bitmap_textures_match_oracle(G, flt, wrap):
    """Texture coordinates in the hit record (skdtree.h:398-405) and `bitmap` textures on reflectance / specularReflectance (level-0
    nearest / bilinear lookups, wrap modes, uv scale and offset, energy-conservation scale): single samples and the film against the
    oracle, through both pipelines."""
    sc = scenes.textured_cornell_box(48, 36, filter=flt, wrap=wrap)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=7)
    cfg = integ.config(4)
    rng = np.random.default_rng(3)
    for _ in range(40):
        px, py, s = int(rng.integers(0, 48)), int(rng.integers(0, 36)), int(rng.integers(0, 4))
        g, o = S.evaluate_point(cfg, px, py, s), O.evaluate_point(go.config(maxDepth=7, spp=4), px, py, s)
        for k in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[k], o[k], rtol=1e-10, atol=1e-14), (px, py, s, k)
    oacc, orays = O.render(go.config(maxDepth=7, spp=4))
    for stages in (0, 2):
        F = G.Film(S); F.set_pipeline(stages)
        integ.renderBlock(S, F, cfg, (0, 0, 48, 36))
        acc, st = F.accum(), F.stats()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (stages, G.BUFFER_NAMES[b])
        F.close()
    untextured = scenes.textured_cornell_box(48, 36)
    untextured.textures, untextured.material_textures = None, None
    ref, _ = go.Scene(untextured).render(go.config(maxDepth=7, spp=4))
    assert not close(oacc[1], ref[1], 1e-3)                     # the textures do change the image
    S.close(); O.close()


@pytest.mark.parametrize("variant,textured", [("bent", False), ("smooth", True)])
def test_uv_tangents_orient_the_shading_frames(G, variant, textured):
    """A mesh with texture coordinates gets UV tangents (TriMesh::configure -> computeUVTangents, trimesh.cpp:362-386,683-735) and its
    shading frames follow the texture's u axis (its.dpdu, skdtree.h:373-380) -- with or without a texture that reads the coordinates.
    Spheres with interpolated normals + random per-triangle coordinates (some with a degenerate parameterization: the
    coordinateSystem branch), rough-conductor so that the frame's orientation decides the sampled directions; samples and film
    against the oracle, and the coordinates must matter."""
    W, H, spp, md = 40, 32, 4, 6
    sc = scenes.cornell_box(W, H, variant)
    nt = sc.ntri
    rng = np.random.default_rng(11)
    uvs = rng.uniform(-1.5, 2.5, (nt, 6)); has = (rng.random(nt) < 0.8).astype(np.uint8)
    degenerate = np.nonzero(has)[0][::7]
    uvs[degenerate, 2:4] = uvs[degenerate, 0:2] + 0.25; uvs[degenerate, 4:6] = uvs[degenerate, 0:2] + 0.5      # collinear coordinates: determinant 0
    sc.uvs, sc.tri_has_uv = uvs, has
    mats = list(sc.materials)
    rough = len(mats); mats.append(scenes.roughconductor(0.2, **scenes.CU)); sc.materials = mats
    tm = np.array(sc.tri_material, np.int32).copy(); tm[np.abs(np.asarray(sc.normals)).sum(1) > 0] = rough; sc.tri_material = tm   # the spheres
    if textured:
        sc.textures = [scenes.bitmap_texture(scenes.checker_rgb(8, 6, 5), wrap=scenes.TEXWRAP_MIRROR, filter=scenes.TEXFILTER_BILINEAR)]
        mt = [-1] * len(mats); mt[rough] = 0; sc.material_textures = mt
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=(variant == "bent"))
    cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp, strictNormals=(variant == "bent"))
    for _ in range(40):
        px, py, k = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = S.evaluate_point(cfg, px, py, k), O.evaluate_point(ocfg, px, py, k)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[key], o[key], rtol=1e-10, atol=1e-14), (px, py, k, key)
    oacc, orays = O.render(ocfg)
    for stages in (0, 2):
        F = G.Film(S); F.set_pipeline(stages)
        integ.renderBlock(S, F, cfg, (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (stages, G.BUFFER_NAMES[b])
        F.close()
    plain = scenes.cornell_box(W, H, variant); plain.materials, plain.tri_material = sc.materials, sc.tri_material
    ref, _ = go.Scene(plain).render(ocfg)
    assert not close(oacc[1], ref[1], 1e-6)                      # same geometry and materials without the coordinates: other frames, other samples
    S.close(); O.close()


@pytest.mark.parametrize("flt,wrap,uvscale,aniso,size", [(scenes.TEXFILTER_EWA, scenes.TEXWRAP_REPEAT, 9.0, 20.0, (37, 23)), (scenes.TEXFILTER_EWA, scenes.TEXWRAP_MIRROR, 25.0, 2.0, (64, 48)),
                                                          (scenes.TEXFILTER_TRILINEAR, scenes.TEXWRAP_ZERO, 14.0, 20.0, (33, 65)), (scenes.TEXFILTER_EWA, scenes.TEXWRAP_ONE, 60.0, 8.0, (16, 12))])
def test_mip_filtered_textures_match_oracle(G, flt, wrap, uvscale, aniso, size):
    """filterType trilinear / ewa (the reference's default): the MIP pyramid (Lanczos-resampled levels, non-power-of-two sizes, every
    wrap mode as boundary condition) and the filtered lookup at camera-ray hits -- UV partials from the ray differentials, ellipse,
    anisotropy clamp, level selection, EWA over two levels / trilinear -- with textures fine enough that levels above 0 and
    anisotropic footprints occur (the floor at a grazing angle).  Samples and the film against the oracle, both pipelines; and the
    filtering must matter against the same scene with bilinear lookups."""
    W, H, spp, md = 48, 36, 3, 5
    sc = scenes.textured_cornell_box(W, H, filter=flt, wrap=wrap, uvscale=uvscale, maxAnisotropy=aniso, size=size)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md)
    cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp)
    rng = np.random.default_rng(17)
    for _ in range(60):
        px, py, k = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = S.evaluate_point(cfg, px, py, k), O.evaluate_point(ocfg, px, py, k)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[key], o[key], rtol=1e-10, atol=1e-14), (px, py, k, key)
    oacc, orays = O.render(ocfg)
    for stages in (0, 2):
        F = G.Film(S); F.set_pipeline(stages)
        integ.renderBlock(S, F, cfg, (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (stages, G.BUFFER_NAMES[b])
        F.close()
    plain = scenes.textured_cornell_box(W, H, filter=scenes.TEXFILTER_BILINEAR, wrap=wrap, uvscale=uvscale, size=size)
    ref, _ = go.Scene(plain).render(ocfg)
    assert not close(oacc[1], ref[1], 1e-4)
    S.close(); O.close()


@pytest.mark.parametrize("variant,strict,rot,size,index", [("diffuse", False, 0.0, (32, 16), -1), ("glossy", False, 0.7, (37, 19), 0), ("smooth", True, 2.1, (64, 32), 1), ("glass", False, -1.3, (16, 9), -1)])
def test_environment_map_matches_oracle(G, variant, strict, rot, size, index):
    """`<emitter type="envmap">` (src/emitters/envmap.cpp) in place of the constant environment: latitude-longitude map in a
    half-precision MIP pyramid, EWA lookup through the camera ray's differentials where the background is seen directly, level-0
    bilinear for every other ray, importance sampling through the float cdf tables with the tent offset, its pdf, a rotated
    `toWorld`, non-power-of-two maps, the emitter's position in the list.  The box is opened (no ceiling light blocking: the front
    is open anyway) so that the map lights the scene; samples and the film against the oracle, both pipelines."""
    W, H, spp, md = 40, 32, 4, 6
    sc = scenes.cornell_box(W, H, variant)
    c, s_ = np.cos(rot), np.sin(rot)
    R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]) @ np.array([[1, 0, 0], [0, np.cos(0.3), -np.sin(0.3)], [0, np.sin(0.3), np.cos(0.3)]])
    sc.environment_map = dict(rgb=scenes.sky_map(size[0], size[1]), scale=0.8, toWorld=R, index=index)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp, strictNormals=strict)
    rng = np.random.default_rng(23)
    for _ in range(60):
        px, py, k = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = S.evaluate_point(cfg, px, py, k), O.evaluate_point(ocfg, px, py, k)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[key], o[key], rtol=1e-10, atol=1e-14), (px, py, k, key)
    oacc, orays = O.render(ocfg)
    for stages in (0, 2):
        F = G.Film(S); F.set_pipeline(stages)
        integ.renderBlock(S, F, cfg, (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (stages, G.BUFFER_NAMES[b])
        F.close()
    assert oacc[4][..., :3].max() > 0                          # the map is seen directly somewhere (very direct = the EWA lookup)
    flat = scenes.cornell_box(W, H, variant); flat.environment = ((0.4, 0.5, 0.6), len(flat.emitters))
    ref, _ = go.Scene(flat).render(ocfg)
    assert not close(oacc[1], ref[1], 1e-3)
    S.close(); O.close()


def test_environment_map_in_the_hbm_builds_and_bad_maps(G, monkeypatch):
    """The same through the 4-wave (HBM-scene) builds, where the lookups are inlined instead of called; black / oversized / non-finite maps
    are refused as the plugin refuses them."""
    monkeypatch.setenv("GDPT_SCENE_IN_HBM", "1")
    W, H, spp = 32, 24, 3
    sc = scenes.cornell_box(W, H, "glossy")
    sc.environment_map = dict(rgb=scenes.sky_map(24, 12), scale=1.5, index=-1)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=5)
    F = G.Film(S)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    acc, st = F.accum(), F.stats()
    oacc, orays = O.render(go.config(maxDepth=5, spp=spp))
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
    for b in range(5):
        assert close(acc[b], oacc[b]), G.BUFFER_NAMES[b]
    F.close(); S.close(); O.close()
    # the constant environment through the same builds, both pipelines, deep paths (the 4-wave builds once had an environment-only variant
    # that faulted from the second bounce on: DESIGN.md)
    flat = scenes.cornell_box(W, H, "glossy"); flat.environment = ((0.4, 0.5, 0.6), len(flat.emitters))
    S, O = G.Scene(flat), go.Scene(flat)
    oacc, orays = O.render(go.config(maxDepth=7, spp=spp))
    for stages in (0, 2):
        F = G.Film(S); F.set_pipeline(stages)
        if stages == 0: F.set_occupancy(4)
        G.GradientPathIntegrator(maxDepth=7).renderBlock(S, F, G.GradientPathIntegrator(maxDepth=7).config(spp), (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (stages, G.BUFFER_NAMES[b])
        F.close()
    S.close(); O.close()
    monkeypatch.delenv("GDPT_SCENE_IN_HBM")
    sc.environment_map = dict(rgb=np.zeros((8, 16, 3)))
    with pytest.raises(RuntimeError, match="completely black"):
        G.Scene(sc)
    bad = scenes.sky_map(16, 8); bad[2, 3, 1] = np.inf
    sc.environment_map = dict(rgb=bad)
    with pytest.raises(RuntimeError, match="invalid floating"):
        G.Scene(sc)
    sc.environment_map = dict(rgb=scenes.sky_map(16, 8)); sc.environment = ((1, 1, 1), 0)
    with pytest.raises(ValueError, match="one environment"):
        G.Scene(sc)


def _with_rectangle_light(sc, corner, eu, ev, radiance=(18.0, 14.0, 9.0)):
    """Appends a `rectangle` shape with an area emitter to a scene description: the two triangles of Rectangle::createTriMesh
    (rectangle.cpp:158-193) and the emitter as (firstTri, 2, radiance, toWorld 3x4, normal).  objectToWorld maps (-1,-1,0) to `corner`."""
    eu, ev, corner = np.asarray(eu, float), np.asarray(ev, float), np.asarray(corner, float)
    c = corner + eu + ev                                             # image of the origin; columns: eu, ev, n, c
    n = np.cross(eu, ev); n /= np.linalg.norm(n)
    M = np.stack([eu, ev, n, c], 1)                                  # 3x4
    P = lambda x, y: M @ np.array([x, y, 0.0, 1.0])
    v = [P(-1, -1), P(1, -1), P(1, 1), P(-1, 1)]
    tris = np.array([[v[0], v[1], v[2]], [v[2], v[3], v[0]]]).reshape(2, 9)
    first = sc.ntri
    sc.verts = np.concatenate([np.asarray(sc.verts, np.float64).reshape(-1, 9), tris])
    sc.tri_material = np.concatenate([np.asarray(sc.tri_material, np.int32), np.zeros(2, np.int32)])
    if getattr(sc, "normals", None) is not None:
        sc.normals = np.concatenate([np.asarray(sc.normals), np.zeros((2, 9))])
    sc.emitters = list(sc.emitters) + [(first, 2, tuple(radiance), M, tuple(n))]
    return sc


@pytest.mark.parametrize("variant,strict", [("diffuse", False), ("glossy", True)])
def test_rectangle_light_is_sampled_as_the_shape_samples_itself(G, variant, strict):
    """The light of a `rectangle` shape: hit as two triangles, SAMPLED as Rectangle::samplePosition does (objectToWorld(2u - 1, 2v - 1, 0),
    the frame's normal, pdf 1 / (|dpdu| |dpdv|): rectangle.cpp:100-121,200-206) -- not through a triangle cdf.  A second, tilted and
    non-square light next to the Cornell box's own mesh light; samples and the film against the oracle, both pipelines; and the
    same two triangles declared as a plain mesh light give other samples (same distribution, another map from random numbers to points)."""
    W, H, spp, md = 40, 32, 4, 6
    base = scenes.cornell_box(W, H, variant)
    sc = _with_rectangle_light(scenes.cornell_box(W, H, variant), (120.0, 300.0, 150.0), (45.0, 6.0, 0.0), (-3.0, 10.0, 60.0))
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp, strictNormals=strict)
    rng = np.random.default_rng(31)
    for _ in range(60):
        px, py, k = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = S.evaluate_point(cfg, px, py, k), O.evaluate_point(ocfg, px, py, k)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[key], o[key], rtol=1e-10, atol=1e-14), (px, py, k, key)
    oacc, orays = O.render(ocfg)
    for stages in (0, 2):
        F = G.Film(S); F.set_pipeline(stages)
        integ.renderBlock(S, F, cfg, (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (stages, G.BUFFER_NAMES[b])
        F.close()
    mesh = _with_rectangle_light(scenes.cornell_box(W, H, variant), (120.0, 300.0, 150.0), (45.0, 6.0, 0.0), (-3.0, 10.0, 60.0))
    mesh.emitters = mesh.emitters[:-1] + [mesh.emitters[-1][:3]]
    ref, _ = go.Scene(mesh).render(ocfg)
    assert not close(oacc[1], ref[1], 1e-6)
    assert abs(oacc[1][..., :3].mean() - ref[1][..., :3].mean()) < 0.1 * ref[1][..., :3].mean()        # (the same light in expectation)
    S.close(); O.close()


def test_texture_arguments_are_checked(G):
    sc = scenes.textured_cornell_box(16, 12)
    sc.textures[0]["filter"] = 4
    with pytest.raises(RuntimeError, match="Invalid filter type"):
        G.Scene(sc)
    sc = scenes.textured_cornell_box(16, 12, filter=scenes.TEXFILTER_EWA, maxAnisotropy=0.5)
    with pytest.raises(RuntimeError, match="maxAnisotropy"):
        G.Scene(sc)
    sc = scenes.textured_cornell_box(16, 12)
    sc.material_textures[0] = 7
    with pytest.raises(RuntimeError, match="out of range"):
        G.Scene(sc)


def _gate_scenes():
    """One scene per feature set the render kernels are built for (flat | + environment | + per-vertex normals), each with the settings the two
    unexplained -O3 findings of round 2 needed: strictNormals with near-specular / glossy lobes at maxDepth 4-8, a constant environment at maxDepth >= 3."""
    return [("flat-nearspecular-strict", lambda: scenes.cornell_box(40, 30, "nearspecular"), dict(maxDepth=6, strictNormals=True)),
            ("flat-glossy-strict", lambda: scenes.cornell_box(40, 30, "glossy"), dict(maxDepth=8, strictNormals=True)),
            ("environment-strict", lambda: scenes.cornell_box(48, 30, "glossy", environment=(0.7, 0.9, 1.2)), dict(maxDepth=5, strictNormals=True)),
            ("environment-deep", lambda: scenes.cornell_box(48, 30, "diffuse", environment=(0.7, 0.9, 1.2)), dict(maxDepth=4)),
            ("pervertex-bent-strict", lambda: scenes.cornell_box(40, 30, "bent"), dict(maxDepth=7, strictNormals=True))]


@pytest.mark.parametrize("name,builder,kw", _gate_scenes(), ids=[c[0] for c in _gate_scenes()])
@pytest.mark.parametrize("hbm", [False, True], ids=["lds-scene", "hbm-scene"])
def test_every_shipped_instantiation_agrees_with_the_general_kernel_and_the_oracle(G, monkeypatch, name, builder, kw, hbm):
    """The regression gate ADVICE r2 asked for: EVERY instantiation of the render kernels the dispatcher can select for a scene -- staged
    (k_render<STAGED> + k_continue) with the sums in LDS or in registers, the same with the GENERAL k_render in the staged pipeline
    (GDPT_DEV_GENERAL_KERNEL), the single-kernel form built for 2 and for 4 waves per SIMD -- for the LDS-resident build and (GDPT_SCENE_IN_HBM:
    small scenes through the HBM-scene builds, incl. the scratch-resident Lane of the 4-wave kernels) the HBM-resident one, against the oracle:
    identical ray counts, films to 1e-9.  A silent wrong value of the kind found in round 2 (one 16-byte unit of an offset's throughput in one
    instantiation, strictNormals + maxDepth >= 4 only) fails here."""
    if hbm:
        monkeypatch.setenv("GDPT_SCENE_IN_HBM", "1")
    sc = builder()
    W, H, spp = sc.width, sc.height, 4
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(**kw)
    cfg = integ.config(spp)
    oacc, orays = O.render(go.config(spp=spp, **kw))
    variants = [("staged", dict(pipeline=2, occ=2), {}), ("staged-register-sums", dict(pipeline=2, occ=-2), {}),
                ("staged-general-kernel", dict(pipeline=2, occ=2), {"GDPT_DEV_GENERAL_KERNEL": "1"}),
                ("single-2-waves", dict(pipeline=0, occ=2), {}), ("single-2-waves-register-sums", dict(pipeline=0, occ=-2), {}),
                ("single-4-waves", dict(pipeline=0, occ=4), {})]
    ref = None
    for vname, knobs, env in variants:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        F = G.Film(S)
        F.set_pipeline(knobs["pipeline"]); F.set_occupancy(knobs["occ"])
        integ.renderBlock(S, F, cfg, (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        F.close()
        for k in env:
            monkeypatch.delenv(k)
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays, (name, hbm, vname)
        for b in range(5):
            assert close(acc[b], oacc[b]), (name, hbm, vname, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
        if ref is None:
            ref = acc
        else:
            assert np.allclose(acc, ref, rtol=1e-12, atol=1e-12), (name, hbm, vname)
    S.close(); O.close()


@pytest.mark.parametrize("variant,md,strict,lens", [("diffuse", -1, False, (25.0, 700.0)), ("glossy", 9, True, (40.0, 900.0)), ("glass", 10, False, (8.0, 1100.0)),
                                                    ("bent", 6, True, (25.0, 820.0))])
def test_thinlens_sensor_samples_and_films_match_oracle(G, variant, md, strict, lens):
    """`<sensor type="thinlens">` (src/sensors/thinlens.cpp; the aperture sample of gpt.cpp:1262-1264, shared by the base ray and its four offsets):
    single samples and whole films against the oracle, ray for ray, through the staged pipeline and the single kernel, for the LDS-resident and the
    HBM-resident builds' feature sets (flat, per-vertex normals)."""
    W, H = 40, 32
    sc = scenes.cornell_box(W, H, variant); sc.thinlens = lens
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    rng = np.random.default_rng(29)
    for _ in range(30):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = S.evaluate_point(integ.config(64), px, py, s)
        o = O.evaluate_point(go.config(maxDepth=md, spp=64, strictNormals=strict), px, py, s)
        for k in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[k], o[k], rtol=1e-10, atol=1e-14), (variant, px, py, s, k)
        assert (g["raysTraced"], g["shadowRaysTraced"]) == (o["raysTraced"], o["shadowRaysTraced"])
    spp = 3
    oacc, orays = O.render(go.config(maxDepth=md, spp=spp, strictNormals=strict))
    for pipeline in (2, 0):
        F = G.Film(S); F.set_pipeline(pipeline)
        integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        F.close()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (pipeline, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    # a lens is not a pinhole: the film differs from the perspective render of the same scene
    pin = scenes.cornell_box(W, H, variant)
    pacc, _ = go.Scene(pin).render(go.config(maxDepth=md, spp=spp, strictNormals=strict))
    assert not close(oacc[1], pacc[1], rel=1e-3)


@pytest.mark.parametrize("variant,md,lens", [("diffuse", 7, None), ("glossy", -1, (30.0, 800.0))])
def test_shutter_interval_draws_the_time_sample(G, variant, md, lens):
    """`shutterOpen` / `shutterClose` on the sensor (Sensor::Sensor, sensor.cpp:26-38): an interval of positive length makes needsTimeSample() true and every
    sample draws its time sample after the film position and the aperture sample (gpt.cpp:1261-1267).  Transforms are static, so the time moves nothing --
    the draw shifts the rest of the sample's random stream, and the device must shift with the oracle: single samples and films ray for ray, in the staged
    pipeline and the single kernel; an interval of zero length (EDeltaTime) draws nothing; a negative one is refused with the reference's message."""
    W, H = 40, 32
    def build(shutter):
        sc = scenes.cornell_box(W, H, variant); sc.thinlens = lens; sc.shutter = shutter
        return sc
    S, O = G.Scene(build((0.0, 1.0 / 30))), go.Scene(build((0.0, 1.0 / 30)))
    S0, Sz = G.Scene(build(None)), G.Scene(build((0.25, 0.25)))
    integ = G.GradientPathIntegrator(maxDepth=md)
    rng = np.random.default_rng(41)
    moved = 0
    for _ in range(30):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = S.evaluate_point(integ.config(64), px, py, s)
        o = O.evaluate_point(go.config(maxDepth=md, spp=64), px, py, s)
        g0, gz = S0.evaluate_point(integ.config(64), px, py, s), Sz.evaluate_point(integ.config(64), px, py, s)
        for k in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[k], o[k], rtol=1e-10, atol=1e-14), (variant, px, py, s, k)
            assert np.array_equal(g0[k], gz[k]), (variant, px, py, s, k)
        assert (g["raysTraced"], g["shadowRaysTraced"]) == (o["raysTraced"], o["shadowRaysTraced"])
        moved += not np.allclose(g["throughput"], g0["throughput"])
    assert moved > 15                                   # the stream behind the draw is another one
    spp = 3
    oacc, orays = O.render(go.config(maxDepth=md, spp=spp))
    for pipeline in (2, 0):
        F = G.Film(S); F.set_pipeline(pipeline)
        integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        F.close()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (pipeline, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    with pytest.raises(RuntimeError, match="Shutter opening time"):
        G.Scene(build((1.0, 0.5)))
    for x in (S, S0, Sz, O): x.close()


@pytest.mark.parametrize("variant,W,H,spp,md,bs,seed,lens,shutter", [("diffuse", 48, 40, 2, 6, 32, 5489, None, None), ("glossy", 40, 33, 2, -1, 16, 5489, (30.0, 800.0), (0.0, 0.1)),
                                                                   ("glass", 36, 36, 3, 9, 32, 77, None, None), ("bent", 70, 20, 1, 5, 32, 5489, None, None)])
def test_serial_render_consumes_the_reference_stream_in_the_reference_order(G, variant, W, H, spp, md, bs, seed, lens, shutter):
    """gdpt_render_serial: the HIP sampler fed by ONE SFMT-19937 stream (the product's generator: pinned to the reference test's own table in
    tests/test_serial_host.py) in the order a one-worker run of the reference visits the film -- SURVEY 8a rows 19 (IndependentSampler + SFMT) and 30 (spiral
    blocks + Hilbert order) on the device.  Every draw of every sample has to sit where the oracle's render_serial has it, or everything behind it is
    another image: films to fp64 rounding of the sums (the device adds per pixel record, the oracle per put), ray counts exactly."""
    sc = scenes.cornell_box(W, H, variant); sc.thinlens = lens; sc.shutter = shutter
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md)
    F = G.Film(S)
    draws = integ.renderSerial(S, F, integ.config(spp), blockSize=bs, parentSeed=seed)
    acc, st = F.accum(), F.stats()
    oacc, orays = O.render_serial(go.config(maxDepth=md, spp=spp), block_size=bs, parent_seed=seed)
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
    for b in range(5):
        assert close(acc[b], oacc[b]), (G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    assert draws >= (2 + (2 if lens else 0) + (1 if shutter else 0)) * W * H * spp
    # not the per-sample streams' film, and another parent seed is another film
    F.clear()
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H)); F.sync()
    assert not close(F.accum()[1], oacc[1], rel=1e-3)
    F.clear()
    integ.renderSerial(S, F, integ.config(spp), blockSize=bs, parentSeed=seed + 1)
    assert not close(F.accum()[1], oacc[1], rel=1e-3)
    F.close()
    # scope: whole-image films with the box filter
    F2 = G.Film(S, 0, H // 2)
    with pytest.raises(RuntimeError, match="whole image"):
        integ.renderSerial(S, F2, integ.config(1))
    F2.close(); S.close(); O.close()


def _serial_block(seed):
    return (8, 16, 32)[seed % 3]


@pytest.mark.parametrize("seed", [3, 6, 11, 12, 14, 17, 22, 31])
def test_serial_render_on_fuzzed_scenes(G, seed):
    """gdpt_render_serial on the scenes of tools/gpu_fuzz_campaign.py (random materials incl. glass and near-specular lobes; every third seed an environment, seeds
    = 1 mod 5 vertex normals, = 2 mod 5 a point light, multiples of 7 the atrium, = 4 mod 9 a thin lens; random depth, roulette depth, strict normals, threshold):
    the serial stream has no per-sample reset, so ONE draw out of place anywhere in the sampler's features shifts every later sample of the film."""
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(17, 44)), int(rng.integers(9, 34))
    kind = "random"
    kw = dict(seed=seed, environment=(0.5, 0.7, 0.9) if seed % 3 == 0 else None)
    if seed % 5 == 1:
        kind = "smooth" if seed % 2 else "bent"; kw = dict(environment=kw["environment"])
    if seed % 5 == 2:
        kw["point_light"] = ((float(rng.uniform(100, 450)), float(rng.uniform(200, 500)), float(rng.uniform(100, 450))), (4e4, 3e4, 2e4), bool(seed % 2))
    sc = scenes.atrium(W, H, columns=int(rng.integers(4, 12)), segments=int(rng.integers(6, 16))) if seed % 7 == 0 else scenes.cornell_box(W, H, kind, **kw)
    if seed % 9 == 4:
        sc.thinlens = (float(rng.uniform(2.0, 60.0)), float(rng.uniform(300.0, 1500.0)))
    md = int(rng.choice([-1, 2, 3, 5, 9])); rr = int(rng.choice([1, 3, 5])); strict = bool(rng.random() < 0.35); thr = float(rng.choice([0.001, 0.02, 0.0]))
    spp = int(rng.integers(1, 4))
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, rrDepth=rr, strictNormals=strict, shiftThreshold=thr)
    F = G.Film(S)
    integ.renderSerial(S, F, integ.config(spp), blockSize=_serial_block(seed))
    acc, st = F.accum(), F.stats()
    F.close()
    oacc, orays = O.render_serial(go.config(maxDepth=md, rrDepth=rr, strictNormals=strict, spp=spp, shiftThreshold=thr), block_size=_serial_block(seed))
    assert (st["raysTraced"], st["shadowRaysTraced"]) == orays, seed
    for b in range(5):
        assert close(acc[b], oacc[b], rel=1e-9), (seed, G.BUFFER_NAMES[b], np.abs(acc[b] - oacc[b]).max())
    S.close(); O.close()


def test_thinlens_sensor_argument_checks_and_scope(G):
    sc = scenes.cornell_box(16, 12, "diffuse"); sc.thinlens = (0.0, 500.0)
    with pytest.raises(RuntimeError, match="apertureRadius"):
        G.Scene(sc)
    sc.thinlens = (10.0, 0.0)
    with pytest.raises(RuntimeError, match="focusDistance"):
        G.Scene(sc)
    # lookups filtered by a camera ray's differentials take their footprint from the pinhole: refused with a lens, never evaluated with the wrong one
    tx = scenes.textured_cornell_box(16, 12, filter=scenes.TEXFILTER_EWA); tx.thinlens = (10.0, 500.0)
    with pytest.raises(RuntimeError, match="thinlens"):
        G.Scene(tx)
    tb = scenes.textured_cornell_box(16, 12, filter=scenes.TEXFILTER_BILINEAR); tb.thinlens = (10.0, 500.0)
    S, O = G.Scene(tb), go.Scene(tb)
    integ = G.GradientPathIntegrator(maxDepth=4)
    g = S.evaluate_point(integ.config(8), 7, 6, 3); o = O.evaluate_point(go.config(maxDepth=4, spp=8), 7, 6, 3)
    assert np.allclose(g["throughput"], o["throughput"], rtol=1e-10, atol=1e-14)
    # G-BDPT samples the sensor itself (aperture position, importance): carried since round 5 (tests/test_gbdpt_gpu.py::test_thinlens_sensor_*); its extra
    # emitter step costs one record, so the depth cap is one lower with a lens
    import gradientdomain_mitsuba_amd.gbdpt as B
    rough = scenes.cornell_box(16, 12, "rough"); rough.thinlens = (10.0, 500.0)
    Sb = G.Scene(rough)
    assert np.isfinite(B.GBDPTIntegrator(maxDepth=4).render(Sb, 1)["-primal"]).all()
    with pytest.raises(RuntimeError, match="maxDepth up to 19 with the thinlens"):
        B.GBDPTIntegrator(maxDepth=20).render(Sb, 1)


def test_intersection_record_matches_reference_dgeom_vectors(G):
    """SURVEY 8a rows 20, 22, 23 on the HIP path against the vectors the REFERENCE'S OWN test holds (src/tests/test_dgeom.cpp:35-178 ->
    tests/golden/dgeom_reference.json; tolerances and the two assertions this fork's own code contradicts: tests/test_dgeom_golden.py), then the
    same record against the oracle on a scene with per-vertex normals and texture coordinates, ray by ray."""
    from test_dgeom_golden import load_cases, scene_of, check_record
    for case in load_cases():
        sc = scene_of(case, with_emitter=True)
        S, O = G.Scene(sc), go.Scene(sc)
        prim, rec = S.intersect_record([case["ray"]["o"]], [case["ray"]["d"]])
        assert prim[0] == 0 and rec["t"][0] == 1.0
        one = {k: v[0] for k, v in rec.items()}
        dn = O.normal_derivative(case["ray"]["o"], case["ray"]["d"])        # (getNormalDerivative lives on the G-BDPT side: tests/test_gbdpt_gpu.py holds the device's)
        assert check_record(case, one, dn) >= 8
        orec = O.intersect_record(case["ray"]["o"], case["ray"]["d"])
        for k in ("p", "uv", "geoFrame.n", "shFrame.n", "shFrame.s", "dpdu", "dpdv", "wi"):
            assert np.allclose(one[k], orec[k], rtol=1e-14, atol=1e-15), (case["name"], k)
        S.close()
    sc = scenes.textured_cornell_box(32, 24)
    smooth = scenes.cornell_box(32, 24, "bent")
    for desc in (sc, smooth):
        S, O = G.Scene(desc), go.Scene(desc)
        rng = np.random.default_rng(5)
        o = np.tile(np.array([[278.0, 273.0, -500.0]]), (400, 1)) + rng.normal(size=(400, 3)) * 40.0
        d = rng.normal(size=(400, 3)); d[:, 2] = np.abs(d[:, 2]) + 0.5; d /= np.linalg.norm(d, axis=1, keepdims=True)
        prim, rec = S.intersect_record(o, d)
        hits = 0
        for i in range(400):
            orec = O.intersect_record(o[i], d[i])
            assert (orec is None) == (prim[i] < 0)
            if orec is None:
                continue
            hits += 1
            assert rec["t"][i] == pytest.approx(orec["t"], rel=1e-13)
            for k in ("p", "uv", "geoFrame.n", "shFrame.n", "shFrame.s", "dpdu", "dpdv", "wi"):
                assert np.allclose(rec[k][i], orec[k], rtol=1e-11, atol=1e-11), (desc.name, i, k, rec[k][i], orec[k])
        assert hits > 100
        S.close()


@pytest.mark.parametrize("variant,md,crop,lens", [("diffuse", -1, (10, 7, 64, 48), None), ("glossy", 8, (0, 0, 48, 40), None), ("bent", 6, (23, 17, 50, 38), (25.0, 820.0))])
def test_crop_window_rays_come_from_the_full_films_raster(G, variant, md, crop, lens):
    """A film with a crop window (film.cpp:34-48): the rendered image is the crop, the sensor's rays, aspect and pixel differentials are the full
    film's (perspective.cpp:126-163) and BlockedRenderProcess hands out crop-relative blocks (renderproc.cpp:158-181).  Camera rays of the crop ==
    those of the full film's pixels; samples and films against the oracle through both pipelines; an invalid window is refused as film.cpp refuses it."""
    W, H = 24, 18
    sc = scenes.cornell_box(W, H, variant); sc.crop = crop; sc.thinlens = lens
    full = scenes.cornell_box(crop[2], crop[3], variant); full.thinlens = lens
    S, O, OF = G.Scene(sc), go.Scene(sc), go.Scene(full)
    for (x, y) in ((0.5, 0.5), (3.25, 7.75), (W - 0.5, H - 0.5)):
        a, b = O.camera_ray(x, y), OF.camera_ray(x + crop[0], y + crop[1])
        assert all(np.allclose(u, v, rtol=1e-15, atol=0) for u, v in zip(a, b))
    integ = G.GradientPathIntegrator(maxDepth=md)
    rng = np.random.default_rng(31)
    for _ in range(20):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = S.evaluate_point(integ.config(64), px, py, s)
        o = O.evaluate_point(go.config(maxDepth=md, spp=64), px, py, s)
        for k in ("veryDirect", "throughput", "gradients", "neighbours"):
            assert np.allclose(g[k], o[k], rtol=1e-10, atol=1e-14), (variant, px, py, s, k)
        assert (g["raysTraced"], g["shadowRaysTraced"]) == (o["raysTraced"], o["shadowRaysTraced"])
    spp = 3
    oacc, orays = O.render(go.config(maxDepth=md, spp=spp))
    for pipeline in (2, 0):
        F = G.Film(S); F.set_pipeline(pipeline)
        integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        F.close()
        assert (st["raysTraced"], st["shadowRaysTraced"]) == orays
        for b in range(5):
            assert close(acc[b], oacc[b]), (pipeline, G.BUFFER_NAMES[b])
    # the crop is not the uncropped film of its own size: another field of view
    pacc, _ = go.Scene(scenes.cornell_box(W, H, variant)).render(go.config(maxDepth=md, spp=spp))
    assert not close(oacc[1], pacc[1], rel=1e-3)
    bad = scenes.cornell_box(W, H, variant); bad.crop = (50, 0, 64, 48)
    with pytest.raises(RuntimeError, match="Invalid crop window specification!"):
        G.Scene(bad)
    import gradientdomain_mitsuba_amd.gbdpt as B
    rough = scenes.cornell_box(W, H, "rough"); rough.crop = crop
    with pytest.raises(RuntimeError, match="crop window"):
        B.GBDPTIntegrator(maxDepth=4).render(G.Scene(rough), 1)


@pytest.mark.parametrize("variant,md,strict,env,rfilter", [("diffuse", -1, False, None, None), ("rough", 9, True, (0.3, 0.4, 0.5), None), ("smooth", 7, True, None, None),
                                                          ("twosided", 12, False, None, None), ("diffuse", 6, False, None, scenes.RFILTER_DEFAULTS[scenes.RFILTER_GAUSSIAN]),
                                                          ("bent", 5, False, (0.3, 0.4, 0.5), None)])
@pytest.mark.parametrize("hbm", [False, True])
def test_deferred_continuation_and_pipelined_chunks_equal_the_in_place_kernels(G, monkeypatch, variant, md, strict, env, rfilter, hbm):
    """Round 6: for LDS-resident scenes without glossy vertices the continuation runs DEFERRED (k_walk: the base paths alone, logging what their joined offsets would have
    read; k_replay: the offsets and sums from the log; k_continue for paths longer than two rounds of four bounces) and the chunks of a render are PIPELINED over two streams
    and two sets of queue buffers.  Films, ray and path counters: bit-identical to the in-place continuation (GDPT_NO_DEFERRED=1) and to the unpipelined form
    (GDPT_NO_PIPE=1), with a queue budget that cuts the render into one-sample chunks (set reuse, the waits between the streams) -- and equal to the oracle's.
    (a filter wider than box: the deferred form without the pipeline)
    hbm: the scene's tables in HBM (the builds of configs 3 / 4).  There the two forms also differ in WHERE a sample is handed over: with the deferred continuation after its
    first bounce (k_first; offsets RAY_RECENTLY_CONNECTED), with the in-place one when every offset is RAY_CONNECTED (k_render<STAGED>) -- the films are the same bits.
    And the deferred path of an HBM-resident scene runs the build of exactly the features it uses: "smooth" = per-vertex normals only, "rough" + environment = special emitters
    only, "bent" + environment = both (at three waves per SIMD), against the in-place kernels' folded <true, true> build."""
    if hbm:
        monkeypatch.setenv("GDPT_SCENE_IN_HBM", "1")
    W, H, spp = 44, 36, 7
    sc = scenes.cornell_box(W, H, variant, environment=env)
    if rfilter is not None:
        sc.rfilter = rfilter
    S = G.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict, rrDepth=3)
    cfg = integ.config(spp)
    res = {}
    variants = (("deferred+pipe", {}), ("one-sample chunks", {"GDPT_QUEUE_MB": "2"}), ("no pipe", {"GDPT_NO_PIPE": "1", "GDPT_QUEUE_MB": "2"}), ("in place", {"GDPT_NO_DEFERRED": "1"}),
                       ("in place, one-sample chunks", {"GDPT_NO_DEFERRED": "1", "GDPT_QUEUE_MB": "2"}))
    for name, envs in variants:
        for k in ("GDPT_QUEUE_MB", "GDPT_NO_PIPE", "GDPT_NO_DEFERRED"):
            monkeypatch.delenv(k, raising=False)
        for k, v in envs.items():
            monkeypatch.setenv(k, v)
        F = G.Film(S)
        for rep in range(2):                     # (twice into one film object: the second render reuses the buffers, streams and events of the first)
            F.clear()
            integ.renderBlock(S, F, cfg, (0, 0, W, H))
        res[name] = (F.accum(), F.stats())
        F.close()
    ref_acc, ref_st = res["in place"]
    chunk_acc = res["in place, one-sample chunks"][0]
    if rfilter is None:                          # (the box filter's fold continues from the pixel's record: the same association however the render is cut)
        assert np.array_equal(chunk_acc, ref_acc)
    else:                                        # (a wider filter's gather adds a chunk's sum to the record: the association follows the chunks -- 6e-16 of the film, with either continuation)
        assert np.abs(chunk_acc - ref_acc).max() <= 1e-14 * np.abs(ref_acc).max()
    for name, (acc, st) in res.items():
        assert st == ref_st, (name, st, ref_st)
        assert np.array_equal(acc, chunk_acc if "GDPT_QUEUE_MB" in dict(variants)[name] else ref_acc), name
    oacc, orays = go.Scene(sc).render(go.config(maxDepth=md, spp=spp, strictNormals=strict, rrDepth=3))
    assert (ref_st["raysTraced"], ref_st["shadowRaysTraced"]) == orays
    for b in range(5):
        assert close(ref_acc[b], oacc[b]), G.BUFFER_NAMES[b]
    S.close()


@pytest.mark.gpu
def test_hbm_scene_with_vertex_normals_runs_the_exact_builds_to_the_same_bits(G, monkeypatch):
    """Round 6 (r06g): an HBM-resident scene with per-vertex data but no special emitters runs <ENV, SMOOTH> = <false, true> builds -- of k_first / k_walk / the k_continue tail on
    the deferred path (no glossy vertices), of k_render<STAGED> / k_continue on the in-place path (glossy vertices) -- instead of the folded <true, true> build
    (GDPT_NO_EXACT_BUILDS=1): the same bits, and the oracle's film."""
    monkeypatch.setenv("GDPT_SCENE_IN_HBM", "1")
    W, H, spp = 40, 30, 5
    for variant, md in (("diffuse", 7), ("glossy", 9), ("glass", 8)):
        sc = scenes.cornell_box(W, H, variant)
        v = np.asarray(sc.verts, np.float64).reshape(-1, 3, 3)
        n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True)
        tilt = np.array([0.05, -0.03, 0.04])             # (shading normals that differ from the face normals, per vertex)
        nv = [n + tilt * (k + 1) * 0.5 for k in range(3)]
        sc.normals = np.concatenate([a / np.linalg.norm(a, axis=1, keepdims=True) for a in nv], axis=1)
        for e in sc.emitters:
            if not isinstance(e[0], str):
                sc.normals[int(e[0]):int(e[0]) + int(e[1])] = 0.0
        S = G.Scene(sc)
        integ = G.GradientPathIntegrator(maxDepth=md, rrDepth=3)
        cfg = integ.config(spp)
        res = {}
        for name, envs in (("exact", {}), ("folded", {"GDPT_NO_EXACT_BUILDS": "1"})):
            monkeypatch.delenv("GDPT_NO_EXACT_BUILDS", raising=False)
            for k, val in envs.items():
                monkeypatch.setenv(k, val)
            F = G.Film(S)
            integ.renderBlock(S, F, cfg, (0, 0, W, H))
            res[name] = (F.accum(), F.stats())
            F.close()
        monkeypatch.delenv("GDPT_NO_EXACT_BUILDS", raising=False)
        assert res["exact"][1] == res["folded"][1], variant
        assert np.array_equal(res["exact"][0], res["folded"][0]), variant
        oacc, orays = go.Scene(sc).render(go.config(maxDepth=md, spp=spp, rrDepth=3))
        assert (res["exact"][1]["raysTraced"], res["exact"][1]["shadowRaysTraced"]) == orays, variant
        for b in range(5):
            assert close(res["exact"][0][b], oacc[b]), (variant, G.BUFFER_NAMES[b])
        S.close()
