"""GPU parity tests of the gfx950 G-BDPT sampler (csrc/gbdpt_kernels.hip.h), called through the C-ABI (include/gdpt_tracer.h, "G-BDPT").

Checker: oracle/gbdpt_oracle.hpp (PARITY UNPINNED -- a line-cited fp64 restatement of GBDPTRenderer::process / evaluate over libbidir, see its
header and DESIGN.md).  Both sides draw the same counter-based random numbers in the reference's order, so they build IDENTICAL subpaths,
offset paths and connections; what differs is libm (ocml vs glibc), std::pow against x * x in the MIS weights, and the order of the fp64
film sums (atomics).  Bar: single samples rtol 1e-9 (observed ~1e-14), films 1e-9 of the buffer scale per pixel, ray counts identical."""
import numpy as np
import pytest

from gradientdomain_mitsuba_amd import scenes
from oracle import gpt_oracle as go
from oracle import poisson_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(gpu_required):
    import gradientdomain_mitsuba_amd.gpt as G
    return G


@pytest.fixture(scope="module")
def B(gpu_required):
    import gradientdomain_mitsuba_amd.gbdpt as B
    return B


def builders():
    return {"diffuse": lambda w, h: scenes.cornell_box(w, h, "diffuse"), "twosided": lambda w, h: scenes.cornell_box(w, h, "twosided"),
            "rough": lambda w, h: scenes.cornell_box(w, h, "rough"), "smooth": lambda w, h: scenes.cornell_box(w, h, "smooth"),
            "textured": lambda w, h: scenes.textured_cornell_box(w, h), "veach": lambda w, h: scenes.veach_bidir(w, h),
            # round 4 (stage C): specular chains -- a solid glass block + a mirror block, a mirror back wall, a rough conductor below shiftThreshold,
            # glass and mirror spheres with interpolated normals (the manifold's normal derivatives)
            "glass": lambda w, h: scenes.cornell_box(w, h, "glass"), "glossy": lambda w, h: scenes.cornell_box(w, h, "glossy"),
            "nearspecular": lambda w, h: scenes.cornell_box(w, h, "nearspecular"), "veach_specular": lambda w, h: scenes.veach_bidir(w, h, specular=True)}


def compare_sample(g, o, what):
    assert o["unsupported"] == 0, what
    assert (g["raysTraced"], g["shadowRaysTraced"]) == (o["raysTraced"], o["shadowRaysTraced"]), what
    assert np.allclose(g["position"], o["position"], rtol=1e-14, atol=0), what
    # (a value the reference's own arithmetic makes NaN or infinite -- DESIGN.md "G-BDPT: endpoints" -- has to be NaN or infinite on both sides)
    fin = lambda a: np.nan_to_num(np.asarray(a, float), nan=0.0, posinf=0.0, neginf=0.0)
    scale = max(np.abs(fin(o["primal"])).max(), np.abs(fin(o["gradients"])).max(), 1e-300)
    assert np.allclose(g["primal"], o["primal"], rtol=1e-9, atol=1e-12 * scale, equal_nan=True), (what, g["primal"], o["primal"])
    assert np.allclose(g["gradients"], o["gradients"], rtol=1e-9, atol=1e-12 * scale, equal_nan=True), (what, g["gradients"], o["gradients"])
    assert g["light"].shape == o["light"].shape, what
    if len(o["light"]):
        assert np.array_equal(g["light"][:, 2], o["light"][:, 2]), what
        assert np.allclose(g["light"][:, :2], o["light"][:, :2], rtol=1e-12, atol=1e-9), what
        ls = np.abs(fin(o["light"][:, 3:])).max() + 1e-300
        assert np.allclose(g["light"][:, 3:], o["light"][:, 3:], rtol=1e-9, atol=1e-12 * ls, equal_nan=True), what


@pytest.mark.parametrize("name,md,li", [("diffuse", 5, True), ("diffuse", -1, True), ("diffuse", 3, False), ("twosided", 7, True), ("rough", 6, True),
                                         ("rough", -1, False), ("smooth", 6, True), ("textured", 5, True), ("veach", -1, True), ("veach", 4, False),
                                         ("glass", 7, True), ("glass", -1, False), ("glossy", 6, True), ("glossy", -1, False), ("nearspecular", 6, True), ("nearspecular", -1, False),
                                         ("veach_specular", -1, True), ("veach_specular", 5, False)])
def test_samples_match_oracle(G, B, name, md, li):
    W, H = 40, 30
    sc = builders()[name](W, H)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, lightImage=li)
    cfg, ocfg = integ.config(64), go.gbdpt_config(maxDepth=md, lightImage=li, spp=64)
    rng = np.random.default_rng(17)
    nonzero = lights = general = 0
    specular = name in ("glass", "glossy", "nearspecular", "veach_specular")
    for _ in range(90 if specular else 60):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = integ.evaluate_sample(S, cfg, px, py, s)
        o = O.gbdpt_sample(ocfg, px, py, s)
        compare_sample(g, o, (name, md, li, px, py, s))
        assert g["overflow"] == 0
        nonzero += bool(o["primal"].any()); lights += len(o["light"]); general += g["general"]
    assert nonzero > 20 and (lights > 0) == li
    # samples that meet a specular vertex run the general form (csrc/gbdpt_general.hip.h: propagatePerturbation, manifold walks, generalized
    # geometry terms); scenes without one never enter it
    assert (general > 10) if specular else (general == 0), (name, general)
    S.close(); O.close()


@pytest.mark.parametrize("name,W,H,spp,md,li", [("diffuse", 48, 36, 4, 6, True), ("rough", 40, 30, 3, -1, True), ("veach", 64, 36, 2, -1, True), ("twosided", 32, 24, 5, 5, False),
                                                ("glass", 40, 30, 3, 8, True), ("glossy", 40, 30, 3, -1, True), ("nearspecular", 32, 24, 3, 6, False), ("veach_specular", 64, 36, 2, -1, True)])
def test_film_matches_oracle(G, B, name, W, H, spp, md, li):
    """The five camera blocks and five light images of a whole film (GBDPTWorkResult + processResult) against the oracle's, the developed
    buffers (GBDPTProcess::develop) and the ray counters."""
    sc = builders()[name](W, H)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, lightImage=li)
    F = B.Film(S)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    block, light = F.accum()
    st = F.stats()
    ob, ol, oc = O.gbdpt_render(go.gbdpt_config(maxDepth=md, lightImage=li, spp=spp))
    assert oc["unsupported"] == 0 and st["invalidPuts"] == oc["invalidPuts"] == 0
    assert (st["raysTraced"], st["shadowRaysTraced"], st["samples"]) == (oc["raysTraced"], oc["shadowRaysTraced"], W * H * spp)
    cs = F.chain_stats()
    assert cs["overflows"] == 0 and ((cs["generalSamples"] > 100) if name in ("glass", "glossy", "nearspecular", "veach_specular") else (cs["generalSamples"] == 0)), cs
    if oc["manifoldWalks"]:
        assert oc["manifoldWalksConverged"] > 0.4 * oc["manifoldWalks"]
    for b in range(5):
        assert np.abs(block[b] - ob[b]).max() <= 1e-9 * (np.abs(ob[b]).max() + 1e-300), (name, "block", b)
        assert np.abs(light[b] - ol[b]).max() <= 1e-9 * (np.abs(ol[b]).max() + 1e-300), (name, "light", b)
    dev = go.gbdpt_develop(ob, ol, spp)
    for b in range(5):
        d = F.develop(b, spp)
        assert np.abs(d - dev[b]).max() <= 1e-9 * (np.abs(dev[b]).max() + 1e-300), (name, "develop", b)
    # tiles rendered one after the other into the same film == the one-call film (sums to rounding: atomics)
    F.clear()
    for (x0, y0, x1, y1) in ((0, 0, W // 2, H), (W // 2, 0, W, H // 2), (W // 2, H // 2, W, H)):
        integ.renderBlock(S, F, integ.config(spp), (x0, y0, x1, y1))
    b2, l2 = F.accum()
    assert np.allclose(b2, block, rtol=1e-12, atol=1e-12) and np.allclose(l2, light, rtol=1e-12, atol=1e-12) and F.stats() == st
    F.close(); S.close(); O.close()


@pytest.mark.parametrize("name,md,li,lens", [("diffuse", 5, True, (40.0, 1100.0)), ("diffuse", -1, False, (120.0, 600.0)), ("rough", 6, True, (25.0, 900.0)), ("glass", 7, True, (40.0, 1100.0)),
                                              ("glossy", -1, True, (60.0, 700.0)), ("veach_specular", 6, True, (0.35, 9.0)), ("veach", 19, True, (0.2, 12.0))])
def test_thinlens_sensor_samples_and_films_match_oracle(G, B, name, md, li, lens):
    """`<sensor type="thinlens">` under G-BDPT (round 5; refused until then).  The sensor sample is a point of the aperture disk (EArea: the sensor end of a path
    becomes connectable, thinlens.cpp:363-384), every direction query goes through the pixel's point of the focus plane (importance / getSamplePosition,
    thinlens.cpp:231-291,536-557), the shift aims at the focus point of the offset pixel from the base path's OWN aperture point (mut_manifold.cpp:957-971),
    and the emitter subpath takes one more step because the sensor is no longer a point (gbdpt_proc.cpp:117-118).  Samples and a small film against the oracle."""
    W, H = 40, 30
    sc = builders()[name](W, H); sc.thinlens = lens
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, lightImage=li)
    cfg, ocfg = integ.config(64), go.gbdpt_config(maxDepth=md, lightImage=li, spp=64)
    rng = np.random.default_rng(23)
    nonzero = general = 0
    for _ in range(60):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = integ.evaluate_sample(S, cfg, px, py, s)
        o = O.gbdpt_sample(ocfg, px, py, s)
        compare_sample(g, o, (name, md, li, lens, px, py, s))
        assert g["overflow"] == 0
        nonzero += bool(o["primal"].any()); general += g["general"]
    assert nonzero > 20
    assert (general > 5) == (name in ("glass", "glossy", "veach_specular")), (name, general)
    spp = 2
    F = B.Film(S)
    cfg, ocfg = integ.config(spp), go.gbdpt_config(maxDepth=md, lightImage=li, spp=spp)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    blk, light = F.accum()
    st = F.stats()
    oblk, olight, ocnt = O.gbdpt_render(ocfg)
    assert ocnt["unsupported"] == 0 and F.chain_stats()["overflows"] == 0
    if name not in ("glass", "glossy", "veach_specular"):      # (specular chains: a ray count can sit on a knife edge, see test_paths_deeper_than_twelve_match_oracle)
        assert (st["raysTraced"], st["shadowRaysTraced"]) == (ocnt["raysTraced"], ocnt["shadowRaysTraced"])
    for a, b in ((blk, oblk), (light, olight)):
        scale = np.abs(b).max() + 1e-300
        assert np.abs(a - b).max() <= 1e-9 * scale, (name, np.abs(a - b).max() / scale)
    F.close(); S.close(); O.close()


@pytest.mark.parametrize("name,md,li,mode", [("diffuse", 5, True, "both"), ("diffuse", -1, False, "first"), ("rough", 6, True, "only"), ("glass", 7, True, "both"),
                                              ("glossy", -1, True, "only"), ("twosided", 4, True, "only"), ("diffuse", 5, True, "nan"), ("rough", 4, True, "nan_lens")])
def test_point_emitters_samples_and_films_match_oracle(G, B, name, md, li, mode):
    """`point` emitters under G-BDPT (round 5; refused until then): a position sample with a discrete measure (the emitter end of a path is not connectable, so no
    strategy ever 'hits' the light: point.cpp:79-95), directions uniform over the sphere without a cosine (not EOnSurface: point.cpp:97-115, vertex.h:592-596), and
    -- when EVERY emitter is a point -- no extra sensor step (Scene::hasDegenerateEmitters, gbdpt_proc.cpp:120-122).  Beside, before, and instead of the area light."""
    W, H = 40, 30
    sc = builders()[name](W, H)
    pl = ("point", (278.0, 400.0, 279.5), (4e4, 3e4, 2e4))
    # "nan": a point light just below the area light.  The light image sees it (s = 1, t = 1), and the ray of that splat's upward shift ends ON the area light, so the
    # offset path connects -- and the reference's half-Jacobian of the base path is G / G with the point sample's zero normal: 0 / 0 (DESIGN.md "G-BDPT: endpoints").
    # The NaN must come out of both sides, and the film must drop the same puts.
    under = ("point", (298.0, 535.0, 280.0), (4e4, 3e4, 2e4))
    sc.emitters = {"both": sc.emitters + [pl], "first": [pl] + sc.emitters, "only": [pl, ("point", (120.0, 90.0, 140.0), (1e4, 2e4, 3e4))],
                   "nan": sc.emitters + [under], "nan_lens": [under] + sc.emitters}[mode]
    if mode == "nan_lens": sc.thinlens = (30.0, 1080.0)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, lightImage=li)
    cfg, ocfg = integ.config(64), go.gbdpt_config(maxDepth=md, lightImage=li, spp=64)
    rng = np.random.default_rng(29)
    nonzero = nans = 0
    for _ in range(60):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = integ.evaluate_sample(S, cfg, px, py, s)
        o = O.gbdpt_sample(ocfg, px, py, s)
        compare_sample(g, o, (name, md, li, mode, px, py, s))
        assert g["overflow"] == 0
        nonzero += bool(o["primal"].any())
        nans += bool(len(o["light"]) and np.isnan(o["light"]).any())
    assert nonzero > 20 and (nans > 5) == mode.startswith("nan"), (nonzero, nans)
    spp = 2
    F = B.Film(S)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    blk, light = F.accum()
    st = F.stats()
    oblk, olight, ocnt = O.gbdpt_render(go.gbdpt_config(maxDepth=md, lightImage=li, spp=spp))
    assert ocnt["unsupported"] == 0 and F.chain_stats()["overflows"] == 0
    assert st["invalidPuts"] == ocnt["invalidPuts"] and (ocnt["invalidPuts"] > 100) == mode.startswith("nan"), (st["invalidPuts"], ocnt["invalidPuts"])
    # (ray counts: a sample whose weight underflows to exactly 0 on one side and to 1e-31 on the other traces its four offset paths on one side only -- two of the
    #  2400 samples of the two-point-lights film, located with tools/gpu_gbdpt_point_locate.py; outputs equal.  The fuzz tool counts these as "ray-count knife edges")
    assert abs(st["raysTraced"] - ocnt["raysTraced"]) <= 1e-3 * ocnt["raysTraced"] and abs(st["shadowRaysTraced"] - ocnt["shadowRaysTraced"]) <= 1e-3 * ocnt["shadowRaysTraced"]
    if mode != "only" and name not in ("glass", "glossy"):
        assert (st["raysTraced"], st["shadowRaysTraced"]) == (ocnt["raysTraced"], ocnt["shadowRaysTraced"])
    for a, b in ((blk, oblk), (light, olight)):
        scale = np.abs(b).max() + 1e-300
        assert np.abs(a - b).max() <= 1e-9 * scale, (name, np.abs(a - b).max() / scale)
    F.close(); S.close(); O.close()


@pytest.mark.parametrize("name,md,li,where", [("diffuse", 5, True, "last"), ("diffuse", -1, False, "first"), ("rough", 6, True, "only"), ("glass", 7, True, "last"),
                                               ("glossy", -1, True, "first"), ("nearspecular", 5, False, "only"),
                                               ("diffuse", 5, True, "map_last"), ("rough", -1, False, "map_first"), ("glass", 6, True, "map_only"), ("nearspecular", 5, True, "map_only")])
def test_constant_environment_samples_and_films_match_oracle(G, B, name, md, li, where):
    """The `constant` environment emitter under G-BDPT (round 5; refused until then).  libbidir sees it as a SHAPE: a sphere 1.5x the bounding sphere of kd-tree +
    sensor, flipped normals, an all-absorbing diffuse BSDF (scene.cpp:397-408, constant.cpp:67-93, Shape::configure), tested after the kd-tree by rayIntersectAll --
    so a subpath that leaves the geometry ends in a connectable, black SURFACE vertex that `cast()` turns into an emitter sample, and emitter subpaths start on that
    sphere with a cosine lobe about the inward normal (constant.cpp:110-160).  Last, first, or the only emitter of the scene; both forms.
    `map_*`: the `envmap` environment (the last endpoint kind, round 5): the same sphere and uniform positions, but directions importance-sampled from the map whatever
    the position, a coloured evalDirection and m_power from the map's normalization (envmap.cpp:326-328,412-498); a rotated map in the `only` cases."""
    W, H = 40, 30
    sc = builders()[name](W, H)
    env = (0.6, 0.7, 0.9)
    if where.endswith("only"): sc.emitters = []
    index = 0 if where.endswith("first") else len(sc.emitters)
    if where.startswith("map"):
        a = 0.7
        rot = [[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]] if where == "map_only" else np.eye(3)
        sc.environment_map = dict(rgb=scenes.sky_map(16, 8), scale=1.5, index=index, toWorld=rot)
    else: sc.environment = (env, index)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, lightImage=li)
    cfg, ocfg = integ.config(64), go.gbdpt_config(maxDepth=md, lightImage=li, spp=64)
    rng = np.random.default_rng(31)
    nonzero = general = 0
    for _ in range(60):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = integ.evaluate_sample(S, cfg, px, py, s)
        o = O.gbdpt_sample(ocfg, px, py, s)
        compare_sample(g, o, (name, md, li, where, px, py, s))
        assert g["overflow"] == 0
        nonzero += bool(o["primal"].any()); general += g["general"]
    assert nonzero > 20 and (general > 5) == (name in ("glass", "glossy", "nearspecular")), (nonzero, general)
    spp = 2
    F = B.Film(S)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    blk, light = F.accum()
    st = F.stats()
    oblk, olight, ocnt = O.gbdpt_render(go.gbdpt_config(maxDepth=md, lightImage=li, spp=spp))
    assert ocnt["unsupported"] == 0 and F.chain_stats()["overflows"] == 0 and st["invalidPuts"] == ocnt["invalidPuts"]
    assert abs(st["raysTraced"] - ocnt["raysTraced"]) <= 1e-3 * ocnt["raysTraced"] and abs(st["shadowRaysTraced"] - ocnt["shadowRaysTraced"]) <= 1e-3 * ocnt["shadowRaysTraced"]
    if name in ("diffuse", "rough"):
        assert (st["raysTraced"], st["shadowRaysTraced"]) == (ocnt["raysTraced"], ocnt["shadowRaysTraced"])
    for a, b in ((blk, oblk), (light, olight)):
        scale = np.abs(b).max() + 1e-300
        assert np.abs(a - b).max() <= 1e-9 * scale, (name, np.abs(a - b).max() / scale)
    F.close(); S.close(); O.close()


def test_shutter_interval_draws_the_time_sample_first(G, B):
    """gbdpt_proc.cpp:156-157: with needsTimeSample() the time sample is the FIRST draw of a sample (before the random walks); the subpaths' time moves
    nothing (static transforms).  Samples and a film against the oracle, with the general form in play (glass)."""
    W, H = 40, 30
    sc = scenes.cornell_box(W, H, "glass"); sc.shutter = (0.0, 0.02)
    S, O = G.Scene(sc), go.Scene(sc)
    S0 = G.Scene(scenes.cornell_box(W, H, "glass"))
    integ = B.GBDPTIntegrator(maxDepth=7)
    cfg, ocfg = integ.config(64), go.gbdpt_config(maxDepth=7, lightImage=True, spp=64)
    rng = np.random.default_rng(5)
    moved = 0
    for _ in range(40):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = integ.evaluate_sample(S, cfg, px, py, s)
        compare_sample(g, O.gbdpt_sample(ocfg, px, py, s), ("shutter", px, py, s))
        moved += not np.allclose(g["position"], integ.evaluate_sample(S0, cfg, px, py, s)["position"])
    assert moved == 40                                   # the film position is drawn behind the time sample: another one for every sample
    F = B.Film(S)
    integ.renderBlock(S, F, integ.config(2), (0, 0, W, H))
    block, light = F.accum()
    ob, ol, oc = O.gbdpt_render(go.gbdpt_config(maxDepth=7, lightImage=True, spp=2))
    st = F.stats()
    assert (st["raysTraced"], st["shadowRaysTraced"]) == (oc["raysTraced"], oc["shadowRaysTraced"])
    for b in range(5):
        assert np.abs(block[b] - ob[b]).max() <= 1e-9 * (np.abs(ob[b]).max() + 1e-300)
        assert np.abs(light[b] - ol[b]).max() <= 1e-9 * (np.abs(ol[b]).max() + 1e-300)
    F.close(); S.close(); S0.close(); O.close()


def test_small_workspace_chunks_give_the_same_film(G, B):
    """gdpt_gbdpt_render_rect walks a rectangle's samples in chunks of its workspace (sized from the memory the device has free; the `first +=
    chunk` loop).  Forced down to chunks that cut through pixels (GDPT_BD_CHUNK = 777 and 5000 samples of 48 x 36 x 4 = 6912): the same samples,
    ray counts identical, film equal to the one-chunk film to the rounding of the fp64 atomic splats, and to the oracle."""
    import os
    W, H, spp = 48, 36, 4
    sc = scenes.cornell_box(W, H, "rough")
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=6, lightImage=True)
    F = B.Film(S)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    block, light = F.accum(); st = F.stats()
    F.close()
    ob, ol, oc = O.gbdpt_render(go.gbdpt_config(maxDepth=6, lightImage=True, spp=spp))
    for forced in ("777", "5000"):
        os.environ["GDPT_BD_CHUNK"] = forced
        try:
            F = B.Film(S)
            integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
            b2, l2 = F.accum(); st2 = F.stats()
            F.close()
        finally:
            del os.environ["GDPT_BD_CHUNK"]
        assert st2 == st and (st2["raysTraced"], st2["shadowRaysTraced"]) == (oc["raysTraced"], oc["shadowRaysTraced"]), forced
        assert np.allclose(b2, block, rtol=1e-12, atol=1e-12) and np.allclose(l2, light, rtol=1e-12, atol=1e-12), forced
        for b in range(5):
            assert np.abs(b2[b] - ob[b]).max() <= 1e-9 * (np.abs(ob[b]).max() + 1e-300), (forced, "block", b)
            assert np.abs(l2[b] - ol[b]).max() <= 1e-9 * (np.abs(ol[b]).max() + 1e-300), (forced, "light", b)
    S.close(); O.close()


def test_small_general_form_passes_give_the_same_film(G, B):
    """The general form of a chunk runs in passes of at most `gsCap` samples (one GSamp record per sample of a pass; gdpt_gbdpt_render_rect's
    `gFirst += gsCap` loop).  Forced down to 300 and 1 samples per pass on a scene where most samples are general: identical ray counts,
    the film of the default pass size to the rounding of the fp64 atomics, and the oracle's."""
    import os
    W, H, spp = 40, 30, 2
    sc = scenes.cornell_box(W, H, "glass")
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=7, lightImage=True)
    F = B.Film(S)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
    block, light = F.accum(); st = F.stats(); ch = F.chain_stats()
    F.close()
    assert ch["generalSamples"] > 0.3 * W * H * spp and ch["overflows"] == 0
    ob, ol, oc = O.gbdpt_render(go.gbdpt_config(maxDepth=7, lightImage=True, spp=spp))
    for forced in ("300", "1"):
        os.environ["GDPT_BD_GENERAL_PASS"] = forced
        try:
            F = B.Film(S)
            integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
            b2, l2 = F.accum(); st2 = F.stats(); ch2 = F.chain_stats()
            F.close()
        finally:
            del os.environ["GDPT_BD_GENERAL_PASS"]
        assert st2 == st and ch2 == ch and (st2["raysTraced"], st2["shadowRaysTraced"]) == (oc["raysTraced"], oc["shadowRaysTraced"]), forced
        assert np.allclose(b2, block, rtol=1e-12, atol=1e-12) and np.allclose(l2, light, rtol=1e-12, atol=1e-12), forced
        for b in range(5):
            assert np.abs(b2[b] - ob[b]).max() <= 1e-9 * (np.abs(ob[b]).max() + 1e-300), (forced, "block", b)
            assert np.abs(l2[b] - ol[b]).max() <= 1e-9 * (np.abs(ol[b]).max() + 1e-300), (forced, "light", b)
    S.close(); O.close()


def test_scope_and_property_errors(G, B):
    from gradientdomain_mitsuba_amd._lib import GdptError
    with pytest.raises(RuntimeError, match="two reconstructions"):
        B.GBDPTIntegrator(reconstructL1=True, reconstructL2=True)
    with pytest.raises(RuntimeError, match="reconstructAlpha"):
        B.GBDPTIntegrator(reconstructAlpha=0.0)
    with pytest.raises(RuntimeError, match="rrDepth"):
        B.GBDPTIntegrator(rrDepth=0)
    with pytest.raises(RuntimeError, match="maxDepth"):
        B.GBDPTIntegrator(maxDepth=0)
    assert B.GBDPTIntegrator().outNames() == ["-L1", "-gradientNegY", "-gradientNegX", "-gradientPosX", "-gradientPosY", "-L2", "-primal"]
    assert B.GBDPTIntegrator(reconstructL1=False, reconstructL2=True).outNames()[0] == "-L2"
    for variant in ("glossy", "nearspecular"):       # a mirror; a roughness below the threshold: refused until round 4, rendered (general form) since
        S = G.Scene(scenes.cornell_box(16, 12, variant))
        F = B.Film(S)
        integ = B.GBDPTIntegrator(maxDepth=4)
        integ.renderBlock(S, F, integ.config(1), (0, 0, 16, 12))
        assert F.chain_stats()["generalSamples"] > 0 and F.chain_stats()["overflows"] == 0
        F.close(); S.close()
    sc = scenes.cornell_box(16, 12, "diffuse"); sc.environment_map = dict(rgb=scenes.sky_map(16, 8), scale=1.0, index=-1)     # (both environment kinds are carried since round 5)
    S = G.Scene(sc)
    assert np.isfinite(B.GBDPTIntegrator(maxDepth=4).render(S, 1)["-primal"]).all()
    S.close()
    sc = scenes.cornell_box(16, 12, "diffuse"); sc.thinlens = (10.0, 800.0)       # (the extra emitter step of a sensor with an aperture needs one more record)
    S = G.Scene(sc)
    F = B.Film(S)
    with pytest.raises(GdptError, match="maxDepth up to 19 with the thinlens"):
        B.GBDPTIntegrator(maxDepth=20).renderBlock(S, F, B.GBDPTIntegrator(maxDepth=20).config(1), (0, 0, 16, 12))
    F.close(); S.close()
    S = G.Scene(scenes.cornell_box(16, 12, "diffuse"))
    F = B.Film(S)
    with pytest.raises(GdptError, match="maxDepth up to 20"):     # (round 5: 12 until then; a sample's record holds whole subpaths, so there is a cap, and it is said)
        B.GBDPTIntegrator(maxDepth=21).renderBlock(S, F, B.GBDPTIntegrator(maxDepth=21).config(1), (0, 0, 16, 12))
    F.close(); S.close()


@pytest.mark.parametrize("name,md,rr", [("diffuse", 20, 18), ("glass", 17, 16), ("rough", 14, 13), ("veach_specular", 20, 19)])
def test_paths_deeper_than_twelve_match_oracle(G, B, name, md, rr):
    """The reference takes any positive maxDepth (gbdpt.cpp:102-103; only -1 renders as 12, gbdpt_proc.cpp:103-106).  Rounds 3-4 refused more than 12; the
    records now hold subpaths of up to 20 + 2 vertices.  With Russian roulette starting just below maxDepth the subpaths really get that long:
    single samples and a film, connectable scenes (the fast form: 230 connections per sample at depth 20) and specular ones (the general form's pools)."""
    W, H, spp = 24, 18, 2
    sc = builders()[name](W, H)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, rrDepth=rr)
    cfg, ocfg = integ.config(spp), go.gbdpt_config(maxDepth=md, rrDepth=rr, spp=spp)
    rng = np.random.default_rng(md)
    for _ in range(30):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = integ.evaluate_sample(S, cfg, px, py, s), O.gbdpt_sample(ocfg, px, py, s)
        if name not in ("diffuse", "rough"):
            g = dict(g, raysTraced=o["raysTraced"] if abs(g["raysTraced"] - o["raysTraced"]) <= 0.15 * o["raysTraced"] else g["raysTraced"])   # (see below)
        compare_sample(g, o, (name, md, px, py, s))
    F = B.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    blk, lgt = F.accum(); st = F.stats(); ch = F.chain_stats()
    F.close()
    ob, ol, oc = O.gbdpt_render(ocfg)
    assert oc["unsupported"] == 0 and ch["overflows"] == 0
    if name in ("diffuse", "rough"):
        assert (st["raysTraced"], st["shadowRaysTraced"]) == (oc["raysTraced"], oc["shadowRaysTraced"])
    else:
        # a manifold walk stops on thresholds (step size, 20 iterations): on the last bit of its input it takes a Newton step -- one re-traced chain -- more or
        # less and arrives at the same vertex.  Sample (9, 3, 0) of the depth-17 glass frame: 183 closest-hit rays here, 172 in the oracle, values equal
        # to 2e-16 -- and the ORACLE's own count for it runs from 171 to 202 over 41 scalings of the geometry by 1 +- k 2^-50 (tools/gpu_gbdpt_depth_locate.py)
        assert st["shadowRaysTraced"] == oc["shadowRaysTraced"] and abs(st["raysTraced"] - oc["raysTraced"]) <= 1e-3 * oc["raysTraced"]
    for b in range(5):
        assert np.abs(blk[b] - ob[b]).max() <= 1e-9 * (np.abs(ob[b]).max() + 1e-300), (name, "block", b)
        assert np.abs(lgt[b] - ol[b]).max() <= 1e-9 * (np.abs(ol[b]).max() + 1e-300), (name, "light", b)
    # paths of this depth occur: the deepest pair of subpaths of the frame traces more closest-hit rays than maxDepth 12 allows a sample (2 x 13 + offsets)
    assert st["raysTraced"] / float(W * H * spp) > 20.0
    S.close(); O.close()


def test_integrator_end_to_end_matches_oracle_pipeline(G, B):
    """GBDPTIntegrator::render: sampler -> develop -> prepareDataForSolver -> L2D and L1D, against the same sequence on the oracle side."""
    W, H, spp = 64, 48, 8
    sc = scenes.cornell_box(W, H, "diffuse")
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=6)
    out = integ.render(S, spp)
    ob, ol, oc = O.gbdpt_render(go.gbdpt_config(maxDepth=6, spp=spp))
    dev = go.gbdpt_develop(ob, ol, spp)
    for i, name in enumerate(B.SAMPLER_BUFFERS):
        assert np.abs(out[name] - dev[i]).max() <= 1e-9 * (np.abs(dev[i]).max() + 1e-300), name
    o2, o1 = po.gbdpt_reconstruct(*[dev[i] for i in range(5)], W, H, alpha=0.2)
    assert np.abs(out["-L2"].ravel() - o2).max() <= 5e-5 * max(1.0, np.abs(o2).max())
    assert np.abs(out["-L1"].ravel() - o1).max() <= 5e-4 * max(1.0, np.abs(o1).max())
    assert integ.stats["samples"] == W * H * spp and integ.stats["raysTraced"] == oc["raysTraced"]
    S.close(); O.close()


def test_gradient_domain_reconstruction_beats_the_primal_image(G, B):
    """What G-BDPT is for (Manzi et al. 2015): at equal sample count the reconstructions are closer to the converged image than the primal."""
    W, H, spp = 96, 54, 8
    S = G.Scene(scenes.veach_bidir(W, H))
    ref = B.GBDPTIntegrator(maxDepth=8).render(S, 64 * spp, seed=99, reconstruct=False)["-primal"]
    out = B.GBDPTIntegrator(maxDepth=8).render(S, spp)
    rel = lambda img: float(np.mean((img - ref) ** 2 / (ref ** 2 + 1e-3)))
    e0, e1, e2 = rel(out["-primal"]), rel(out["-L1"]), rel(out["-L2"])
    assert np.isfinite(out["-L1"]).all() and np.isfinite(out["-L2"]).all()
    assert e1 < 0.7 * e0 and e2 < 0.8 * e0, (e0, e1, e2)
    S.close()


def test_config5_veach_1280x720_frame_matches_oracle(G, B):
    """BASELINE config 5's resolution and scene class (Veach-bidir stand-in, G-BDPT, 1280x720) at 1 of its 128 spp: every pixel of the five
    camera blocks and light images and both ray counters against the oracle; then samples of the 128 spp configuration spot-checked."""
    W, H = 1280, 720
    sc = scenes.veach_bidir(W, H)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=-1)
    F = B.Film(S)
    integ.renderBlock(S, F, integ.config(1), (0, 0, W, H))
    block, light = F.accum(); st = F.stats()
    ob, ol, oc = O.gbdpt_render(go.gbdpt_config(maxDepth=-1, spp=1))
    # Ray counts: a connection between two vertices of the SAME plane (both subpaths keep landing on the wall behind the sconce) runs a
    # visibility ray inside that plane -- hit or miss is decided by the last bit, the connection's geometry term is ~1e-27 either way; a
    # Beckmann lobe at its 1e-20 cut-off (microfacet.h:191-234) does the same.  42 of the 921 600 samples of this frame differ by 1-10
    # closest-hit rays with outputs equal to 1e-25 (tools/gpu_gbdpt_locate.py, tools/gpu_gbdpt_trace.py); everything else is identical.
    assert oc["unsupported"] == 0 and st["samples"] == W * H and st["shadowRaysTraced"] == oc["shadowRaysTraced"]
    assert abs(st["raysTraced"] - oc["raysTraced"]) <= 2e-6 * oc["raysTraced"]
    for b in range(5):
        assert np.abs(block[b] - ob[b]).max() <= 1e-9 * (np.abs(ob[b]).max() + 1e-300), ("block", b)
        assert np.abs(light[b] - ol[b]).max() <= 1e-9 * (np.abs(ol[b]).max() + 1e-300), ("light", b)
    cfg, ocfg = integ.config(128), go.gbdpt_config(maxDepth=-1, spp=128)
    rng = np.random.default_rng(3)
    for _ in range(30):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 128))
        compare_sample(integ.evaluate_sample(S, cfg, px, py, s), O.gbdpt_sample(ocfg, px, py, s), (px, py, s))
    F.close(); S.close(); O.close()


def test_config5_specular_veach_1280x720_frame_matches_oracle(G, B):
    """The scene `bench.py --config 5` renders -- veach_bidir(specular=True): glass egg, wall mirror, polished copper -- at config 5's resolution and
    1 of its 128 spp, every pixel of the ten buffers against the oracle: 31 % of its samples run the general form (round 5: k_bdg_shift /
    k_bdg_connect / k_bdg_light), ~110 000 manifold walks per frame.  (Round 4 held this scene at 64x36 only, VERDICT r4 weak #5.)"""
    W, H = 1280, 720
    sc = scenes.veach_bidir(W, H, specular=True)
    S, O = G.Scene(sc), go.Scene(sc)
    assert abs(S.bsphere_radius() - O.bsphere_radius()) <= 1e-12 * O.bsphere_radius()
    integ = B.GBDPTIntegrator(maxDepth=-1)
    F = B.Film(S)
    integ.renderBlock(S, F, integ.config(1), (0, 0, W, H))
    block, light = F.accum(); st = F.stats(); ch = F.chain_stats()
    ob, ol, oc = O.gbdpt_render(go.gbdpt_config(maxDepth=-1, spp=1))
    assert oc["unsupported"] == 0 and st["samples"] == W * H and ch["overflows"] == 0
    assert 0.2 * W * H < ch["generalSamples"] < 0.5 * W * H and oc["manifoldWalks"] > 50000
    assert abs(st["raysTraced"] - oc["raysTraced"]) <= 2e-6 * oc["raysTraced"] and abs(st["shadowRaysTraced"] - oc["shadowRaysTraced"]) <= 2e-6 * oc["shadowRaysTraced"]
    for b in range(5):
        assert np.abs(block[b] - ob[b]).max() <= 1e-9 * (np.abs(ob[b]).max() + 1e-300), ("block", b)
        assert np.abs(light[b] - ol[b]).max() <= 1e-9 * (np.abs(ol[b]).max() + 1e-300), ("light", b)
    F.close(); S.close(); O.close()


@pytest.mark.parametrize("seed,variant", [(9001, "random_connectable"), (9002, "random_connectable"), (9003, "random_connectable"), (9004, "random_connectable"),
                                          (9006, "random_connectable"), (9007, "random_connectable"),
                                          # round 4: materials from everything the path carries (smooth conductors, dielectrics, rough conductors on both sides of
                                          # shiftThreshold): samples with specular chains -- the general form (GBDPT_FUZZ_SPECULAR=1 tools/gpu_gbdpt_fuzz.py)
                                          (740001, "random"), (740002, "random"), (740003, "random"), (740004, "random"), (740006, "random"), (740007, "random")])
def test_fuzzed_connectable_scenes_match_oracle(G, B, seed, variant):
    """A few seeds of tools/gpu_gbdpt_fuzz.py (which ran 34 000 of them on the round-3 binary: 816 000 single samples and 34 000 films, no difference
    beyond 1.2e-12 of a sample's scale): the four free surfaces of the box draw connectable materials from the seed (diffuse, rough conductors of
    all three distributions, anisotropic, one- or two-sided), random depth / Russian-roulette depth / light image; single samples through the probe
    entry and one film through the wavefront kernels."""
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(12, 36)), int(rng.integers(8, 28))
    sc = scenes.cornell_box(W, H, variant, seed=seed)
    md = int(rng.choice([-1, 1, 2, 3, 5, 8, 12])); rr = int(rng.choice([1, 3, 5])); li = bool(rng.random() < 0.7)
    spp = int(rng.integers(1, 4))
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, rrDepth=rr, lightImage=li)
    cfg = integ.config(spp, 5489 + seed); ocfg = go.gbdpt_config(maxDepth=md, rrDepth=rr, lightImage=li, spp=spp, seed=5489 + seed)
    for _ in range(24):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g = integ.evaluate_sample(S, cfg, px, py, s); o = O.gbdpt_sample(ocfg, px, py, s)
        assert o["unsupported"] == 0
        scale = max(np.abs(o["primal"]).max(), np.abs(o["gradients"]).max(), 1e-300)
        assert np.abs(np.asarray(g["primal"]) - o["primal"]).max() <= 1e-9 * scale + 1e-13, (seed, px, py, s)
        assert np.abs(np.asarray(g["gradients"]) - o["gradients"]).max() <= 1e-9 * scale + 1e-13, (seed, px, py, s)
        gl, ol = np.asarray(g["light"]).reshape(-1, 6), np.asarray(o["light"]).reshape(-1, 6)
        assert gl.shape == ol.shape and (not len(ol) or (np.array_equal(gl[:, 2], ol[:, 2]) and np.abs(gl[:, 3:] - ol[:, 3:]).max() <= 1e-9 * max(np.abs(ol[:, 3:]).max(), 1e-300) + 1e-13))
    F = B.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H)); F.sync()
    blk, lgt = F.accum()
    F.close()
    oblk, olgt, _ = O.gbdpt_render(ocfg)
    assert np.abs(blk - oblk).max() <= 1e-9 * max(np.abs(oblk).max(), 1e-300)
    assert np.abs(lgt - olgt).max() <= 1e-9 * max(np.abs(olgt).max(), 1e-300)


def test_untextured_uv_mesh_normal_derivative(G, B):
    """TriMesh::getNormalDerivative reparameterizes dndu / dndv by the texture coordinates of ANY mesh that has them (trimesh.cpp:800-820), textured or not --
    the input of the manifold walk (manifold.cpp:101-122).  Spheres with interpolated normals, random texture coordinates, NO texture, glass and mirror BSDFs:
    the general form's samples against the oracle.  (Until round 6 the device only saw the coordinates of textured scenes: found by holding the intersection
    record to the reference's own src/tests/test_dgeom.cpp vectors, which include this derivative.)"""
    W, H, md = 40, 30, 7
    sc = scenes.cornell_box(W, H, "smooth")
    nt = sc.ntri
    rng = np.random.default_rng(23)
    sc.uvs = rng.uniform(-1.0, 2.0, (nt, 6)); sc.tri_has_uv = (rng.random(nt) < 0.85).astype(np.uint8)
    mats = list(sc.materials)
    glass, mirror = len(mats), len(mats) + 1
    mats += [scenes.dielectric(), scenes.conductor(**scenes.AL)]; sc.materials = mats
    tm = np.array(sc.tri_material, np.int32).copy()
    sph = np.nonzero(np.abs(np.asarray(sc.normals)).sum(1) > 0)[0]
    tm[sph[: len(sph) // 2]] = glass; tm[sph[len(sph) // 2:]] = mirror; sc.tri_material = tm
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, lightImage=True)
    cfg, ocfg = integ.config(64), go.gbdpt_config(maxDepth=md, lightImage=True, spp=64)
    general = walks = 0
    for _ in range(120):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        g = integ.evaluate_sample(S, cfg, px, py, s)
        o = O.gbdpt_sample(ocfg, px, py, s)
        compare_sample(g, o, ("uv-smooth-specular", px, py, s))
        general += g["general"]
    assert general > 20
    # and the coordinates must matter to the oracle: the same scene without them gives other samples somewhere
    plain = scenes.cornell_box(W, H, "smooth"); plain.materials, plain.tri_material = sc.materials, sc.tri_material
    P = go.Scene(plain)
    differs = 0
    rng = np.random.default_rng(23); rng.uniform(-1.0, 2.0, (nt, 6)); rng.random(nt)
    for _ in range(120):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
        a, b = O.gbdpt_sample(ocfg, px, py, s), P.gbdpt_sample(ocfg, px, py, s)
        differs += not np.allclose(a["gradients"], b["gradients"], rtol=1e-9, atol=0, equal_nan=True)
    assert differs > 0
    S.close(); O.close(); P.close()
