"""Closed forms of the G-PT estimator, derived by hand from /root/reference/src/integrators/gpt/gpt.cpp on a scene where every quantity has one:
a finite diffuse floor (two triangles), one triangular area light above it, a perspective camera, paths of depth 2 (one light sample + one BSDF
sample per base path).  `model()` below is written against gpt.cpp directly -- it shares NO code with oracle/ or csrc/ -- and is asserted against
BOTH the oracle (CPU) and the HIP path (`-m gpu`), so a misreading common to those two (same author, same reading of the reference) would show:

  * light-sample term: MIS numerator `main.pdf * dRec.pdf`, denominator `main.pdf^2 (dRec.pdf^2 + bsdfPdf^2)` (gpt.cpp:599-600); per offset the SAME
    light point re-sampled from the offset vertex, Jacobian `|cos'_L d^2| / (Epsilon + |cos_L d'^2|)` with Epsilon (not D_EPSILON) (:695), weight
    `numerator / (D_EPSILON + shiftedDenominator + mainDenominator)` with `(J shifted.pdf)^2 (dRecPdf'^2 + bsdfPdf'^2)` (:698-699);
  * emitter hit by the BSDF sample: numerator `prevPdf * bsdfPdf`, denominator `prevPdf^2 (lumPdf^2 + bsdfPdf^2)` (:819-820); reconnection shift with
    `J = |cos' d^2| / (D_EPSILON + |cos d'^2|)` (:316-345), `shifted.throughput *= f' J`, `shifted.pdf *= p' J` (:939-940), and a shifted denominator
    WITHOUT the Jacobian (`shiftedPreviousPdf^2 (lumPdf'^2 + bsdfPdf'^2)`, :980);
  * a dead offset (its primary ray misses the floor): weight `numerator / (D_EPSILON + mainDenominator)`, shifted contribution 0 (:708-717,1131-1136);
  * every term is added to the base path once PER offset (`main.addRadiance` inside the offset loop, :723,1141), which is what the put
    factors of renderBlock assume (:1314-1352);
  * the random numbers: SplitMix64 streams keyed by (seed, pixel, sample), consumed as samplePos(2), lightSample(2), bsdfSample(2)."""
import math

import numpy as np
import pytest

from gradientdomain_mitsuba_amd import scenes

EPSILON, D_EPSILON = 1e-7, 1e-14                       # constants.h:25 (double build), gpt.cpp:63
RHO = np.array([0.6, 0.5, 0.4])
LE = np.array([9.0, 7.0, 5.0])
FLOOR = [np.array(p, float) for p in ((-2, 0, -2), (-2, 0, 2), (2, 0, 2), (2, 0, -2))]
LIGHT = [np.array(p, float) for p in ((-0.5, 2, -0.5), (0.5, 2, -0.5), (0.0, 2, 0.6))]
CAM_O, CAM_T, FOV, W, H = (0.0, 1.2, -3.5), (0.0, 0.0, 0.0), 40.0, 16, 12
SEED = 5489


def scene():
    verts = np.array([np.concatenate([FLOOR[0], FLOOR[1], FLOOR[2]]), np.concatenate([FLOOR[0], FLOOR[2], FLOOR[3]]), np.concatenate(LIGHT)])
    return scenes.Scene(verts, np.array([0, 0, 1], np.int32), [scenes.diffuse(tuple(RHO)), scenes.diffuse((0.5, 0.5, 0.5))], [(2, 1, tuple(LE))],
                        scenes.lookat(CAM_O, CAM_T, (0, 1, 0)), FOV, near=0.1, far=100.0, width=W, height=H, name="closed-form")


# ---- the counter-based generator (the specification shared by oracle and device; DESIGN.md "Stated deviation: random numbers") ----
M64 = (1 << 64) - 1


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def draws(pixel, sample, n):
    s = mix64((SEED + 0x9E3779B97F4A7C15 * (pixel + 1)) & M64)
    s = mix64(s ^ ((0xD1B54A32D192ED03 * (sample + 1)) & M64))
    out = []
    for _ in range(n):
        s = (s + 0x9E3779B97F4A7C15) & M64
        bits = (mix64(s) >> 12) | 0x3FF0000000000000
        out.append(np.frombuffer(np.uint64(bits).tobytes(), np.float64)[0] - 1.0)       # Random::nextFloat, double build (random.cpp)
    return out


# ---- geometry helpers -------------------------------------------------------------------------------------------------------------------
def camera_dir(px, py):
    """PerspectiveCamera::sampleRayDifferential (perspective.cpp:271-298): the ray through film position (px, py) in pixels."""
    o, t = np.array(CAM_O), np.array(CAM_T)
    d = (t - o) / np.linalg.norm(t - o)
    left = np.cross((0.0, 1.0, 0.0), d); left /= np.linalg.norm(left)
    up = np.cross(d, left)
    th = math.tan(math.radians(FOV) / 2)
    v = (1 - 2 * px / W) * th * left + (1 - 2 * py / H) * th / (W / H) * up + d
    return v / np.linalg.norm(v)


def hit_floor(o, d):
    if d[1] >= 0:
        return None
    p = o + d * (-o[1] / d[1])
    return p if (abs(p[0]) <= 2 and abs(p[2]) <= 2) else None


def floor_frame(p):
    """skdtree.h:367-397 for a flat triangle: n = face normal, s from dpdu = p1 - p0 (computeShadingFrame, util.cpp:603-608), t = n x s."""
    n = np.array([0.0, 1.0, 0.0])
    dpdu = (FLOOR[1] - FLOOR[0]) if p[2] >= p[0] else (FLOOR[2] - FLOOR[0])          # triangle (a, b, c) covers z >= x, (a, c, d) the rest
    s = dpdu - n * np.dot(n, dpdu); s /= np.linalg.norm(s)
    return s, np.cross(n, s), n


def hit_light(o, d):
    e1, e2 = LIGHT[1] - LIGHT[0], LIGHT[2] - LIGHT[0]
    pv = np.cross(d, e2); det = np.dot(e1, pv)
    tv = o - LIGHT[0]; u = np.dot(tv, pv) / det
    qv = np.cross(tv, e1); v = np.dot(d, qv) / det
    t = np.dot(e2, qv) / det
    return (o + d * t) if (u >= 0 and v >= 0 and u + v <= 1 and t > 1e-9) else None


LIGHT_N = np.cross(LIGHT[1] - LIGHT[0], LIGHT[2] - LIGHT[0]); LIGHT_AREA = 0.5 * np.linalg.norm(LIGHT_N); LIGHT_N = LIGHT_N / np.linalg.norm(LIGHT_N)


def cosine_hemisphere(u1, u2):
    """warp::squareToCosineHemisphere over the concentric disk (warp.cpp:43-52,81-102)."""
    r1, r2 = 2 * u1 - 1, 2 * u2 - 1
    if r1 == 0 and r2 == 0:
        r = phi = 0.0
    elif r1 * r1 > r2 * r2:
        r, phi = r1, (math.pi / 4) * (r2 / r1)
    else:
        r, phi = r2, (math.pi / 2) - (r1 / r2) * (math.pi / 4)
    x, y = r * math.cos(phi), r * math.sin(phi)
    return np.array([x, y, math.sqrt(max(0.0, 1 - x * x - y * y))])


def model(px, py, sample):
    """GradientPathTracer::evaluatePoint at maxDepth 2 on this scene, from gpt.cpp.  -> throughput[3], gradients[4,3], neighbours[4,3]"""
    u = draws(py * W + px, sample, 6)
    sx, sy = px + u[0], py + u[1]
    o = np.array(CAM_O)
    T, G, N = np.zeros(3), np.zeros((4, 3)), np.zeros((4, 3))
    x1 = hit_floor(o, camera_dir(sx, sy))
    if x1 is None:
        return T, G, N                                                                   # no environment: nothing is added (gpt.cpp:494-497)
    shifts = ((1, 0), (0, 1), (-1, 0), (0, -1))                                           # gpt.cpp:410-415
    xo = [hit_floor(o, camera_dir(sx + a, sy + b)) for a, b in shifts]                    # dead offsets: None (:508-513)
    n = np.array([0.0, 1.0, 0.0])
    main_thr, main_pdf = np.ones(3), 1.0
    off_thr, off_pdf = [np.ones(3) for _ in range(4)], [1.0] * 4
    # ---- light sample (gpt.cpp:565-730) ----
    a = math.sqrt(max(0.0, 1 - u[2]))                                                    # squareToUniformTriangle (warp.cpp:76-79) after two identity sampleReuse
    y = LIGHT[0] + (LIGHT[1] - LIGHT[0]) * (1 - a) + (LIGHT[2] - LIGHT[0]) * (a * u[3])

    def from_vertex(x):
        e = y - x; d2 = float(np.dot(e, e)); d = e / math.sqrt(d2)
        cos_l = float(np.dot(LIGHT_N, -d))                                               # "opposing cosine" of :596 / :684
        pdf_light = (1.0 / LIGHT_AREA) * d2 / abs(cos_l)                                  # Shape::sampleDirect, shape.cpp:102-116
        cos_s = float(np.dot(n, d))
        ok = cos_s >= 0 and np.dot(d, LIGHT_N) < 0                                        # AreaLight::sampleDirect's acceptance, area.cpp:166
        return d2, cos_l, (pdf_light if ok else 0.0), (RHO / math.pi * cos_s if cos_s > 0 else np.zeros(3)), (cos_s / math.pi if cos_s > 0 else 0.0), (LE if ok else np.zeros(3))

    d2, cos_l, p_l, f, p_b, le = from_vertex(x1)
    num, den = main_pdf * p_l, main_pdf ** 2 * (p_l ** 2 + p_b ** 2)
    main_c = main_thr * (f * le)
    for i in range(4):
        if xo[i] is None:
            w = num / (D_EPSILON + den); sh_c = np.zeros(3)                               # :708-717
        else:
            d2s, cos_ls, p_ls, fs, p_bs, les = from_vertex(xo[i])
            J = abs(cos_ls * d2) / (EPSILON + abs(cos_l * d2s))                           # :695
            sden = (J * off_pdf[i]) ** 2 * (p_ls ** 2 + p_bs ** 2)                        # :698
            w = num / (D_EPSILON + sden + den)
            sh_c = J * off_thr[i] * (fs * les)
        T += w * main_c; N[i] += w * sh_c; G[i] += w * (sh_c - main_c)                    # :723-726
    # ---- BSDF sample (gpt.cpp:737-1146) ----
    s, t, _ = floor_frame(x1)
    wl = cosine_hemisphere(u[4], u[5])
    wo = s * wl[0] + t * wl[1] + n * wl[2]
    bsdf_pdf, weight = wl[2] / math.pi, RHO                                               # diffuse.cpp:141-151
    x2 = hit_light(x1, wo)
    if x2 is None or bsdf_pdf <= 0:
        return T, G, N                                                                   # nothing hit, no environment (:800-803)
    prev_pdf = main_pdf
    main_thr = main_thr * (weight * bsdf_pdf); main_pdf *= bsdf_pdf                       # :810-811
    e = x2 - x1; dist2 = float(np.dot(e, e))
    lum_pdf = (1.0 / LIGHT_AREA) * dist2 / abs(float(np.dot(wo, LIGHT_N)))                # pdfEmitterDirect, scene.cpp:976-979
    num, den = prev_pdf * bsdf_pdf, prev_pdf ** 2 * (lum_pdf ** 2 + bsdf_pdf ** 2)         # :819-820
    main_c = main_thr * LE
    for i in range(4):
        if xo[i] is None:
            w = num / (D_EPSILON + den); sh_c = np.zeros(3)                               # :1131-1136
        else:
            me, se = x1 - x2, xo[i] - x2                                                  # reconnectShift, :316-345
            swo = -se / math.sqrt(np.dot(se, se))
            J = abs(float(np.dot(swo, LIGHT_N)) * np.dot(me, me)) / (D_EPSILON + abs(float(np.dot(me, LIGHT_N)) / math.sqrt(np.dot(me, me)) * np.dot(se, se)))
            cos_o = float(np.dot(n, swo))
            fs, ps = RHO / math.pi * cos_o, cos_o / math.pi
            thr = off_thr[i] * fs * J                                                     # :939-940
            lum_s = (1.0 / LIGHT_AREA) * float(np.dot(se, se)) / abs(float(np.dot(swo, LIGHT_N)))   # :956-966
            sden = off_pdf[i] ** 2 * (lum_s ** 2 + ps ** 2)                               # :980 (no Jacobian)
            w = num / (D_EPSILON + sden + den)
            sh_c = thr * LE
        T += w * main_c; N[i] += w * sh_c; G[i] += w * (sh_c - main_c)                    # :1141-1146
    return T, G, N


def cases():
    """(px, py, sample) covering: BSDF sample misses / hits the light, a dead offset at the floor's far edge, a base path that misses."""
    out, kinds = [], set()
    o = np.array(CAM_O)
    for py in range(H):
        for px in range(W):
            for s in range(6):
                u = draws(py * W + px, s, 6)
                sx, sy = px + u[0], py + u[1]
                x1 = hit_floor(o, camera_dir(sx, sy))
                dead = sum(hit_floor(o, camera_dir(sx + a, sy + b)) is None for a, b in ((1, 0), (0, 1), (-1, 0), (0, -1)))
                if x1 is None:
                    kind = "miss"
                else:
                    st, tt, nn = floor_frame(x1); wl = cosine_hemisphere(u[4], u[5])
                    hit = hit_light(x1, st * wl[0] + tt * wl[1] + nn * wl[2]) is not None
                    kind = ("hit" if hit else "nohit") + ("-dead%d" % dead if dead else "")
                if sum(k == kind for k, _ in out) < 3:
                    out.append((kind, (px, py, s))); kinds.add(kind)
    assert {"miss", "hit", "nohit"} <= kinds and any("dead" in k for k in kinds), kinds
    return out


CASES = cases()


def check(evaluate_point, cfg):
    assert len(CASES) >= 6
    for kind, (px, py, s) in CASES:
        T, G, N = model(px, py, s)
        r = evaluate_point(cfg, px, py, s)
        assert not np.asarray(r["veryDirect"]).any(), kind
        for got, want, what in ((r["throughput"], T, "throughput"), (r["gradients"], G, "gradients"), (r["neighbours"], N, "neighbours")):
            assert np.allclose(got, want, rtol=1e-9, atol=1e-13), (kind, px, py, s, what, got, want)
        if kind.startswith("hit"):
            assert T.max() > 0 and np.abs(G).max() > 0


def test_closed_forms_against_the_oracle():
    from oracle import gpt_oracle as go
    O = go.Scene(scene())
    check(O.evaluate_point, go.config(maxDepth=2, spp=6))


@pytest.mark.gpu
def test_closed_forms_against_the_hip_path(gpu_required):
    import gradientdomain_mitsuba_amd.gpt as G
    S = G.Scene(scene())
    check(S.evaluate_point, G.GradientPathIntegrator(maxDepth=2).config(6))
    S.close()


@pytest.mark.gpu
def test_device_and_serial_sfmt_order_agree_statistically(gpu_required):
    """The HIP path draws counter-based streams; the reference draws ONE serial SFMT-19937 stream in spiral-block x Hilbert order (oracle:
    render_serial, pinned to the reference's own generator table).  The two cannot agree sample for sample; their ESTIMATES must: the
    means of the five developed buffers over a 64x48 Cornell frame at 1024 spp agree within 4 sigma, sigma from eight independent device
    renders of 128 spp each."""
    import gradientdomain_mitsuba_amd.gpt as G
    from oracle import gpt_oracle as go
    Wd, Hd, spp = 64, 48, 1024
    sc = scenes.cornell_box(Wd, Hd, "diffuse")
    S = G.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=6, reconstructL1=False, reconstructL2=False)
    parts = []
    for k in range(8):
        out = integ.render(S, spp // 8, seed=1000 + k)
        parts.append(np.array([out[n].astype(np.float64).mean(axis=(0, 1)) for n in G.BUFFER_NAMES[1:]]))
    parts = np.array(parts)                                             # [8, 4 buffers, 3]
    mean, sigma = parts.mean(axis=0), parts.std(axis=0, ddof=1) / math.sqrt(8)
    acc, _ = go.Scene(sc).render_serial(go.config(maxDepth=6, spp=spp))
    ser = go.develop(acc)[1:].mean(axis=(1, 2))                          # throughput, dx, dy, direct
    # the serial estimate has the same variance as the device's pooled estimate: compare with sqrt(2) sigma
    z = np.abs(mean - ser) / (math.sqrt(2.0) * sigma + 1e-12)
    assert (z < 4.0).all(), (z, mean, ser)
    assert np.abs(mean[0] - ser[0]).max() < 0.02 * ser[0].max()          # and the throughput means within 2 % outright
    S.close()
