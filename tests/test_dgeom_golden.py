"""SURVEY 8a rows 20, 22, 23 (ShapeKDTree::rayIntersect, TriAccel, fillIntersectionRecord<true>) and TriMesh::getNormalDerivative (row f1's manifold
walk) held to the vectors the REFERENCE'S OWN test holds: tests/golden/dgeom_reference.json <- src/tests/test_dgeom.cpp:35-178 (made by
tests/golden/make_dgeom_golden.py).  The oracle here on the CPU; the HIP path in test_gpt_gpu.py::test_intersection_record_matches_reference_dgeom_vectors.

What these vectors pin: the position, texture coordinates, geometric and shading normal, shading tangent, dpdu / dpdv of a hit, and the normal derivative
-- for one axis-aligned triangle and one ray.  What they do not: anything of rows 5-12 (the shift mappings) or 32-37 (the solver)."""
import json
import os

import numpy as np
import pytest

from gradientdomain_mitsuba_amd import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dgeom_reference.json")
# Not checked, with the reason:
#  time                      ray.time is not carried (static scenes: no animated transform reads it)
#  dnd*[shadingFrame=false]  the G-BDPT manifold walk only ever asks with shadingFrame = true (manifold.cpp:101-122); false is the constant 0
SKIPPED = ("time", "dndu[shadingFrame=false]", "dndv[shadingFrame=false]")
# TWO assertions of the reference's test cannot hold against this fork's own code, both in test02 ("UV coords, but no explicit parameterization"), both the
# v-derivative, both by the same factor 1 / (0.9f - 0.1f) = the v-extent of test02's texture coordinates:
#  dpdv   TriMesh::configure() of gradientdomain-mitsuba calls computeUVTangents() unconditionally (src/librender/trimesh.cpp:383-385, "For manifold
#         exploration: always compute UV tangents"), so a mesh WITH texture coordinates gets dpdv = (-dUV2.x dP1 + dUV1.x dP2) / det (:729-731)
#         = (p2 - p0) / 0.8, not the (p2 - p0) that test02 (written for upstream Mitsuba, whose configure() computes tangents only for anisotropic
#         BSDFs) still asserts;
#  dndv   TriMesh::getNormalDerivative converts to the texture parameterization whenever the mesh has texture coordinates (:800-820, `if (m_texcoords)`),
#         which divides test02's "from mathematica" dndv (taken in the triangle's own parameterization) by the same 0.8.
# dpdu / dndu of the same case are unchanged by that parameterization (u spans exactly 1).  The restatements follow the fork's CODE there: the expected
# value is the test's vector times that factor (derived by hand from the lines cited; test03, whose coordinates span 1 in u and v, holds unscaled).
V_EXTENT_TEST02 = float(np.float32(0.9)) - float(np.float32(0.1))
STALE_IN_FORK = {("test02_trimesh_2", "dpdv"): 1.0 / V_EXTENT_TEST02, ("test02_trimesh_2", "dndv[shadingFrame=true]"): 1.0 / V_EXTENT_TEST02}


def load_cases():
    return json.load(open(GOLDEN))["cases"]


def scene_of(case, with_emitter=False):
    """The one-triangle mesh of a test case as a scene description (the camera and the material play no part in an intersection record).  with_emitter: a
    second, flat, untextured triangle far off the ray carries an area light -- the HIP scene constructor asks for at least one emitter."""
    tris = [np.asarray(case["vertices"], np.float64).reshape(9)]
    if with_emitter:
        tris.append(np.array([100.0, 100.0, 5.0, 101.0, 100.0, 5.0, 100.0, 101.0, 5.0]))
    n = len(tris)
    sc = scenes.Scene(verts=np.asarray(tris).reshape(n, 9), tri_material=np.zeros(n, np.int32), materials=[scenes.diffuse((0.5, 0.5, 0.5))],
                      emitters=[(1, 1, (1.0, 1.0, 1.0))] if with_emitter else [],
                      to_world=scenes.lookat((0.3, 0.3, -3.0), (0.3, 0.3, 0.0), (0, 1, 0)), fov_x=40.0, near=0.1, far=100.0, width=8, height=8, name=case["name"])
    if case["normals"] is not None:
        sc.normals = np.zeros((n, 9), np.float64)
        sc.normals[0] = np.asarray(case["normals"], np.float64).reshape(9)
    if case["texcoords"] is not None:
        sc.uvs = np.zeros((n, 6), np.float64)
        sc.uvs[0] = np.asarray(case["texcoords"], np.float64).reshape(6)
        sc.tri_has_uv = np.array([1] + [0] * (n - 1), np.uint8)
    return sc


def check_record(case, rec, dn):
    """rec: field -> array (p, uv, geoFrame.n, shFrame.n, shFrame.s, dpdu, dpdv); dn: (dndu, dndv) with shadingFrame = true.  The test's own tolerances."""
    assert case["hit"] and rec is not None
    checked = 0
    for e in case["expect"]:
        f = e["field"]
        if f in SKIPPED:
            continue
        if f.startswith("dnd"):
            actual = dn[0] if f.startswith("dndu") else dn[1]
        else:
            actual = rec[f]
        if "relation" in e:
            assert e["relation"] == "normalize(its.dpdu - its.shFrame.n * dot(its.dpdu, its.shFrame.n))"
            n, dpdu = rec["shFrame.n"], rec["dpdu"]
            w = dpdu - n * np.dot(dpdu, n)
            expected = w / np.sqrt(np.dot(w, w))
        else:
            expected = np.asarray(e["expected"]) * STALE_IN_FORK.get((case["name"], f), 1.0)
        assert np.all(np.abs(np.asarray(actual) - expected) <= e["eps"]), (case["name"], f, actual, expected, e["eps"])
        checked += 1
    return checked


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c["name"])
def test_oracle_intersection_record_matches_reference_dgeom_vectors(case):
    from oracle import gpt_oracle as go
    for with_emitter in (False, True):
        S = go.Scene(scene_of(case, with_emitter))
        rec = S.intersect_record(case["ray"]["o"], case["ray"]["d"])
        dn = S.normal_derivative(case["ray"]["o"], case["ray"]["d"])
        n = check_record(case, rec, dn)
        assert n >= 8
        assert rec["t"] == 1.0 and rec["prim"] == 0


def test_golden_file_is_what_the_extractor_writes():
    """(in the build container only: the reference is not on the GPU box)"""
    if not os.path.exists("/root/reference/src/tests/test_dgeom.cpp"):
        pytest.skip("/root/reference is not here")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_dgeom_golden", os.path.join(os.path.dirname(GOLDEN), "make_dgeom_golden.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    saved = open(GOLDEN).read()
    try:
        m.main()
        assert open(GOLDEN).read() == saved
    finally:
        open(GOLDEN, "w").write(saved)
