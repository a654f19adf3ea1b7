"""Extracts the golden vectors the reference's own differential-geometry test holds for triangle meshes --
/root/reference/src/tests/test_dgeom.cpp:35-178 (TestDGeom::test01_trimesh_1, test02_trimesh_2, test03_trimesh_3: a one-triangle TriMesh, one ray
through ShapeKDTree::rayIntersect, assertions on its.p / uv / geoFrame.n / shFrame.n / shFrame.s / dpdu / dpdv / time and on
Shape::getNormalDerivative) -- into tests/golden/dgeom_reference.json.  That is ShapeKDTree::rayIntersect + TriAccel + fillIntersectionRecord<true>
 and TriMesh::getNormalDerivative (used by the G-BDPT manifold walk, row f1).

DATA only: the meshes' inputs, the ray, and every expected value with the test's own tolerance (assertEquals = exact, assertEqualsEpsilon(..., eps) =
per component |actual - expected| <= eps, testcase.cpp:92-100; Epsilon = 1e-7 in the double-precision build, constants.h:25).  An expected value the test
writes as an expression of the INPUTS (normalize(normals[0]*.7f + ...), vertices[1]-vertices[0]) is evaluated here, with the test's float literals
(0.1f = the double nearest the float); one that it writes as a relation between OUTPUTS (shFrame.s against dpdu and shFrame.n) is kept as a relation.
Runs in the build container, where /root/reference exists; the JSON travels to the GPU box, the reference does not.

    python tests/golden/make_dgeom_golden.py
"""
import json
import os
import re

import numpy as np

SRC = "/root/reference/src/tests/test_dgeom.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dgeom_reference.json")
EPSILON = 1e-7          # include/mitsuba/core/constants.h:25 (DOUBLE_PRECISION build: Float = double)


def F(s):
    return float(np.float32(float(s)))


def fix_literals(expr):
    expr = re.sub(r"(?<![\w.])((?:\d*\.\d+|\d+\.?\d*)(?:e[+-]?\d+)?)f\b", lambda m: 'F("%s")' % m.group(1), expr)
    return expr


def vec(*a):
    return np.array([float(x) for x in a], np.float64)


def normalize(v):
    return v / np.sqrt(np.dot(v, v))


def evaluate(expr, env):
    return eval(fix_literals(expr), {"__builtins__": {}}, dict(env, F=F, Point=vec, Point2=vec, Vector=lambda *a: vec(*a) if len(a) > 1 else vec(a[0], a[0], a[0]),
                                                          Normal=vec, normalize=normalize, Epsilon=EPSILON))


def main():
    text = open(SRC).read()
    lines = text.split("\n")
    cases = []
    for name in ("test01_trimesh_1", "test02_trimesh_2", "test03_trimesh_3"):
        a = text.index("void %s() {" % name)
        b = text.index("\n\t}\n", a)
        body = text[a:b]
        first_line = text[:a].count("\n") + 1
        last_line = text[:b].count("\n") + 2
        env = {"vertices": [None] * 3, "normals": [None] * 3, "uv": [None] * 3}
        for arr in ("vertices", "normals", "uv"):
            for m in re.finditer(r"\b%s\[(\d)\] = (\w+\([^;]*\));" % arr, body):
                env[arr][int(m.group(1))] = evaluate(m.group(2), {})
        has_normals, has_uv = env["normals"][0] is not None, env["uv"][0] is not None
        m = re.search(r"Ray ray\((Point\([^)]*\)), (Vector\([^)]*\)), ([^)]*)\);", body)
        ray_o, ray_d, ray_time = evaluate(m.group(1), {}), evaluate(m.group(2), {}), evaluate(m.group(3), {})
        expect = []
        # getNormalDerivative(its, dndu, dndv, shadingFrame): the assertions that follow each call belong to it
        mode = None
        for st in re.sub(r"//[^\n]*", "", body).split(";"):     # statement by statement (some assertions span two lines)
            s = re.sub(r"\s+", " ", st).strip() + ";"
            m = re.match(r"its\.shape->getNormalDerivative\(its, dndu, dndv, (true|false)\);", s)
            if m:
                mode = m.group(1)
                continue
            for m in re.finditer(r"assert(Equals|EqualsEpsilon)\(([\w.]+), (.*?)(?:, ([\w.\-]+))?\);", s):
                kind, field, expr, eps = m.group(1), m.group(2), m.group(3), m.group(4)
                if kind == "Equals" and eps is not None:          # (a two-argument constructor swallowed by the optional group)
                    expr, eps = expr + ", " + eps, None
                tol = 0.0 if kind == "Equals" else float(evaluate(eps, {}))
                field = field[4:] if field.startswith("its.") else field
                if field in ("dndu", "dndv"):
                    field = "%s[shadingFrame=%s]" % (field, mode)
                if "its." in expr:                                  # a relation between outputs
                    expect.append({"field": field, "relation": re.sub(r"\s+", " ", expr), "eps": tol})
                else:
                    val = evaluate(expr, env)
                    expect.append({"field": field, "expected": [float(x) for x in np.atleast_1d(val)], "eps": tol})
        cases.append({
            "name": name, "source": "src/tests/test_dgeom.cpp:%d-%d" % (first_line, last_line),
            "vertices": [[float(x) for x in v] for v in env["vertices"]],
            "normals": [[float(x) for x in v] for v in env["normals"]] if has_normals else None,
            "texcoords": [[float(x) for x in v] for v in env["uv"]] if has_uv else None,
            "explicit_computeUVTangents": "trimesh->computeUVTangents();" in body,
            "ray": {"o": [float(x) for x in ray_o], "d": [float(x) for x in ray_d], "time": float(ray_time)},
            "hit": "assertTrue(kdtree->rayIntersect(ray, its));" in body,
            "expect": expect})
    json.dump({"source": "src/tests/test_dgeom.cpp:35-178 (TestDGeom: test01_trimesh_1, test02_trimesh_2, test03_trimesh_3)",
               "precision": "double (Float = double, Epsilon = 1e-7); the test's float literals are the doubles nearest those floats",
               "tolerance": "per component |actual - expected| <= eps (testcase.cpp:92-100); eps 0 = assertEquals",
               "cases": cases}, open(OUT, "w"), indent=1)
    print("wrote", OUT, [(c["name"], len(c["expect"])) for c in cases])


if __name__ == "__main__":
    main()
