"""Wireframe drawing (M_WIRE of M_render_polygon, EngBase.cpp:393-431; G_line,
GraphicsBase.cpp:165-253): the CPU restatement against digests of frames the reference
build drew, and against the reference build itself."""
import json
import os
import sys

import numpy as np
import pytest

import oracle.port as P
import oracle.ref as R
from raytracing_amd import scenes as S

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG  # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wire_kats.json")))
needs_ref = pytest.mark.skipif(not R.available(MG.WIRE_W, MG.WIRE_H), reason="reference build not present")


@pytest.mark.parametrize("name", sorted(GOLD))
def test_restatement_matches_golden_digests(name):
    c, z = MG.run_wire_case(MG.wire_cases()[name], P.Oracle(MG.WIRE_W, MG.WIRE_H))
    assert S.frame_digest(c) == GOLD[name]["colour"] and S.frame_digest(z) == GOLD[name]["depth"]
    assert int((c != c[0, 0]).sum()) == GOLD[name]["drawn"]


@needs_ref
@pytest.mark.parametrize("seed", range(4))
def test_wire_objects_match_reference_build(seed):
    """objects placed so that many polygons cross the near plane and the screen borders"""
    sc = S.gen_fuzz_polys(200, MG.WIRE_W, MG.WIRE_H, 7, 5100 + seed, render_type=S.M_FLAT)
    steps = [("clear", 0x010203), ("object", sc, S.M_WIRE, (seed * 9, 250 - seed, 7, 3, -2, 10 * seed))]
    a = MG.run_wire_case(steps, R.Ref(MG.WIRE_W, MG.WIRE_H))
    b = MG.run_wire_case(steps, P.Oracle(MG.WIRE_W, MG.WIRE_H))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert (a[0] != 0x010203).sum() > 500


@needs_ref
def test_lines_over_filled_polygons_match_reference_build():
    """unclipped lines are depth-tested against what the polygons left and leave the depth alone"""
    sc = S.gen_c2(1500, MG.WIRE_W, MG.WIRE_H, 7, radius=30, seed=15)
    steps = [("clear", 0), ("object", sc, S.M_SHADED, (0, 0, 0, 0, 0, 0)),
             ("lines", MG.wire_lines(31, 500, 1, -2, -7000, -50))]
    a = MG.run_wire_case(steps, R.Ref(MG.WIRE_W, MG.WIRE_H))
    b = MG.run_wire_case(steps, P.Oracle(MG.WIRE_W, MG.WIRE_H))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    before = MG.run_wire_case(steps[:2], P.Oracle(MG.WIRE_W, MG.WIRE_H))
    assert np.array_equal(before[1], b[1])                      # depth untouched by the lines
    changed = (before[0] != b[0]).sum()
    assert 1000 < changed < 500 * 200                           # some points pass, some are hidden
