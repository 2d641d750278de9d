"""Pins the CPU oracle: (1) against the reference's own known answers (lmdemo.c problems; committed fixture
generated from the compiled reference, cross-checked with the printed table of SURVEY.md section 4), and
(2) bit-for-bit against oracle/_ref when that library is present."""
import json
import os

import numpy as np
import pytest

from tests import oracle_libs as L
from tests.kat_problems import PROBLEMS, SURVEY_TABLE, run_problem

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "lmdemo_kat.json")))
pytestmark = pytest.mark.skipif(L.ref is None, reason="lmdemo problem functions live in oracle/_ref (not built)")


def _hex(v):
    return np.array([float.fromhex(s) for s in v])


@pytest.mark.parametrize("kat", GOLD["kats"], ids=lambda k: f"{k['problem']}-{k['name']}-{k['entry']}")
def test_oracle_matches_reference_fixture(kat):
    pr = PROBLEMS[kat["problem"]]
    r, p, info, covar = run_problem(L.orc, "orc_", pr, ref_lib=L.ref)
    assert r == kat["ret"]
    assert np.array_equal(p, _hex(kat["p"]))
    assert np.array_equal(info, _hex(kat["info"]))
    if kat["covar"] is not None:
        assert np.array_equal(covar, _hex(kat["covar"]))


@pytest.mark.parametrize("pid", sorted(SURVEY_TABLE))
def test_fixture_matches_lmdemo_printout(pid):
    """the committed fixture reproduces what the reference's lmdemo prints"""
    kat = next(k for k in GOLD["kats"] if k["problem"] == pid)
    printed, tail = SURVEY_TABLE[pid]
    assert " ".join("%.7g" % v for v in _hex(kat["p"])) == printed
    assert tuple(int(v) for v in _hex(kat["info"])[5:10]) == tail


def test_meyer_covariance_and_info():
    """Meyer (dlevmar_dif with caller work + covar, lmdemo.c:905-913): info[0..4] and covar row 0 as printed"""
    kat = next(k for k in GOLD["kats"] if k["problem"] == 4)
    info, covar = _hex(kat["info"]), _hex(kat["covar"])
    assert ["%g" % v for v in info[:5]] == ["1308.25", "8.79459e-05", "1.0794e-07", "9.26027e-34", "67472.7"]
    assert ["%g" % v for v in covar[:3]] == ["0.00483514", "-0.00162445", "-0.000548114"]


@pytest.mark.parametrize("pid", sorted(PROBLEMS))
def test_oracle_bit_exact_vs_compiled_reference(pid):
    pr = PROBLEMS[pid]
    a = run_problem(L.ref, "", pr)
    b = run_problem(L.orc, "orc_", pr, ref_lib=L.ref)
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    if a[3] is not None:
        assert np.array_equal(a[3], b[3])
