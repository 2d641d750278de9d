"""The C-ABI library loads and exports every symbol include/brdf_levmar.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "brdf_levmar.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^(?:int|void|double|float|long long|const char \*)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_symbols_are_exported():
    import brdf_amd
    declared = _declared_symbols()
    assert {"dlevmar_dif", "dlevmar_bc_dif", "BRDFFunc_hip", "brdf_hip_fit_dev", "brdf_hip_fit_batch_dev"} <= set(declared)
    lib = C.CDLL(brdf_amd.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/brdf_levmar.h but not exported"
    # and the Python binding table covers exactly the header
    from brdf_amd._lib import ABI
    assert sorted(ABI) == declared


def test_library_is_self_contained_hip_code():
    """the product must not link the CPU oracle (or the reference) in any form"""
    import subprocess
    import brdf_amd
    out = subprocess.run(["ldd", brdf_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "levmar" not in out
    assert "amdhip64" in out


def test_argument_errors_return_lm_error_without_a_gpu(capfd):
    """argument validation happens before any HIP call: m out of range, m != 3 for the BRDF models, n < m"""
    import brdf_amd
    from brdf_amd._lib import D, ExtraData, MODEL_FUNC, lib
    angles = np.zeros(30)
    x = np.zeros(10)
    p = np.array([0.5, 1.0, 1.0])
    ed = ExtraData(angles.ctypes.data_as(D), 1)

    @MODEL_FUNC
    def user_func(p_, hx_, m_, n_, adata_):  # a host callback the library has never been told about
        pass

    fptr = C.cast(user_func, C.c_void_p)
    # an unregistered callback is legal (generic path: host evaluates it); too many parameters is refused up front
    rc = lib.dlevmar_dif(fptr, np.zeros(17).ctypes.data_as(D), np.zeros(20).ctypes.data_as(D), 17, 20, 100, None, None, None, None, None)
    assert rc == -1 and "1 <= m <= 16" in brdf_amd.last_error()
    hip = C.cast(lib.BRDFFunc_hip, C.c_void_p)
    rc = lib.dlevmar_dif(hip, p.ctypes.data_as(D), x.ctypes.data_as(D), 4, 10, 100, None, None, None, None, C.byref(ed))
    assert rc == -1 and "exactly 3 parameters" in brdf_amd.last_error()
    rc = lib.dlevmar_bc_dif(hip, p.ctypes.data_as(D), x.ctypes.data_as(D), 3, 2, None, None, None, 100, None, None,
                            None, None, C.byref(ed))
    assert rc == -1 and "fewer measurements" in brdf_amd.last_error()
    assert lib.brdf_hip_register_model(fptr) == 0 and lib.brdf_hip_unregister_model(fptr) == 0
    assert lib.brdf_hip_unregister_model(fptr) == -1
    capfd.readouterr()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    import brdf_amd._lib as m
    monkeypatch.setenv("BRDF_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        importlib.reload(m)
    monkeypatch.delenv("BRDF_HIP_LIB")
    importlib.reload(m)


def test_covariance_helpers():
    """dlevmar_stddev / dlevmar_corcoef (misc_core.c:598-611) on the Meyer covariance printed by the reference's
    lmdemo: sigma_0 = sqrt(0.00483514)"""
    from brdf_amd._lib import D, lib
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "lmdemo_kat.json")))["kats"]
    covar = np.array([float.fromhex(s) for s in next(k for k in gold if k["problem"] == 4)["covar"]])
    assert lib.dlevmar_stddev(covar.ctypes.data_as(D), 3, 0) == np.sqrt(covar[0])
    assert lib.dlevmar_corcoef(covar.ctypes.data_as(D), 3, 0, 1) == covar[1] / np.sqrt(covar[0] * covar[4])


def test_integration_guide_only_names_declared_entry_points():
    """INTEGRATION.md is the maintainer's recipe: every library symbol it mentions must be declared in the header"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "brdf_levmar.h")).read()
    guide = open(os.path.join(root, "INTEGRATION.md")).read()
    names = set(re.findall(r"\b(brdf_hip_[a-z_]+|dlevmar_[a-z_]+|BRDF[A-Za-z]+_hip)\b", guide))
    names -= {"brdf_hip"}  # "-lbrdf_hip"
    assert len(names) >= 10
    missing = sorted(n for n in names if not re.search(r"\b" + re.escape(n) + r"\s*\(", header))
    assert not missing, missing


def test_lu_solver_matches_the_restated_reference_bit_for_bit():
    """dAx_eq_b_LU_noLapack (levmar.h:336, Axb_core.c:1140-1277) is a host-side scalar utility of the drop-in library:
    same Crout LU as the oracle's restatement (which the lmdemo known answers pin), for run-time m."""
    import ctypes as C
    import numpy as np
    from brdf_amd._lib import D, lib
    from tests import oracle_libs as L
    rng = np.random.default_rng(5)
    for m in (1, 2, 3, 5, 8, 12):
        for trial in range(20):
            A = rng.normal(size=(m, m))
            if trial == 3 and m > 1:
                A[0, 0] = 0.0  # forces a row exchange
            if trial == 4 and m > 2:
                A[:, 1] = A[:, 0]  # singular: the zero pivot becomes DBL_EPSILON, as in the reference
            B = rng.normal(size=m)
            x, x_ref = np.zeros(m), np.zeros(m)
            Ac, Bc = A.copy().reshape(-1), B.copy()
            r = lib.dAx_eq_b_LU_noLapack(Ac.ctypes.data_as(D), Bc.ctypes.data_as(D), x.ctypes.data_as(D), m)
            r_ref = L.orc.orc_lu_solve(L.ptr(A.copy().reshape(-1)), L.ptr(B.copy()), L.ptr(x_ref), m)
            assert r == r_ref == 1 and np.array_equal(x, x_ref, equal_nan=True)
            assert np.array_equal(Ac.reshape(m, m), A) and np.array_equal(Bc, B)  # inputs untouched
    Z = np.zeros(9)
    assert lib.dAx_eq_b_LU_noLapack(Z.ctypes.data_as(D), np.ones(3).ctypes.data_as(D), np.zeros(3).ctypes.data_as(D), 3) == 0  # zero row
    assert lib.dAx_eq_b_LU_noLapack(None, None, None, 0) == 1  # the reference's "free the retained buffer" call


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside a launcher starts two ranks itself (before anything touches a GPU) and relays
    rank 0's line.  Here over gloo with the fit stubbed (BRDF_BENCH_STUB: plumbing only, no GPU in this container): the
    gathered output of two ranks must equal the one-rank output (surfels sharded contiguously, one gather per step)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, BRDF_BENCH_BACKEND="gloo", BRDF_BENCH_STUB="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    lines = {}
    for gpus in (1, 2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--workload", "c5", "--surfels", "37",
                              "--samples", "16", "--steps", "2", "--warmup", "1", "--no-cpu"], capture_output=True, text=True, env=env,
                             cwd=ROOT, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        js = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(js) == 1 and len(out.stdout.strip().splitlines()) == 1, out.stdout  # exactly one line on stdout: rank 0's JSON
        lines[gpus] = json.loads(js[0])
    assert lines[1]["n_gpus"] == 1 and lines[2]["n_gpus"] == 2
    assert lines[2]["config"]["surfels"] == 37 and lines[2]["steps"] == 2
    assert lines[1]["stub_checksum"] == lines[2]["stub_checksum"]  # ragged shards (19 + 18), gathered in surfel order
    # under a launcher with the wrong world size the bench refuses instead of printing n_gpus: 1
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "c5", "--no-cpu"], capture_output=True,
                         text=True, env=dict(env, RANK="0", WORLD_SIZE="1"), cwd=ROOT, timeout=600)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
