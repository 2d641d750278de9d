"""Regenerates tests/golden/*.json from the REFERENCE itself (oracle/_ref = levmar 2.6 compiled from
/root/reference/levmar in this container) -- run here, where the reference exists; the JSON travels.

  brdf_fits.json   p[3] + info[10] + return value of the reference's dlevmar_dif / dlevmar_bc_dif / dlevmar_bc_der driving
                   the BRDF callback on the seeded synthetic sets of brdf_amd/synth.py (inputs are
                   regenerated from the seed, not stored).  This pins the fits no reference-owned test pins.
  lmdemo_kat.json  the reference's own known answers: lmdemo.c problems run through the compiled
                   reference (the same numbers as SURVEY.md section 4, here with full precision).
  slevmar_kat.json the single-precision twins: the compiled reference's slevmar_* on float test problems (ours, ref_shim.c)
  model_values.json 64 model values per BRDF model.  Phong and Blinn-Phong: from `ref_BRDFFunc`, the REFERENCE's own
                   BRDFFunc compiled from its text (brdfdata.cpp:962-989, cut out in place by oracle/Makefile's `ref`
                   target; only OpenCV's CV_PI literal is supplied) -- this is what pins the model values.  Ward is
                   build-defined: from the restated callback.
  The BRDF fits of models 0 / 1 are driven by that same `ref_BRDFFunc` (oracle/ref_shim.c).
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from brdf_amd import synth  # noqa: E402
from tests import oracle_libs as L  # noqa: E402
from tests.kat_problems import PROBLEMS, OPTS, SOPTS, SPROBLEMS, run_problem, run_sproblem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    assert L.ref is not None, "build oracle/_ref first (make -C oracle)"
    fits = []
    for n in (16, 64, 341, 342, 1000, 10000):
        for model in (0, 1, 2):
            angles, x, _ = synth.make_single(model, n)
            for method in (0, 1, 2, 3):  # 2 / 3 = the reference's dlevmar_bc_der / dlevmar_der driven by the analytic Jacobian (f3)
                r, p, info = L.brdf_fit("ref", method, model, angles, x, synth.P0[model], synth.ITMAX, synth.OPTS,
                                        synth.LB, synth.UB)
                fits.append({"n": n, "model": model, "method": method, "ret": int(r), "p": [float.hex(v) for v in p],
                             "info": [float.hex(v) for v in info]})
    json.dump({"generator": "brdf_amd.synth.make_single(model, n), seed %d" % synth.SEED, "fits": fits},
              open(os.path.join(HERE, "brdf_fits.json"), "w"), indent=1)

    kats = []
    for pid, pr in PROBLEMS.items():
        r, p, info, covar = run_problem(L.ref, "", pr)
        kats.append({"problem": pid, "name": pr["f"], "entry": pr["kind"], "ret": int(r),
                     "p": [float.hex(v) for v in p], "info": [float.hex(v) for v in info],
                     "covar": None if covar is None else [float.hex(v) for v in covar]})
    json.dump({"opts": list(OPTS), "kats": kats}, open(os.path.join(HERE, "lmdemo_kat.json"), "w"), indent=1)

    skats = []  # single precision: the reference's slevmar_* on the float problems of oracle/ref_shim.c
    for name, pr in SPROBLEMS.items():
        r, p, info, covar = run_sproblem(L.ref, pr)
        skats.append({"name": name, "entry": pr["kind"], "ret": int(r), "p": [float.hex(float(v)) for v in p],
                      "info": [float.hex(float(v)) for v in info], "covar": None if covar is None else [float.hex(float(v)) for v in covar]})
    json.dump({"opts": list(SOPTS), "kats": skats}, open(os.path.join(HERE, "slevmar_kat.json"), "w"), indent=1)

    vals = []
    for model in (0, 1, 2):
        angles, _, _ = synth.make_single(model, 64)
        for p in (synth.P0[model], synth.TRUTH[model]):
            hx = L.ref_model_values(model, angles, p)  # models 0 / 1: ref_BRDFFunc; Ward: the restatement (not in the reference)
            vals.append({"model": model, "p": list(p), "source": "ref_BRDFFunc (brdfdata.cpp:962-989 compiled in place)" if model < 2
                         else "oracle/brdf_models_oracle.c (Ward is build-defined)", "hx": [float.hex(v) for v in hx]})
    json.dump({"n": 64, "values": vals}, open(os.path.join(HERE, "model_values.json"), "w"), indent=1)
    print("wrote", len(fits), "fits,", len(kats), "kats,", len(vals), "model value sets")


if __name__ == "__main__":
    main()
