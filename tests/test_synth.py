import numpy as np

from brdf_amd import synth


def test_counter_stream_is_shardable_and_reproducible():
    a, x, t = synth.make_surfels(1, 64, first=0, count=8)
    a2, x2, t2 = synth.make_surfels(1, 64, first=3, count=2)
    assert np.array_equal(a[3:5], a2) and np.array_equal(x[3:5], x2) and np.array_equal(t[3:5], t2)
    assert a.min() >= 0.05 and a.max() <= 1.0
    assert abs(float(np.mean(a)) - 0.525) < 0.02


def test_single_material_truth_and_noise():
    for model in (0, 1, 2):
        a, x, t = synth.make_single(model, 1000)
        f = synth.model_value(model, t, a[0], a[1], a[2])
        assert np.all(np.abs(x - f) <= 0.005 + 1e-15)
