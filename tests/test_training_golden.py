"""CPU: the training-surface helpers against vectors generated from the reference's own Python (tests/golden/make_golden.py):
optimisation defaults [REF arguments/__init__.py:72-99], the learning-rate schedule [REF utils/general_utils.py:29-62] in the five
configurations training_setup builds [REF scene/gaussian_model.py:394-491], inverse_sigmoid, build_rotation."""
import os

import numpy as np
import torch

from gaussianprediction_amd import training

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_optimisation_defaults_are_the_references():
    g = np.load(os.path.join(G, "training.npz"))
    ref = dict(zip([str(n) for n in g["default_names"]], g["default_values"]))
    mine = vars(training.default_training_args())
    assert set(mine) <= set(ref), set(mine) - set(ref)
    for k, v in mine.items():
        assert float(v) == ref[k], (k, v, ref[k])
    assert len(mine) >= 23


def test_learning_rate_schedules_match_the_reference():
    g = np.load(os.path.join(G, "training.npz"))
    steps = g["steps"]
    for name in ("xyz", "mlp", "hash", "mfeature", "plain", "off"):
        lr_init, lr_final, delay_steps, delay_mult, max_steps = [float(v) for v in g[name + "_kw"]]
        f = training.get_expon_lr_func(lr_init, lr_final, lr_delay_steps=int(delay_steps), lr_delay_mult=delay_mult, max_steps=int(max_steps))
        got = np.array([f(int(t)) for t in steps])
        np.testing.assert_allclose(got, g[name], rtol=1e-13, atol=0.0, err_msg=name)


def test_build_rotation_and_inverse_sigmoid_match_the_reference():
    c = np.load(os.path.join(G, "cov3d.npz"))
    R = training.build_rotation(torch.tensor(c["quats"]))
    np.testing.assert_allclose(R.numpy(), c["R"], rtol=0, atol=2e-7)
    g = np.load(os.path.join(G, "training.npz"))
    x = torch.tensor(g["inv_sig_x"])
    np.testing.assert_allclose(torch.log(x / (1 - x)).numpy(), g["inv_sig_y"], rtol=1e-6)
    if hasattr(training, "inverse_sigmoid"):
        np.testing.assert_allclose(training.inverse_sigmoid(x).numpy(), g["inv_sig_y"], rtol=1e-6)
