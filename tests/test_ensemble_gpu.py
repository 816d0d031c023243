"""End-to-end ensembles on the device (the reference's product: `model accuracy : ...` per member and
`ensemble accuracy : ...`, /root/reference/DDL.py:168-174; transfer generation /root/reference/DDL_Normal3.py:240-248).

Full-schedule runs of the three workflows are kept under profiles/ensemble_e2e/ (tools/r3_e2e.sh); these tests pin
the same properties on a small configuration: the top-k vote is no worse than its best member, transfer children
start from their parent's weights and end near its accuracy, and device members track the fp32 CPU oracle."""
import numpy as np
import pytest
import torch

from distributeddeeplearning_amd import config
from distributeddeeplearning_amd.parallel.ensemble import EnsembleModel

pytestmark = pytest.mark.gpu
DATA = {"synthetic": True, "n_train": 2400, "n_test": 400}


def _run(device, overrides, preset="ddl_normal3"):
    sp = config.get_space(preset, overrides)
    ens = EnsembleModel(sp, mode="local", device=device, data_kw=DATA, verbose=False)
    ens.make_and_train_new_models()
    n_random = len(ens.accuracy_with_models)
    if sp.transfer_model_size:
        ens.make_and_train_transfer_models()
    val = {e[1]._a.model.number: float(e[0]) for e in ens.accuracy_with_models}
    test = ens.member_test_accuracies()                 # in pool order
    numbers = [e[1]._a.model.number for e in ens.accuracy_with_models]
    vote = ens.get_accuracy()
    topk = [e[1]._a.model.number for e in ens.accuracy_with_models[:sp.predict_model_size]]
    return ens, n_random, val, dict(zip(numbers, test)), vote, topk


def test_vote_is_no_worse_than_its_best_member_and_transfer_children_keep_up():
    ov = ["data_train_size=2000", "random_model_size=4", "transfer_model_size=2", "predict_model_size=3",
          "epochs=8", "seed=5", "conv2d_size=1,3", "dense_size=0,2", "kernel_size=2,6", "optimizers=Adam",
          "lr_exponent=-3,-2"]
    ens, n_random, val, test, vote, topk = _run("cuda", ov)
    assert len(val) == 6 and n_random == 4
    assert all(np.isfinite(v) for v in val.values())
    best_in_vote = max(test[n] for n in topk)
    assert vote >= best_in_vote - 0.01, (vote, test, topk)
    assert max(val.values()) > 0.9                      # the synthetic glyphs are learnable: something must learn them
    # transfer children (numbers 4, 5) were built from the two best random members; they inherit every layer but the
    # cut head, so after their own training they sit near their parent
    parents = sorted((n for n in val if n < 4), key=lambda n: -val[n])[:2]
    for child, parent in zip((4, 5), parents):
        assert val[child] >= val[parent] - 0.15, (child, parent, val)


def test_device_members_track_the_cpu_oracle():
    """Same seeds, same splits, same shuffles on both devices: well-conditioned members (the oracle itself reaches a
    non-trivial accuracy) must land within a few points of the oracle; every member must be finite."""
    ov = ["data_train_size=1600", "random_model_size=3", "transfer_model_size=0", "predict_model_size=3",
          "epochs=8", "early_stop_rate=-1.0", "seed=9", "conv2d_size=1,2", "dense_size=0,2", "kernel_size=2,5",
          "optimizers=Adam", "lr_exponent=-3,-2", "activations=relu,tanh"]
    _, _, gval, gtest, gvote, _ = _run("cuda", ov)
    _, _, cval, ctest, cvote, _ = _run("cpu", ov)
    assert set(gval) == set(cval)
    compared = 0
    for n in cval:
        assert np.isfinite(gval[n])
        if cval[n] > 0.6:                               # a member the oracle trains well
            assert abs(gval[n] - cval[n]) < 0.08, (n, gval, cval)
            assert abs(gtest[n] - ctest[n]) < 0.08, (n, gtest, ctest)
            compared += 1
    assert compared >= 1, (gval, cval)
    assert abs(gvote - cvote) < 0.08 or gvote > 0.95
