"""Pins oracle/magat_oracle.py to the reference-made golden vectors (CPU only)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_paths, load_layer_fixture, load_model_fixture
from oracle import magat_oracle as orc

LAYER = golden_paths("gat_")
MODEL = golden_paths("model_")


def test_fixtures_present():
    assert len(LAYER) == 27 and len(MODEL) == 9      # 14 + the 9 directed-GSO fixtures of round 3 + 4 small-graph ones (round 4)


@pytest.mark.parametrize("path", LAYER, ids=[os.path.basename(p)[:-4] for p in LAYER])
def test_layer_oracle_matches_reference(path):
    z, p = load_layer_fixture(path)
    x, S = torch.from_numpy(z["x"]), torch.from_numpy(z["S"])
    mode = str(z["mode"])
    for concat, key in ((True, "y_concat"), (False, "y_mean")):
        y, aij = orc.gat_layer_forward(x, S, p, mode, concat)
        np.testing.assert_allclose(y.numpy(), z[key], rtol=0, atol=2e-6)
        np.testing.assert_allclose(aij.numpy(), z["aij"], rtol=0, atol=1e-6)
    nin = int(z["nin"])
    y, _ = orc.gat_layer_forward(x[:, :, :nin].contiguous(), S, p, mode, True)
    np.testing.assert_allclose(y.numpy(), z["y_concat_nin"], rtol=0, atol=2e-6)


def _small_enough_for_loops(path):
    """(the edge-by-edge restatement is pure Python: the N >= 100, G = 128 fixtures would take minutes)"""
    import re
    n, g = map(int, re.search(r"_N(\d+)_G(\d+)_", os.path.basename(path)).groups())
    return n * g < 100 * 128


@pytest.mark.parametrize("path", [p for p in LAYER if _small_enough_for_loops(p)],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_loop_oracle_matches_reference(path):
    """The edge-by-edge float64 restatement agrees too (independent derivation)."""
    z, p = load_layer_fixture(path)
    mode = str(z["mode"])
    for concat, key in ((True, "y_concat"), (False, "y_mean")):
        y, aij = orc.gat_layer_forward_loops(z["x"], z["S"], p, mode, concat)
        np.testing.assert_allclose(y, z[key], rtol=0, atol=5e-6)
        np.testing.assert_allclose(aij, z["aij"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("path", MODEL, ids=[os.path.basename(p)[:-4] for p in MODEL])
def test_model_oracle_matches_reference(path):
    z, sd, cfg = load_model_fixture(path)
    x = torch.from_numpy(z["x"].astype(np.float32))
    S = torch.from_numpy(z["S"].copy())
    logits, parts = orc.planner_forward(x, S, sd, cfg, return_parts=True)
    np.testing.assert_allclose(logits.numpy(), z["logits"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(parts["aij"].numpy(), z["aij"], rtol=0, atol=1e-6)
    # addGSO's in-place mutation of the caller's tensor is part of the contract
    np.testing.assert_array_equal(np.nan_to_num(S.numpy(), nan=-7.0), np.nan_to_num(z["S_after"], nan=-7.0))


def test_init_state_dict_shapes_match_reference_fixture():
    for path in MODEL:
        z, sd, cfg = load_model_fixture(path)
        mine = orc.init_state_dict(cfg, seed=1)
        assert set(mine) == set(sd), (path, set(mine) ^ set(sd))
        for k in sd:
            assert tuple(mine[k].shape) == tuple(sd[k].shape), (path, k)


GNN = golden_paths("gnn_")


def test_gnn_fixture_inventory():
    assert len(GNN) == 4


@pytest.mark.parametrize("path", GNN, ids=[os.path.basename(p)[:-4] for p in GNN])
def test_graph_filter_batch_oracle_matches_reference(path):
    z = np.load(path)
    y = orc.graph_filter_batch_forward(torch.from_numpy(z["x"]), torch.from_numpy(z["S"]), torch.from_numpy(z["p_weight"]),
                                       torch.from_numpy(z["p_bias"]))
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=0, atol=2e-6)


GNNMODEL = golden_paths("gnnmodel_")


@pytest.mark.parametrize("path", GNNMODEL, ids=[os.path.basename(p)[:-4] for p in GNNMODEL])
def test_gnn_model_oracle_matches_reference(path):
    """DecentralPlannerNet (graphs/models/decentralplanner.py; oracle/make_golden.py --gnn-model): the oracle's restatement
    against logits and the mutated GSO made by the real reference."""
    z, sd, cfg = load_model_fixture(path)
    assert len(GNNMODEL) == 5
    x = torch.from_numpy(z["x"].astype(np.float32))
    S = torch.from_numpy(z["S"].copy())
    got = orc.planner_gnn_forward(x, S, sd, cfg)
    np.testing.assert_allclose(got.numpy(), z["logits"], rtol=0, atol=2e-6 * max(1.0, float(np.abs(z["logits"]).max())))
    np.testing.assert_array_equal(S.numpy(), z["S_after"])


EDGE = golden_paths("edge_")
ACTIVATIONS = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
               "leaky_relu": lambda t: torch.nn.functional.leaky_relu(t, 0.1), "identity": lambda t: t}


@pytest.mark.parametrize("path", EDGE, ids=[os.path.basename(p)[:-4] for p in EDGE])
def test_edge_feature_and_nonlinearity_oracle_matches_reference(path):
    """E > 1 edge features / a nonlinearity other than ReLU (oracle/make_golden.py --edge: the reference's own forward)."""
    assert len(EDGE) == 7
    z, p = load_layer_fixture(path)
    x, S = torch.from_numpy(z["x"]), torch.from_numpy(z["S"])
    mode, act = str(z["mode"]), ACTIVATIONS[str(z["act"])]
    assert S.shape[1] == int(z["E"])
    for concat, key in ((True, "y_concat"), (False, "y_mean")):
        y, aij = orc.gat_layer_forward(x, S, p, mode, concat, nonlinearity=act)
        np.testing.assert_allclose(y.numpy(), z[key], rtol=0, atol=2e-6)
        np.testing.assert_allclose(aij.numpy(), z["aij"], rtol=0, atol=1e-6)
    nin = int(z["nin"])
    y, _ = orc.gat_layer_forward(x[:, :, :nin].contiguous(), S, p, mode, True, nonlinearity=act)
    np.testing.assert_allclose(y.numpy(), z["y_concat_nin"], rtol=0, atol=2e-6)
