"""A TRAINING step of the whole model against vectors made by the real reference (oracle/make_golden.py --train: train mode,
cross-entropy, loss.backward(), float64): the module containers' training semantics on the CPU (torch composite, host test) and
the HIP training path on the GPU (convolutions: train_cnn.py, graph layer: its HIP forward / backward)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as tnf

from conftest import golden_paths, load_model_fixture

TRAIN = golden_paths("train_")


def _step(path, device, dtype):
    from magat_pathplanning_amd import DecentralPlannerGATNet
    z, sd, cfg = load_model_fixture(path)
    cfg.device = str(device)
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net = net.to(device=device, dtype=dtype).train()
    x = torch.from_numpy(z["x"]).to(device=device, dtype=dtype)
    S = torch.from_numpy(z["S"]).to(device)
    tgt = torch.from_numpy(z["target"]).to(device)
    net.addGSO(S.clone())
    logits = net(x)
    loss = tnf.cross_entropy(logits, tgt)
    loss.backward()
    return z, net, logits, loss


def _check(z, net, logits, loss, tol):
    np.testing.assert_allclose(logits.detach().cpu().double().numpy(), z["logits"], rtol=0, atol=tol * max(1.0, float(np.abs(z["logits"]).max())))
    assert abs(float(loss.detach()) - float(z["loss"])) < tol
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("g/"))
    for k, v in net.named_parameters():
        want = z["g/" + k]
        got = np.zeros_like(want) if v.grad is None else v.grad.detach().cpu().double().numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=tol * (10 * float(np.abs(want).max()) + 1e-2 * gmax), err_msg=k)
    for k, v in net.named_buffers():
        want = z["b/" + k]
        if v.dtype.is_floating_point:
            np.testing.assert_allclose(v.detach().cpu().double().numpy(), want, rtol=0, atol=tol * max(1.0, float(np.abs(want).max())), err_msg=k)
        else:
            assert int(v) == int(want), k


def test_train_fixtures_present():
    assert len(TRAIN) == 2


@pytest.mark.parametrize("path", TRAIN, ids=[os.path.basename(p)[:-4] for p in TRAIN])
def test_training_step_semantics_match_the_reference_on_the_cpu(path):
    """Host test (no GPU): the package's parameter containers + torch composite of the graph layer in float64 reproduce the
    reference's training step - what the HIP training path is then compared with on the GPU."""
    z, net, logits, loss = _step(path, torch.device("cpu"), torch.float64)
    _check(z, net, logits, loss, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("path", TRAIN, ids=[os.path.basename(p)[:-4] for p in TRAIN])
def test_training_step_on_the_hip_path_matches_the_reference(gpu_device, path, monkeypatch):
    """The HIP training path (float32 matrix-core convolutions forward / dX / dW, the graph layer's HIP forward / backward)
    against the REFERENCE's own training step: logits, loss, every parameter gradient, the BatchNorm buffers."""
    monkeypatch.setenv("MAGAT_TRAIN_CNN", "hip")
    z, net, logits, loss = _step(path, gpu_device, torch.float32)
    _check(z, net, logits, loss, 2e-4)
