"""CPU-only checks: the C-ABI library loads and exports every declared symbol, host-side folding /
module surface / state_dict layout, the training composite against the pinned oracle."""
import ctypes
import os
import pickle
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_paths, load_layer_fixture, load_model_fixture
from oracle import magat_oracle as orc

MODEL = golden_paths("model_")
LAYER = golden_paths("gat_")


def test_library_exports_every_declared_symbol():
    from magat_pathplanning_amd import _native as nat
    hdr = open(os.path.join(ROOT, "include", "magat_hip.h")).read()
    declared = set(re.findall(r"\b(magat_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"magat_conv_gemm_desc", "magat_encoder_desc"}
    assert declared == set(nat.EXPORTED_SYMBOLS), declared ^ set(nat.EXPORTED_SYMBOLS)
    lib = nat.lib()                       # loads without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.magat_abi_version() == 9
    assert lib.magat_build_flavor() == 0          # a release build: no timing-experiment switches compiled in
    assert lib.magat_error_string(-2).decode().startswith("unsupported")
    # pure host queries work without a device
    nc = 4 * 128 + 4 * 3 * 128            # fp32 [NC][G] + column bias [NC] + bf16x3 planes [3][NC][G] + f16x2 planes + scale
    base = nc * 129 + 3 * nc * 128 // 2 + nc * 128 + 4
    # + the f16x2 planes once more in matrix-core fragment order (128 features: the one-launch KeyQuery layer, gat_mfma.hip)
    # + (round 6) the bf16 fragments of the fused CSR layer (gat_csr_fused.hip): NC * G bf16 = NC * G / 2 floats + 4
    assert lib.magat_gat_packed_floats(128, 128, 3, 4, 0) == base + nc * 128 + nc * 64 + 4
    nc64 = 4 * 64 + 4 * 3 * 64
    assert lib.magat_gat_packed_floats(64, 64, 3, 4, 0) == nc64 * 65 + 3 * nc64 * 64 // 2 + nc64 * 64 + 4
    assert lib.magat_gat_workspace_bytes(2, 10, 128, 128, 2, 1, 0, 1) >= 2 * 10 * 384 * 4


@pytest.mark.parametrize("path", MODEL, ids=[os.path.basename(p)[:-4] for p in MODEL])
def test_state_dict_layout_and_training_path_match_reference(path):
    """Reference checkpoints load strictly; the autograd (training) composite reproduces the
    reference logits on CPU in eval-BN mode."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    z, sd, cfg = load_model_fixture(path)
    cfg.device = "cpu"
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd, strict=True)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.eval()
    x = torch.from_numpy(z["x"].astype(np.float32))
    S = torch.from_numpy(z["S"].copy())
    net.addGSO(S)
    y = net(x)                      # grad enabled -> differentiable composite
    assert y.requires_grad
    np.testing.assert_allclose(y.detach().numpy(), z["logits"], rtol=0, atol=5e-6)
    np.testing.assert_array_equal(np.nan_to_num(S.numpy(), nan=-7.0), np.nan_to_num(z["S_after"], nan=-7.0))
    y.sum().backward()
    assert net.GFL[0].filterWeight.grad is not None and torch.isfinite(net.GFL[0].filterWeight.grad).all()


def test_encoder_fold_matches_unfolded_cnn():
    """BN folding + avgpool/fc/Linear folding reproduce conv_layers_forward when evaluated densely on CPU."""
    import torch.nn.functional as tnf
    from magat_pathplanning_amd import encoder as enc
    from magat_pathplanning_amd.synthetic import fov_states, make_config
    for mode in ("ResNetLarge_withMLP", "ResNetSlim_withMLP", "ResNetLarge", "ResNetSlim"):
        cfg = make_config(CNN_mode=mode, device="cpu")
        sd = orc.init_state_dict(cfg, seed=3)
        lin = (sd["ConvLayers.3.weight"], sd["ConvLayers.3.bias"]) if mode.endswith("_withMLP") else None
        pack, offs, meta = enc.fold_resnet(sd, 11, 11, "ConvLayers.0", lin,
                                           (sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
        assert all(o % 4 == 0 for o in offs)
        x = fov_states(1, 6, seed=9).reshape(6, 3, 11, 11)
        want = orc.conv_layers_forward(x, sd, mode)

        def seg(slot, *shape):
            n = int(np.prod(shape))
            return pack[offs[slot]:offs[slot] + n].reshape(*shape)

        y = torch.relu(tnf.conv2d(x, seg(0, 32, 3, 3, 3), seg(1, 32), 1, 1))
        cin = 32
        for l, (cout, stride) in enumerate([(32, 2), (64, 1), (128, 1)][:3 if meta["variant"] == 0 else 2]):
            w1 = seg(2 + 4 * l, cout, 3, 3, cin).permute(0, 3, 1, 2)
            h = torch.relu(tnf.conv2d(y, w1, seg(3 + 4 * l, cout), stride, 1))
            wcat = seg(4 + 4 * l, cout, 9 * cout + cin)
            w2 = wcat[:, :9 * cout].reshape(cout, 3, 3, cout).permute(0, 3, 1, 2)
            wd = wcat[:, 9 * cout:].reshape(cout, cin, 1, 1)
            y = torch.relu(tnf.conv2d(h, w2, seg(5 + 4 * l, cout), 1, 1) + tnf.conv2d(y, wd, None, stride))
            cin = cout
        wh = seg(14, meta["n_feat"], 3, 3, cin).permute(0, 3, 1, 2)      # x 1/4; kernel sum-pools 2x2 on load
        feat = tnf.conv2d(4.0 * tnf.avg_pool2d(y, 2), wh, seg(15, meta["n_feat"])).flatten(1)
        np.testing.assert_allclose(feat.numpy(), want.numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("path", [p for p in LAYER if "N12" in p or "N10" in p],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_layer_training_composite_matches_reference(path):
    from magat_pathplanning_amd import GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin
    z, p = load_layer_fixture(path)
    G, K, P = int(z["G"]), int(z["K"]), int(z["P"])
    cls = GraphFilterBatchAttentional_Origin if str(z["mode"]) == "GAT_origin" else GraphFilterBatchAttentional
    layer = cls(G, G, K, P, attentionMode=str(z["mode"]))
    layer.load_state_dict(p)
    layer.addGSO(torch.from_numpy(z["S"]))
    y = layer(torch.from_numpy(z["x"]))
    np.testing.assert_allclose(y.detach().numpy(), z["y_concat"], rtol=0, atol=5e-6)
    assert "attentionMode=%s" % str(z["mode"]) in repr(layer)


GRAD = golden_paths("grad_")


@pytest.mark.parametrize("path", GRAD, ids=[os.path.basename(p)[5:-4] for p in GRAD])
def test_training_composite_gradients_match_reference_made_gradients(path):
    """The float64 composite the GPU gradient tests use as their tight second gate, pinned on the CPU to gradients made by
    the REAL reference's autograd (oracle/make_golden.py --grad; directed GSOs, all three attention modes, K = 1..4):
    y, dL/dx and every parameter gradient to 1e-6 of the gradient's scale (the fixture stores float32)."""
    from magat_pathplanning_amd import GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin
    z = np.load(path, allow_pickle=False)
    mode, concat = str(z["mode"]), bool(int(z["concat"]))
    G, K, P = int(z["G"]), int(z["K"]), int(z["P"])
    cls = GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else GraphFilterBatchAttentional
    layer = cls(G, G, K, P, concatenate=concat, attentionMode=mode)
    layer.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p_")})
    layer = layer.double()
    x = torch.from_numpy(z["x"]).double().requires_grad_(True)
    layer.addGSO(torch.from_numpy(z["S"]))
    y = layer(x)
    (y * torch.from_numpy(z["wgt"]).double()).sum().backward()

    def close(a, b, what):
        b = torch.from_numpy(b).double()
        scale = max(1.0, float(b.abs().max()))
        assert float((a.detach() - b).abs().max()) <= 1e-6 * scale, (what, float((a.detach() - b).abs().max()), scale)

    close(y, z["y"], "y")
    close(x.grad, z["dx"], "dx")
    for k in z.files:
        if k.startswith("g_"):
            g = getattr(layer, k[2:]).grad
            close(torch.zeros_like(getattr(layer, k[2:])) if g is None else g, z[k], k)


def test_module_pickles_without_device_state():
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import make_config
    net = DecentralPlannerGATNet(make_config(device="cpu"))
    blob = pickle.dumps(net)
    clone = pickle.loads(blob)
    for (k1, v1), (k2, v2) in zip(net.state_dict().items(), clone.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert clone._rt.key is None and clone.GFL[0]._scratch.packed is None


def test_synthetic_inputs_are_seeded_and_well_formed():
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states
    x1, x2 = fov_states(3, 20, seed=5), fov_states(3, 20, seed=5)
    assert torch.equal(x1, x2) and set(x1.unique().tolist()) <= {0.0, 1.0}
    assert torch.all(x1[:, :, 1].sum(dim=(2, 3)) == 1) and torch.all(x1[:, :, 2, 5, 5] == 1)
    assert float(x1[:, :, :, 0, :].abs().max()) == 0.0
    S = comm_gso(4, 20, 28, seed=6, dtype=torch.float64)
    assert torch.equal(S, S.transpose(1, 2)) and float(torch.diagonal(S, dim1=1, dim2=2).abs().max()) == 0.0


def test_default_cnn_fold_matches_unfolded_stack():
    """CNN_mode=Default: conv bias + BN folded per layer, max-pools applied to the layer INPUT (pool-on-load)."""
    import torch.nn.functional as tnf
    from magat_pathplanning_amd import encoder as enc
    from magat_pathplanning_amd.synthetic import fov_states, make_config
    cfg = make_config(CNN_mode="Default", device="cpu")
    sd = orc.init_state_dict(cfg, seed=4)
    pack, offs, meta = enc.fold_default_cnn(sd, 11, 11, "ConvLayers", (sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
    assert meta["variant"] == 2 and all(o % 4 == 0 for o in offs)
    x = fov_states(1, 5, seed=2).reshape(5, 3, 11, 11)
    want = orc.default_cnn_forward(x, sd).flatten(1)

    def seg(slot, *shape):
        n = int(np.prod(shape))
        return pack[offs[slot]:offs[slot] + n].reshape(*shape)

    y = torch.relu(tnf.conv2d(x, seg(0, 32, 3, 3, 3), seg(1, 32), 1, 1))
    chans = [32, 32, 64, 64, 128]
    for l in range(1, 5):
        if (l - 1) % 2 == 0:
            y = tnf.max_pool2d(y, 2)
        w = seg(2 + 2 * (l - 1), chans[l], 3, 3, chans[l - 1]).permute(0, 3, 1, 2)
        y = torch.relu(tnf.conv2d(y, w, seg(3 + 2 * (l - 1), chans[l]), 1, 1))
    feat = tnf.max_pool2d(y, 2).flatten(1) @ seg(14, 128, 128).t()
    np.testing.assert_allclose(feat.numpy(), want.numpy(), rtol=0, atol=2e-5)


def test_graph_filter_batch_state_dict_and_composite():
    """GraphFilterBatch (GNN baseline): reference parameter names / shapes, and the autograd composite (CPU) equals the
    reference-made fixture."""
    from conftest import golden_paths
    from magat_pathplanning_amd import GraphFilterBatch
    for path in golden_paths("gnn_")[:2]:
        z = np.load(path)
        layer = GraphFilterBatch(int(z["G"]), int(z["F"]), int(z["K"]))
        assert {k: tuple(v.shape) for k, v in layer.state_dict().items()} == {"weight": z["p_weight"].shape, "bias": z["p_bias"].shape}
        layer.load_state_dict({"weight": torch.from_numpy(z["p_weight"]), "bias": torch.from_numpy(z["p_bias"])})
        layer.addGSO(torch.from_numpy(z["S"]))
        y = layer(torch.from_numpy(z["x"]).requires_grad_(True))
        np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=0, atol=5e-6)
        with pytest.raises(Exception):
            with torch.no_grad():
                layer(torch.from_numpy(z["x"]))        # inference is HIP-only: CPU tensors fail loudly


def test_f16x2_weight_block_reconstructs_weights():
    """encoder.split_f16x2: two half planes of w * 2^e + the inverse scale reproduce w to 2^-22 relative (absolute floor
    2^-25 * 2^-e for the tiny entries)."""
    from magat_pathplanning_amd.encoder import split_f16x2
    g = torch.Generator().manual_seed(3)
    for amp in (0.05, 3.0, 1e-4):
        w = torch.randn(64, 288, generator=g) * amp
        w[0, :4] = torch.tensor([0.0, 1e-9, -amp * 5, amp * 1e-5])
        blk, e = split_f16x2(w)
        n = w.numel()
        h = blk[:n].view(torch.int16).view(torch.float16).float()
        assert blk.numel() == n + 1 and blk[-1].item() == 2.0 ** (-e)
        rec = (h[:n] + h[n:]) * blk[-1]
        err = (rec - w.reshape(-1)).abs()
        assert float(err.max()) <= max(2.0 ** -22 * float(w.abs().max()), 2.0 ** -24 * 2.0 ** (-e)), (amp, float(err.max()))


def test_fold_activation_scales_is_exact_and_consistent():
    """encoder.fold_activation_scales (include/magat_hip.h "Activation scales"): every entry of the block is the pack's value
    times a power of two; a block's input, conv1 output and residual input share one exponent; the 1 / weight-scale floats
    carry the exponent DIFFERENCES, so that a chain of layers ends at the true scale."""
    import math
    import torch
    from magat_pathplanning_amd import encoder as enc
    from magat_pathplanning_amd.synthetic import make_config
    from oracle import magat_oracle as orc
    cfg = make_config(num_agents=4)
    sd = orc.init_state_dict(cfg, seed=3)
    pack, offs, meta = enc.fold_resnet(sd, 11, 11, "ConvLayers.0", (sd["ConvLayers.3.weight"], sd["ConvLayers.3.bias"]),
                                       (sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
    absmax = [3e-4, 2e-4, 7.0, 900.0, 0.02, 0.05, 1.5e-6, 4e3, 0.3] + [0.0] * 7
    blk, info = enc.fold_activation_scales(pack, offs, meta, absmax)
    assert blk.numel() == enc.SCALED_BLOCK_FLOATS and blk.dtype == torch.float32
    s1, s2, s3 = info["s1"], info["s2"], info["s3"]
    assert 2 ** 9 <= max(absmax[0], absmax[1]) * 2.0 ** s1 <= 2 ** 10 or s1 <= 11 - 4 + 9     # (or the stem-weight cap)
    assert 2 ** 9 <= max(absmax[2], absmax[3]) * 2.0 ** s2 <= 2 ** 10
    assert 2 ** 9 <= max(absmax[4], absmax[5]) * 2.0 ** s3 <= 2 ** 10
    w0 = pack[offs[0]:offs[0] + 864]
    assert torch.equal(blk[0:864], w0 * 2.0 ** s1)                      # exact: a power of two
    assert float((blk[0:864].abs().max()) * 16) < 65504                 # the stem kernel's f16 planes of 16 w'
    assert torch.equal(blk[928:960], pack[offs[5]:offs[5] + 32] * 2.0 ** s2)
    assert torch.equal(blk[1216:1344], pack[offs[13]:offs[13] + 128])   # the chain ends at the true scale
    ch = meta["chain"]
    sA, sC = float(pack[ch + 10240]), float(pack[ch + 10240 + 4 + 18432 + 4 + 38912])
    assert float(blk[1344]) == sA * 2.0 ** (s2 - s1) and float(blk[1346]) == sC * 2.0 ** (s3 - s2)
    c3 = meta["chain3"]
    s32 = float(pack[c3 + 73728 + 4 + 73728 + 4 + 81920])
    assert float(blk[1348]) == s32 * 2.0 ** (-s3)
    # scale ratios telescope: (s2 - s1) + (s3 - s2) + (0 - s3) = -s1, what the stem put in
    tot = math.log2(float(blk[1344]) / sA) + math.log2(float(blk[1346]) / sC) + math.log2(float(blk[1348]) / s32)
    assert tot == -s1
    for k in ("head_in", "feat_in"):
        assert float(blk[1349 if k == "head_in" else 1350]) == 2.0 ** info[k]


def test_weights_key_sees_every_way_the_weights_can_change():
    """planner._weights_key (the per-forward check that decides whether the packed weights of the HIP path are still valid):
    equal for an untouched module; different after an in-place update, load_state_dict, a replaced Parameter, a replaced
    submodule, a BatchNorm buffer update, and for another device."""
    import copy
    import torch
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import make_config
    cfg = make_config(num_agents=10, nGraphFilterTaps=2, nAttentionHeads=2, device="cpu")
    net = DecentralPlannerGATNet(cfg).eval()
    dev = torch.device("cpu")
    k0 = net._weights_key(dev)
    assert net._weights_key(dev) == k0
    n_tensors = sum(1 for _ in net.parameters()) + sum(1 for _ in net.buffers())
    assert len(k0) == 3 and len(k0[1]) == n_tensors and len(k0[2]) == n_tensors      # every tensor is in it (addresses, versions)
    with torch.no_grad():
        next(net.parameters()).add_(1.0)
    k1 = net._weights_key(dev)
    assert k1 != k0
    net.load_state_dict(copy.deepcopy(net.state_dict()))
    k2 = net._weights_key(dev)
    assert k2 != k1
    lin = net.compressMLP[0]
    lin.weight = torch.nn.Parameter(lin.weight.detach().clone())
    k3 = net._weights_key(dev)
    assert k3 != k2
    net.compressMLP[0] = torch.nn.Linear(lin.in_features, lin.out_features)
    k4 = net._weights_key(dev)
    assert k4 != k3
    bn = next(m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
    with torch.no_grad():
        bn.running_mean.mul_(0.5)
    assert net._weights_key(dev) != k4
    assert net._weights_key(torch.device("cuda:0"))[0] != k4[0]


def test_gnn_model_class_surface_matches_the_reference_fixture():
    """DecentralPlannerNet: reference-made state_dicts load strict=True (same keys and shapes), the module pickles, and the
    autograd path (torch composite on CPU tensors, host tests only) reproduces the reference's logits."""
    from magat_pathplanning_amd import DecentralPlannerNet
    for path in golden_paths("gnnmodel_"):
        z, sd, cfg = load_model_fixture(path)
        cfg.device = "cpu"
        net = DecentralPlannerNet(cfg)
        net.load_state_dict(sd, strict=True)
        net.train(False)
        pickle.loads(pickle.dumps(net))
        x = torch.from_numpy(z["x"].astype(np.float32))
        S = torch.from_numpy(z["S"].copy())
        net.addGSO(S)
        np.testing.assert_array_equal(S.numpy(), z["S_after"])
        for p_ in net.parameters():
            p_.requires_grad_(True)
        y = net(x)          # grad enabled -> the differentiable composite
        np.testing.assert_allclose(y.detach().numpy(), z["logits"], rtol=0, atol=5e-6 * max(1.0, float(np.abs(z["logits"]).max())))


def test_graft_entry_build_passes():
    """The driver's "does it build" check is __graft_entry__.build(): run it here too (incremental: the library the other host
    tests load is already built), so that an ABI bump or a new source file cannot leave it behind the suite."""
    import importlib
    sys.path.insert(0, ROOT)
    ge = importlib.import_module("__graft_entry__")
    ge.build()


def test_reset_option_restores_the_environments_value():
    """magat_reset_option goes back to what the PROCESS started with - MAGAT_<NAME> from the environment when it was set, else the
    built-in default (a deployment's setting must survive a tool that flips an option and resets it).  The
    library reads the environment once, so this needs a fresh process."""
    import subprocess
    import sys
    code = ("from magat_pathplanning_amd import _native as n\n"
            "assert n.get_option('HEAD_SPLITK') == 777 and n.get_option('GAT_PACK') == 1\n"
            "n.set_option('HEAD_SPLITK', 5); n.set_option('GAT_PACK', 0)\n"
            "n.reset_option('HEAD_SPLITK'); n.reset_option('GAT_PACK')\n"
            "assert n.get_option('HEAD_SPLITK') == 777 and n.get_option('GAT_PACK') == 1\n"
            "print('ok')\n")
    env = dict(os.environ, MAGAT_HEAD_SPLITK="777")
    env.pop("MAGAT_GAT_PACK", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]


def test_fragment_major_head_pack_holds_the_row_major_planes():
    """ABI 8: encoder.fold_resnet appends the head's and compressMLP's f16 planes once more fragment-major (block_lat.hip fetches
    1 KB blocks straight into registers).  They must be the SAME halves and the same scale as the row-major planes the batched
    kernels read (head16 / comp16) - that is what makes the one-launch encoder of a batch-1 step bit-identical to the batched
    forms - in the long-K head's k order (32-channel slab outer, pooled cell inner)."""
    from magat_pathplanning_amd import encoder as enc
    from magat_pathplanning_amd.synthetic import make_config
    cfg = make_config(device="cpu")
    sd = orc.init_state_dict(cfg, seed=2)
    pack, offs, meta = enc.fold_resnet(sd, 11, 11, "ConvLayers.0", (sd["ConvLayers.3.weight"], sd["ConvLayers.3.bias"]),
                                       (sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
    assert meta["headfrag"] > 0 and meta["compfrag"] > 0 and meta["headfrag"] % 4 == 0 and meta["compfrag"] % 4 == 0
    for frag_off, row_off, K, steps in (
            (meta["headfrag"], meta["head16"], 1152, [cell * 128 + 32 * cs + 16 * ks for cs in range(4) for cell in range(9) for ks in range(2)]),
            (meta["compfrag"], meta["comp16"], 128, [16 * ks for ks in range(8)])):
        planes = pack[row_off:row_off + 128 * K].view(torch.int16).view(2, 128, K)
        scale_rows = float(pack[row_off + 128 * K])
        ns = len(steps)
        blk = pack[frag_off:frag_off + 4 * ns * 1024 // 2].view(torch.int16).view(4, ns, 2, 64, 8)
        assert float(pack[frag_off + 4 * ns * 512]) == scale_rows
        lane = torch.arange(64)
        for ct in range(4):
            for s_, k0 in enumerate(steps):
                for pl in range(2):
                    want = planes[pl][(32 * ct + (lane & 31)).view(64, 1), (k0 + 8 * (lane >> 5)).view(64, 1) + torch.arange(8).view(1, 8)]
                    assert torch.equal(blk[ct, s_, pl], want)

