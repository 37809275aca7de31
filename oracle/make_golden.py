"""Generates tests/golden/*.npz by running the REAL reference (imported from
/root/reference, build container only) on seeded inputs.  TEST INFRASTRUCTURE.

    python oracle/make_golden.py            # rewrites every fixture

The fixtures hold data only: inputs, parameters (float32 arrays) and the outputs the
reference's own code produced for them:
  gat_<mode>_N<N>_G<G>_K<K>_P<P>.npz   GraphFilterBatchAttentional.forward
                                       (utils/graphUtils/graphML.py:4636-4671), both merges
  model_<name>.npz                     DecentralPlannerGATNet.addGSO + forward
                                       (graphs/models/decentralplanner_GAT_bottleneck*.py)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle._ref_import import import_reference, make_config  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def tricky_gso(gen, B, N, density, f64):
    """Symmetric random graph + the edge cases SURVEY.md section 4 lists: an isolated node,
    an asymmetric (directed) edge pair, tiny values around the 1e-9 threshold, 1/lambda_max
    scaling, and one NaN entry (layer level: |NaN|>1e-9 is False -> no edge)."""
    W = (torch.rand(B, N, N, generator=gen) < density).double()
    W = torch.triu(W, 1)
    W = W + W.transpose(1, 2)
    for b in range(B):
        iso = (3 + b) % N
        W[b, iso, :] = 0
        W[b, :, iso] = 0
        i, j = (1 + b) % N, (5 + 2 * b) % N
        if i != j and i != iso and j != iso:
            W[b, i, j] = 1.0
            W[b, j, i] = 0.0           # asymmetric mask
        lam = float(np.max(np.real(np.linalg.eigvals(W[b].numpy())))) or 1.0
        W[b] = W[b] / max(lam, 1e-6)
        k, l = (2 + b) % N, (7 + b) % N
        if k != l and iso not in (k, l):
            W[b, k, l] = 5e-10         # below zeroTolerance -> not an edge
            W[b, l, k] = -3e-9         # |.| above zeroTolerance -> an edge
    W[0, 0, (N - 1)] = float("nan")
    return W if f64 else W.float()


def layer_fixture(gml, mode, N, G, K, P, seed, density, f64, gso=None):
    gen = torch.Generator().manual_seed(seed)
    B, F = 2, G
    layers = {}
    cls = gml.GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else gml.GraphFilterBatchAttentional
    for concat in (True, False):
        torch.manual_seed(seed)
        layers[concat] = cls(G, F, K, P, 1, True, concatenate=concat, attentionMode=mode)
    ref = layers[True]
    with torch.no_grad():
        if mode != "GAT_origin":
            ref.weight_bias.uniform_(-0.3, 0.3, generator=gen)   # reference init is 0; exercise it
        layers[False].load_state_dict(ref.state_dict())
    x = torch.randn(B, G, N, generator=gen) * 0.7
    S = (tricky_gso(gen, B, N, density, f64) if gso is None else gso(B, N)).unsqueeze(1)
    out = {}
    with torch.no_grad():
        for concat, lay in layers.items():
            lay.addGSO(S)
            y = lay(x)
            out["y_concat" if concat else "y_mean"] = y.numpy()
        out["aij"] = ref.aij.astype(np.float32)
        # Nin < N zero-padding path (graphML.py:4642-4646, 4669-4670)
        nin = max(1, N - 3)
        out["y_concat_nin"] = ref(x[:, :, :nin].contiguous()).numpy()
        out["nin"] = np.int64(nin)
    out.update(x=x.numpy(), S=S.numpy(), mode=np.array(mode), N=N, G=G, K=K, P=P)
    for k, v in ref.state_dict().items():
        out["p_" + k] = v.numpy()
    return out


def fov_states(gen, B, N):
    """Binary 3-channel FOV tensors shaped like AgentState.toInputTensor output
    (dataloader/statetransformer_Guidance.py:185-239): 9x9 FOV inside a zero 1-px border."""
    x = torch.zeros(B, N, 3, 11, 11)
    x[:, :, 0, 1:10, 1:10] = (torch.rand(B, N, 9, 9, generator=gen) < 0.1).float()
    gi = torch.randint(0, 81, (B, N), generator=gen)
    goal = torch.zeros(B, N, 121)
    goal.scatter_(2, ((gi // 9 + 1) * 11 + gi % 9 + 1).unsqueeze(-1), 1.0)
    x[:, :, 1] = goal.view(B, N, 11, 11)
    x[:, :, 2, 1:10, 1:10] = (torch.rand(B, N, 9, 9, generator=gen) < 0.08).float()
    x[:, :, 2, 5, 5] = 1.0
    return x


def model_fixture(classes, name, seed, B, **cfgkw):
    cfg = make_config(**cfgkw)
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = classes[cfg.bottleneckMode](cfg).eval()
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.2, generator=gen)
                mod.running_var.uniform_(0.5, 1.5, generator=gen)
                mod.bias.normal_(0, 0.1, generator=gen)
        if hasattr(model.GFL[0], "weight_bias"):
            model.GFL[0].weight_bias.uniform_(-0.3, 0.3, generator=gen)
        for n_, p_ in model.named_parameters():
            if n_.endswith(".bias") and p_.dim() == 1 and "bn" not in n_ and "downsample" not in n_:
                p_.normal_(0, 0.05, generator=gen)
    N = cfg.num_agents
    if cfg.FOV == 9:
        x = fov_states(gen, B, N)
    else:
        from magat_pathplanning_amd.synthetic import fov_states as fov_states_any
        x = fov_states_any(B, N, seed=seed + 1, fov=cfg.FOV)
    f64 = cfgkw.get("_f64", True)
    S = tricky_gso(gen, B, N, 0.25 if N <= 20 else 0.08, True)
    if cfg.bottleneckMode != "BottomNeck_only":
        S[torch.isnan(S)] = 0.3        # Skip* variants do not scrub NaN (file diff); keep finite
    S_in = S.clone()
    model.addGSO(S)
    with torch.no_grad():
        logits = model(x)
    out = dict(x=x.numpy().astype(np.uint8), S=S_in.numpy(), S_after=S.numpy(), logits=logits.numpy(),
               aij=model.GFL[0].aij.astype(np.float32), cfg=np.array(repr(vars(cfg))))
    for k, v in model.state_dict().items():
        out["sd/" + k] = v.numpy()
    return out


def main_origin():
    """GAT_origin fixtures (added after the first set; generated separately so the earlier files stay byte-identical)."""
    gml, classes = import_reference()
    os.makedirs(OUT, exist_ok=True)
    for si, (N, G, K, P) in enumerate([(10, 128, 2, 1), (20, 128, 3, 4), (100, 32, 3, 4), (12, 16, 3, 4)]):
        fx = layer_fixture(gml, "GAT_origin", N, G, K, P, seed=2337 + 13 * si, density=0.3 if N <= 20 else 0.1,
                           f64=(si % 2 == 1))
        path = os.path.join(OUT, "gat_GAT_origin_N%d_G%d_K%d_P%d.npz" % (N, G, K, P))
        np.savez_compressed(path, **fx)
        print("wrote", path, os.path.getsize(path) // 1024, "KB")
    fx = model_fixture(classes, "origin_skipconcat", 5151, 2, num_agents=16, nGraphFilterTaps=3, nAttentionHeads=4,
                       attentionMode="GAT_origin", bottleneckMode="BottomNeck_skipConcat", bottleneckFeature=64)
    path = os.path.join(OUT, "model_origin_skipconcat.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


def main():
    gml, classes = import_reference()
    os.makedirs(OUT, exist_ok=True)
    shapes = [(10, 128, 2, 1), (20, 128, 3, 4), (100, 128, 3, 4), (100, 32, 3, 4), (12, 16, 3, 4)]
    for mode in ("KeyQuery", "GAT_modified"):
        for si, (N, G, K, P) in enumerate(shapes):
            fx = layer_fixture(gml, mode, N, G, K, P, seed=1337 + 17 * si + (0 if mode == "KeyQuery" else 5),
                               density=0.3 if N <= 20 else 0.1, f64=(si % 2 == 0))
            path = os.path.join(OUT, "gat_%s_N%d_G%d_K%d_P%d.npz" % (mode, N, G, K, P))
            np.savez_compressed(path, **fx)
            print("wrote", path, os.path.getsize(path) // 1024, "KB")
    models = [
        ("bottleneck_c1", dict(num_agents=10, nGraphFilterTaps=2, nAttentionHeads=1, B=2)),
        ("skipconcat_c3", dict(num_agents=100, nGraphFilterTaps=3, nAttentionHeads=4, B=2,
                               bottleneckMode="BottomNeck_skipConcat")),
        ("skipconcatgnn_b32_mean", dict(num_agents=20, nGraphFilterTaps=2, nAttentionHeads=4, B=3,
                                        bottleneckFeature=32, AttentionConcat=False,
                                        bottleneckMode="BottomNeck_skipConcatGNN", GSO_mode="dist_GSO_one")),
        ("skipadd_slim_modified", dict(num_agents=12, nGraphFilterTaps=3, nAttentionHeads=2, B=2,
                                       AttentionConcat=False, attentionMode="GAT_modified",
                                       CNN_mode="ResNetSlim", bottleneckMode="BottomNeck_skipAddGNN")),
        ("default_cnn", dict(num_agents=10, nGraphFilterTaps=3, nAttentionHeads=4, B=2, CNN_mode="Default",
                             attentionMode="GAT_modified")),
        ("legacy_gat_dropout", dict(num_agents=8, nGraphFilterTaps=2, nAttentionHeads=2, B=2,
                                    bottleneckMode="", use_dropout=True, numInputFeatures=64,
                                    CNN_mode="ResNetSlim_withMLP", GSO_mode="full_GSO")),
    ]
    for i, (name, kw) in enumerate(models):
        B = kw.pop("B")
        fx = model_fixture(classes, name, 4242 + i, B, **kw)
        path = os.path.join(OUT, "model_%s.npz" % name)
        np.savez_compressed(path, **fx)
        print("wrote", path, os.path.getsize(path) // 1024, "KB")


def main_gnn():
    """GraphFilterBatch (non-attentional GNN baseline, graphML.py:5581-5700) fixtures, generated separately."""
    gml, _ = import_reference()
    os.makedirs(OUT, exist_ok=True)
    for si, (N, G, F, K) in enumerate([(10, 128, 128, 2), (20, 128, 64, 3), (100, 32, 32, 3), (150, 64, 128, 4)]):
        gen = torch.Generator().manual_seed(3337 + 11 * si)
        torch.manual_seed(3337 + 11 * si)
        layer = gml.GraphFilterBatch(G, F, K, 1, True)
        B = 2
        x = torch.randn(B, G, N, generator=gen) * 0.7
        S = tricky_gso(gen, B, N, 0.3 if N <= 20 else 0.1, f64=(si % 2 == 0))
        S[torch.isnan(S)] = 0.25                      # the GSO values are multiplied in: keep them finite
        S = S.unsqueeze(1)
        with torch.no_grad():
            layer.addGSO(S)
            y = layer(x)
        out = dict(x=x.numpy(), S=S.numpy(), y=y.numpy(), N=N, G=G, F=F, K=K)
        for k, v in layer.state_dict().items():
            out["p_" + k] = v.numpy()
        path = os.path.join(OUT, "gnn_N%d_G%d_F%d_K%d.npz" % (N, G, F, K))
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KB")


def main_directed():
    """Round 3: layer and model fixtures over fully DIRECTED GSOs (magat_pathplanning_amd.synthetic.directed_gso: every
    ordered pair drawn independently, a one-way edge into an otherwise isolated node, threshold entries, a NaN, float64
    1/lambda_max values) at the shapes the one-launch kernel covers and at its hand-over (N = 102 | 103 | 128), plus the
    published B32-P4 / head-mean settings (scripts/train_DMap.sh).  Generated separately: the earlier files stay
    byte-identical."""
    from magat_pathplanning_amd.synthetic import directed_gso
    gml, classes = import_reference()
    cases = [("KeyQuery", 100, 128, 3, 4, 0.05, True), ("KeyQuery", 102, 128, 2, 2, 0.3, False),
             ("KeyQuery", 103, 128, 3, 4, 0.05, True), ("KeyQuery", 128, 128, 3, 4, 0.04, False),
             ("KeyQuery", 64, 128, 3, 4, 0.1, True), ("KeyQuery", 100, 32, 2, 4, 0.05, True),
             ("KeyQuery", 100, 64, 3, 4, 0.05, False), ("GAT_modified", 100, 128, 3, 4, 0.05, True),
             ("GAT_modified", 100, 32, 2, 4, 0.08, False)]
    for si, (mode, N, G, K, P, dens, f64) in enumerate(cases):
        seed = 6337 + 19 * si
        fx = layer_fixture(gml, mode, N, G, K, P, seed=seed, density=dens, f64=f64,
                           gso=lambda B, N_: directed_gso(B, N_, dens, seed=seed + 1,
                                                          dtype=torch.float64 if f64 else torch.float32))
        path = os.path.join(OUT, "gat_%s_directed_N%d_G%d_K%d_P%d.npz" % (mode, N, G, K, P))
        np.savez_compressed(path, **fx)
        print("wrote", path, os.path.getsize(path) // 1024, "KB")


def main_fov():
    """Round 3: CNN_mode Default at fields of view other than 9 (decentralplanner_GAT_bottleneck*.py:118-147 sizes its
    feature map from config.FOV): 19 x 19 maps leave 2 x 2 pooled cells (512 features, Flatten is (channel, cell)-ordered),
    9 x 9 maps one."""
    _, classes = import_reference()
    models = [("default_cnn_fov17_skipconcat", dict(num_agents=6, nGraphFilterTaps=3, nAttentionHeads=2, B=2, FOV=17,
                                                     CNN_mode="Default", bottleneckMode="BottomNeck_skipConcat")),
              ("default_cnn_fov7", dict(num_agents=7, nGraphFilterTaps=2, nAttentionHeads=2, B=2, FOV=7,
                                        CNN_mode="Default", attentionMode="GAT_modified", bottleneckFeature=64))]
    for i, (name, kw) in enumerate(models):
        B = kw.pop("B")
        fx = model_fixture(classes, name, 7373 + i, B, **kw)
        path = os.path.join(OUT, "model_%s.npz" % name)
        np.savez_compressed(path, **fx)
        print("wrote", path, os.path.getsize(path) // 1024, "KB")


def main_gnn_model():
    """Round 4: DecentralPlannerNet (graphs/models/decentralplanner.py:14-398: encoder + ONE GraphFilterBatch + ReLU + action
    MLP, the paper's GNN baseline and the first command of scripts/train_DMap.sh:30) - model-level fixtures: the published
    setting (10 agents, K = 2, 128 features, ResNetLarge_withMLP, dist_GSO), one with use_dropout / ResNetSlim / dist_GSO_one
    / K = 3, one with the Default CNN and no_ReLU."""
    from oracle._ref_import import import_reference_gnn_model
    cls = import_reference_gnn_model()
    cases = [("gnn_published", dict(num_agents=10, nGraphFilterTaps=2), 3),
             ("gnn_dropout_slim_one", dict(num_agents=12, nGraphFilterTaps=3, use_dropout=True, CNN_mode="ResNetSlim",
                                           GSO_mode="dist_GSO_one", numInputFeatures=64), 2),
             ("gnn_default_cnn_norelu", dict(num_agents=9, nGraphFilterTaps=4, CNN_mode="Default", no_ReLU=True), 2)]
    for i, (name, kw, B) in enumerate(cases):
        cfg = make_config(use_dilated=False, no_ReLU=kw.pop("no_ReLU", False), **kw)
        seed = 9191 + i
        gen = torch.Generator().manual_seed(seed)
        torch.manual_seed(seed)
        model = cls(cfg).eval()
        with torch.no_grad():
            for mod in model.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.normal_(0, 0.2, generator=gen)
                    mod.running_var.uniform_(0.5, 1.5, generator=gen)
                    mod.bias.normal_(0, 0.1, generator=gen)
        N = cfg.num_agents
        x = fov_states(gen, B, N)
        S = tricky_gso(gen, B, N, 0.3, True)
        S_in = S.clone()
        model.addGSO(S)
        with torch.no_grad():
            logits = model(x)
        out = dict(x=x.numpy().astype(np.uint8), S=S_in.numpy(), S_after=model.S[:, 0].numpy(), logits=logits.numpy(),
                   cfg=np.array(repr(vars(cfg))))
        for k, v in model.state_dict().items():
            out["sd/" + k] = v.numpy()
        path = os.path.join(OUT, "gnnmodel_%s.npz" % name[4:])
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KB", "max|logit| %.3g" % float(np.abs(out["logits"]).max()))


def main_gnn_dilated():
    """Round 6: DecentralPlannerNet with config.use_dilated (graphs/models/decentralplanner.py:57-86, 138-162: the dilated CNNs,
    use_dilated_version 1 and 2) - model-level fixtures made by the real reference."""
    from oracle._ref_import import import_reference_gnn_model
    cls = import_reference_gnn_model()
    cases = [("gnn_dilated_v1", dict(num_agents=9, nGraphFilterTaps=2), 1, 2),
             ("gnn_dilated_v2", dict(num_agents=11, nGraphFilterTaps=3, GSO_mode="dist_GSO_one"), 2, 3)]
    for i, (name, kw, version, B) in enumerate(cases):
        cfg = make_config(use_dilated=True, no_ReLU=False, **kw)
        cfg.use_dilated_version = version
        seed = 9393 + i
        gen = torch.Generator().manual_seed(seed)
        torch.manual_seed(seed)
        model = cls(cfg).eval()
        with torch.no_grad():
            for mod in model.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.normal_(0, 0.2, generator=gen)
                    mod.running_var.uniform_(0.5, 1.5, generator=gen)
                    mod.bias.normal_(0, 0.1, generator=gen)
        N = cfg.num_agents
        x = fov_states(gen, B, N)
        S = tricky_gso(gen, B, N, 0.3, True)
        S_in = S.clone()
        model.addGSO(S)
        with torch.no_grad():
            logits = model(x)
        out = dict(x=x.numpy().astype(np.uint8), S=S_in.numpy(), S_after=model.S[:, 0].numpy(), logits=logits.numpy(),
                   cfg=np.array(repr(vars(cfg))))
        for k, v in model.state_dict().items():
            out["sd/" + k] = v.numpy()
        path = os.path.join(OUT, "gnnmodel_%s.npz" % name[4:])
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KB", "max|logit| %.3g" % float(np.abs(out["logits"]).max()))


def main_small():
    """Round 4: layer fixtures at the PUBLISHED widths on small graphs (scripts/train_DMap.sh:42-46: 10 agents, bottleneckFeature
    32, four heads, K = 2) - the shapes the wave-per-instance one-launch kernel (csrc/gat_small.hip) takes: N <= 32,
    G = F in {32, 64}, KeyQuery; directed GSOs with threshold entries and a NaN."""
    from magat_pathplanning_amd.synthetic import directed_gso
    gml, _ = import_reference()
    cases = [(10, 32, 2, 4, 0.3, True), (20, 64, 3, 4, 0.25, False), (32, 32, 3, 2, 0.15, True), (7, 64, 2, 1, 0.5, False)]
    for si, (N, G, K, P, dens, f64) in enumerate(cases):
        seed = 7337 + 29 * si
        fx = layer_fixture(gml, "KeyQuery", N, G, K, P, seed=seed, density=dens, f64=f64,
                           gso=lambda B, N_: directed_gso(B, N_, dens, seed=seed + 1,
                                                          dtype=torch.float64 if f64 else torch.float32))
        path = os.path.join(OUT, "gat_KeyQuery_small_N%d_G%d_K%d_P%d.npz" % (N, G, K, P))
        np.savez_compressed(path, **fx)
        print("wrote", path, os.path.getsize(path) // 1024, "KB")


def main_grad():
    """Round 4: GRADIENT fixtures made by the real reference's autograd (the training step's loss.backward(),
    agents/decentralplannerlocal_OnlineExpert_GAT.py:560-567, at the layer level): for the seven shapes of
    tests/test_gpu_kernels.py::test_gat_training_backward_* - all three attention modes, concat and mean, K = 1..4 - over
    DIRECTED GSOs (synthetic.directed_gso: asymmetric masks, a one-way edge into an isolated node, threshold entries, a NaN),
    loss = sum(y * wgt) with a seeded weighting: y, dL/dx and dL/d(parameter) of GraphFilterBatchAttentional(_Origin) run
    in float64 (the reference's own code on double tensors: a reference tighter than float32 rounding, so the HIP backward
    is compared with the REFERENCE's gradients, not with this package's composite)."""
    from magat_pathplanning_amd.synthetic import directed_gso
    gml, _ = import_reference()
    cases = [("KeyQuery", True, 12, 64, 3, 2), ("KeyQuery", False, 20, 128, 2, 4), ("GAT_modified", True, 9, 32, 4, 3),
             ("KeyQuery", True, 30, 16, 1, 2), ("GAT_modified", False, 40, 128, 3, 2), ("GAT_origin", True, 14, 32, 3, 4),
             ("GAT_origin", False, 25, 64, 2, 2)]
    for si, (mode, concat, N, G, K, P) in enumerate(cases):
        seed = 8337 + 23 * si
        gen = torch.Generator().manual_seed(seed)
        torch.manual_seed(seed)
        cls = gml.GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else gml.GraphFilterBatchAttentional
        layer = cls(G, G, K, P, 1, True, concatenate=concat, attentionMode=mode)
        with torch.no_grad():
            if mode != "GAT_origin":
                layer.weight_bias.uniform_(-0.3, 0.3, generator=gen)
        layer = layer.double()          # (parameters, x and wgt are float32 values: the fixture stores them losslessly as float32)
        B = 3
        x = (torch.randn(B, G, N, generator=gen) * 0.6).double().requires_grad_(True)
        S = directed_gso(B, N, 0.3 if N <= 20 else 0.12, seed=seed + 1, dtype=torch.float64).unsqueeze(1)
        wgt = torch.randn(B, P * G if concat else G, N, generator=gen).double()
        layer.addGSO(S)
        y = layer(x)
        (y * wgt).sum().backward()
        f32 = lambda t: t.detach().numpy().astype(np.float32)
        assert all(np.array_equal(f32(t).astype(np.float64), t.detach().numpy()) for t in [x, wgt] + list(layer.parameters()))
        out = dict(x=f32(x), S=S.numpy(), wgt=f32(wgt), y=f32(y), dx=f32(x.grad),
                   mode=np.array(mode), concat=np.int64(concat), N=N, G=G, K=K, P=P)
        for k, v in layer.named_parameters():
            out["p_" + k] = f32(v)
            out["g_" + k] = f32(v.grad if v.grad is not None else torch.zeros_like(v))
        path = os.path.join(OUT, "grad_%s_%s_N%d_G%d_K%d_P%d.npz" % (mode, "concat" if concat else "mean", N, G, K, P))
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KB", {k: float(np.abs(v).max()) for k, v in out.items()
                                                                    if k.startswith("g_") or k == "dx"})


ACTIVATIONS = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
               "leaky_relu": lambda t: torch.nn.functional.leaky_relu(t, 0.1), "identity": lambda t: t}


def main_edge():
    """Round 4: the layer's two remaining constructor arguments - E > 1 edge features (the union of the E GSOs is the edge mask,
    every (head, edge feature) has its own score and tap weights, the taps of all edge features are summed:
    graphML.py:1262-1286, 1744-1775) and a nonlinearity other than ReLU (graphML.py:4654-4667) - through the reference's own
    GraphFilterBatchAttentional(_Origin).forward; directed GSOs per edge feature, concat and mean, Nin < N."""
    from magat_pathplanning_amd.synthetic import directed_gso
    gml, _ = import_reference()
    cases = [("KeyQuery", 2, "relu", 12, 16, 3, 2), ("KeyQuery", 1, "tanh", 10, 32, 2, 4), ("KeyQuery", 3, "sigmoid", 20, 128, 3, 2),
             ("GAT_modified", 2, "tanh", 14, 32, 3, 3), ("GAT_modified", 1, "leaky_relu", 9, 64, 2, 2),
             ("GAT_origin", 2, "relu", 11, 16, 3, 2), ("GAT_origin", 1, "identity", 16, 32, 2, 4)]
    for si, (mode, E, act, N, G, K, P) in enumerate(cases):
        seed = 9337 + 31 * si
        gen = torch.Generator().manual_seed(seed)
        B, F = 2, G
        cls = gml.GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else gml.GraphFilterBatchAttentional
        layers = {}
        for concat in (True, False):
            torch.manual_seed(seed)
            layers[concat] = cls(G, F, K, P, E, True, nonlinearity=ACTIVATIONS[act], concatenate=concat, attentionMode=mode)
        ref = layers[True]
        with torch.no_grad():
            if mode != "GAT_origin":
                ref.weight_bias.uniform_(-0.3, 0.3, generator=gen)
            layers[False].load_state_dict(ref.state_dict())
        x = torch.randn(B, G, N, generator=gen) * 0.7
        f64 = si % 2 == 1
        S = torch.stack([directed_gso(B, N, 0.25, seed=seed + 1 + e, dtype=torch.float64 if f64 else torch.float32)
                         for e in range(E)], dim=1)
        out = {}
        with torch.no_grad():
            for concat, lay in layers.items():
                lay.addGSO(S)
                out["y_concat" if concat else "y_mean"] = lay(x).numpy()
            out["aij"] = ref.aij.astype(np.float32)
            nin = max(1, N - 3)
            out["y_concat_nin"] = ref(x[:, :, :nin].contiguous()).numpy()
            out["nin"] = np.int64(nin)
        out.update(x=x.numpy(), S=S.numpy(), mode=np.array(mode), act=np.array(act), N=N, G=G, K=K, P=P, E=E)
        for k, v in ref.state_dict().items():
            out["p_" + k] = v.numpy()
        path = os.path.join(OUT, "edge_%s_E%d_%s_N%d_G%d_K%d_P%d.npz" % (mode, E, act, N, G, K, P))
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KB", float(np.abs(out["y_concat"]).max()))


def main_train():
    """Round 4: a TRAINING step of the whole model by the real reference (agents/decentralplannerlocal_OnlineExpert_GAT.py:
    556-567: train mode, cross-entropy, loss.backward()): logits, loss, the gradient of every parameter and the BatchNorm
    buffers behind the forward (batch statistics), run in float64 (the reference's own modules on double tensors; inputs and
    parameters are float32 values).  Dropout probabilities are set to 0 (its masks would depend on the generator of the
    device that draws them)."""
    _, classes = import_reference()
    cases = [("train_skipconcat_keyquery", 9437, 3, dict(num_agents=8, nGraphFilterTaps=3, nAttentionHeads=2, bottleneckFeature=64,
                                                          attentionMode="KeyQuery", bottleneckMode="BottomNeck_skipConcat")),
             ("train_only_slim_modified", 9471, 2, dict(num_agents=10, nGraphFilterTaps=2, nAttentionHeads=4, bottleneckFeature=32,
                                                        attentionMode="GAT_modified", bottleneckMode="BottomNeck_only",
                                                        CNN_mode="ResNetSlim_withMLP", AttentionConcat=False))]
    for name, seed, B, kw in cases:
        cfg = make_config(**kw)
        gen = torch.Generator().manual_seed(seed)
        torch.manual_seed(seed)
        model = classes[cfg.bottleneckMode](cfg).train()
        with torch.no_grad():
            for mod in model.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.weight.uniform_(0.5, 1.5, generator=gen)
                    mod.bias.normal_(0, 0.1, generator=gen)
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            if hasattr(model.GFL[0], "weight_bias"):
                model.GFL[0].weight_bias.uniform_(-0.3, 0.3, generator=gen)
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        model = model.double()
        N = cfg.num_agents
        x = fov_states(gen, B, N)
        from magat_pathplanning_amd.synthetic import comm_gso
        S = comm_gso(B, N, 20, seed=seed + 1, dtype=torch.float64)
        target = torch.randint(0, 5, (B * N,), generator=gen)
        model.addGSO(S.clone())
        logits = model(x.double())
        loss = torch.nn.functional.cross_entropy(logits, target)
        loss.backward()
        f32 = lambda t: t.detach().numpy().astype(np.float32)
        out = dict(x=x.numpy().astype(np.uint8), S=S.numpy(), target=target.numpy(), logits=f32(logits), loss=np.float64(loss.item()),
                   cfg=np.array(repr(vars(cfg))))
        for k, v in sd0.items():
            out["sd/" + k] = v.numpy()
        for k, v in model.named_parameters():
            out["g/" + k] = f32(v.grad if v.grad is not None else torch.zeros_like(v))
        for k, v in model.named_buffers():
            out["b/" + k] = v.detach().numpy().astype(np.float32 if v.dtype.is_floating_point else np.int64)
        path = os.path.join(OUT, "%s.npz" % name)
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KB", "loss", float(loss),
              "max|g|", max(float(np.abs(v).max()) for k, v in out.items() if k.startswith("g/")))


if __name__ == "__main__":
    if "--train" in sys.argv:
        main_train()
    elif "--edge" in sys.argv:
        main_edge()
    elif "--grad" in sys.argv:
        main_grad()
    elif "--small" in sys.argv:
        main_small()
    elif "--gnn-model" in sys.argv:
        main_gnn_model()
    elif "--gnn-dilated" in sys.argv:
        main_gnn_dilated()
    elif "--fov" in sys.argv:
        main_fov()
    elif "--directed" in sys.argv:
        main_directed()
    elif "--gnn" in sys.argv:
        main_gnn()
    elif "--origin" in sys.argv:
        main_origin()
    else:
        main()
