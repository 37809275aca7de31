"""Training of the per-agent CNN on the HIP convolution kernels (train_cnn.py, csrc/conv_train.hip) against torch's own
float64 autograd of the same layers on the CPU: convolution forward / input gradient / weight gradient at every geometry the
ResNet trunks use, the whole trunk in training mode (batch-statistics BatchNorm, running-stat updates), and a training step of
the planner with either convolution backend."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as tnf

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


# (Cin, Cout, k, stride, pad, H)   stem; layer1.conv1 (stride 2); its 1x1 downsample; the 3x3s of the chain; the head's 1x1
GEOMS = [(3, 32, 3, 1, 1, 11), (32, 32, 3, 2, 1, 11), (32, 32, 1, 2, 0, 11), (32, 64, 3, 1, 1, 6), (64, 64, 3, 1, 1, 6),
         (64, 128, 1, 1, 0, 6), (128, 128, 3, 1, 1, 6), (128, 128, 1, 1, 0, 3), (32, 32, 3, 2, 1, 12)]


@pytest.mark.parametrize("M", [77, 150])
@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "c%d_%d_k%d_s%d_p%d_h%d" % g)
def test_conv_forward_dgrad_wgrad_match_torch(gpu_device, geom, M):
    from magat_pathplanning_amd.train_cnn import _ConvPixelMajor, _pad4
    cin, cout, k, s, p, H = geom
    g = torch.Generator().manual_seed(11 + cin + cout + k + s + M)
    x = torch.randn(M, cin, H, H, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.2
    ho = (H + 2 * p - k) // s + 1
    wgt = torch.randn(M, cout, ho, ho, generator=g)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = tnf.conv2d(x64, w64, None, s, p)
    (y64 * wgt.double()).sum().backward()
    xd = _pad4(x.permute(2, 3, 0, 1).reshape(H * H, M, cin)).contiguous().to(gpu_device).requires_grad_(True)
    wd = w.to(gpu_device).requires_grad_(True)
    y = _ConvPixelMajor.apply(xd, wd, H, H, s, p)
    assert tuple(y.shape) == (ho * ho, M, cout)
    (y * wgt.permute(2, 3, 0, 1).reshape(ho * ho, M, cout).to(gpu_device)).sum().backward()
    ynchw = y.detach().cpu().view(ho, ho, M, cout).permute(2, 3, 0, 1)
    assert _rel(ynchw.double(), y64.detach()) < 2e-6
    dx = xd.grad.cpu()[..., :cin].view(H, H, M, cin).permute(2, 3, 0, 1)
    assert _rel(dx.double(), x64.grad) < 2e-6
    assert float(xd.grad[..., cin:].abs().max()) == 0.0 if cin % 4 else True
    assert _rel(wd.grad.cpu().double(), w64.grad) < 5e-6


@pytest.mark.parametrize("slim", [False, True], ids=["ResNetLarge", "ResNetSlim"])
def test_resnet_training_step_matches_torch(gpu_device, slim):
    """Trunk in TRAINING mode: output, the gradient of every parameter and of the input, and the BatchNorm running statistics
    after the step against the module's own torch forward in float64 on the CPU."""
    from magat_pathplanning_amd.resnet import ResNet, ResNetSlim
    from magat_pathplanning_amd.train_cnn import resnet_forward
    torch.manual_seed(3)
    body = (ResNetSlim if slim else ResNet)()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in body.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5, generator=g)
                m.bias.normal_(0, 0.2, generator=g)
    M = 45
    x = (torch.rand(M, 3, 11, 11, generator=g) < 0.3).float() + 0.1 * torch.randn(M, 3, 11, 11, generator=g)
    ref = copy.deepcopy(body).double().train()
    dev = copy.deepcopy(body).to(gpu_device).train()
    x64 = x.double().requires_grad_(True)
    y64 = ref(x64)
    wgt = torch.randn(y64.shape, generator=g)
    (y64 * wgt.double()).sum().backward()
    xd = x.to(gpu_device).requires_grad_(True)
    y = resnet_forward(dev, xd)
    assert tuple(y.shape) == tuple(y64.shape)
    (y * wgt.to(gpu_device)).sum().backward()
    assert _rel(y.detach().cpu().double(), y64.detach()) < 1e-5
    assert _rel(xd.grad.cpu().double(), x64.grad) < 1e-4
    pr = dict(ref.named_parameters())
    for k, v in dev.named_parameters():
        assert v.grad is not None, k
        assert _rel(v.grad.cpu().double(), pr[k].grad) < 2e-4, k
    br = dict(ref.named_buffers())
    for k, v in dev.named_buffers():
        if v.dtype.is_floating_point:
            assert _rel(v.cpu().double(), br[k]) < 1e-5, k
        else:
            assert int(v) == int(br[k]), k
    # evaluation mode under autograd: running statistics, no updates
    dev.eval(); ref.eval()
    y = resnet_forward(dev, x.to(gpu_device))
    assert _rel(y.detach().cpu().double(), ref(x.double()).detach()) < 1e-5


def test_planner_training_step_hip_convolutions_vs_torch_convolutions(gpu_device, monkeypatch):
    """loss.backward() through DecentralPlannerGATNet in training mode (agents/..._GAT.py:556-567): the HIP convolution
    backend (default) against torch's (MAGAT_TRAIN_CNN=torch) from identical weights - logits, every gradient, the BatchNorm
    buffers - and one optimiser step of each ends in the same parameters."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 6, 10
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
    torch.manual_seed(21)
    base = DecentralPlannerGATNet(cfg)
    for m in base.modules():              # (Dropout draws from the device generator: the two runs must see the same stream)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    x = fov_states(B, N, seed=5).to(gpu_device)
    S = comm_gso(B, N, 20, seed=6).to(gpu_device)
    tgt = torch.randint(0, 5, (B * N,), generator=torch.Generator().manual_seed(7)).to(gpu_device)
    res = {}
    for backend in ("hip", "torch"):
        monkeypatch.setenv("MAGAT_TRAIN_CNN", backend)
        net = copy.deepcopy(base).to(gpu_device).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.05)
        net.addGSO(S.clone())
        logits = net(x)
        loss = tnf.cross_entropy(logits, tgt)
        opt.zero_grad()
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}
        opt.step()
        res[backend] = (logits.detach(), grads, {k: v.detach().clone() for k, v in net.state_dict().items()})
    lh, gh, sh = res["hip"]
    lt, gt, st = res["torch"]
    assert _rel(lh, lt) < 1e-4
    assert gh.keys() == gt.keys() and any(k.startswith("ConvLayers.0.layer3") for k in gh)
    for k in gh:
        assert _rel(gh[k], gt[k]) < 2e-3, k
    for k in sh:
        if sh[k].dtype.is_floating_point:
            assert _rel(sh[k], st[k]) < 1e-4, k


def test_default_cnn_training_step_matches_torch(gpu_device):
    """CNN_mode 'Default' (conv3x3 + bias, BatchNorm, ReLU, MaxPool2d stacks) in training mode on the HIP convolution kernels
    against the same nn.Sequential in float64 on the CPU: output, every gradient, the BatchNorm buffers."""
    import torch.nn as nn
    from magat_pathplanning_amd.train_cnn import conv_stack_forward, _is_conv_stack
    torch.manual_seed(8)
    chans = [3, 32, 32, 64, 64, 128]
    layers = []
    for l in range(5):
        layers += [nn.Conv2d(chans[l], chans[l + 1], 3, 1, 1, bias=True), nn.BatchNorm2d(chans[l + 1]), nn.ReLU(inplace=True)]
        if l % 2 == 0:
            layers.append(nn.MaxPool2d(kernel_size=2))
    seq = nn.Sequential(*layers)
    assert _is_conv_stack(seq)
    g = torch.Generator().manual_seed(9)
    M = 37
    x = (torch.rand(M, 3, 11, 11, generator=g) < 0.3).float() + 0.1 * torch.randn(M, 3, 11, 11, generator=g)
    ref = copy.deepcopy(seq).double().train()
    dev = copy.deepcopy(seq).to(gpu_device).train()
    x64 = x.double().requires_grad_(True)
    y64 = ref(x64)
    wgt = torch.randn(y64.shape, generator=g)
    (y64 * wgt.double()).sum().backward()
    xd = x.to(gpu_device).requires_grad_(True)
    y = conv_stack_forward(dev, xd)
    assert tuple(y.shape) == tuple(y64.shape)
    (y * wgt.to(gpu_device)).sum().backward()
    assert _rel(y.detach().cpu().double(), y64.detach()) < 1e-5
    assert _rel(xd.grad.cpu().double(), x64.grad) < 2e-4
    pr = dict(ref.named_parameters())
    gmax = max(float(p_.grad.abs().max()) for p_ in pr.values())
    for k, v in dev.named_parameters():
        # (a convolution's bias in front of a training-mode BatchNorm has a gradient of exactly zero in exact arithmetic:
        #  an absolute floor relative to the largest gradient of the stack)
        err = float((v.grad.cpu().double() - pr[k].grad).abs().max())
        assert err < 5e-4 * float(pr[k].grad.abs().max()) + 1e-6 * gmax, (k, err)
    br = dict(ref.named_buffers())
    for k, v in dev.named_buffers():
        if v.dtype.is_floating_point:
            assert _rel(v.cpu().double(), br[k]) < 1e-5, k


def test_batchnorm_training_statistics_with_a_large_mean(gpu_device):
    """channels with |mean| >> std (a conv bias in front of the BatchNorm: CNN_mode Default, or drifting activations).
    The sums are taken of (x - pivot) with a per-channel pivot from the data, so the variance does not cancel: mean / std = 1e3
    here, where E[x^2] - mean^2 in float32 partials is off by ~10 % in the variance.  Against float64 on the same float32 values;
    the bounds left are the float32 rounding of the saved mean (6e-5 of a standard deviation)."""
    from magat_pathplanning_amd.train_cnn import _BatchNormTrain
    rows, C = 23040, 64
    g = torch.Generator().manual_seed(77)
    x = torch.randn(rows, C, generator=g) * (torch.rand(C, generator=g) + 0.5) + 1000.0
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm, rv = torch.zeros(C), torch.ones(C)
    x64 = x.double()
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y64 = tnf.batch_norm(x64, rm64, rv64, gamma.double(), beta.double(), True, 0.1, 1e-5)
    rmd, rvd = rm.to(gpu_device), rv.to(gpu_device)
    y = _BatchNormTrain.apply(x.to(gpu_device), gamma.to(gpu_device), beta.to(gpu_device), rmd, rvd, 0.1, 1e-5, False)
    assert _rel(rvd.cpu().double(), rv64) < 1e-5          # (was ~1e-1 relative in the variance with unshifted float32 sums)
    assert _rel(rmd.cpu().double(), rm64) < 1e-6
    assert float((y.cpu().double() - y64).abs().max()) < 5e-4


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("rows,C", [(77 * 36, 32), (1001, 64), (23040, 128), (5, 128)])
def test_batchnorm_training_kernels_match_torch(gpu_device, rows, C, relu):
    """magat_bn_train_{forward,backward}_f32 through train_cnn._BatchNormTrain against torch.nn.functional.batch_norm (+ relu) in
    float64: output, dx / dgamma / dbeta, and the running statistics (momentum 0.1, unbiased variance)."""
    from magat_pathplanning_amd.train_cnn import _BatchNormTrain
    g = torch.Generator().manual_seed(rows + C + int(relu))
    x = torch.randn(rows, C, generator=g) * 1.7 + 0.4
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    wgt = torch.randn(rows, C, generator=g)
    x64, g64, b64 = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y64 = tnf.batch_norm(x64, rm64, rv64, g64, b64, True, 0.1, 1e-5)
    y64 = torch.relu(y64) if relu else y64
    (y64 * wgt.double()).sum().backward()
    xd, gd, bd = (t.to(gpu_device).requires_grad_(True) for t in (x, gamma, beta))
    rmd, rvd = rm.to(gpu_device), rv.to(gpu_device)
    y = _BatchNormTrain.apply(xd, gd, bd, rmd, rvd, 0.1, 1e-5, relu)
    (y * wgt.to(gpu_device)).sum().backward()
    assert _rel(y.detach().cpu().double(), y64.detach()) < 2e-6
    assert _rel(xd.grad.cpu().double(), x64.grad) < 2e-5
    assert _rel(gd.grad.cpu().double(), g64.grad) < 2e-5
    assert _rel(bd.grad.cpu().double(), b64.grad) < 2e-5
    assert _rel(rmd.cpu().double(), rm64) < 1e-6 and _rel(rvd.cpu().double(), rv64) < 1e-6


def test_gnn_baseline_training_step_hip_vs_torch_convolutions(gpu_device, monkeypatch):
    """DecentralPlannerNet (the GNN-baseline model class, graphs/models/decentralplanner.py) in training mode: its CNN trains on
    the HIP kernels too - logits and gradients against torch's convolutions from identical weights."""
    from magat_pathplanning_amd import DecentralPlannerNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 4, 10
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, device="cuda:0")
    torch.manual_seed(31)
    base = DecentralPlannerNet(cfg)
    for m in base.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    x = fov_states(B, N, seed=5).to(gpu_device)
    S = comm_gso(B, N, 20, seed=6).to(gpu_device)
    tgt = torch.randint(0, 5, (B * N,), generator=torch.Generator().manual_seed(7)).to(gpu_device)
    res = {}
    for backend in ("hip", "torch"):
        monkeypatch.setenv("MAGAT_TRAIN_CNN", backend)
        net = copy.deepcopy(base).to(gpu_device).train()
        net.addGSO(S.clone())
        logits = net(x)
        tnf.cross_entropy(logits, tgt).backward()
        res[backend] = (logits.detach(), {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None})
    assert _rel(res["hip"][0], res["torch"][0]) < 1e-4
    assert res["hip"][1].keys() == res["torch"][1].keys()
    for k in res["hip"][1]:
        assert _rel(res["hip"][1][k], res["torch"][1][k]) < 2e-3, k


def test_training_backend_is_chosen_by_batch_size(gpu_device, monkeypatch):
    """MAGAT_TRAIN_CNN=auto (the default): torch's convolutions below train_cnn.TRAIN_HIP_MIN_AGENTS agents - the reference's own
    training batch (64 x 10 agents, scripts/train_DMap.sh:30-46) is launch-bound and must not get slower by dropping this
    module in - the HIP convolution kernels from there on; hip / torch force one side."""
    from magat_pathplanning_amd import train_cnn as tc
    small = torch.zeros(640, 3, 11, 11, device=gpu_device)
    large = torch.zeros(tc.TRAIN_HIP_MIN_AGENTS, 3, 11, 11, device=gpu_device)
    monkeypatch.delenv("MAGAT_TRAIN_CNN", raising=False)
    assert not tc._use_hip_convs(small) and tc._use_hip_convs(large) and not tc._use_hip_convs(large.cpu())
    monkeypatch.setenv("MAGAT_TRAIN_CNN", "hip")
    assert tc._use_hip_convs(small)
    monkeypatch.setenv("MAGAT_TRAIN_CNN", "torch")
    assert not tc._use_hip_convs(large)


def test_every_model_variant_trains_and_infers(gpu_device, monkeypatch):
    """One training step (cross-entropy, backward) and one inference forward of every (skip variant, CNN_mode, attention mode)
    combination of DecentralPlannerGATNet and every CNN_mode of DecentralPlannerNet: finite logits and gradients, nothing raises
    (this sweep found the ResNetLarge / ResNetSlim trunks handing out a non-contiguous map and the GNN-baseline class refusing
    to train on the GPU)."""
    import itertools
    from magat_pathplanning_amd import DecentralPlannerGATNet, DecentralPlannerNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    monkeypatch.setenv("MAGAT_TRAIN_CNN", "hip")      # (21 agents: `auto` would take torch's convolutions)
    B, N = 3, 7
    x = fov_states(B, N, seed=5).to(gpu_device)
    S = comm_gso(B, N, 20, seed=6).to(gpu_device)
    tgt = torch.randint(0, 5, (B * N,), generator=torch.Generator().manual_seed(2)).to(gpu_device)
    cnns = ["ResNetLarge_withMLP", "ResNetSlim_withMLP", "ResNetLarge", "ResNetSlim", "Default"]
    for skip, cnn, att in itertools.product(["BottomNeck_only", "BottomNeck_skipConcat", "BottomNeck_skipConcatGNN",
                                             "BottomNeck_skipAddGNN", ""], cnns, ["KeyQuery", "GAT_origin"]):
        concat = skip != "BottomNeck_skipAddGNN"            # (that variant ADDS the bottleneck feature: head mean, as the reference needs)
        cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=2, bottleneckMode=skip, CNN_mode=cnn, attentionMode=att,
                          AttentionConcat=concat, device="cuda:0")
        torch.manual_seed(1)
        net = DecentralPlannerGATNet(cfg).to(gpu_device).train()
        net.addGSO(S.clone())
        tnf.cross_entropy(net(x), tgt).backward()
        assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None), (skip, cnn, att)
        net.eval()
        with torch.no_grad():
            net.addGSO(S.clone())
            assert torch.isfinite(net(x)).all(), (skip, cnn, att)
    for cnn in cnns:
        cfg = make_config(num_agents=N, nGraphFilterTaps=3, CNN_mode=cnn, device="cuda:0")
        net = DecentralPlannerNet(cfg).to(gpu_device).train()
        net.addGSO(S.clone())
        tnf.cross_entropy(net(x), tgt).backward()
        assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None), cnn


@pytest.mark.parametrize("backend", ["hip", "auto"])
def test_inference_sees_what_training_changed(gpu_device, monkeypatch, backend):
    """Between training steps the inference path must run on the module's CURRENT parameters and BatchNorm statistics - also when
    only the buffers moved (a training-mode forward without an optimiser step: the HIP BatchNorm kernels update the running
    statistics in place and bump their version counters, which planner._weights_key watches): inference logits against the
    oracle evaluated on the state_dict of the moment."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    monkeypatch.setenv("MAGAT_TRAIN_CNN", backend)     # (40 agents: `auto` = torch's convolutions, `hip` = the HIP kernels)
    B, N = 4, 10
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=2, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
    torch.manual_seed(5)
    net = DecentralPlannerGATNet(cfg).to(gpu_device)
    x, S = fov_states(B, N, seed=1), comm_gso(B, N, 20, seed=2)
    tgt = torch.randint(0, 5, (B * N,), generator=torch.Generator().manual_seed(3)).to(gpu_device)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    for step in range(4):
        net.eval()
        with torch.no_grad():
            net.addGSO(S.clone().to(gpu_device))
            got = net(x.to(gpu_device)).cpu().numpy()
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        ref = orc.planner_forward(x, S.clone(), sd, cfg).numpy()
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max())), step
        net.train()
        net.addGSO(S.clone().to(gpu_device))
        if step == 1:           # buffers only
            with torch.no_grad():
                net(x.to(gpu_device))
        else:
            loss = tnf.cross_entropy(net(x.to(gpu_device)), tgt)
            opt.zero_grad()
            loss.backward()
            opt.step()
