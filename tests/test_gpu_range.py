"""Range guard of the split arithmetic (include/magat_hip.h "Range guard"): the f16x3 convolutions / maps carry every value
as two half-precision planes, exact within +-65504 (the fused stem: 4094); a checkpoint whose activations leave that range
must not silently lose the 1e-4 parity.  The guard is on the device: the split kernels OR a flag when they clamp, the
encoder / the graph layer's maps re-run on the float32 MFMA kernels in the same stream predicated on it, and a status
word says so.  Tolerances are RELATIVE to the logit scale with the north star's absolute 1e-4 as the floor:
|hip - oracle| <= 1e-4 * max(1, max|oracle logits|)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scaled_model(device, scale, N=20, K=3, P=4, seed=11, where="stem"):
    """Reference-initialised weights with one BatchNorm's gamma / beta scaled: every activation behind it scales with
    it (eval-mode BN does not renormalise), like a trained checkpoint with a small running_var / large gamma would."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import make_config
    from oracle import magat_oracle as orc
    cfg = make_config(num_agents=N, nGraphFilterTaps=K, nAttentionHeads=P, bottleneckMode="BottomNeck_skipConcat",
                      device=str(device))
    sd = orc.init_state_dict(cfg, seed=seed)
    key = {"stem": "ConvLayers.0.bn1", "layer2": "ConvLayers.0.layer2.0.bn1",
           "compress": None}[where]
    if key is not None:
        sd[key + ".weight"] = sd[key + ".weight"] * scale
        sd[key + ".bias"] = sd[key + ".bias"] * scale
    else:       # the compressMLP output X feeds the graph layer's f16x3 maps
        sd["compressMLP.0.weight"] = sd["compressMLP.0.weight"] * scale
        sd["compressMLP.0.bias"] = sd["compressMLP.0.bias"] * scale
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd)
    return cfg, sd, net.to(device).eval()


def _run(net, cfg, sd, device, B=4, N=20):
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states
    from oracle import magat_oracle as orc
    x, S = fov_states(B, N, seed=5), comm_gso(B, N, 28, seed=6)
    ref = orc.planner_forward(x.double(), S.clone().double(), {k: (v.double() if v.is_floating_point() else v)
                                                                for k, v in sd.items()}, cfg).float()
    with torch.no_grad():
        net.addGSO(S.clone().to(device))
        got = net(x.to(device)).cpu()
    return got, ref


@pytest.mark.parametrize("where,scale,expect_enc,expect_gat", [
    ("stem", 1.0, False, False),          # the reference's init: nothing near the range limits
    ("stem", 3.0e3, False, False),        # activations up to ~3e3 everywhere: inside the planes' range, no re-run
    ("stem", 1.0e4, True, None),          # stem output ~1e4 > 4094 (the fused stem carries it 16x): re-run in fp32
    ("layer2", 2.0e3, False, False),      # block-internal BN, maps up to ~3e3: in range
    ("layer2", 1.0e5, True, None),        # layer2 / layer3 maps ~1e5 > 65504: re-run in fp32
    ("stem", 1.0e-4, False, False),       # tiny activations: absolute error floor of the f16 planes (3e-8 per value)
    ("compress", 1.0e4, False, False),    # graph-layer input X ~1.7e4: in range
    ("compress", 1.0e5, False, True),     # X ~1.7e5 > 65504: encoder fine, the layer's maps re-run in fp32
])
def test_scaled_checkpoints_keep_parity(gpu_device, monkeypatch, where, scale, expect_enc, expect_gat):
    """The range guard ALONE (MAGAT_ACT_SCALE=0: every plane carries its layer at the true scale): what leaves the planes' range
    is re-run in float32.  With the activation scales on (the default, next test) none of these checkpoints re-runs."""
    monkeypatch.setenv("MAGAT_ACT_SCALE", "0")
    cfg, sd, net = _scaled_model(gpu_device, scale, where=where)
    got, ref = _run(net, cfg, sd, gpu_device)
    st = net.range_status()
    lim = 1e-4 * max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    print("range case %s x%g: max|logit| %.3g, err %.3g (limit %.3g), status %s" % (where, scale, float(ref.abs().max()), err, lim, st))
    assert torch.isfinite(got).all()
    assert err <= lim, (where, scale, err, lim, st)
    assert st["encoder_rerun"] == expect_enc, st
    if expect_gat is not None:
        assert st["gat_rerun"] == expect_gat, st


@pytest.mark.parametrize("where,scale", [("stem", 1.0), ("stem", 3.0e3), ("stem", 1.0e4), ("layer2", 2.0e3), ("layer2", 1.0e5),
                                         ("stem", 1.0e-4), ("stem", 1.0e-7), ("layer2", 1.0e-5), ("compress", 1.0e4),
                                         ("compress", 1.0e5), ("compress", 1.0e-5)])
def test_activation_scales_keep_every_checkpoint_on_the_fast_path(gpu_device, where, scale):
    """Default arithmetic: the first forward calibrates (one float32 pass), the power-of-two activation scales are folded, and
    the SECOND forward - the fast path - keeps the parity without any re-run, whether the checkpoint's activations are 1e5
    or 1e-7 (include/magat_hip.h "Activation scales")."""
    cfg, sd, net = _scaled_model(gpu_device, scale, where=where)
    got0, ref = _run(net, cfg, sd, gpu_device)             # calibration pass (float32 kernels)
    assert net.range_status()["act_scales"] is not None
    got, ref = _run(net, cfg, sd, gpu_device)              # fast path with the folded scales
    st = net.range_status()
    lim = 1e-4 * max(1.0, float(ref.abs().max()))
    err, err0 = float((got - ref).abs().max()), float((got0 - ref).abs().max())
    print("act-scale case %s x%g: max|logit| %.3g, err fast %.3g calibration pass %.3g (limit %.3g), %s" % (
        where, scale, float(ref.abs().max()), err, err0, lim, st["act_scales"]))
    assert err <= lim and err0 <= lim, (where, scale, err, err0, lim)
    assert not st["encoder_rerun"], st
    # (the graph layer's input is only ever scaled UP: at |X| ~ 1.7e5 the scores are ~1e10, float32 steps of 1e3 - that
    #  regime stays with the guard's float32 form, which rounds in the reference's order)
    assert st["gat_rerun"] == (where == "compress" and scale >= 1.0e5), st


@pytest.mark.parametrize("where", ["tail", "layer1"])
@pytest.mark.parametrize("down", [1.0e-3, 1.0e-4, 1.0e-6])
@pytest.mark.parametrize("act_scale", ["1", "0"])
def test_small_activations_reamplified_later(gpu_device, monkeypatch, down, where, act_scale):
    """part of the network carried at a SMALL magnitude and re-amplified behind it, so that the logits are
    exactly those of the unscaled checkpoint (O(1)) while an absolute error floor on the small maps would be amplified into them.
      tail:   layer3's output BatchNorms (bn2, downsample) and the head's biases x `down` -> the pooled map, the head's input and
              feat are all `down` times smaller; compressMLP's weight x 1 / down undoes it.
      layer1: layer1's output BatchNorms x `down`; layer2's input-side BatchNorms (bn1, downsample) take running_mean x down,
              running_var x down^2: layer1's output map (the residual input of layer2 as well) is small, everything else is not.
    Tolerance relative to the output scale: 1e-4 * max(1, |logit|).  With the activation scales (default) the fast path holds
    it; with MAGAT_ACT_SCALE=0 the planes' 2^-25 floor is what remains (reported; not asserted)."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import make_config
    from oracle import magat_oracle as orc
    monkeypatch.setenv("MAGAT_ACT_SCALE", act_scale)
    cfg = make_config(num_agents=20, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcatGNN",
                      device=str(gpu_device))
    sd = orc.init_state_dict(cfg, seed=11)
    pre = "ConvLayers.0."
    if where == "tail":
        for bn in ("layer3.0.bn2", "layer3.0.downsample.1"):
            for k in ("weight", "bias"):
                sd[pre + bn + "." + k] = sd[pre + bn + "." + k] * down
        sd[pre + "fc.bias"] = sd[pre + "fc.bias"] * down
        sd["ConvLayers.3.bias"] = sd["ConvLayers.3.bias"] * down
        sd["compressMLP.0.weight"] = sd["compressMLP.0.weight"] / down
    else:
        for bn in ("layer1.0.bn2", "layer1.0.downsample.1"):
            for k in ("weight", "bias"):
                sd[pre + bn + "." + k] = sd[pre + bn + "." + k] * down
        for bn in ("layer2.0.bn1", "layer2.0.downsample.1"):
            sd[pre + bn + ".running_mean"] = sd[pre + bn + ".running_mean"] * down
            sd[pre + bn + ".running_var"] = sd[pre + bn + ".running_var"] * down * down
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval()
    _run(net, cfg, sd, gpu_device)
    got, ref = _run(net, cfg, sd, gpu_device)
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    print("re-amplified (%s x %g), ACT_SCALE=%s: max|logit| %.3g err %.3g (%.2g of the scale) %s" % (
        where, down, act_scale, float(ref.abs().max()), err, err / scale, net.range_status()["act_scales"]))
    assert 0.05 < float(ref.abs().max()) < 1e3
    if act_scale == "1":
        assert err <= 1e-4 * scale, (err, scale)
        assert not net.range_status()["encoder_rerun"]


def test_drift_after_calibration_still_reruns(gpu_device):
    """The scales come from the calibration batch; inputs that later drive a layer 64x beyond what was measured leave the planes'
    range again - the guard catches it as before (float32 re-run, status word)."""
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states
    from oracle import magat_oracle as orc
    cfg, sd, net = _scaled_model(gpu_device, 1.0, where="stem")
    _run(net, cfg, sd, gpu_device)
    x, S = fov_states(4, 20, seed=5) * 3000.0, comm_gso(4, 20, 28, seed=6)
    ref = orc.planner_forward(x.double(), S.clone().double(), {k: (v.double() if v.is_floating_point() else v)
                                                                for k, v in sd.items()}, cfg).float()
    with torch.no_grad():
        net.addGSO(S.clone().to(gpu_device))
        got = net(x.to(gpu_device)).cpu()
    st = net.range_status()
    assert st["encoder_rerun"], st
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_guard_off_shows_what_it_protects_from(gpu_device, libopt, monkeypatch):
    """With the guard disabled the same out-of-range checkpoint silently loses parity (the clamp is real) - and the status
    stays clear because nothing is tracked.  (MAGAT_ACT_SCALE=0: with the activation scales this checkpoint is in range.)"""
    monkeypatch.setenv("MAGAT_ACT_SCALE", "0")
    cfg, sd, net = _scaled_model(gpu_device, 1.0e4, where="stem")
    libopt.set("MAGAT_RANGE_GUARD", 0)
    got, ref = _run(net, cfg, sd, gpu_device)
    lim = 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) > lim
    libopt.reset("MAGAT_RANGE_GUARD")
    got, ref = _run(net, cfg, sd, gpu_device)
    assert float((got - ref).abs().max()) <= lim
    assert net.range_status()["encoder_rerun"]


def test_ill_conditioned_checkpoint_follows_float32(gpu_device, libopt, monkeypatch):
    """Activations in the thousands (maps of 1e3..1e5 cancelling down to logits of 3e2): inside the f16 planes' range - no
    re-run - and as accurate as the float32 arithmetic allows there.  (MAGAT_ACT_SCALE=0: every plane at its layer's true scale.)"""
    monkeypatch.setenv("MAGAT_ACT_SCALE", "0")
    cfg, sd, net = _scaled_model(gpu_device, 1000.0, where="layer2")
    got, ref = _run(net, cfg, sd, gpu_device)
    err, scale = float((got - ref).abs().max()), max(1.0, float(ref.abs().max()))
    assert not net.range_status()["encoder_rerun"]
    # ... and the bound FOLLOWS the float32 arithmetic instead of the implementation: the same checkpoint on the
    # strict float32-MFMA kernels (CONV_SPLIT = 0: the reference's own precision, another summation order) sets the yardstick.
    # The split products carry 22 significand bits where float32 carries 24 - on this ill-conditioned checkpoint (maps of
    # 1e3..1e5 cancelling down to logits of 3e2; measured 1.74e-4 of the logit scale against float32's 5.9e-5) that is the
    # whole difference: at most 4 x what float32 itself loses here (two bits), and never beyond 2e-4 of the logit scale
    libopt.set("MAGAT_CONV_SPLIT", 0)
    libopt.set("MAGAT_HEAD_F16", 0)
    got32, _ = _run(net, cfg, sd, gpu_device)
    err32 = float((got32 - ref).abs().max())
    print("ill-conditioned checkpoint: f16x3 %.3g, strict f32 %.3g of scale %.3g" % (err, err32, scale))
    assert err <= 2e-4 * scale and err <= 4.0 * max(err32, 5.0e-5 * scale), (err, err32, scale)


def test_split_gemm_reports_clamps_through_the_c_abi(gpu_device):
    """magat_conv_gemm_desc.range_flag / run_if: the f16x3 GEMM ORs the flag exactly when an input left +-65504, and the
    float32 kernel with run_if pointing at a zero word does not touch its output."""
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.encoder import split_f16x2
    lib = nat.lib()
    M, K, Nc = 256, 128, 128
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(Nc, K, generator=g) / K ** 0.5).contiguous()
    ws = split_f16x2(w)[0].to(gpu_device)
    wd, bd = w.to(gpu_device), torch.zeros(Nc, device=gpu_device)
    for big in (False, True):
        x = torch.randn(M, K, generator=g)
        if big:
            x[17, 5] = 7.0e4
        xd = x.to(gpu_device)
        flag = torch.zeros(2, dtype=torch.int32, device=gpu_device)
        out = torch.empty(M, Nc, device=gpu_device)
        d = nat.ConvGemmDesc()
        d.inp, d.wt, d.bias, d.out = xd.data_ptr(), ws.data_ptr(), bd.data_ptr(), out.data_ptr()
        d.M, d.Cin, d.lda, d.Hin, d.Win, d.kH, d.kW, d.stride, d.pad = M, K, K, 1, 1, 1, 1, 1, 0
        d.Hout, d.Wout, d.Cout, d.ldc, d.in_fmt = 1, 1, Nc, Nc, 4
        d.range_flag = flag.data_ptr()
        nat.check(lib.magat_conv_gemm_f32(ctypes.byref(d), nat.current_stream(gpu_device)), "f16x3 gemm")
        torch.cuda.synchronize()
        assert int(flag[0]) == (1 if big else 0)
        # predicated float32 kernel: runs iff the flag is set
        out2 = torch.full((M, Nc), -7.0, device=gpu_device)
        d2 = nat.ConvGemmDesc()
        d2.inp, d2.wt, d2.bias, d2.out = xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out2.data_ptr()
        d2.M, d2.Cin, d2.lda, d2.Hin, d2.Win, d2.kH, d2.kW, d2.stride, d2.pad = M, K, K, 1, 1, 1, 1, 1, 0
        d2.Hout, d2.Wout, d2.Cout, d2.ldc = 1, 1, Nc, Nc
        d2.run_if = flag.data_ptr()
        nat.check(lib.magat_conv_gemm_f32(ctypes.byref(d2), nat.current_stream(gpu_device)), "predicated f32 gemm")
        torch.cuda.synchronize()
        if big:
            ref = x.double() @ w.double().t()
            assert float((out2.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        else:
            assert bool((out2 == -7.0).all())


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf"), 1.0e5])
def test_non_finite_inputs_are_not_clamped_into_finite_numbers(gpu_device, bad):
    """the plane split's clamp is a v_med3, which maps a NaN to a finite number, and fmaxf drops NaNs from the
    running maxima - a NaN / Inf entry of the state tensor must still raise the flag (negated compares on the kernels' inputs),
    and the float32 re-run hands it on (its ReLU keeps NaN like torch.relu): the agent with the bad input gets non-finite
    logits as in the reference, every OTHER planning instance keeps its parity, and the status word says what happened.
    (Inside the affected instance the reference spreads the NaN to every agent through `aij * mask`, graphML.py:1286; the
    kernels spread it along graph edges only - not asserted.)"""
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states
    from oracle import magat_oracle as orc
    cfg, sd, net = _scaled_model(gpu_device, 1.0, where="stem")
    B, N = 4, 20
    x, S = fov_states(B, N, seed=5), comm_gso(B, N, 28, seed=6)
    with torch.no_grad():      # (a clean first forward: the calibration pass)
        net.addGSO(S.clone().to(gpu_device))
        net(x.to(gpu_device))
    x[2, 7, 1, 4, 6] = bad
    ref = orc.planner_forward(x, S.clone(), sd, cfg).view(B, N, 5)
    with torch.no_grad():
        net.addGSO(S.clone().to(gpu_device))
        got = net(x.to(gpu_device)).cpu().view(B, N, 5)
    st = net.range_status()
    assert st["encoder_rerun"], st
    finite_ref = torch.isfinite(ref[2, 7]).all()
    assert torch.isfinite(got[2, 7]).all() == finite_ref, (got[2, 7], ref[2, 7])
    if finite_ref:      # (1e5: an ordinary out-of-range value - full parity through the re-run)
        assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    keep = [0, 1, 3]
    assert torch.isfinite(got[keep]).all()
    assert float((got[keep] - ref[keep]).abs().max()) <= 1e-4
    # a clean batch afterwards: no re-run, status clear
    x[2, 7, 1, 4, 6] = 0.0
    with torch.no_grad():
        got = net(x.to(gpu_device)).cpu().view(B, N, 5)
    assert not net.range_status()["encoder_rerun"] and torch.isfinite(got).all()


@pytest.mark.parametrize("B,N", [(300, 20), (1100, 10), (40, 100)])
def test_graph_layer_rerun_with_many_instances(gpu_device, B, N):
    """The guard's float32 re-run of the graph layer walks the instances with a capped grid (a predicated launch pays for
    every workgroup it dispatches: 4096 of them cost 76 us per forward at BASELINE config 2).  A layer input beyond the planes'
    range at instance counts far above the cap: the re-run must rewrite EVERY instance's rows (checked against the oracle)."""
    from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso
    from oracle import magat_oracle as orc
    import ctypes
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(B + N)
    layer = GraphFilterBatchAttentional(128, 128, 3, 4, concatenate=True, attentionMode="KeyQuery")
    with torch.no_grad():                 # small weights: the scores stay where float32 follows the reference
        layer.weight.mul_(1e-4)
    S = comm_gso(B, N, {100: 50, 20: 28, 10: 20}[N], seed=3)
    x = torch.randn(B, 128, N, generator=g) * 0.5
    x[::7] *= 4.0e5                       # every seventh instance leaves the f16 range
    y_ref, _ = orc.gat_layer_forward(x.double(), S.unsqueeze(1).double(),
                                     {k: v.detach().double() for k, v in layer.state_dict().items()}, "KeyQuery", True)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        y = layer(x.to(gpu_device)).cpu()
    st = (ctypes.c_int32 * 2)()
    ws = layer._scratch.workspace
    nat.check(nat.lib().magat_gat_read_status(nat.ptr(ws), st, nat.current_stream(ws.device)), "magat_gat_read_status")
    assert st[0] == 1, list(st)
    scale = y_ref.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1.0)
    assert float(((y.double() - y_ref).abs() / scale).max()) <= 1e-4


def test_encoder_rerun_with_more_agent_blocks_than_the_capped_grid(gpu_device, monkeypatch):
    """The predicated float32 stem of the guard's re-run is launched with at most 256 workgroups that walk the 32-agent blocks
    (a no-op launch still pays for every workgroup it dispatches).  9 000 agents = 282 blocks with an out-of-range stem: the
    re-run must rewrite every agent (oracle on a sample of instances)."""
    monkeypatch.setenv("MAGAT_ACT_SCALE", "0")
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states
    from oracle import magat_oracle as orc
    cfg, sd, net = _scaled_model(gpu_device, 1.0e4, where="stem", N=100)
    B, N = 90, 100
    x, S = fov_states(B, N, seed=8), comm_gso(B, N, 50, seed=9)
    with torch.no_grad():
        net.addGSO(S.clone().to(gpu_device))
        got = net(x.to(gpu_device)).cpu()
    assert net.range_status()["encoder_rerun"]
    for b in (0, 41, 89):
        ref = orc.planner_forward(x[b:b + 1].double(), S[b:b + 1].clone().double(),
                                  {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, cfg).float()
        lim = 1e-4 * max(1.0, float(ref.abs().max()))
        assert float((got[b * N:(b + 1) * N] - ref).abs().max()) <= lim, b
