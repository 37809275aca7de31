"""One-launch graph layer for the published feature widths on graphs of 33 .. 128 agents (csrc/gat_mid.hip, round 6; VERDICT r05
item 3a): G = F in {32, 64}, K = 2 | 3, KeyQuery - the released F-32-P4 / B-32-P4 checkpoints on the README's 30 .. 100-robot
sets (README.md:372-390, scripts/train_DMap.sh:42-46).  Against the pinned CPU oracle over every row-tile count (N = 33 .. 128),
directed graphs, isolated agents, float64 GSOs, both merges; WHICH kernel ran is asserted (launch tag + form counter); the range
guard's float32 re-run behind it."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ONE_LAUNCH = "gat_layer (one launch)"


def _layer_and_ref(G, K, P, concat, x, S, seed):
    from magat_pathplanning_amd import GraphFilterBatchAttentional
    from oracle import magat_oracle as orc
    torch.manual_seed(seed)
    layer = GraphFilterBatchAttentional(G, G, K, P, attentionMode="KeyQuery", concatenate=concat)
    with torch.no_grad():
        layer.bias.uniform_(-0.1, 0.1)
    params = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    y_ref, _ = orc.gat_layer_forward(x, S.unsqueeze(1), params, "KeyQuery", concat)
    return layer, y_ref


@pytest.mark.parametrize("N,G,K,P,concat,f64", [(33, 32, 2, 4, False, False), (64, 64, 3, 4, True, True), (65, 32, 3, 2, True, False),
                                                (96, 64, 2, 1, False, False), (97, 32, 2, 4, False, True), (100, 32, 2, 4, False, False),
                                                (100, 64, 3, 4, True, False), (128, 64, 3, 4, True, False), (128, 32, 3, 4, False, True),
                                                (50, 32, 2, 4, False, False), (60, 64, 2, 4, True, False)])
def test_mid_layer_against_the_oracle(gpu_device, tag_counts, N, G, K, P, concat, f64):
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import directed_gso
    B = 5
    g = torch.Generator().manual_seed(N * 7 + G + K)
    x = torch.randn(B, G, N, generator=g) * 0.7
    S = torch.nan_to_num(directed_gso(B, N, 8.0 / N, seed=N + G, dtype=torch.float64 if f64 else torch.float32))
    S[0, 3, :] = 0            # an agent without out-edges (its attention row is all zeros)
    S[1, :, 5] = 0            # ... one nobody listens to
    S[2] = 0                  # an instance without any edge
    S[3, N - 1, 0] = 5e-10    # below the 1e-9 threshold: not an edge
    layer, y_ref = _layer_and_ref(G, K, P, concat, x, S, seed=N + P)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    assert nat.lib().magat_gat_one_launch_supported(N, G, G, K, 0, 1 if concat else 0)
    nat.lib().magat_form_reset()
    with torch.no_grad(), tag_counts() as tc:
        y = layer(x.to(gpu_device)).cpu()
    assert tc[ONE_LAUNCH] == 1 and tc["gat_maps_gemm"] == 0 and tc["gat_graph"] == 0, tc.counts
    assert int(nat.lib().magat_form_count(nat.FORMS["gat_mid"])) == 1
    assert tuple(y.shape) == tuple(y_ref.shape)
    err = float((y - y_ref).abs().max())
    assert err <= 1e-5 * max(1.0, float(y_ref.abs().max())), err
    # Nin < N: the layer pads the signal with zero agents and trims its output (graphML.py:4641-4646, 4670-4671)
    nin = N - 2
    from oracle import magat_oracle as orc
    params = {k: v.detach().cpu() for k, v in layer.state_dict().items()}
    y2_ref, _ = orc.gat_layer_forward(x[:, :, :nin].contiguous(), S.unsqueeze(1), params, "KeyQuery", concat)
    with torch.no_grad():
        y2 = layer(x[:, :, :nin].contiguous().to(gpu_device)).cpu()
    assert float((y2 - y2_ref).abs().max()) <= 1e-5 * max(1.0, float(y2_ref.abs().max()))


def test_mid_layer_many_instances_and_run_to_run(gpu_device):
    """More planning instances than workgroups (the persistent instance loop), twice: bit-identical, and the sampled instances
    equal the oracle's."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso
    B, N, G, K, P = 1100, 100, 32, 2, 4
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, G, N, generator=g) * 0.6
    S = comm_gso(B, N, 50, seed=3)
    layer, _ = _layer_and_ref(G, K, P, False, x[:2], S[:2], seed=9)
    params = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        y1 = layer(x.to(gpu_device)).cpu()
        y2 = layer(x.to(gpu_device)).cpu()
    assert torch.equal(y1, y2)
    pick = [0, 255, 256, 777, 1099]
    y_ref, _ = orc.gat_layer_forward(x[pick], S[pick].unsqueeze(1), params, "KeyQuery", False)
    assert float((y1[pick] - y_ref).abs().max()) <= 1e-5 * max(1.0, float(y_ref.abs().max()))


def test_mid_layer_range_guard_rerun(gpu_device):
    """Inputs beyond the f16 planes' range raise the flag; the predicated float32 form (two launches; its LDS tiles reach
    N = 128 at these widths) rewrites the output in the same stream: still the oracle's numbers."""
    from magat_pathplanning_amd.synthetic import comm_gso
    B, N, G, K, P = 3, 128, 64, 3, 4
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, G, N, generator=g) * 3.0e4          # |x| up to ~1e5 > 65504
    S = comm_gso(B, N, 50, seed=5)
    layer, y_ref = _layer_and_ref(G, K, P, True, x, S, seed=1)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        y = layer(x.to(gpu_device)).cpu()
    scale = float(y_ref.abs().max())
    assert bool(torch.isfinite(y).all())
    assert float((y - y_ref).abs().max()) <= 2e-5 * max(1.0, scale)


def test_published_checkpoint_shape_on_100_robots(gpu_device, tag_counts):
    """The whole module with the published hyper-parameters (G = F = 32, P = 4, K = 2, head mean, BottomNeck_only) on 100 agents:
    logits within 1e-4 of the oracle, the graph layer as one launch."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    from oracle import magat_oracle as orc
    B, N = 6, 100
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=4, bottleneckFeature=32, bottleneckMode="BottomNeck_only",
                      AttentionConcat=False, device=str(gpu_device))
    sd = orc.init_state_dict(cfg, seed=77)
    x, S = fov_states(B, N, seed=1), comm_gso(B, N, 50, seed=2, dtype=torch.float64)
    ref = orc.planner_forward(x, S.clone(), sd, cfg)
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval()
    with torch.no_grad():
        net.addGSO(S.clone().to(gpu_device))
        net(x.to(gpu_device))
        with tag_counts() as tc:
            net.addGSO(S.clone().to(gpu_device))
            got = net(x.to(gpu_device)).cpu()
    assert tc[ONE_LAUNCH] == 1 and tc["gat_maps_gemm"] == 0, tc.counts
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


# ---------------------------------------------------------------- 128 features on 103 .. 128 agents
@pytest.mark.parametrize("N,K,P,concat,f64,B", [(103, 3, 4, True, False, 5), (105, 2, 4, False, True, 5), (106, 3, 4, True, False, 5),
                                                (110, 3, 4, False, False, 5), (117, 2, 1, True, False, 3), (128, 3, 4, True, True, 5),
                                                (128, 3, 4, False, False, 40), (128, 2, 4, True, False, 40), (127, 3, 2, False, False, 2)])
def test_wide_layer_one_launch_up_to_128_agents(gpu_device, tag_counts, libopt, N, K, P, concat, f64, B):
    """G = F = 128 beyond gat_mfma.hip's 102 agents: the row-tile kernel with the X fragments in registers (before: two launches up
    to 105 agents, the CSR kernels above).  Small batches take its head-split form, 40 instances the plain one.  Option
    GAT_WIDE_FROM (103) moves the hand-over up: with 106 the sizes 103 .. 105 are two launches again."""
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import directed_gso
    G = 128
    if N < 106:
        libopt.set("MAGAT_GAT_WIDE_FROM", 106)
        assert not nat.lib().magat_gat_one_launch_supported(N, G, G, K, 0, 1 if concat else 0)
        libopt.restore()
    g = torch.Generator().manual_seed(N * 11 + K + P)
    x = torch.randn(B, G, N, generator=g) * 0.5
    S = torch.nan_to_num(directed_gso(B, N, 8.0 / N, seed=N + K, dtype=torch.float64 if f64 else torch.float32))
    S[0, 3, :] = 0
    S[1, :, 5] = 0
    if B > 2:
        S[2] = 0
    S[B - 1, N - 1, 0] = 5e-10
    layer, y_ref = _layer_and_ref(G, K, P, concat, x, S, seed=N + P)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    assert nat.lib().magat_gat_one_launch_supported(N, G, G, K, 0, 1 if concat else 0)
    nat.lib().magat_form_reset()
    with torch.no_grad(), tag_counts() as tc:
        y = layer(x.to(gpu_device)).cpu()
    assert tc[ONE_LAUNCH] == 1 and tc["gat_maps_gemm"] == 0 and tc["gat_graph"] == 0, tc.counts
    assert int(nat.lib().magat_form_count(nat.FORMS["gat_mid"])) == 1
    assert int(nat.lib().magat_form_count(nat.FORMS["gat_hsplit"])) == (1 if P > 1 and B * P <= 64 else 0)
    err = float((y - y_ref).abs().max())
    assert err <= 1e-5 * max(1.0, float(y_ref.abs().max())), err
    with torch.no_grad():
        y_again = layer(x.to(gpu_device)).cpu()
    assert torch.equal(y, y_again)


def test_one_launch_supported_for_every_size_up_to_128(libopt):
    """`magat_gat_one_launch_supported(N, 32 | 64 | 128, .)` for every N <= 128 (KeyQuery, K = 2 | 3); with GAT_WIDE_FROM = 106 the
    sizes 103 .. 105 at 128 features drop out - the predicate says what the dispatcher does."""
    from magat_pathplanning_amd import _native as nat
    lib = nat.lib()
    for wide_from, hole in ((None, []), (106, [103, 104, 105])):
        if wide_from is not None:
            libopt.set("MAGAT_GAT_WIDE_FROM", wide_from)
        for G in (32, 64, 128):
            for K in (2, 3):
                for concat in (0, 1):
                    missing = [N for N in range(1, 129) if not lib.magat_gat_one_launch_supported(N, G, G, K, 0, concat)]
                    assert missing == (hole if G == 128 else []), (G, K, concat, missing)
                assert not lib.magat_gat_one_launch_supported(129, G, G, K, 0, 1)


@pytest.mark.parametrize("N,K,concat", [(104, 3, True), (106, 3, True), (128, 3, False), (128, 2, True), (120, 3, False)])
def test_wide_layer_range_guard_rerun(gpu_device, tag_counts, libopt, N, K, concat):
    """Inputs beyond the f16 planes' range at the sizes where gat_dense_kernel's tiles no longer fit (N >= 106): the predicated
    gat_slim_kernel rewrites the output - still the oracle's numbers, the re-run counted once, and a sane forward afterwards
    runs no float32 work."""
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso
    B, G, P = 3, 128, 4
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(B, G, N, generator=g) * 3.0e4
    S = comm_gso(B, N, 50, seed=5)
    S[1, 7, :] = 0
    layer, y_ref = _layer_and_ref(G, K, P, concat, x, S, seed=1)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        y = layer(x.to(gpu_device)).cpu()
    scale = float(y_ref.abs().max())
    assert bool(torch.isfinite(y).all())
    assert float((y - y_ref).abs().max()) <= 2e-5 * max(1.0, scale)
    def status():
        import ctypes
        st = (ctypes.c_int32 * 2)()
        ws = layer._scratch.workspace
        with torch.cuda.device(ws.device):
            nat.check(nat.lib().magat_gat_read_status(nat.ptr(ws), st, nat.current_stream(ws.device)), "magat_gat_read_status")
        return int(st[0]), int(st[1])
    st = status()
    assert st[0] == 1 and st[1] == 1, st      # this forward's flag, re-runs so far
    x_ok = torch.randn(B, G, N, generator=g) * 0.5
    _, y_ok_ref = _layer_and_ref(G, K, P, concat, x_ok, S, seed=1)
    with torch.no_grad():
        y_ok = layer(x_ok.to(gpu_device)).cpu()
    assert float((y_ok - y_ok_ref).abs().max()) <= 1e-5 * max(1.0, float(y_ok_ref.abs().max()))
    assert status() == (0, 1)


@pytest.mark.parametrize("N,B", [(128, 6), (120, 1), (106, 40)])
def test_whole_model_on_106_to_128_agents_is_one_graph_launch(gpu_device, tag_counts, N, B):
    """The default module (128 features, K = 3, P = 4, skip-concat) on more than 105 agents: the graph layer is ONE launch (no
    CSR structure is built at addGSO any more), logits within 1e-4 of the oracle; the step plan and the general path agree bit
    for bit."""
    from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    from oracle import magat_oracle as orc
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device=str(gpu_device))
    sd = orc.init_state_dict(cfg, seed=N)
    x, S = fov_states(B, N, seed=1), comm_gso(B, N, 50, seed=2, dtype=torch.float64)
    ref = orc.planner_forward(x, S.clone(), sd, cfg)
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval()
    dx = x.to(gpu_device)
    with torch.no_grad():
        net.addGSO(S.clone().to(gpu_device))
        first = net(dx).cpu()                      # general path (allocates, calibrates)
        nat.lib().magat_form_reset()
        with tag_counts() as tc:
            net.addGSO(S.clone().to(gpu_device))
            got = net(dx).cpu()                    # step plan
    assert tc[ONE_LAUNCH] == 1 and tc["gat_maps_gemm"] == 0 and tc["gso_to_csr"] == 0, tc.counts
    assert int(nat.lib().magat_form_count(nat.FORMS["gat_mid"])) == 1
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert torch.equal(first, got)
    st = net.range_status()
    assert not st["gat_rerun"] and not st["encoder_rerun"]


def test_step_plan_survives_a_route_changing_option(gpu_device, tag_counts, libopt):
    """A step plan built while 128 agents ran as one graph launch, then GAT_WIDE_FROM = 129 (never): the next forward drops the
    plan, builds the CSR structure and runs the CSR kernels - same logits to float32 accuracy - and back again."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 3, 128
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device=str(gpu_device))
    torch.manual_seed(5)
    net = DecentralPlannerGATNet(cfg).to(gpu_device).eval()
    x, S = fov_states(B, N, seed=1).to(gpu_device), comm_gso(B, N, 50, seed=2, dtype=torch.float64).to(gpu_device)
    with torch.no_grad():
        for _ in range(2):
            net.addGSO(S.clone())
            one = net(x).cpu()
        libopt.set("MAGAT_GAT_WIDE_FROM", 129)
        with tag_counts() as tc:
            net.addGSO(S.clone())
            csr = net(x).cpu()
        assert tc[ONE_LAUNCH] == 0, tc.counts
        libopt.restore()
        with tag_counts() as tc:
            net.addGSO(S.clone())
            back = net(x).cpu()
        assert tc[ONE_LAUNCH] == 1, tc.counts
    assert float((one - csr).abs().max()) <= 2e-5 * max(1.0, float(one.abs().max()))
    assert torch.equal(one, back)


@pytest.mark.parametrize("N,G,K,P,concat,B,f64", [(10, 32, 2, 4, False, 1, True), (20, 64, 3, 4, True, 2, False), (100, 32, 2, 4, False, 1, False),
                                                  (100, 128, 3, 4, True, 1, True), (64, 128, 2, 2, False, 3, False), (33, 64, 3, 1, False, 5, False),
                                                  (7, 128, 3, 4, False, 16, False)])
def test_few_instances_rerun_is_one_launch(gpu_device, tag_counts, N, G, K, P, concat, B, f64):
    """The range guard's float32 re-run behind a one-launch graph kernel, few instances (instances x heads <= 64: the closed-loop
    step): ONE predicated launch (gat_rerun_small_kernel forms the maps itself) instead of maps GEMM + graph kernel (+ head
    mean).  With inputs beyond the f16 planes' range it rewrites the output - the oracle's numbers, the re-run counted; with
    sane inputs it returns at once and the status says so."""
    import ctypes
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import directed_gso
    g = torch.Generator().manual_seed(N + G + K)
    x_big = torch.randn(B, G, N, generator=g) * 3.0e4
    x_ok = torch.randn(B, G, N, generator=g) * 0.5
    S = torch.nan_to_num(directed_gso(B, N, min(1.0, 8.0 / N), seed=N, dtype=torch.float64 if f64 else torch.float32))
    layer, y_big_ref = _layer_and_ref(G, K, P, concat, x_big, S, seed=3)
    _, y_ok_ref = _layer_and_ref(G, K, P, concat, x_ok, S, seed=3)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))

    def status():
        st = (ctypes.c_int32 * 2)()
        ws = layer._scratch.workspace
        with torch.cuda.device(ws.device):
            nat.check(nat.lib().magat_gat_read_status(nat.ptr(ws), st, nat.current_stream(ws.device)), "magat_gat_read_status")
        return int(st[0]), int(st[1])
    with torch.no_grad(), tag_counts() as tc:
        y_big = layer(x_big.to(gpu_device)).cpu()
    assert tc[ONE_LAUNCH] == 1 and tc["range_guard"] == 1 and tc["gat_maps_gemm"] == 0 and tc["gat_graph"] == 0 and tc["head_mean"] == 0, tc.counts
    assert bool(torch.isfinite(y_big).all())
    assert float((y_big - y_big_ref).abs().max()) <= 2e-5 * max(1.0, float(y_big_ref.abs().max()))
    assert status() == (1, 1)
    with torch.no_grad(), tag_counts() as tc:
        y_ok = layer(x_ok.to(gpu_device)).cpu()
    assert tc[ONE_LAUNCH] == 1 and tc["range_guard"] == 1, tc.counts
    assert float((y_ok - y_ok_ref).abs().max()) <= 1e-5 * max(1.0, float(y_ok_ref.abs().max()))
    assert status() == (0, 1)
