"""The one-launch matrix-core form of the KeyQuery graph-attention layer (csrc/gat_mfma.hip) against the oracle
(oracle/magat_oracle.py, restating utils/graphUtils/graphML.py:4636-4671 / 1724-1827), and against the two-launch form."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer_and_inputs(B, N, K, P, concat, seed, density=None, bias=True):
    from magat_pathplanning_amd import GraphFilterBatchAttentional
    from magat_pathplanning_amd.synthetic import comm_gso, random_gso
    torch.manual_seed(1000 + seed)          # (the layer's initialisation draws from the global generator)
    g = torch.Generator().manual_seed(seed)
    G = 128
    layer = GraphFilterBatchAttentional(G, G, K, P, bias=bias, concatenate=concat, attentionMode="KeyQuery")
    if density is None:
        S = comm_gso(B, N, {100: 50, 20: 28, 10: 20}.get(N, 40), seed=seed)
    else:
        S = random_gso(B, N, density, seed=seed)
    x = torch.randn(B, G, N, generator=g) * 0.7
    return layer, S, x


@pytest.mark.parametrize("N,K,P,concat,dt", [(100, 3, 4, True, torch.float64), (100, 3, 4, False, torch.float32),
                                              (101, 2, 2, True, torch.float32), (97, 3, 1, True, torch.float64),
                                              (64, 3, 4, True, torch.float32), (50, 2, 4, False, torch.float64),
                                              (33, 3, 2, True, torch.float32), (32, 3, 4, True, torch.float32),
                                              (20, 3, 4, True, torch.float64), (10, 2, 1, True, torch.float32),
                                              (1, 3, 2, True, torch.float32), (17, 3, 4, False, torch.float32),
                                              (70, 3, 4, True, torch.float32)])
def test_gat_mfma_vs_oracle(gpu_device, libopt, N, K, P, concat, dt):
    from oracle import magat_oracle as orc
    B = 5
    layer, S, x = _layer_and_inputs(B, N, K, P, concat, seed=100 + N)
    S = S.to(dt)
    y_ref, _ = orc.gat_layer_forward(x, S.unsqueeze(1).float(), {k: v.detach() for k, v in layer.state_dict().items()},
                                     "KeyQuery", concat)
    layer = layer.to(gpu_device).eval()
    out = {}
    for fused in (1, 0):
        libopt.set("GAT_MFMA", fused)
        layer.addGSO(S.unsqueeze(1).to(gpu_device))
        with torch.no_grad():
            out[fused] = layer(x.to(gpu_device)).cpu()
    np.testing.assert_allclose(out[1].numpy(), y_ref.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(out[1].numpy(), out[0].numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("density", [0.0, 1.0, 0.3])
def test_gat_mfma_dense_empty_and_mixed_graphs(gpu_device, density):
    """Fully connected graphs (every softmax row has N entries), empty graphs (rows without edges are exact zeros, not NaN:
    the output is relu(U_0 + bias)) and random ones."""
    from oracle import magat_oracle as orc
    B, N, K, P = 3, 100, 3, 4
    layer, S, x = _layer_and_inputs(B, N, K, P, True, seed=7, density=density)
    y_ref, _ = orc.gat_layer_forward(x, S.unsqueeze(1), {k: v.detach() for k, v in layer.state_dict().items()},
                                     "KeyQuery", True)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        y = layer(x.to(gpu_device)).cpu()
    assert torch.isfinite(y).all()
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=0, atol=3e-5)


def test_gat_mfma_many_instances_per_workgroup(gpu_device):
    """More instances than compute units: every workgroup walks several instances (the X planes, masks and the Q / U^T
    region are rebuilt per instance); results must not depend on the position in the walk."""
    B, N, K, P = 700, 20, 3, 4
    layer, S, x = _layer_and_inputs(B, N, K, P, True, seed=3)
    S[350:] = S[:350]
    x[350:] = x[:350]
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        y = layer(x.to(gpu_device)).cpu()
    assert torch.equal(y[:350], y[350:])


def test_gat_mfma_range_guard_reruns_in_float32(gpu_device):
    """Features beyond the f16 range: the fused kernel raises the flag, the predicated float32 two-launch form
    re-writes Y in the same stream (magat_hip.h, range guard)."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.graphml import gat_forward_rows
    B, N, K, P = 4, 100, 3, 4
    layer, S, x = _layer_and_inputs(B, N, K, P, True, seed=21)
    x[1, 5, 7] = 9.0e4
    with torch.no_grad():
        # scores through a 9e4 feature are ill-conditioned in float32 on ANY path (their differences sit below the float32
        # resolution at that magnitude): the key / query weights are scaled far down so that the comparison tests the
        # re-run, not the conditioning of the softmax
        for prm in layer.parameters():
            prm.mul_(0.05)
        layer.weight.mul_(1e-6)
    y_ref, _ = orc.gat_layer_forward(x, S.unsqueeze(1), {k: v.detach() for k, v in layer.state_dict().items()},
                                     "KeyQuery", True)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        y = layer(x.to(gpu_device)).cpu()
    scale = float(y_ref.abs().max())
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=0, atol=2e-6 * max(scale, 1.0))


ONE_LAUNCH = "gat_layer (one launch)"


@pytest.mark.parametrize("N,K,P,concat,dt,density", [
    (100, 3, 4, True, torch.float64, 0.05), (100, 3, 4, False, torch.float32, 0.08), (102, 3, 4, True, torch.float64, 0.05),
    (102, 2, 2, False, torch.float64, 0.3), (103, 3, 4, True, torch.float64, 0.05), (107, 3, 4, True, torch.float64, 0.05), (64, 3, 4, True, torch.float32, 0.1),
    (65, 2, 3, True, torch.float64, 0.1), (33, 3, 2, False, torch.float64, 0.2), (20, 3, 4, True, torch.float64, 0.25),
    (12, 3, 4, True, torch.float32, 0.3), (6, 2, 1, True, torch.float64, 0.5)])
def test_gat_mfma_directed_graphs_vs_oracle(gpu_device, tag_counts, N, K, P, concat, dt, density):
    """Directed masks (M != M^T) through the DEFAULT kernel: the attention is a row softmax (row i over its out-edges j) that
    the hops apply column-wise (z_k[j] = sum_i a_ij z_{k-1}[i], graphML.py:1757, 1274-1286) - a transposed mask or hop would
    pass every symmetric-GSO test.  synthetic.directed_gso adds a one-way edge into an otherwise isolated node, threshold
    entries (5e-10 / -3e-9), a NaN and float64 1/lambda_max values.  The profiling tag asserts which kernel ran: the
    one-launch matrix-core kernel up to N = 102, the row-tile one-launch kernel from N = 103 (the hand-over)."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import directed_gso
    B = 4
    layer, _, x = _layer_and_inputs(B, N, K, P, concat, seed=300 + N)
    S = directed_gso(B, N, density, seed=17 + N, dtype=dt)
    assert N < 6 or not torch.equal(torch.nan_to_num(S) != 0, torch.nan_to_num(S).transpose(1, 2) != 0)
    y_ref, a_ref = orc.gat_layer_forward(x, S.unsqueeze(1), {k: v.detach() for k, v in layer.state_dict().items()},
                                         "KeyQuery", concat)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    nat.lib().magat_form_reset()
    with tag_counts() as tc, torch.no_grad():
        y = layer(x.to(gpu_device)).cpu()
    expect_one = bool(nat.lib().magat_gat_one_launch_supported(N, 128, 128, K, nat.MODE_KEYQUERY, int(concat)))
    # (round 6: every N <= 128 - gat_mfma.hip up to 102 agents, gat_mid.hip's 128-wide form above: option GAT_WIDE_FROM = 103)
    assert expect_one
    assert int(nat.lib().magat_form_count(nat.FORMS["gat_mid"])) == (1 if N >= 103 else 0)
    assert (tc[ONE_LAUNCH] > 0) == expect_one, tc.counts
    assert tc["gat_graph"] == (0 if expect_one else 1), tc.counts
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=0, atol=2e-5)
    # and the attention tensor of the two-launch form for the same directed graph (incl. the empty row of the sink node)
    layer.return_attention = True
    with tag_counts() as tc, torch.no_grad():
        y2 = layer(x.to(gpu_device)).cpu()
    assert tc[ONE_LAUNCH] == 0
    np.testing.assert_allclose(y2.numpy(), y_ref.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(layer.aij.cpu().numpy(), a_ref.numpy(), rtol=0, atol=2e-6)


def test_gat_mfma_orientation_is_observable(gpu_device, tag_counts):
    """A graph with ONE directed edge i -> j: under the reference's orientation only node j's output differs from the
    edge-free result (it aggregates x_i with weight a_ij = 1); node i's does not.  Checked against the oracle AND as a
    property, on the one-launch kernel."""
    from oracle import magat_oracle as orc
    B, N, K, P = 2, 40, 3, 4
    layer, _, x = _layer_and_inputs(B, N, K, P, True, seed=9)
    S = torch.zeros(B, N, N, dtype=torch.float64)
    S[0, 7, 31] = 0.4
    S[1, 31, 7] = -2e-9
    sd = {k: v.detach() for k, v in layer.state_dict().items()}
    y_ref, _ = orc.gat_layer_forward(x, S.unsqueeze(1), sd, "KeyQuery", True)
    y_empty, _ = orc.gat_layer_forward(x, torch.zeros_like(S).unsqueeze(1), sd, "KeyQuery", True)
    layer = layer.to(gpu_device).eval()
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    with tag_counts() as tc, torch.no_grad():
        y = layer(x.to(gpu_device)).cpu()
    assert tc[ONE_LAUNCH] == 1, tc.counts
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=0, atol=2e-5)
    changed = (y - y_empty).abs().amax(dim=1) > 1e-4            # (B, N): nodes whose output the edge changed
    assert changed[0].nonzero().flatten().tolist() == [31]
    assert changed[1].nonzero().flatten().tolist() == [7]


@pytest.mark.parametrize("mode", ["KeyQuery", "GAT_modified", "GAT_origin"])
@pytest.mark.parametrize("N,K,P,concat,dt,B", [(20, 3, 4, True, torch.float64, 7), (10, 2, 1, True, torch.float32, 9),
                                               (32, 3, 4, False, torch.float32, 5), (1, 3, 2, True, torch.float32, 4),
                                               (17, 2, 4, False, torch.float64, 6), (12, 3, 4, True, torch.float32, 13)])
def test_gat_mfma_packed_instances_same_bits_as_unpacked(gpu_device, libopt, tag_counts, mode, N, K, P, concat, dt, B):
    """Round 4, small graphs (N <= 32): four planning instances per pass of the one-launch kernel, each in its own 32-row slot
    (option GAT_PACK: 2 = always, 0 = never, 1 = when the batch fills the chip).  An instance's arithmetic does not depend on
    its slot or on its pack mates: packed and unpacked forms agree BIT FOR BIT - batches that do not fill their last pack,
    directed GSOs with threshold entries and a NaN, all three attention modes, both merges - and both match the oracle."""
    from magat_pathplanning_amd import GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin
    from magat_pathplanning_amd.synthetic import directed_gso
    from oracle import magat_oracle as orc
    torch.manual_seed(77 + N)
    g = torch.Generator().manual_seed(N + B)
    cls = GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else GraphFilterBatchAttentional
    layer = cls(128, 128, K, P, concatenate=concat, attentionMode=mode)
    if mode == "GAT_modified":
        with torch.no_grad():
            layer.weight_bias.uniform_(-0.3, 0.3, generator=g)
    S = directed_gso(B, N, 0.3, seed=N + 3, dtype=dt)
    x = torch.randn(B, 128, N, generator=g) * 0.7
    y_ref, _ = orc.gat_layer_forward(x, S.unsqueeze(1), {k: v.detach() for k, v in layer.state_dict().items()}, mode, concat)
    layer = layer.to(gpu_device).eval()
    out = {}
    for pack in (2, 0):
        libopt.set("GAT_PACK", pack)
        layer.addGSO(S.unsqueeze(1).to(gpu_device))
        with tag_counts() as tc, torch.no_grad():
            out[pack] = layer(x.to(gpu_device)).cpu()
        assert tc["gat_layer (one launch)"] > 0, tc.counts
    assert torch.isfinite(out[2]).all()
    assert torch.equal(out[2], out[0])
    np.testing.assert_allclose(out[2].numpy(), y_ref.numpy(), rtol=0, atol=3e-5)
    # ... and the position inside the batch (= the slot) does not matter either
    libopt.set("GAT_PACK", 2)
    perm = torch.randperm(B, generator=g)
    layer.addGSO(S[perm].unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        yp = layer(x[perm].to(gpu_device)).cpu()
    assert torch.equal(yp, out[2][perm])


def test_gat_mfma_packed_at_benchmark_size(gpu_device, libopt):
    """BASELINE config 2's graph layer (1024 instances of 20 agents, K = 3, P = 4): the packed form is what runs by default at
    this size (GAT_PACK = 1) and equals the unpacked form bit for bit."""
    B, N, K, P = 1024, 20, 3, 4
    layer, S, x = _layer_and_inputs(B, N, K, P, True, seed=11)
    layer = layer.to(gpu_device).eval()
    out = {}
    for pack in (1, 0):
        libopt.set("GAT_PACK", pack)
        layer.addGSO(S.unsqueeze(1).to(gpu_device))
        with torch.no_grad():
            out[pack] = layer(x.to(gpu_device)).cpu()
    assert torch.equal(out[1], out[0])


@pytest.mark.parametrize("N,G,K,P,concat,dt,B", [(10, 32, 2, 4, False, torch.float32, 70), (10, 32, 2, 4, True, torch.float64, 5),
                                                 (20, 64, 3, 4, True, torch.float32, 33), (32, 32, 3, 2, False, torch.float64, 9),
                                                 (1, 32, 2, 1, True, torch.float32, 3), (27, 64, 2, 3, False, torch.float32, 1030)])
def test_small_graph_kernel_vs_oracle_and_two_launch_form(gpu_device, libopt, tag_counts, N, G, K, P, concat, dt, B):
    """csrc/gat_small.hip (round 4): the published widths (G = F = 32 | 64) on graphs of at most 32 agents as ONE launch, a wave
    per planning instance.  Against the oracle on directed GSOs (threshold entries, a NaN), against the two-launch form, and the
    instance-independence property: a permuted batch gives the permuted rows bit for bit (an instance's result depends on
    neither its wave nor its workgroup)."""
    from magat_pathplanning_amd import GraphFilterBatchAttentional
    from magat_pathplanning_amd.synthetic import directed_gso
    from oracle import magat_oracle as orc
    torch.manual_seed(31 + N + G)
    g = torch.Generator().manual_seed(N * G + B)
    layer = GraphFilterBatchAttentional(G, G, K, P, concatenate=concat, attentionMode="KeyQuery")
    Bo = min(B, 40)                       # (the oracle on the first instances only: it is a CPU restatement)
    S = directed_gso(B, N, 0.3, seed=N + G, dtype=dt)
    x = torch.randn(B, G, N, generator=g) * 0.7
    y_ref, _ = orc.gat_layer_forward(x[:Bo], S[:Bo].unsqueeze(1), {k: v.detach() for k, v in layer.state_dict().items()},
                                     "KeyQuery", concat)
    layer = layer.to(gpu_device).eval()
    out = {}
    for fused in (1, 0):
        libopt.set("GAT_MFMA", fused)
        layer.addGSO(S.unsqueeze(1).to(gpu_device))
        with tag_counts() as tc, torch.no_grad():
            out[fused] = layer(x.to(gpu_device)).cpu()
        assert (tc["gat_layer (one launch)"] > 0) == bool(fused), tc.counts
    assert torch.isfinite(out[1]).all()
    np.testing.assert_allclose(out[1][:Bo].numpy(), y_ref.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(out[1].numpy(), out[0].numpy(), rtol=0, atol=2e-5)
    libopt.set("GAT_MFMA", 1)
    perm = torch.randperm(B, generator=g)
    layer.addGSO(S[perm].unsqueeze(1).to(gpu_device))
    with torch.no_grad():
        yp = layer(x[perm].to(gpu_device)).cpu()
    assert torch.equal(yp, out[1][perm])


@pytest.mark.parametrize("N,K,P", [(100, 3, 4), (10, 2, 4), (64, 3, 2), (20, 3, 3)])
def test_gat_mfma_head_mean_head_split_equals_the_unsplit_form(gpu_device, tag_counts, N, K, P):
    """Head MEAN - what the reference runs unless `--AttentionConcat` is given (main.py:115, utils/config.py:122) - for few
    instances: a workgroup per (instance, head), the heads' rows through the workspace's scratch rows and the mean kernel
    (before: one workgroup walked the P heads of an instance, 83 against 37 us per layer call at one instance of 100 agents).
    Same sum in the same order as the unsplit form's read-add-write of Y: an instance alone equals its rows in a 1100-instance
    batch (more instances than the round-count model splits) bit for bit; against the oracle."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat
    B = 1100
    layer, S, x = _layer_and_inputs(B, N, K, P, False, seed=70 + N)
    sd = {k: v.detach() for k, v in layer.state_dict().items()}
    layer = layer.to(gpu_device).eval()
    lib = nat.lib()
    with torch.no_grad():
        layer.addGSO(S.unsqueeze(1).to(gpu_device))
        lib.magat_form_reset()
        big = layer(x.to(gpu_device)).cpu()
        assert lib.magat_form_count(nat.FORMS["gat_hsplit"]) == 0
        for pick in ([0], [7, 8], [1099]):
            layer.addGSO(S[pick].unsqueeze(1).to(gpu_device))
            lib.magat_form_reset()
            with tag_counts() as tc:
                small = layer(x[pick].to(gpu_device)).cpu()
            assert tc[ONE_LAUNCH] == 1, tc.counts
            assert lib.magat_form_count(nat.FORMS["gat_hsplit"]) == 1
            assert torch.equal(small, big[pick])
    y_ref, _ = orc.gat_layer_forward(x[:3], S[:3].unsqueeze(1), sd, "KeyQuery", False)
    np.testing.assert_allclose(big[:3].numpy(), y_ref.numpy(), rtol=0, atol=2e-5)
