"""The bf16-storage CSR layer with the maps INSIDE the two graph kernels (csrc/gat_csr_fused.hip; BASELINE config 5: KeyQuery,
K = 2, G = F = 128, concat): q' = W^T x and the tap contraction on the matrix cores, the hop on the node features - the
reference's own order (graphML.py:1757, 1768-1770).  Checked against (a) the oracle's emulation of exactly that order with
bf16 rounding where the kernels round, (b) the pinned float32 oracle within the bf16 budget, (c) the split form (maps GEMM +
tiled score / hop kernels) of the same library on adversarial graphs, and for WHICH kernels ran (form counter, launch tags)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
G5 = 128


def _forms(nat):
    return {k: int(nat.lib().magat_form_count(i)) for k, i in nat.FORMS.items()}


@pytest.mark.parametrize("N,P", [(1000, 4), (333, 2), (40, 1)])
def test_fused_layer_against_the_oracle(gpu_device, tag_counts, N, P):
    from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso
    from oracle import magat_oracle as orc
    B = 2
    g = torch.Generator().manual_seed(15 + N)
    layer = GraphFilterBatchAttentional(G5, G5, 2, P, attentionMode="KeyQuery")
    with torch.no_grad():
        layer.bias.copy_(torch.randn(G5, 1, generator=g) * 0.05)       # (the reference initialises it to zero: make it visible)
    x = torch.randn(B, G5, N, generator=g) * 0.5
    S = comm_gso(B, N, int(5.0 * N ** 0.5) + 2, seed=8)
    S[1, 3, :] = 0                                                         # a row without edges
    params = {k: v.detach() for k, v in layer.state_dict().items()}
    y_ref, a_ref = orc.gat_layer_forward(x, S.unsqueeze(1), params, "KeyQuery", True)
    y_emul, _ = orc.gat_layer_forward_bf16_fused(x, S.unsqueeze(1), params)
    layer = layer.to(gpu_device).eval()
    layer.storage_dtype = torch.bfloat16
    layer.return_attention = True
    layer.addGSO(S.unsqueeze(1).to(gpu_device))
    nat.lib().magat_form_reset()
    with torch.no_grad(), tag_counts() as tc:
        y = layer(x.to(gpu_device)).cpu()
    forms = _forms(nat)
    assert forms["csr_fused"] == 1 and tc["gat_maps_gemm"] == 0 and tc["gat_graph"] == 3, (forms, tc.counts)
    assert y.dtype == torch.float32 and tuple(y.shape) == (B, P * G5, N)
    scale = float(y_ref.abs().max())
    e_emul = float((y - y_emul).abs().max())
    e_ref = float((y - y_ref).abs().max())
    print("fused CSR layer N=%d P=%d: scale %.3g, vs emulation %.3g, vs fp32 oracle %.3g" % (N, P, scale, e_emul, e_ref))
    assert e_emul <= 2.0 ** -7 * scale, (e_emul, scale)
    assert e_ref <= 2e-2 * scale, (e_ref, scale)
    # the attention tensor (float32 row softmax over scores of bf16 operands) and its exact zeros
    a = layer.aij.cpu() if torch.is_tensor(layer.aij) else torch.as_tensor(layer.aij)
    assert tuple(a.shape) == tuple(a_ref.shape)
    assert float((a - a_ref).abs().max()) <= 2e-2
    assert float(a[1, :, 0, 3, :].abs().max()) == 0.0
    mask = (S.abs() > 1e-9)
    assert float(a[:, 0, 0][~mask].abs().max()) == 0.0
    rows = mask.sum(-1) > 0
    assert float((a[:, 0, 0].sum(-1)[rows] - 1).abs().max()) <= 1e-5


@pytest.mark.parametrize("N,B,P,kind,f32out", [(1000, 2, 4, "hubs", False), (255, 5, 4, "hubs", True), (9, 5, 2, "empty", False),
                                               (1, 3, 4, "sparse", False), (31, 9, 1, "dense", True), (33, 2, 4, "dense", False),
                                               (1024, 1, 2, "sparse", False), (129, 17, 4, "empty", True)])
def test_fused_against_the_split_form(gpu_device, libopt, N, B, P, kind, f32out):
    """Same library, same inputs: the fused form against maps GEMM + score / hop kernels (which round Q and U to bf16 where
    the fused form rounds q' and z: a bf16 budget, not bit equality) on hub rows AND columns (hundreds of edges: far past one
    slot per lane), empty rows, dense neighbourhoods, N below / at / above a 32-row group, more instances than XCDs, a
    float32 result buffer with a wider row stride; and run to run the fused form is bit-identical (the degree ranking's
    order inside a bin is not, and must not matter)."""
    from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
    from magat_pathplanning_amd.graphml import CsrStructure, gat_forward_rows_csr
    g = torch.Generator().manual_seed(N + P)
    dens = {"sparse": 5.0 / N, "dense": min(0.5, 40.0 / N), "hubs": 3.0 / N, "empty": 1.0 / N}[kind]
    S = (torch.rand(B, N, N, generator=g) < dens).float()
    if kind == "hubs":
        S[:, N // 3, :] = 1.0
        S[:, :, N // 2] = 1.0
    if kind == "empty":
        S[:, : N // 2, :] = 0.0
    S = S.to(gpu_device)
    torch.manual_seed(N)
    layer = GraphFilterBatchAttentional(G5, G5, 2, P, attentionMode="KeyQuery").to(gpu_device).eval()
    with torch.no_grad():
        layer.bias.copy_(torch.randn(G5, 1) * 0.05)
    X = (torch.randn(B, N, G5, device=gpu_device) * 0.5).to(torch.bfloat16)
    st = CsrStructure().build(S.clone(), 0)
    nnz = st.ready(gpu_device)
    csc = (st.cscptr, st.csc[0], st.csc[1])
    width = P * G5

    def run(fused):
        libopt.set("CSR_FUSED", fused)
        nat.lib().magat_form_reset()
        if f32out:
            out = torch.full((B * N, width + 4), -7.0, dtype=torch.float32, device=gpu_device)
        else:
            out = torch.empty(B * N, width, dtype=torch.bfloat16, device=gpu_device)
        _, att = gat_forward_rows_csr(X, st.rowptr, st.colidx, nnz, layer, out=out, csc=csc, want_attention=True)
        torch.cuda.synchronize()
        assert int(nat.lib().magat_form_count(nat.FORMS["csr_fused"])) == (1 if fused else 0)
        if f32out:
            assert bool((out[:, width:] == -7.0).all())
            assert torch.equal(out[:, :width], out[:, :width].to(torch.bfloat16).float())       # bf16-representable values
        return out[:, :width].float().clone(), att[:, :nnz].clone()

    a, att_a = run(1)
    a2, att_a2 = run(1)
    b, att_b = run(0)
    assert not bool(torch.isnan(a).any())
    assert torch.equal(a, a2) and torch.equal(att_a, att_a2)
    scale = float(b.abs().max()) + 1e-6
    assert float((a - b).abs().max()) <= 2e-2 * scale, (float((a - b).abs().max()), scale)
    if nnz:
        assert float((att_a - att_b).abs().max()) <= 2e-2


def test_fused_model_matches_split_model_at_config5_shape(gpu_device, libopt):
    """The whole module with bf16 storage at 2 x 1000 agents: logits of the fused and the split form within the bf16 budget of
    each other and of the float32 oracle; the action head reads the fused kernel's bf16 rows directly."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    from oracle import magat_oracle as orc
    B, N = 2, 1000
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=4, gat_storage="bf16", device=str(gpu_device))
    sd = orc.init_state_dict(cfg, seed=21)
    x = fov_states(B, N, seed=5)
    S = comm_gso(B, N, 160, seed=6)
    ref = orc.planner_forward(x, S.clone(), sd, cfg)
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval()
    got = {}
    for fused in (1, 0):
        libopt.set("CSR_FUSED", fused)
        with torch.no_grad():
            net.addGSO(S.to(gpu_device))
            got[fused] = net(x.to(gpu_device)).cpu()
    scale = max(1.0, float(ref.abs().max()))
    for fused in (1, 0):
        err = float((got[fused] - ref).abs().max())
        agree = float((got[fused].argmax(1) == ref.argmax(1)).float().mean())
        print("fused=%d: max|dlogit| %.3e of scale %.3g, argmax agreement %.4f" % (fused, err, scale, agree))
        assert err <= 2e-2 * scale and agree >= 0.97
    assert float((got[1] - got[0]).abs().max()) <= 2e-2 * scale


@pytest.mark.parametrize("B,N,drift", [(2, 200, False), (40, 1000, False), (3, 200, True)])
def test_bf16_rows_of_compressmlp_leave_the_encoder(gpu_device, tag_counts, B, N, drift):
    """ABI 7 (last clause): with bf16 storage in the graph layer compressMLP's rows come out of the encoder
    as bf16 as well - from the epilogue that produces them when compressMLP rides in the head's launch (40 000 agents), by a
    cast pass inside the encoder call otherwise - and there is no cast launch of the planner's own in front of the layer.  The
    bf16 rows are RNE(comp) bit for bit, also behind a range-guard re-run (inputs driven 3000x beyond the calibration)."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=4, gat_storage="bf16", device=str(gpu_device))
    torch.manual_seed(3)
    net = DecentralPlannerGATNet(cfg).to(gpu_device).eval()
    x = fov_states(B, N, seed=5).to(gpu_device)
    S = comm_gso(B, N, int(6 * N ** 0.5), seed=6).to(gpu_device)
    with torch.no_grad():
        net.addGSO(S.clone())
        net(x)
        if drift:
            x = x * 3000.0
        with tag_counts() as tc:
            net.addGSO(S.clone())
            out = net(x)
    torch.cuda.synchronize()
    comp, comp16 = net._rt.buffers["comp"], net._rt.buffers["comp16"]
    assert comp16.dtype == torch.bfloat16 and tuple(comp16.shape) == (B * N, 128)
    assert torch.equal(comp16, comp.to(torch.bfloat16))
    assert bool(torch.isfinite(out).all()) and float(comp.abs().max()) > 0
    if drift:
        assert net.range_status()["encoder_rerun"]
    if B * N >= 32768:
        assert tc["gat_cast"] == 0 and tc["compressMLP"] == 0, tc.counts          # one launch: head + compressMLP + the bf16 rows
    else:
        assert tc["gat_cast"] >= 1, tc.counts                                     # the cast pass inside the encoder call
