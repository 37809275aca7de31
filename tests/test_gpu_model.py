"""End-to-end parity of DecentralPlannerGATNet (HIP inference path) against the reference-made golden
logits and, at larger sizes, against the pinned CPU oracle.  Gate: max|dlogits| <= 1e-4 (north star)."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import golden_paths, load_model_fixture

pytestmark = pytest.mark.gpu
MODEL = golden_paths("model_")
TOL = 1e-4


def _build(cfg, sd, device):
    from magat_pathplanning_amd import DecentralPlannerGATNet
    cfg.device = str(device)
    net = DecentralPlannerGATNet(cfg)
    missing = net.load_state_dict(sd, strict=True)
    return net.to(device).eval()


@pytest.mark.parametrize("want_att", [False, True], ids=["default_kernels", "with_attention"])
@pytest.mark.parametrize("path", MODEL, ids=[os.path.basename(p)[:-4] for p in MODEL])
def test_model_vs_reference_golden(gpu_device, tag_counts, path, want_att):
    """want_att=False: the launch sequence inference runs by default (for KeyQuery models the graph layer is the one-launch
    matrix-core kernel - asserted through the profiling tags); want_att=True: config.return_attentionGSO, which
    materialises the attention tensor (two-launch form) and checks it as well."""
    from magat_pathplanning_amd import _native as nat
    z, sd, cfg = load_model_fixture(path)
    net = _build(cfg, sd, gpu_device)
    x = torch.from_numpy(z["x"].astype(np.float32)).to(gpu_device)
    S = torch.from_numpy(z["S"].copy()).to(gpu_device)
    cfg.return_attentionGSO = want_att
    with tag_counts() as tc, torch.no_grad():
        net.addGSO(S)
        logits = net(x)
    torch.cuda.synchronize()
    assert logits.shape == z["logits"].shape and logits.dtype == torch.float32
    err = np.abs(logits.cpu().numpy() - z["logits"]).max()
    assert err <= TOL, err
    # addGSO mutates the caller's tensor exactly like the reference
    np.testing.assert_array_equal(np.nan_to_num(S.cpu().numpy(), nan=-7.0), np.nan_to_num(z["S_after"], nan=-7.0))
    layer = net.GFL[0]
    one_launch = bool(nat.lib().magat_gat_one_launch_supported(S.shape[1], layer.G, layer.F, layer.K,
                                                               nat._MODE_IDS[layer.attentionMode], int(layer.concatenate)))
    if want_att:
        assert tc["gat_layer (one launch)"] == 0
        np.testing.assert_allclose(net.returnAttentionGSO(), z["aij"].mean(axis=1), rtol=0, atol=5e-6)
    else:
        assert (tc["gat_layer (one launch)"] > 0) == one_launch, (tc.counts, one_launch)
        st = net.range_status()
        assert not st["encoder_rerun"] and not st["gat_rerun"], st       # the fast pass produced these logits, not the re-run


@pytest.mark.parametrize("B,N,K,P,mode,concat,skip", [
    (8, 100, 3, 4, "KeyQuery", True, "BottomNeck_skipConcat"),
    (16, 20, 3, 4, "KeyQuery", True, "BottomNeck_only"),
    (64, 10, 2, 1, "KeyQuery", True, "BottomNeck_only"),
    (4, 100, 3, 4, "GAT_modified", False, "BottomNeck_skipConcatGNN"),
    (4, 100, 3, 4, "GAT_origin", True, "BottomNeck_skipConcat"),
])
def test_model_vs_oracle_benchmark_shapes(gpu_device, B, N, K, P, mode, concat, skip):
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=N, nGraphFilterTaps=K, nAttentionHeads=P, attentionMode=mode,
                      AttentionConcat=concat, bottleneckMode=skip)
    sd = orc.init_state_dict(cfg, seed=7)
    x = fov_states(B, N, seed=3)
    S = comm_gso(B, N, {10: 20, 20: 28, 100: 50}[N], seed=4, dtype=torch.float64)
    ref = orc.planner_forward(x, S.clone(), sd, cfg)
    net = _build(cfg, sd, gpu_device)
    with torch.no_grad():
        net.addGSO(S.to(gpu_device))
        got = net(x.to(gpu_device))
    err = (got.cpu() - ref).abs().max().item()
    assert err <= TOL, err


def test_shard_equivalence_and_pickle(gpu_device):
    """Instances are independent: running two half-batches equals the full batch bit-for-bit, and a
    pickled copy (how test_multi ships the model to workers) reproduces the logits."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=20, nGraphFilterTaps=3, nAttentionHeads=4)
    net = _build(cfg, orc.init_state_dict(cfg, seed=11), gpu_device)
    x = fov_states(6, 20, seed=1).to(gpu_device)
    S = comm_gso(6, 20, 28, seed=2).to(gpu_device)
    with torch.no_grad():
        net.addGSO(S)
        full = net(x).clone()
        net.addGSO(S[:3].contiguous())
        a = net(x[:3]).clone()
        net.addGSO(S[3:].contiguous())
        b = net(x[3:]).clone()
    assert torch.equal(full, torch.cat((a, b)))
    clone = pickle.loads(pickle.dumps(net))
    with torch.no_grad():
        clone.addGSO(S)
        assert torch.equal(clone(x), full)


def _two_unlike_shards(device, B=6, N=20):
    """A batch whose halves drive the layers to clearly different magnitudes (sparse maps first, saturated maps second): under
    a first-batch calibration the two halves would fold different activation-scale exponents."""
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states
    x = fov_states(B, N, seed=21)
    h = B // 2
    x[:h, :, 0] *= 0.0                                   # no obstacles, few agents
    x[:h, :, 2, 1:-1, 1:-1] *= 0.0
    x[h:, :, 0, 1:-1, 1:-1] = 1.0                        # every cell an obstacle / an agent
    x[h:, :, 2, 1:-1, 1:-1] = 1.0
    S = comm_gso(B, N, 28, seed=22)
    return x.to(device), S.to(device), h


def test_shard_order_and_process_history_do_not_change_the_bits(gpu_device):
    """VERDICT r03 item 1 / SURVEY.md section 8(e): the logits of a planning instance must not depend on which shard a
    process saw FIRST.  Two fresh modules with one state_dict, one fed shard A first, the other shard B first, and a third
    that sees the whole batch: concat(a, b) == full bit for bit, and all three fold the same activation-scale exponents
    (they come from the canonical calibration batch, a function of the config alone).  The same with pickled clones whose
    first batch differs from the parent's - how torch.multiprocessing.spawn workers receive the model."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import make_config
    cfg = make_config(num_agents=20, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    sd = orc.init_state_dict(cfg, seed=31)
    x, S, h = _two_unlike_shards(gpu_device)
    nets = [_build(cfg, sd, gpu_device) for _ in range(3)]
    with torch.no_grad():
        nets[0].addGSO(S[:h].contiguous())
        a = nets[0](x[:h]).clone()                       # rank 0: shard A is its first batch
        nets[1].addGSO(S[h:].contiguous())
        b = nets[1](x[h:]).clone()                       # rank 1: shard B is its first batch
        nets[2].addGSO(S.clone())
        full = nets[2](x).clone()                        # single process: the whole batch
        # ... and each of them on the OTHER shard afterwards
        nets[0].addGSO(S[h:].contiguous())
        b0 = nets[0](x[h:]).clone()
        nets[1].addGSO(S[:h].contiguous())
        a1 = nets[1](x[:h]).clone()
    assert torch.equal(torch.cat((a, b)), full)
    assert torch.equal(torch.cat((a1, b0)), full)
    sc = [n.range_status()["act_scales"] for n in nets]
    assert sc[0] is not None and sc[0]["source"] == "canonical"
    assert sc[0] == sc[1] == sc[2], sc
    # a clone pickled BEFORE its parent has seen anything, first batch = shard B; one pickled after, first batch = shard A
    parent = _build(cfg, sd, gpu_device)
    early = pickle.loads(pickle.dumps(parent))
    with torch.no_grad():
        parent.addGSO(S[:h].contiguous())
        pa = parent(x[:h]).clone()
        early.addGSO(S[h:].contiguous())
        eb = early(x[h:]).clone()
        late = pickle.loads(pickle.dumps(parent))
        late.addGSO(S[h:].contiguous())
        lb = late(x[h:]).clone()
    assert torch.equal(torch.cat((pa, eb)), full) and torch.equal(lb, eb)
    assert late.range_status()["act_scales"] == sc[0]


def test_explicit_calibration_travels_with_the_module(gpu_device):
    """model.calibrate(x): magnitudes measured on the caller's own data (source 'user').  They are pickled with the module,
    so a clone reproduces the parent's bits on ANY batch without having seen the calibration data; a weight change drops
    them (back to the canonical batch)."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import make_config
    cfg = make_config(num_agents=20, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    sd = orc.init_state_dict(cfg, seed=32)
    x, S, h = _two_unlike_shards(gpu_device)
    net = _build(cfg, sd, gpu_device)
    info = net.calibrate(x[h:])
    assert info["source"] == "user"
    clone = pickle.loads(pickle.dumps(net))
    with torch.no_grad():
        net.addGSO(S.clone())
        full = net(x).clone()
        clone.addGSO(S[:h].contiguous())
        a = clone(x[:h]).clone()
        clone.addGSO(S[h:].contiguous())
        b = clone(x[h:]).clone()
    assert clone.range_status()["act_scales"] == net.range_status()["act_scales"]
    assert torch.equal(torch.cat((a, b)), full)
    ref = orc.planner_forward(x.cpu(), S.cpu().clone(), sd, cfg)
    assert float((full.cpu() - ref).abs().max()) <= 1e-4
    with torch.no_grad():
        net.compressMLP[0].bias.add_(0.0)                # touched, same values: the record is keyed on the CONTENT of the pack
        net.addGSO(S.clone())
        assert torch.equal(net(x), full)
        assert net.range_status()["act_scales"]["source"] == "user"
        net.compressMLP[0].bias.add_(0.25)               # the weights changed: the record no longer belongs to them
        net.addGSO(S.clone())
        net(x)
    assert net.range_status()["act_scales"]["source"] == "canonical"


def test_shard_equivalence_across_the_head_split(gpu_device, monkeypatch, libopt):
    """The one size-dependent piece of arithmetic is the encoder head: below MAGAT_HEAD_SPLITK agents (5120) it sums nine
    per-cell partial products, above it runs one long-K GEMM.  A batch above the threshold cut into shards below it
    therefore agrees to float32 rounding (1e-5 here, the gate is 1e-4), and bit-for-bit once both sides are pinned to
    one form (MAGAT_HEAD_SPLITK=0) - what a deployment that needs bit-exact resharding sets."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 70, 100                                    # 7000 agents: above the threshold; halves of 3500: below
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    net = _build(cfg, orc.init_state_dict(cfg, seed=12), gpu_device)
    x = fov_states(B, N, seed=3).to(gpu_device)
    S = comm_gso(B, N, 50, seed=4).to(gpu_device)
    h = B // 2

    def run():
        with torch.no_grad():
            net.addGSO(S)
            full = net(x).clone()
            net.addGSO(S[:h].contiguous())
            a = net(x[:h]).clone()
            net.addGSO(S[h:].contiguous())
            b = net(x[h:]).clone()
        return full, torch.cat((a, b))
    full, parts = run()
    assert (full - parts).abs().max().item() <= 1e-5
    libopt.set("MAGAT_HEAD_SPLITK", "0")
    full0, parts0 = run()
    assert torch.equal(full0, parts0)
    assert torch.equal(full0, full)                   # above the threshold the default already is the one-GEMM form
    # the column tile of the head GEMM narrows until the launch fills the chip (CONV_BNFILL; 7000 agents: 55 agent tiles ->
    # 32-column tiles): no element's summation order depends on it
    libopt.set("MAGAT_CONV_BNFILL", "0")
    full1, parts1 = run()
    assert torch.equal(full1, full0) and torch.equal(parts1, parts0)


def test_shards_choose_their_forms_on_the_global_agent_count(gpu_device, tag_counts):
    """64 instances of 100 agents cut into 8 shards of 800 agents - every shard far below
    HEAD_SPLITK (5120), the whole batch (6400) above it.  distributed.sharded_forward hands the GLOBAL agent count to the
    encoder (magat_encoder_desc.form_agents, ABI 6), so each shard's head runs the form the whole batch runs and the
    shards concatenate to the single-process logits BIT FOR BIT with default options; without the hint the same shards
    differ in the last bits (which is what makes the assertion meaningful).  Here the eight ranks are walked by one
    process through sharded_forward's own code path (its rank / world come from a stand-in process group)."""
    from unittest import mock
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import distributed as D
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N, world = 64, 100, 8
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    sd = orc.init_state_dict(cfg, seed=21)
    net = _build(cfg, sd, gpu_device)
    x = fov_states(B, N, seed=7).to(gpu_device)
    S = comm_gso(B, N, 50, seed=8).to(gpu_device)
    with torch.no_grad():
        net.addGSO(S.clone())
        whole = net(x).clone()
        parts, plain = [], []
        for rank in range(world):
            with mock.patch.object(D.dist, "is_initialized", return_value=True), \
                    mock.patch.object(D.dist, "get_world_size", return_value=world), \
                    mock.patch.object(D.dist, "get_rank", return_value=rank):
                parts.append(D.sharded_forward(net, x, S.clone(), gather=False).clone())
            b0, b1 = D.shard_range(B, rank, world)
            net.addGSO(S[b0:b1].contiguous())
            plain.append(net(x[b0:b1]).clone())        # the same shard WITHOUT the hint: the few-agent form of the head
        assert net.form_agents == 0                    # (restored behind every shard)
    got = torch.cat(parts)
    assert got.shape == whole.shape and torch.equal(got, whole)
    unhinted = torch.cat(plain)
    assert not torch.equal(unhinted, whole) and (unhinted - whole).abs().max().item() <= 1e-5
    # and against the oracle on a sample of instances (the logits themselves, not only their agreement)
    pick = [0, 9, 33, 63]
    ref = orc.planner_forward(x[pick].cpu(), S[pick].cpu().clone(), sd, cfg)
    rows = torch.cat([whole[b * N:(b + 1) * N] for b in pick]).cpu()
    assert float((rows - ref).abs().max()) <= 1e-4


def test_compress_in_the_head_epilogue_is_bit_identical(gpu_device, libopt):
    """Round 5: compressMLP computed in the encoder head's epilogue (option HEAD_COMPRESS, default on; one launch for both
    layers once the head keeps its 128-column tile, i.e. from 32 768 agents on) against the two launches it replaces: the same
    products in the same order on the same planes - logits equal BIT FOR BIT, so a batch that takes the fused form and its
    shards that do not (narrowed head tiles) still concatenate exactly.  The form counters say which one ran."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 331, 100                                   # 33 100 agents: 259 agent tiles, a ragged last one
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    sd = orc.init_state_dict(cfg, seed=23)
    net = _build(cfg, sd, gpu_device)
    x = fov_states(B, N, seed=9).to(gpu_device)
    S = comm_gso(B, N, 50, seed=10).to(gpu_device)
    lib = nat.lib()
    with torch.no_grad():
        net.addGSO(S.clone())
        net(x)
        lib.magat_form_reset()
        net.addGSO(S.clone())
        fused = net(x).clone()
        assert lib.magat_form_count(nat.FORMS["head_compress"]) == 1
        libopt.set("MAGAT_HEAD_COMPRESS", 0)
        lib.magat_form_reset()
        net.addGSO(S.clone())
        two = net(x).clone()
        assert lib.magat_form_count(nat.FORMS["head_compress"]) == 0
        assert torch.equal(fused, two)
        libopt.reset("MAGAT_HEAD_COMPRESS")
        # halves of the batch (16 500 / 16 600 agents: narrowed head tiles, two launches) with the global form hint
        h = 165
        net.form_agents = B * N
        lib.magat_form_reset()
        net.addGSO(S[:h].contiguous())
        a = net(x[:h]).clone()
        net.addGSO(S[h:].contiguous())
        b = net(x[h:]).clone()
        net.form_agents = 0
        assert lib.magat_form_count(nat.FORMS["head_compress"]) == 0
        assert torch.equal(torch.cat((a, b)), fused)
    pick = [0, 164, 165, 330]
    ref = orc.planner_forward(x[pick].cpu(), S[pick].cpu().clone(), sd, cfg)
    rows = torch.cat([fused[b_ * N:(b_ + 1) * N] for b_ in pick]).cpu()
    assert float((rows - ref).abs().max()) <= 1e-4
    assert not net.range_status()["encoder_rerun"]


def test_stem_binary_input_fast_path_and_general_inputs(gpu_device):
    """Round 5: the stem skips its third split product for a group of eight agents whose state maps are exact in ONE f16 plane
    (the reference's binary 'Project_G' channels): a sum of exact zeros.  General float32 inputs (a second plane that is not
    zero) take all three products.  Both against the oracle; and a binary planning instance gives the same logits whether its
    batch neighbours are binary or not (groups of eight agents are decided one by one)."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 6, 16                                        # 96 agents = 12 groups of eight, two per instance
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    sd = orc.init_state_dict(cfg, seed=29)
    net = _build(cfg, sd, gpu_device)
    xb = fov_states(B, N, seed=3)
    g = torch.Generator().manual_seed(4)
    xn = xb.clone()
    xn[1::2] += 0.37 * torch.randn(xn[1::2].shape, generator=g)      # odd instances: values with a non-zero second plane
    S = comm_gso(B, N, 20, seed=5)
    outs = {}
    for name, x in (("binary", xb), ("mixed", xn)):
        ref = orc.planner_forward(x, S.clone(), sd, cfg)
        with torch.no_grad():
            net.addGSO(S.clone().to(gpu_device))
            got = net(x.to(gpu_device)).cpu()
        assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), name
        outs[name] = got
    for b in range(0, B, 2):                             # the binary instances of the mixed batch: the same numbers
        assert torch.equal(outs["mixed"][b * N:(b + 1) * N], outs["binary"][b * N:(b + 1) * N]), b


def test_shard_equivalence_across_the_encoder_chunk(gpu_device, libopt):
    """The encoder walks a batch in chunks of MAGAT_ENC_CHUNK agents (65 536).  The form of the head is chosen on the
    agent count of the whole call, so the short last chunk sums like the others and a batch that spills into a second
    chunk still equals its own shards bit for bit (found by tools/exp/big_ragged.py at 70 100 agents; here the chunk is
    set to 8192 so that 9000 agents reproduce it in a second)."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 90, 100                                    # 9000 agents: chunks of 8192 + 808 (the latter below HEAD_SPLITK)
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    net = _build(cfg, orc.init_state_dict(cfg, seed=13), gpu_device)
    x = fov_states(B, N, seed=5).to(gpu_device)
    S = comm_gso(B, N, 50, seed=6).to(gpu_device)
    cut = 60                                          # shards of 6000 and 3000 agents; the second is below HEAD_SPLITK, so
    # the shards are compared with the head pinned to one form, and the chunk loop on its own (whole batch chunked against
    # the whole batch in one pass) with the default options
    with torch.no_grad():
        net.addGSO(S)
        whole = net(x).clone()
        libopt.set("MAGAT_ENC_CHUNK", "8192")
        chunked = net(x).clone()
        assert torch.equal(chunked, whole)
        libopt.set("MAGAT_HEAD_SPLITK", "0")
        chunked0 = net(x).clone()
        net.addGSO(S[:cut].contiguous())
        a = net(x[:cut]).clone()
        net.addGSO(S[cut:].contiguous())
        b = net(x[cut:]).clone()
    assert torch.equal(chunked0, whole)
    assert torch.equal(torch.cat((a, b)), whole)


def test_cpu_tensor_inference_fails_loudly(gpu_device):
    from magat_pathplanning_amd import DecentralPlannerGATNet, _native
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    net = DecentralPlannerGATNet(make_config(device="cpu")).eval()
    with torch.no_grad():
        net.addGSO(comm_gso(1, 10, 20))
        with pytest.raises(_native.MagatNativeError):
            net(fov_states(1, 10))


@pytest.mark.parametrize("B,N,G,K,P,skip", [(1, 1, 128, 3, 4, "BottomNeck_only"), (1, 128, 128, 3, 2, "BottomNeck_skipConcat"),
                                           (3, 128, 32, 2, 4, "BottomNeck_skipConcatGNN"), (1, 2, 64, 1, 1, "BottomNeck_skipAddGNN"),
                                           (5, 33, 128, 4, 3, "BottomNeck_only")])
def test_model_edge_sizes_vs_oracle(gpu_device, B, N, G, K, P, skip):
    """Single agent, the largest dense-kernel graph (N=128: G=128 overflows LDS and takes the CSR kernels, G=32
    stays on the LDS kernel), K=1, odd sizes that leave partial GEMM tiles."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    concat = skip != "BottomNeck_skipAddGNN"
    cfg = make_config(num_agents=N, nGraphFilterTaps=K, nAttentionHeads=P, bottleneckFeature=G, bottleneckMode=skip,
                      AttentionConcat=concat, CNN_mode="ResNetLarge" if skip == "BottomNeck_skipAddGNN" else
                      "ResNetLarge_withMLP")
    sd = orc.init_state_dict(cfg, seed=B * 100 + N)
    x = fov_states(B, N, seed=N)
    S = comm_gso(B, N, max(8, int(5 * N ** 0.5)), seed=N + 1, dtype=torch.float64)
    ref = orc.planner_forward(x, S.clone(), sd, cfg)
    net = _build(cfg, sd, gpu_device)
    with torch.no_grad():
        net.addGSO(S.to(gpu_device))
        got = net(x.to(gpu_device))
    assert got.shape == (B * N, 5)
    err = (got.cpu() - ref).abs().max().item()
    assert err <= TOL, err


def test_graph_capture_replay_matches_eager(gpu_device):
    """The whole addGSO+forward is stream-ordered and allocation-free after warm-up, so it can be captured in a
    HIP graph (torch.cuda.CUDAGraph) for the reference's batch-1 closed loop."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=10, nGraphFilterTaps=3, nAttentionHeads=4)
    net = _build(cfg, orc.init_state_dict(cfg, seed=5), gpu_device)
    x, S = fov_states(1, 10, seed=1).to(gpu_device), comm_gso(1, 10, 20, seed=2, dtype=torch.float64).to(gpu_device)
    sx, sS = x.clone(), S.clone()
    with torch.no_grad():
        net.addGSO(S)
        eager = net(x).clone()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(2):
                net.addGSO(sS)
                net(sx)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            net.addGSO(sS)
            out = net(sx)
        x2, S2 = fov_states(1, 10, seed=7).to(gpu_device), comm_gso(1, 10, 20, seed=8, dtype=torch.float64).to(gpu_device)
        sx.copy_(x2); sS.copy_(S2)
        g.replay()
        torch.cuda.synchronize()
        net.addGSO(S2)
        assert torch.equal(out, net(x2))
        sx.copy_(x); sS.copy_(S)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)


def test_model_bf16_gat_storage_config5_shape(gpu_device):
    """BASELINE config 5 in miniature (sparse comm-radius graph beyond the LDS kernel's N, K=2, P=4,
    config.gat_storage='bf16'): logits against the fp32 oracle -- error reported and bounded, greedy actions agree
    (SURVEY.md 8(d) parity gate for the bf16 configuration)."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 2, 400
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=4, gat_storage="bf16")
    sd = orc.init_state_dict(cfg, seed=21)
    x = fov_states(B, N, seed=5)
    S = comm_gso(B, N, 100, seed=6, dtype=torch.float64)
    ref = orc.planner_forward(x, S.clone(), sd, cfg)
    net = _build(cfg, sd, gpu_device)
    assert net.GFL[0].storage_dtype == torch.bfloat16
    with torch.no_grad():
        net.addGSO(S.to(gpu_device))
        got = net(x.to(gpu_device)).cpu()
    err = (got - ref).abs().max().item()
    agree = (got.argmax(1) == ref.argmax(1)).float().mean().item()
    print("bf16 GAT storage: max|dlogit| = %.3e, argmax agreement = %.4f" % (err, agree))
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err
    assert agree >= 0.97, agree


@pytest.mark.parametrize("cnn", ["ResNetLarge_withMLP", "ResNetSlim_withMLP"])
def test_encoder_paths_agree(gpu_device, monkeypatch, cnn, libopt):
    """The encoder's kernel paths are interchangeable: fused stem + layer1.conv1 vs two launches, f16 plane-granule vs
    float32-granule activation tiles, direct vs 2x2 LDS-staged f16x3 GEMM, f16x3 vs bf16x6 vs fp32 MFMA - every
    combination reproduces the oracle within the 1e-4 gate and the default path within 2e-5.  Ragged agent count
    (B*N = 150: a partial 128-agent tile) on purpose."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 15, 10
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=2, CNN_mode=cnn)
    sd = orc.init_state_dict(cfg, seed=13)
    x = fov_states(B, N, seed=5)
    S = comm_gso(B, N, 20, seed=6, dtype=torch.float64)
    ref = orc.planner_forward(x, S.clone(), sd, cfg).numpy()
    net = _build(cfg, sd, gpu_device)
    xd = x.to(gpu_device)
    outs = {}
    variants = [{}, {"MAGAT_L1_FUSED": "0"}, {"MAGAT_L1_FUSED": "1"}, {"MAGAT_CONV_PCHAIN": "0"},
                {"MAGAT_CONV_TM": "1"}, {"MAGAT_CONV_SPLIT": "0"}, {"MAGAT_HEAD_SPLITK": "0"},
                {"MAGAT_BLOCK_FUSED": "0"}, {"MAGAT_BLOCK_FUSED": "1"}, {"MAGAT_GAT_MFMA": "0"}, {"MAGAT_HEAD_COMPRESS": "0"},
                {"MAGAT_HEAD_F16": "0"}, {"MAGAT_SKINNY": "0"}]
    for env in variants:
        for k, v in env.items():
            libopt.set(k, v)
        with torch.no_grad():
            net.addGSO(S.clone().to(gpu_device))
            outs[tuple(sorted(env.items()))] = net(xd).cpu().numpy()
        for k in env:
            libopt.reset(k)
    base = outs[()]
    for key, got in outs.items():
        assert np.abs(got - ref).max() <= TOL, (key, np.abs(got - ref).max())
        assert np.abs(got - base).max() <= 2e-5, (key, np.abs(got - base).max())


@pytest.mark.parametrize("hw,cnn", [(7, "ResNetLarge"), (9, "ResNetLarge"), (12, "ResNetSlim"), (13, "ResNetSlim"),
                                    (15, "ResNetLarge")])
def test_encoder_other_map_sizes(gpu_device, monkeypatch, hw, cnn, libopt):
    """The encoder C entry at map sizes other than the reference's 11x11 (its ResNet heads hard-wire 1152 features, so
    this is below the module level): fused stem + layer1.conv1 (odd / even output widths, maps up to 15 wide: the LDS limit of its windows),
    plane-granule chain, float32-granule chain and the
    two-kernel path, against the oracle's conv stack."""
    import ctypes
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat, encoder as enc
    from magat_pathplanning_amd.synthetic import make_config
    lib = nat.lib()
    M = 150
    cfg = make_config(CNN_mode=cnn)
    sd = orc.init_state_dict(cfg, seed=23)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(M, 3, hw, hw, generator=g) < 0.3).float() + 0.25 * torch.rand(M, 3, hw, hw, generator=g)
    ref = orc.resnet_forward(x.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}).flatten(1)
    pack, offs, meta = enc.fold_resnet(sd, hw, hw, "ConvLayers.0", None, None)
    packd = pack.to(gpu_device)
    d = nat.EncoderDesc()
    d.variant, d.H, d.W, d.n_feat, d.n_comp, d.pack = meta["variant"], hw, hw, meta["n_feat"], 0, packd.data_ptr()
    for i, o in enumerate(offs):
        d.off[i] = o
    assert meta["n_feat"] == ref.shape[1]
    xd = x.to(gpu_device)
    ws = torch.empty(lib.magat_encoder_workspace_bytes(ctypes.byref(d), M), dtype=torch.uint8, device=gpu_device)
    scale = float(ref.abs().max())
    outs = []
    envs = [{}, {"MAGAT_L1_FUSED": "0"}, {"MAGAT_CONV_PCHAIN": "0"}]
    for env in envs:
        for k, v in env.items():
            libopt.set(k, v)
        feat = torch.full((M, meta["n_feat"]), float("nan"), device=gpu_device)
        nat.check(lib.magat_encoder_forward_f32(ctypes.byref(d), nat.ptr(xd), nat.ptr(feat), meta["n_feat"], None, 0,
                                                nat.ptr(ws), ws.numel(), M, nat.current_stream(gpu_device)), "encoder")
        torch.cuda.synchronize()
        for k in env:
            libopt.reset(k)
        got = feat.double().cpu()
        assert float((got - ref).abs().max()) <= 2e-5 * max(scale, 1.0), (env, float((got - ref).abs().max()), scale)
        outs.append(got)


def test_empty_batch_raises_like_the_reference(gpu_device):
    """B = 0: the reference's forward dies with a RuntimeError (flattening zero feature maps with `view(size(0), -1)` is ambiguous,
    decentralplanner_GAT_bottleneck.py:297 - checked against the real module in the build container); here the C ABI
    refuses M = 0 and the module raises MagatNativeError, a RuntimeError.  Nothing is silently returned."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import make_config
    cfg = make_config(num_agents=10)
    net = _build(cfg, orc.init_state_dict(cfg, seed=2), gpu_device)
    with torch.no_grad():
        net.addGSO(torch.zeros(0, 10, 10, device=gpu_device))
        with pytest.raises(RuntimeError):
            net(torch.zeros(0, 10, 3, 11, 11, device=gpu_device))


def test_unusual_inputs_behave_like_the_reference(gpu_device):
    """Probed on the real reference in the build container (and the oracle reproduces both accepted cases exactly):
    a GSO with MORE nodes than x has agents is accepted (the graph layer zero-pads the signal to N and trims the
    output, graphML.py:4641-4650), an integer GSO is accepted (the mask is |S| > 1e-9 whatever the dtype); a GSO
    whose batch differs from x's and a map size the CNN head was not built for raise RuntimeError; forward before
    addGSO raises (AttributeError there, TypeError here - SURVEY 8(b))."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=10)
    sd = orc.init_state_dict(cfg, seed=5)
    x = fov_states(2, 10, seed=1)
    net = _build(cfg, sd, gpu_device)
    with torch.no_grad():
        with pytest.raises((TypeError, AttributeError)):
            net(x.to(gpu_device))
        S11 = comm_gso(2, 11, 20, seed=2)
        ref = orc.planner_forward(x, S11.clone(), sd, cfg)
        net.addGSO(S11.clone().to(gpu_device))
        got = net(x.to(gpu_device)).cpu()
        assert got.shape == ref.shape and (got - ref).abs().max().item() <= TOL
        Si = (comm_gso(2, 10, 20, seed=3) != 0).to(torch.int64)
        ref = orc.planner_forward(x, Si.clone(), sd, cfg)
        net.addGSO(Si.clone().to(gpu_device))
        got = net(x.to(gpu_device)).cpu()
        assert (got - ref).abs().max().item() <= TOL
        net.addGSO(comm_gso(3, 10, 20, seed=4).to(gpu_device))
        with pytest.raises(RuntimeError):
            net(x.to(gpu_device))
        net.addGSO(comm_gso(2, 10, 20, seed=4).to(gpu_device))
        with pytest.raises(RuntimeError):
            net(torch.zeros(2, 10, 3, 9, 9, device=gpu_device))
        with pytest.raises(AssertionError):
            net.addGSO(torch.zeros(10, 10, device=gpu_device))


@pytest.mark.parametrize("N,B", [(13, 7), (10, 64), (100, 33), (100, 300)])
def test_forward_is_deterministic_and_independent_of_batch_composition(gpu_device, N, B):
    """The chain kernels keep an 8-agent group's maps in LDS, run persistent workgroups that prefetch the NEXT group's input and
    hand maps between stages through LDS barriers: the same batch twice must give bit-identical logits (partial groups, fewer and
    more groups than CUs), and an instance's logits must not depend on its neighbours in the batch."""
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat",
                      device=str(gpu_device))
    torch.manual_seed(N + B)
    net = DecentralPlannerGATNet(cfg).to(gpu_device).eval()
    x, S = fov_states(B, N, seed=B).to(gpu_device), comm_gso(B, N, max(8, N // 2), seed=B + 1).to(gpu_device)
    with torch.no_grad():
        net.addGSO(S)
        ref = net(x).clone()
        for _ in range(3):
            net.addGSO(S)
            assert torch.equal(net(x), ref)
        for b in (0, B // 2, B - 1):
            net.addGSO(S[b:b + 1].contiguous())
            alone = net(x[b:b + 1].contiguous())
            # (the encoder head changes its summation form with the agent count: float32 rounding, not bit equality)
            assert float((alone - ref[b * N:(b + 1) * N]).abs().max()) <= 2e-5


GNNMODEL = golden_paths("gnnmodel_")


@pytest.mark.parametrize("path", GNNMODEL, ids=[os.path.basename(p)[:-4] for p in GNNMODEL])
def test_gnn_model_vs_reference_golden(gpu_device, path):
    """DecentralPlannerNet (the GNN-baseline class, graphs/models/decentralplanner.py) on the HIP kernels - encoder, CSR
    graph-filter layer, action head - against logits made by the real reference; addGSO mutates the caller's tensor alike."""
    from magat_pathplanning_amd import DecentralPlannerNet
    z, sd, cfg = load_model_fixture(path)
    cfg.device = str(gpu_device)
    net = DecentralPlannerNet(cfg)
    net.load_state_dict(sd, strict=True)
    net = net.to(gpu_device).eval()
    x = torch.from_numpy(z["x"].astype(np.float32)).to(gpu_device)
    S = torch.from_numpy(z["S"].copy()).to(gpu_device)
    with torch.no_grad():
        net.addGSO(S)
        logits = net(x)
        again = net(x)
    torch.cuda.synchronize()
    assert logits.shape == z["logits"].shape and torch.equal(logits, again)
    assert float(np.abs(logits.cpu().numpy() - z["logits"]).max()) <= TOL
    np.testing.assert_array_equal(S.cpu().numpy(), z["S_after"])
    assert not net.range_status()["encoder_rerun"]


def test_narrow_graph_layer_with_upper_waves_past_n(gpu_device):
    """Regression (found by tools/exp/fuzz_forward.py as a GPU memory fault): G = F = 16 at N = 103 - the two-launch graph kernel's
    narrow form has 16 rows per wave step, the row groups of its upper waves lie past N, and their U-row base pointer was not
    clamped (for the last instance of a batch it pointed past the end of Z).  The configuration that faulted, against the oracle."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 2, 103
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=2, bottleneckFeature=16, bottleneckMode="BottomNeck_skipConcat",
                      CNN_mode="ResNetLarge", attentionMode="KeyQuery", AttentionConcat=True, device="cuda:0")
    sd = orc.init_state_dict(cfg, seed=100)
    x = fov_states(B, N, seed=3)
    S = comm_gso(B, N, 50, seed=4, dtype=torch.float64)
    ref = orc.planner_forward(x, S.clone(), sd, cfg).numpy()
    net = _build(cfg, sd, gpu_device)
    with torch.no_grad():
        net.addGSO(S.clone().to(gpu_device))
        got = net(x.to(gpu_device)).cpu().numpy()
    assert np.abs(got - ref).max() <= TOL
