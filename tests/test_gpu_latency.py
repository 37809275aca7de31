"""Latency forms of the batch-1 step (the reference's own inference loop: `test_batch_size = 1`,
configs/dcpGAT_OE_Random.json:58; agents/decentralplannerlocal_OnlineExpert_GAT.py:1030-1055): few agents take kernels that
spread ONE planning instance over the chip.  Every latency form is held against the form it replaces BIT FOR BIT, against
the pinned oracle at the north star's gate, and the form counters say which one ran."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _build(cfg, sd, device):
    from magat_pathplanning_amd import DecentralPlannerGATNet
    cfg.device = str(device)
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd, strict=True)
    return net.to(device).eval()


@pytest.mark.parametrize("B,N", [(1, 10), (1, 100), (1, 1), (3, 37), (2, 128), (1, 129), (1, 7), (4, 100), (4, 128), (64, 8)])
def test_one_agent_per_workgroup_encoder_equals_the_batched_forms(gpu_device, libopt, tag_counts, B, N):
    """block_lat_kernel - one agent per workgroup and ONE launch for the whole encoder: the stem and layer1.conv1 (stem8_kernel's
    arithmetic), the BasicBlock chain on zero-bordered maps with two row tiles, the encoder head and compressMLP in its epilogue,
    the range guard inside - against the BATCHED forms of the same layers run on the same few agents (options LAT_AGENTS = 0,
    HEAD_SPLITK = 0: stem8_kernel and block_full_p_kernel with eight agents per workgroup and nine row tiles by tap-validity
    class, the long-K f16x3 head, the f16x3 compressMLP): the same products in the same order per output element, the same
    pairing of a pooled cell's four pixels, the same plane splits - logits equal BIT FOR BIT.  Agent counts on both sides of an
    agent tile (128)."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    sd = orc.init_state_dict(cfg, seed=41 + N)
    net = _build(cfg, sd, gpu_device)
    x = fov_states(B, N, seed=5 + N).to(gpu_device)
    S = comm_gso(B, N, 50, seed=6 + N).to(gpu_device)
    lib = nat.lib()
    with torch.no_grad():
        net.addGSO(S.clone())
        net(x)                                           # (the first forward folds the activation scales)
        lib.magat_form_reset()
        with tag_counts() as tc:
            net.addGSO(S.clone())
            lat = net(x).clone()
        # the encoder is one launch: no stem launch, no head / compressMLP launches, no encoder guard launches
        assert tc["conv_first"] == 0 and tc["head(avgpool+fc+linear)"] == 0 and tc["compressMLP"] == 0, tc.counts
        assert tc["layer1.conv2+layer2+layer3 (fused, pooled)"] == 1, tc.counts
        if B * 4 <= 64 and N <= 128:
            # ... the graph layer's predicated float32 re-run is ONE launch (gat_rerun_small_kernel), and the action head rides
            # in it (magat_gat_forward_tail_f32): three launches a step
            assert tc["range_guard"] == 1 and tc["gat_layer (one launch)"] == 1 and sum(tc.counts.values()) == 3, tc.counts
            assert lib.magat_form_count(nat.FORMS["actions_tail"]) == 1
        assert lib.magat_form_count(nat.FORMS["chain_lat"]) == 1 and lib.magat_form_count(nat.FORMS["head_lat"]) == 1
        assert lib.magat_form_count(nat.FORMS["stem_lat"]) == 1 and lib.magat_form_count(nat.FORMS["guard_lat"]) == 1
        assert lib.magat_form_count(nat.FORMS["head_splitk"]) == 0 and lib.magat_form_count(nat.FORMS["head_longk"]) == 0
        libopt.set("MAGAT_LAT_AGENTS", 0)
        libopt.set("MAGAT_HEAD_SPLITK", 0)
        lib.magat_form_reset()
        net.addGSO(S.clone())
        batched = net(x).clone()
        assert lib.magat_form_count(nat.FORMS["chain_lat"]) == 0 and lib.magat_form_count(nat.FORMS["head_longk"]) == 1
    assert torch.equal(lat, batched)
    ref = orc.planner_forward(x.cpu(), S.cpu().clone(), sd, cfg)
    assert float((lat.cpu() - ref).abs().max()) <= TOL
    st = net.range_status()
    assert not st["encoder_rerun"] and not st["gat_rerun"], st


@pytest.mark.parametrize("N,concat", [(40, False), (100, False), (64, True)])
def test_published_widths_batch_one_equals_its_rows_of_a_large_batch(gpu_device, N, concat):
    """The published widths (bottleneck 32, K = 2, four heads) on 33 .. 128 agents: ONE instance runs gat_mid_kernel with a
    workgroup per HEAD (head-mean: pre-activation rows + gat_mid_mean_kernel), a batch runs a workgroup per instance - the same
    logits bit for bit, as for the encoder."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B = 6400 // N
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=4, bottleneckFeature=32, bottleneckMode="BottomNeck_only",
                      AttentionConcat=concat)
    sd = orc.init_state_dict(cfg, seed=13)
    net = _build(cfg, sd, gpu_device)
    x = fov_states(B, N, seed=8).to(gpu_device)
    S = comm_gso(B, N, 50, seed=9).to(gpu_device)
    lib = nat.lib()
    with torch.no_grad():
        net.addGSO(S.clone())
        net(x)
        lib.magat_form_reset()
        net.addGSO(S.clone())
        whole = net(x).clone()
        assert lib.magat_form_count(nat.FORMS["gat_mid"]) == 1 and lib.magat_form_count(nat.FORMS["gat_hsplit"]) == 0
        for b in (0, B // 2, B - 1):
            lib.magat_form_reset()
            net.addGSO(S[b:b + 1].clone())
            alone = net(x[b:b + 1]).clone()
            assert lib.magat_form_count(nat.FORMS["gat_hsplit"]) == 1 and lib.magat_form_count(nat.FORMS["head_lat"]) == 1
            assert torch.equal(alone, whole[b * N:(b + 1) * N]), (N, b, float((alone - whole[b * N:(b + 1) * N]).abs().max()))
    ref = orc.planner_forward(x[:2].cpu(), S[:2].cpu().clone(), sd, cfg)
    assert float((whole[:2 * N].cpu() - ref).abs().max()) <= TOL


@pytest.mark.parametrize("N", [10, 100])
def test_batch_one_step_equals_its_rows_of_a_large_batch(gpu_device, N):
    """The reference's inference loop presents ONE planning instance per step; a benchmark (or a shard of BASELINE config 3)
    presents hundreds.  With the latency forms the logits of an instance do not depend on which: alone (one agent per
    workgroup, head in the chain kernel's epilogue, a workgroup per attention head) or as rows of a 6 400-agent batch
    (eight-agent groups, long-K head, packed / persistent graph kernel) - bit for bit."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B = 6400 // N
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    sd = orc.init_state_dict(cfg, seed=77)
    net = _build(cfg, sd, gpu_device)
    x = fov_states(B, N, seed=8).to(gpu_device)
    S = comm_gso(B, N, 50 if N > 20 else 20, seed=9).to(gpu_device)
    lib = nat.lib()
    with torch.no_grad():
        net.addGSO(S.clone())
        net(x)
        lib.magat_form_reset()
        net.addGSO(S.clone())
        whole = net(x).clone()
        assert lib.magat_form_count(nat.FORMS["chain_lat"]) == 0 and lib.magat_form_count(nat.FORMS["head_longk"]) == 1
        for b in (0, 1, B // 2, B - 1):
            lib.magat_form_reset()
            net.addGSO(S[b:b + 1].clone())
            alone = net(x[b:b + 1]).clone()
            assert lib.magat_form_count(nat.FORMS["head_lat"]) == 1
            assert torch.equal(alone, whole[b * N:(b + 1) * N]), (N, b, float((alone - whole[b * N:(b + 1) * N]).abs().max()))


@pytest.mark.parametrize("G,N,concat", [(32, 10, False), (32, 100, False), (64, 40, True), (64, 16, False)])
def test_published_widths_take_the_one_launch_encoder(gpu_device, libopt, G, N, concat):
    """The published checkpoints (MAGAT F-32-P4 / B-32-P4, scripts/train_DMap.sh:42-46: bottleneckFeature 32, K = 2, four heads,
    head-mean) in the reference's batch-1 loop on the README's 10 .. 100-robot sets: compressMLP has 32 (64) outputs - the first
    one (two) waves of the one-launch encoder compute them - and the graph layer is gat_small / gat_mid.  Bit-identical to the
    batched forms of the same layers, 1e-4 from the oracle."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=4, bottleneckFeature=G, bottleneckMode="BottomNeck_only",
                      AttentionConcat=concat)
    sd = orc.init_state_dict(cfg, seed=90 + G + N)
    net = _build(cfg, sd, gpu_device)
    x = fov_states(1, N, seed=2 + N).to(gpu_device)
    S = comm_gso(1, N, 20 if N <= 20 else 50, seed=3 + N, dtype=torch.float64).to(gpu_device)
    lib = nat.lib()
    with torch.no_grad():
        net.addGSO(S.clone())
        net(x)
        lib.magat_form_reset()
        net.addGSO(S.clone())
        lat = net(x).clone()
        assert lib.magat_form_count(nat.FORMS["head_lat"]) == 1 and lib.magat_form_count(nat.FORMS["stem_lat"]) == 1
        libopt.set("MAGAT_LAT_AGENTS", 0)
        libopt.set("MAGAT_HEAD_SPLITK", 0)
        lib.magat_form_reset()
        net.addGSO(S.clone())
        batched = net(x).clone()
        assert lib.magat_form_count(nat.FORMS["chain_lat"]) == 0
    assert torch.equal(lat, batched)
    ref = orc.planner_forward(x.cpu(), S.cpu().clone(), sd, cfg)
    assert float((lat.cpu() - ref).abs().max()) <= TOL
    st = net.range_status()
    assert not st["encoder_rerun"] and not st["gat_rerun"], st


def test_latency_form_is_chosen_on_the_global_agent_count(gpu_device):
    """A shard of a large batch (form_agents = the global count, distributed.sharded_forward) keeps the batched forms even when
    the shard itself is small: shards and the whole batch stay bit-identical with default options."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 6, 100
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    sd = orc.init_state_dict(cfg, seed=3)
    net = _build(cfg, sd, gpu_device)
    x = fov_states(B, N, seed=1).to(gpu_device)
    S = comm_gso(B, N, 50, seed=2).to(gpu_device)
    lib = nat.lib()
    with torch.no_grad():
        net.addGSO(S.clone())
        net(x)
        lib.magat_form_reset()
        net.addGSO(S.clone())
        whole = net(x).clone()
        assert lib.magat_form_count(nat.FORMS["chain_lat"]) == 0      # 600 agents: above LAT_AGENTS
        net.form_agents = B * N
        lib.magat_form_reset()
        parts = []
        for b in range(B):
            net.addGSO(S[b:b + 1].contiguous())
            parts.append(net(x[b:b + 1]).clone())
        net.form_agents = 0
        assert lib.magat_form_count(nat.FORMS["chain_lat"]) == 0
    assert torch.equal(torch.cat(parts), whole)


def test_latency_chain_range_guard_rerun(gpu_device, monkeypatch):
    """A checkpoint whose maps leave the f16 planes' range inside the chain (layer2 / layer3 maps ~1e5, activation scales off):
    the one-agent-per-workgroup kernel raises the encoder's flag like the eight-agent form, and the predicated float32 re-run
    produces the logits (tests/test_gpu_range.py holds the whole table of cases; 80 agents take the latency form there too)."""
    from magat_pathplanning_amd import _native as nat
    from test_gpu_range import _run, _scaled_model
    monkeypatch.setenv("MAGAT_ACT_SCALE", "0")
    cfg, sd, net = _scaled_model(gpu_device, 1.0e5, where="layer2")
    lib = nat.lib()
    lib.magat_form_reset()
    got, ref = _run(net, cfg, sd, gpu_device)
    assert lib.magat_form_count(nat.FORMS["chain_lat"]) == 1
    st = net.range_status()
    assert st["encoder_rerun"], st
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("skip", ["BottomNeck_skipConcat", "BottomNeck_only", "BottomNeck_skipConcatGNN"])
@pytest.mark.parametrize("B,N,dtype", [(1, 10, torch.float64), (1, 100, torch.float32), (4, 20, torch.float32)])
def test_step_plan_equals_the_general_host_path(gpu_device, B, N, dtype, skip):
    """The step plan (planner._plan_build: buffers, workspaces, packed weights and the ctypes arguments of the three C-ABI calls
    resolved once per (weights, batch shape)) launches the same kernels with the same arguments as the general host path:
    logits bit-identical, also when the inputs live in NEW tensors every step, when the weights change in place (plan
    dropped, rebuilt), and after a forward of another shape in between."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode=skip)
    sd = orc.init_state_dict(cfg, seed=5)
    net = _build(cfg, sd, gpu_device)
    x = fov_states(B, N, seed=1).to(gpu_device)
    S = comm_gso(B, N, 50, seed=2, dtype=dtype).to(gpu_device)
    with torch.no_grad():
        net.addGSO(S.clone()); net(x)                        # general path (calibrates), builds the plan
        assert net._rt.plan is not None
        net.addGSO(S.clone()); planned = net(x.clone()).clone()
        net.step_plan = False
        net.addGSO(S.clone()); general = net(x).clone()
        net.step_plan = True
        assert torch.equal(planned, general)
        # another shape in between drops the plan; coming back rebuilds it
        x2, S2 = fov_states(2, N, seed=3).to(gpu_device), comm_gso(2, N, 50, seed=4, dtype=dtype).to(gpu_device)
        net.addGSO(S2.clone()); net(x2)
        net.addGSO(S.clone()); again = net(x).clone()
        net.addGSO(S.clone()); again2 = net(x).clone()
        assert torch.equal(again, planned) and torch.equal(again2, planned)
        # weights changed in place: the plan belongs to the old ones
        net.actionsMLP[0].bias.add_(1.0)
        net.addGSO(S.clone()); moved = net(x).clone()
        assert torch.allclose(moved, planned + 1.0, atol=1e-5)
    ref = orc.planner_forward(x.cpu(), S.cpu().clone(), sd, cfg)
    assert float((planned.cpu() - ref).abs().max()) <= TOL
