"""Direct parity at the FULL size of every single-GPU BASELINE configuration: the
HIP forward runs on the whole batch - c2 1024 x 20, c3 512 x 100, c5 128 x 1000 with bf16 storage - and a sample of its
planning instances is compared with the pinned CPU oracle run on exactly those instances (planning instances never interact,
so the oracle's rows for a sample are the whole batch's rows).  The kernel FORMS that exist only at these sizes are asserted
next to the numbers: the packed graph kernel (four instances per pass), the long-K encoder head, the persistent chain / graph
workgroups (more groups / instances than CUs), the range guard NOT having re-run anything."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _forms(nat):
    lib = nat.lib()
    return {k: int(lib.magat_form_count(i)) for k, i in nat.FORMS.items()}


def _run_full(cfg, sd, x, S, device, tag_counts):
    from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
    net = DecentralPlannerGATNet(cfg)
    net.load_state_dict(sd)
    net = net.to(device).eval()
    xd, Sd = x.to(device), S.to(device)
    with torch.no_grad():
        net.addGSO(Sd.clone())
        net(xd)                                  # (first call: packs, calibration pass - not the one that is compared)
        nat.lib().magat_form_reset()
        with tag_counts() as tc:
            net.addGSO(Sd.clone())
            got = net(xd)
        torch.cuda.synchronize()
    return net, got.cpu(), tc, _forms(nat)


def _oracle_rows(cfg, sd, x, S, pick):
    from oracle import magat_oracle as orc
    return orc.planner_forward(x[pick], S[pick].clone(), sd, cfg)


def _rows(t, pick, N):
    return torch.cat([t[b * N:(b + 1) * N] for b in pick])


@pytest.mark.parametrize("name,B,N,map_w,K,P,skip", [
    ("c2", 1024, 20, 28, 3, 4, "BottomNeck_only"),
    ("c3", 512, 100, 50, 3, 4, "BottomNeck_skipConcat"),
])
def test_full_batch_against_oracle_on_sampled_instances(gpu_device, tag_counts, name, B, N, map_w, K, P, skip):
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    cfg = make_config(num_agents=N, nGraphFilterTaps=K, nAttentionHeads=P, bottleneckMode=skip, device=str(gpu_device))
    sd = orc.init_state_dict(cfg, seed=41)
    x = fov_states(B, N, seed=11)
    S = comm_gso(B, N, map_w, seed=12)
    net, got, tc, forms = _run_full(cfg, sd, x, S, gpu_device, tag_counts)
    assert tuple(got.shape) == (B * N, 5) and bool(torch.isfinite(got).all())
    pick = [0, 1, B // 7, B // 3, B // 2, B - 129, B - 2, B - 1]          # first / last groups and packs, and the middle
    ref = _oracle_rows(cfg, sd, x, S, pick)
    err = float((_rows(got, pick, N) - ref).abs().max())
    print("%s full batch %d x %d: max|hip - oracle| over %d sampled instances = %.3g" % (name, B, N, len(pick), err))
    assert err <= 1e-4, err                       # the north star's gate, float32
    # which kernels produced these numbers
    assert tc["gat_layer (one launch)"] == 1 and tc["layer1.conv2+layer2+layer3 (fused, pooled)"] == 1, tc.counts
    assert tc["gat_graph"] == 0 and tc["gat_maps_gemm"] == 0, tc.counts
    st = net.range_status()
    assert not st["encoder_rerun"] and not st["gat_rerun"], st
    assert forms["head_longk"] == 1 and forms["head_splitk"] == 0, forms          # B N agents > HEAD_SPLITK
    assert forms["chain_persist"] == 1, forms                                   # B N / 8 groups > CUs
    assert forms["head_compress"] == (1 if B * N >= 32768 else 0), forms        # compressMLP in the head's epilogue (128-column tile)
    if N <= 32:
        assert forms["gat_pack"] == 1 and forms["gat_hsplit"] == 0, forms         # four instances per pass
    else:
        assert forms["gat_pack"] == 0 and forms["gat_persist"] == 1 and forms["gat_hsplit"] == 0, forms


def test_config5_full_batch_bf16_against_oracle_on_sampled_instances(gpu_device, tag_counts):
    """c5: 128 instances x 1000 agents, K = 2, P = 4, bf16 STORAGE inside the graph layer (CSR kernels over the device-built
    structure).  Gate: the one tests/test_gpu_model.py::test_model_bf16_gat_storage_config5_shape states for this storage
    type - |dlogit| <= 2e-2 of the logit scale against the float32 oracle, greedy actions agree on >= 97 % of the agents."""
    from oracle import magat_oracle as orc
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N = 128, 1000
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=4, gat_storage="bf16", device=str(gpu_device))
    sd = orc.init_state_dict(cfg, seed=42)
    x = fov_states(B, N, seed=13)
    S = comm_gso(B, N, 160, seed=14)
    net, got, tc, forms = _run_full(cfg, sd, x, S, gpu_device, tag_counts)
    assert net.GFL[0].storage_dtype == torch.bfloat16
    assert tuple(got.shape) == (B * N, 5) and bool(torch.isfinite(got).all())
    pick = [0, 63, 127]
    ref = _oracle_rows(cfg, sd, x, S, pick)
    rows = _rows(got, pick, N)
    err = float((rows - ref).abs().max())
    agree = float((rows.argmax(1) == ref.argmax(1)).float().mean())
    print("c5 full batch 128 x 1000 (bf16 storage): max|dlogit| = %.3g, argmax agreement = %.4f" % (err, agree))
    assert err <= 2e-2 * max(1.0, float(ref.abs().max())), err
    assert agree >= 0.97, agree
    assert tc["gat_graph"] >= 1 and tc["gat_layer (one launch)"] == 0, tc.counts          # the CSR kernels, not the dense layer
    # round 6: the maps live inside the two graph kernels - no maps GEMM launch, no Z in memory (gat_csr_fused.hip)
    assert forms["csr_fused"] >= 1 and tc["gat_maps_gemm"] == 0 and tc["gat_graph"] == 3, (forms, tc.counts)
    # (128 000 agents = two encoder passes of ENC_CHUNK = 65 536 agents: each with the long-K head and the persistent chain walk)
    assert forms["head_longk"] == 2 and forms["head_splitk"] == 0 and forms["chain_persist"] == 2, forms
