import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import magat_oracle as orc
from magat_pathplanning_amd import _native as nat, DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
for B in (331, 64, 512):
    N = 100
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
    sd = orc.init_state_dict(cfg, seed=23)
    net = DecentralPlannerGATNet(cfg).to(dev).eval(); net.load_state_dict(sd)
    x = fov_states(B, N, seed=9).to(dev); S = comm_gso(B, N, 50, seed=10).to(dev)
    lib = nat.lib()
    with torch.no_grad():
        net.addGSO(S.clone()); net(x)
        lib.magat_form_reset(); net.addGSO(S.clone()); fused = net(x).clone()
        fc = lib.magat_form_count(nat.FORMS["head_compress"])
        st = net.range_status() if hasattr(net, "range_status") else None
        nat.set_option("HEAD_COMPRESS", 0) if hasattr(nat, "set_option") else os.environ.__setitem__("MAGAT_HEAD_COMPRESS", "0")
        net.addGSO(S.clone()); two = net(x).clone()
        nat.set_option("HEAD_COMPRESS", 1) if hasattr(nat, "set_option") else None
    d = (fused - two).abs().view(B * N, -1).max(dim=1).values
    bad = torch.nonzero(d > 0).flatten()
    print("B", B, "fused form count", fc, "rows differing", bad.numel(), "of", B * N, "max diff", float(d.max()), "nan", bool(torch.isnan(fused).any()))
    if bad.numel():
        b = bad.cpu()
        print("  first bad rows", b[:12].tolist(), " last", b[-4:].tolist())
        import collections
        print("  bad rows mod 128 histogram (top):", collections.Counter((b % 128).tolist()).most_common(6))
        print("  bad tiles (row // 128) count", len(set((b // 128).tolist())), "of", (B * N + 127) // 128)
    print("  status", st)
