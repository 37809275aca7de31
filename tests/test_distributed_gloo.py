"""N>1 path on CPU: world_size-2 gloo processes shard the instance axis; the gathered logits equal the
single-process result (the model runs its differentiable torch composite here, because the HIP inference
path needs a GPU; the sharding logic is device-independent)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magat_pathplanning_amd import DecentralPlannerGATNet
        from magat_pathplanning_amd.distributed import shard_range, sharded_forward
        from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
        torch.manual_seed(0)
        net = DecentralPlannerGATNet(make_config(num_agents=6, nGraphFilterTaps=3, nAttentionHeads=2, device="cpu",
                                                 bottleneckFeature=32, bottleneckMode="BottomNeck_skipConcatGNN")).eval()
        x, S = fov_states(B, 6, seed=3), comm_gso(B, 6, 12, seed=4)
        full = sharded_forward(net, x, S.clone(), gather=True)
        local = sharded_forward(net, x, S.clone(), gather=False)
        b0, b1 = shard_range(B, rank, world)
        assert local.shape[0] == (b1 - b0) * 6
        assert torch.equal(full[b0 * 6:b1 * 6], local.detach())
        if rank == 0:
            net.addGSO(S.clone())
            ret["single"] = net(x).detach()
            ret["full"] = full
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5, 1])      # 1: fewer instances than ranks - rank 1 owns an empty shard
def test_two_rank_gloo_shards_equal_single_process(B):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, B, ret), nprocs=2, join=True)
        assert torch.allclose(ret["full"], ret["single"], rtol=0, atol=1e-6)


def test_shard_range_partitions_exactly():
    from magat_pathplanning_amd.distributed import shard_range
    for B in (1, 7, 512, 4096):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
