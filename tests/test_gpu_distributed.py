"""N>1 path on the GPU: instance sharding over RCCL (torch.distributed backend "nccl") with the HIP forward, and the
driver's launch commands for bench.py.  One GPU is enough for world size 1 (RCCL is initialised for real); the
world-size-2 case runs when two devices are visible."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, N, ret, backend="nccl", one_device=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    idx = 0 if one_device else rank
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from magat_pathplanning_amd import DecentralPlannerGATNet, _native
        from magat_pathplanning_amd.distributed import shard_range, sharded_forward
        from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
        from oracle import magat_oracle as orc
        assert _native.library_present()
        cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, device=str(dev),
                          bottleneckMode="BottomNeck_skipConcat")
        net = DecentralPlannerGATNet(cfg)
        net.load_state_dict(orc.init_state_dict(cfg, seed=5))
        net = net.to(dev).eval()
        x, S = fov_states(B, N, seed=3), comm_gso(B, N, 28, seed=4).to(dev)
        # the shards differ in magnitude (sparse maps in the first half of the batch, saturated ones in the second): every
        # rank's FIRST batch is its own shard, and the exponents of the split arithmetic must not follow it
        x[: B // 2, :, 0] *= 0.0
        x[B // 2:, :, 0, 1:-1, 1:-1] = 1.0
        x[B // 2:, :, 2, 1:-1, 1:-1] = 1.0
        x = x.to(dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        assert int(ones.item()) == world
        with torch.no_grad():
            full = sharded_forward(net, x, S.clone(), gather=True)
            scales = [None] * world
            dist.all_gather_object(scales, net.range_status()["act_scales"])
            assert all(s == scales[0] for s in scales) and scales[0]["source"] == "canonical", scales
            local = sharded_forward(net, x, S.clone(), gather=False)
            b0, b1 = shard_range(B, rank, world)
            assert local.shape[0] == (b1 - b0) * N
            assert torch.equal(full[b0 * N:b1 * N], local)
            net.addGSO(S.clone())
            single = net(x)
        # instance shards never interact: the gathered logits ARE the single-process ones, bit for bit
        assert torch.equal(full, single)
        if world > 1:
            # fewer instances than ranks, inputs host-resident: the rank that owns nothing still enters the all_gather,
            # with a placeholder on ITS device (a CPU placeholder under nccl would fail and leave the others blocked)
            with torch.no_grad():
                one = sharded_forward(net, x[:1].cpu(), S[:1].clone(), gather=True)
            assert torch.equal(one, single[:N])
        if rank == 0:
            ret["ok"] = True
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_forward_hip_over_rccl(gpu_device, world):
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d visible GPUs" % world)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), 6, 20, ret), nprocs=world, join=True)
        assert ret.get("ok")


def test_two_processes_share_one_gpu(gpu_device):
    """World size 2 WITHOUT a second GPU: two spawned processes drive the same device (the HIP forward of each rank's shard
    runs on it; gloo carries the barrier and the gather).  Each process builds its own module, folds its own activation
    scales and sees only its own - deliberately unlike - shard first: the gathered logits equal the single-process result
    bit for bit."""
    import torch.multiprocessing as mp
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), 6, 20, ret, "gloo", True), nprocs=2, join=True)
        assert ret.get("ok")


def _one_json_line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_runs_as_the_driver_launches_it(gpu_device):
    """`python bench.py --gpus 1 ...` and the torch.distributed.run form the driver uses for N > 1 (here with one rank:
    RCCL is initialised, ranks_seen comes from an all_reduce of ones) both print exactly one JSON line and exit 0."""
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["value"] > 0 and "roofline" in d
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["value"] > 0


def test_bench_eight_ranks_rehearsed_on_one_gpu(gpu_device):
    """the driver's N = 8 command line - torch.distributed.run with eight ranks of bench.py - run for
    real on a one-GPU box (`--share-gpu`: every rank drives cuda:0, gloo carries the collectives).  What is exercised is
    bench.py's OWN rank logic: RANK / LOCAL_RANK / WORLD_SIZE from the environment, the sum of ones (ranks_seen == 8), the
    barriers either side of the timed region, the gather of every rank's elapsed time (per_rank_ms has eight entries and the
    reported ms_per_step is their MAX), whole-job `value` = 8 x batch x N x steps / that time, one JSON line from rank 0
    only.  The first real 8-GPU run is then not the first run of this code; its numbers are not a scaling measurement."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    steps, batch = 3, 16
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", str(steps), "--warmup", "1",
                        "--batch", str(batch), "--no-cpu-baseline", "--no-kernel-timing", "--share-gpu"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["scaling"] == "weak"
    assert len(d["per_rank_ms"]) == 8 and len(d["ranks"]) == 8
    assert sorted(x["rank"] for x in d["ranks"]) == list(range(8))
    assert abs(d["ms_per_step"] - max(d["per_rank_ms"])) <= 1e-3
    assert d["config"]["global_batch"] == 8 * batch and "rehearsal" in d["config"]
    want = 8 * batch * 100 / (d["ms_per_step"] * 1e-3)
    assert abs(d["value"] - want) <= 1e-3 * want
