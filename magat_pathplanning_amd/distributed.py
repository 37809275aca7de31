"""Instance sharding across the GPUs of a node (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).  Planning instances are independent (the GSO is
block-diagonal over B), so the forward needs NO collective: each rank runs its
contiguous slice of the batch.  Only callers that want the whole batch's logits on every rank pay one
all_gather of (B/R*N, 5) floats (8 MB total at B=4096, N=100)."""
import torch
import torch.distributed as dist


def shard_range(num_instances, rank, world):
    """Contiguous, balanced slice [start, stop) of the instance axis owned by `rank`."""
    base, rem = divmod(num_instances, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def sharded_forward(model, x, S, group=None, gather=True):
    """x (B,N,3,W,H), S (B,N,N): the FULL batch on every rank (or anything indexable the same way).
    Runs addGSO+forward on this rank's instances only.  gather=False -> local logits ((b1-b0)*N, 5);
    gather=True -> full (B*N, 5) logits on every rank, identical to the single-process result."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, N = x.shape[0], x.shape[1]
    b0, b1 = shard_range(B, rank, world)
    if b1 > b0:
        Sl = S[b0:b1].contiguous()
        model.addGSO(Sl)
        # the kernel forms that depend on the batch size (the encoder head's split-K form for few agents) are chosen on the
        # GLOBAL agent count, not on the shard's: 64 instances of 100 agents over 8 ranks (800 agents each) sum in the order
        # the 6400-agent batch does - the shards concatenate to the single-process result BIT FOR BIT with default options
        prev = getattr(model, "form_agents", 0)
        model.form_agents = B * N
        try:
            local = model(x[b0:b1])
        finally:
            model.form_agents = prev
    else:
        # fewer instances than ranks: this rank owns nothing, but it still has to enter the all_gather below (a rank that
        # raised on its empty batch would leave the others blocked in the collective)
        # (on the device and with the width the other ranks' logits have - x may be host-resident)
        dev = next(model.parameters()).device
        width = [m for m in model.actionsMLP if isinstance(m, torch.nn.Linear)][-1].out_features
        local = torch.zeros(0, width, dtype=torch.float32, device=dev)
    if not gather or world == 1:
        return local
    cap = (B + world - 1) // world * N
    # (gloo has no all_gather over device tensors: staged through the host there - the CPU tests, and the two-processes-on-
    #  one-GPU test; under nccl = RCCL the logits never leave the devices)
    via_host = local.is_cuda and dist.get_backend(group) == "gloo"
    pad = torch.zeros(cap, local.shape[1], dtype=local.dtype, device="cpu" if via_host else local.device)
    pad[: local.shape[0]] = local.detach()
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    rows = [shard_range(B, r, world) for r in range(world)]
    return torch.cat([parts[r][: (e - s) * N] for r, (s, e) in enumerate(rows)], dim=0).to(local.device)
