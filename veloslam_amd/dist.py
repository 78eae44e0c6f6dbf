"""Multi-GPU plumbing for the one exchange step of the path: after a
registration round every rank contributes its accepted map increment and all ranks
append all blocks in rank order, so every replica of the map stays identical.

torch.distributed is used as plumbing only (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Two-phase all-gather-v: counts first, then
max-padded blocks -- the blocks are a few hundred KB, latency-bound on the 7 direct
xGMI links, so one padded all-gather beats 7 ring steps of exact sizes."""
import torch
import torch.distributed as dist


def exchange_increments(inc_xyz, count, group=None):
    """inc_xyz: (3, cap) float32 tensor whose first `count` columns are this rank's
    increment (on the device of the process group's backend).  Returns
    (blocks, counts): blocks is a list over ranks of (3, counts[r]) tensors, in rank
    order.  Single-process (no initialised group): returns the local block."""
    if not (dist.is_available() and dist.is_initialized()):
        return [inc_xyz[:, :count]], [int(count)]
    world = dist.get_world_size(group)
    # gloo (CPU tests, one-GPU functional checks) gathers host tensors; nccl = RCCL gathers in HBM
    cdev = inc_xyz.device if dist.get_backend(group) != "gloo" else torch.device("cpu")
    cnt = torch.tensor([int(count)], dtype=torch.int32, device=cdev)
    counts = torch.zeros(world, dtype=torch.int32, device=cdev)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts_h = [int(v) for v in counts.cpu().tolist()]
    pad = max(max(counts_h), 1)
    send = torch.zeros((3, pad), dtype=torch.float32, device=cdev)
    send[:, :count] = inc_xyz[:, :count].to(cdev)
    recv = torch.empty((world, 3, pad), dtype=torch.float32, device=cdev)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    recv = recv.to(inc_xyz.device)
    return [recv[r, :, :counts_h[r]] for r in range(world)], counts_h


def shard_units(n_units, rank, world):
    """Frame-parallel sharding with no data-path collective: unit u -> rank u % world."""
    return [u for u in range(n_units) if u % world == rank]
