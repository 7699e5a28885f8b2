"""Multi-GPU sharding of worlds: one process per GPU, no data-path collective.

The reference is single-device (no NCCL/MPI/sharding anywhere).  Worlds are independent, so a
node's GPUs each own a contiguous block of worlds with a private Model replica and Data shard; the only
cross-GPU traffic is ONE small all-reduce of the timing/health metrics per measurement window (RCCL over
xGMI when the backend is "nccl"; gloo in CPU tests).  Control noise uses the GLOBAL world id
(`Data.world_offset`), so trajectories do not depend on the number of GPUs.
"""

import os

import torch
import torch.distributed as dist


def dist_env():
  """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process per GPU)."""
  return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
  rank, local_rank, world_size = dist_env()
  if world_size > 1 and not dist.is_initialized():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    # MJH_DIST_BACKEND=gloo: developer knob to run several ranks on ONE GPU (RCCL refuses duplicate devices)
    backend = backend or os.environ.get("MJH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
      torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
  return rank, local_rank, world_size


def shard_worlds(nworld_total: int, rank: int, world_size: int):
  """Contiguous block [offset, offset+count) of worlds owned by `rank` (remainder spread over low ranks)."""
  if nworld_total < 0 or world_size < 1 or not (0 <= rank < world_size):
    raise ValueError("bad shard arguments")
  base, rem = divmod(nworld_total, world_size)
  count = base + (1 if rank < rem else 0)
  offset = rank * base + min(rank, rem)
  return offset, count


def barrier():
  if dist.is_initialized():
    dist.barrier()


def reduce_metrics(elapsed_s: float, env_steps: float, nan_worlds: float, overflow_worlds: float, device=None):
  """All-reduce of the per-rank measurement vector: max over ranks of time, sums of the counters.

  Returns (max_elapsed_s, total_env_steps, total_nan_worlds, total_overflow_worlds) on every rank.
  """
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return float(elapsed_s), float(env_steps), float(nan_worlds), float(overflow_worlds)
  if device is None:
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
  t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
  s = torch.tensor([env_steps, nan_worlds, overflow_worlds], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  dist.all_reduce(s, op=dist.ReduceOp.SUM)
  return float(t[0]), float(s[0]), float(s[1]), float(s[2])
