"""world_size-2 gloo tests of the multi-GPU path (runs on CPU): world sharding + the one metrics all-reduce."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mujoco_warp_amd import shard


def test_shard_worlds_partitions_exactly():
  for total, ws in ((8192, 8), (8192, 1), (10, 4), (3, 8), (0, 2)):
    blocks = [shard.shard_worlds(total, r, ws) for r in range(ws)]
    assert sum(c for _, c in blocks) == total
    off = 0
    for o, c in blocks:
      assert o == off
      off += c
    assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
  with pytest.raises(ValueError):
    shard.shard_worlds(8, 2, 2)


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world_size, port, out):
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  r, lr, ws = shard.init_process_group(backend="gloo")
  assert (r, ws) == (rank, world_size)
  off, cnt = shard.shard_worlds(101, r, ws)
  # the global world ids seen by the two ranks tile [0, 101) exactly
  ids = torch.zeros(101, dtype=torch.int64)
  ids[off : off + cnt] = 1
  dist.all_reduce(ids)
  assert bool((ids == 1).all())
  shard.barrier()
  t, steps, nan, ovf = shard.reduce_metrics(elapsed_s=1.0 + rank, env_steps=cnt * 10.0, nan_worlds=rank, overflow_worlds=2.0)
  out[rank] = (t, steps, nan, ovf)
  dist.destroy_process_group()


def test_metrics_allreduce_gloo_world_size_2():
  port = _free_port()
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
  for rank in (0, 1):
    t, steps, nan, ovf = out[rank]
    assert t == 2.0  # max over ranks
    assert steps == 1010.0  # 101 worlds x 10 steps, summed
    assert nan == 1.0 and ovf == 4.0


def test_reduce_metrics_single_process():
  assert shard.reduce_metrics(0.5, 100.0, 0.0, 1.0) == (0.5, 100.0, 0.0, 1.0)
