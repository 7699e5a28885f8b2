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


def test_bench_respawn_builds_the_torchrun_command(monkeypatch):
  """`python bench.py --gpus N` without a torch.distributed environment re-executes itself under torch.distributed.run: one rank per
  GPU on this node, rendezvous on 127.0.0.1, the caller's flags passed through, dmabuf IPC mode kept in the environment."""
  import sys

  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import bench

  seen = {}

  def fake_call(cmd, env=None):
    seen["cmd"], seen["env"] = cmd, env
    return 7

  monkeypatch.setattr(bench.subprocess, "call", fake_call)
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5", "--scaling", "strong"])
  monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
  monkeypatch.setenv("MJH_DIST_BACKEND", "gloo")
  assert bench.respawn_under_torchrun(4) == 7
  cmd = seen["cmd"]
  assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
  assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
  assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
  i = cmd.index(os.path.abspath(bench.__file__))
  assert cmd[i + 1 :] == ["--gpus", "4", "--steps", "20", "--warmup", "5", "--scaling", "strong"]
  assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["MJH_DIST_BACKEND"] == "gloo"


def test_bench_main_respawns_only_without_a_distributed_environment(monkeypatch):
  import sys

  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import bench

  calls = []
  monkeypatch.setattr(bench, "respawn_under_torchrun", lambda n: calls.append(n) or 0)
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  with pytest.raises(SystemExit) as e:
    bench.main()
  assert e.value.code == 0 and calls == [2]
  with pytest.raises(SystemExit):  # --gpus must be positive
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "0"])
    bench.main()


@pytest.mark.gpu
def test_gpu_shard_invariance_with_control_noise():
  """(e): 64 worlds stepped as ONE Data are bitwise the 64 worlds stepped as two Data of 32 with world_offset 0 / 32 -- control noise
  on (it is keyed by the global world id), 50 steps, CG and Newton.  The solver's longest-first schedule and world pairing differ
  between the two layouts; no world's arithmetic may depend on which worlds share its wavefront."""
  import mujoco_warp_amd as mjw
  from tests import conftest

  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  for solver in ("cg", "newton"):
    mjw.override_model(mjm, {"opt.solver": solver})
    m = mjw.put_model(mjm)
    whole = mjw.make_data(mjm, nworld=64, nconmax=24, njmax=64)
    halves = [mjw.make_data(mjm, nworld=32, nconmax=24, njmax=64) for _ in range(2)]
    halves[1].world_offset = 32
    for d in [whole] + halves:
      mjw.reset_data_keyframe(m, d, 0)
    for i in range(50):
      for d in [whole] + halves:
        mjw.ctrl_noise(m, d, i)
        mjw.step(m, d)
    q = np.concatenate([h.qpos.numpy() for h in halves])
    v = np.concatenate([h.qvel.numpy() for h in halves])
    c = np.concatenate([h.ctrl.numpy() for h in halves])
    assert (whole.ctrl.numpy() == c).all(), solver
    assert len(np.unique(whole.qpos.numpy(), axis=0)) == 64  # the noise really decorrelated the worlds
    assert (whole.qpos.numpy() == q).all() and (whole.qvel.numpy() == v).all(), solver
    assert (whole.solver_niter.numpy() == np.concatenate([h.solver_niter.numpy() for h in halves])).all()
