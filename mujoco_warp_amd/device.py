"""DeviceArray: the minimal array handle the reference's callers rely on.

The reference hands out `wp.array`s; callers (testspeed.py:71,261-263,299, tests) use `.numpy()`, `.shape`,
`.dtype`, `.zero_()`, `.fill_()`, `.size` and `wp.copy`.  Here device memory is a torch tensor (plumbing only:
allocator + stream), exposed through the same small surface plus `.ptr` for the C ABI.
"""

import numpy as np
import torch

_DEVICE = None


def default_device():
  """cuda:LOCAL_RANK when a GPU is visible, else cpu (host-logic tests only: kernels refuse to run)."""
  global _DEVICE
  if _DEVICE is None:
    if torch.cuda.is_available():
      import os

      _DEVICE = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
      torch.cuda.set_device(_DEVICE)
    else:
      _DEVICE = torch.device("cpu")
  return _DEVICE


def set_device(dev):
  global _DEVICE
  _DEVICE = torch.device(dev)
  if _DEVICE.type == "cuda":
    torch.cuda.set_device(_DEVICE)


_NP2T = {np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32, np.dtype(np.uint32): torch.int32,
         np.dtype(np.bool_): torch.bool}


class DeviceArray:
  __slots__ = ("t", "_unsigned")

  def __init__(self, tensor, unsigned=False):
    self.t = tensor
    self._unsigned = unsigned

  @staticmethod
  def from_numpy(a, dtype=None, device=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    unsigned = a.dtype == np.uint32
    if unsigned:
      a = a.view(np.int32)
    t = torch.from_numpy(a.copy()).to(device or default_device())
    return DeviceArray(t, unsigned)

  @staticmethod
  def zeros(shape, dtype=np.float32, device=None):
    return DeviceArray(torch.zeros(shape, dtype=_NP2T[np.dtype(dtype)], device=device or default_device()))

  @staticmethod
  def full(shape, value, dtype=np.float32, device=None):
    return DeviceArray(torch.full(shape, value, dtype=_NP2T[np.dtype(dtype)], device=device or default_device()))

  # --- the wp.array surface used by the reference's callers ---
  def numpy(self):
    a = self.t.detach().cpu().numpy()
    return a.view(np.uint32) if self._unsigned else a

  @property
  def shape(self):
    return tuple(self.t.shape)

  @property
  def dtype(self):
    if self._unsigned:
      return np.uint32
    return self.numpy().dtype if self.t.numel() == 0 else {torch.float32: np.float32, torch.int32: np.int32, torch.bool: np.bool_}[self.t.dtype]

  @property
  def size(self):
    return self.t.numel()

  @property
  def capacity(self):
    return self.t.numel() * self.t.element_size()

  @property
  def ptr(self):
    return self.t.data_ptr() if self.t.numel() else 0

  def zero_(self):
    self.t.zero_()
    return self

  def fill_(self, v):
    self.t.fill_(v)
    return self

  def assign(self, a):
    if isinstance(a, DeviceArray):
      self.t.copy_(a.t)
    else:
      src = torch.as_tensor(np.ascontiguousarray(a)).to(self.t.dtype)
      self.t.copy_(src.reshape(self.t.shape))
    return self

  def __getitem__(self, idx):
    return DeviceArray(self.t[idx], self._unsigned)

  def __len__(self):
    return self.t.shape[0]

  def __repr__(self):
    return f"DeviceArray(shape={self.shape}, dtype={self.t.dtype}, device={self.t.device})"


def copy(dst, src):
  """wp.copy equivalent."""
  dst.assign(src)
