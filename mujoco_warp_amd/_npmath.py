"""Small float64 numpy helpers used by the host-side model compiler.

Conventions follow the reference (/root/reference/mujoco_warp/_src/math.py:24-112):
quaternions are (w, x, y, z); rotation matrices are row-major world-from-local.
"""

import numpy as np

MJ_MINVAL = 1e-15


def quat_mul(u, v):
  return np.array([
    u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3],
    u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2],
    u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1],
    u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0],
  ])


def quat_conj(q):
  return np.array([q[0], -q[1], -q[2], -q[3]])


def quat_normalize(q):
  q = np.asarray(q, dtype=np.float64)
  n = np.linalg.norm(q)
  if n < MJ_MINVAL:
    return np.array([1.0, 0.0, 0.0, 0.0])
  return q / n


def quat_to_mat(q):
  w, x, y, z = q
  return np.array([
    [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
    [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
    [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
  ])


def rot_vec_quat(v, q):
  return quat_to_mat(q) @ np.asarray(v, dtype=np.float64)


def axis_angle_to_quat(axis, angle):
  axis = np.asarray(axis, dtype=np.float64)
  s, c = np.sin(angle * 0.5), np.cos(angle * 0.5)
  return np.array([c, axis[0] * s, axis[1] * s, axis[2] * s])


def quat_z2vec(vec):
  """Quaternion rotating the z-axis onto vec (reference math.py:87-112)."""
  vec = np.asarray(vec, dtype=np.float64)
  n = np.linalg.norm(vec)
  if n < MJ_MINVAL:
    return np.array([1.0, 0.0, 0.0, 0.0])
  vec = vec / n
  axis = np.array([-vec[1], vec[0], 0.0])
  a = np.linalg.norm(axis)
  if a < MJ_MINVAL:
    if vec[2] < 0:
      return np.array([0.0, 1.0, 0.0, 0.0])  # 180 deg about x
    return np.array([1.0, 0.0, 0.0, 0.0])
  axis = axis / a
  return axis_angle_to_quat(axis, np.arctan2(a, vec[2]))


def mat_to_quat(m):
  """Rotation matrix to unit quaternion (w,x,y,z)."""
  m = np.asarray(m, dtype=np.float64)
  tr = m[0, 0] + m[1, 1] + m[2, 2]
  if tr > 0:
    s = np.sqrt(tr + 1.0) * 2
    q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
  elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
    s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
    q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
  elif m[1, 1] > m[2, 2]:
    s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
    q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
  else:
    s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
    q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
  return quat_normalize(np.array(q))


def inert_vec(i, v):
  """10-vector spatial inertia times 6-vector motion (reference math.py:121-131)."""
  return np.array([
    i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5],
    i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5],
    i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4],
    i[8] * v[1] - i[7] * v[2] + i[9] * v[3],
    i[6] * v[2] - i[8] * v[0] + i[9] * v[4],
    i[7] * v[0] - i[6] * v[1] + i[9] * v[5],
  ])
