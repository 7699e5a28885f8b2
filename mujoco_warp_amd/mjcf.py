"""Minimal MJCF-subset compiler producing an MjModel-like numpy struct.

Why this exists: the reference's `put_model(mjm: mujoco.MjModel)` consumes a model compiled by
MuJoCo C (/root/reference/mujoco_warp/_src/io.py:259, cli.py:87-95). The `mujoco` package is not
available in this environment, so the hot-path benchmark models (humanoid.xml, and G1/Panda-class
models with primitive colliders and explicit inertials) are compiled here.  Attribute names and
array shapes follow MuJoCo's MjModel so `put_model` can `getattr` them exactly like the reference
does (io.py:426).  If a real `mujoco.MjModel` is handed to `put_model`, it is used as is.

Derived constants (`mj_setConst`): body_subtreemass, dof_invweight0, body_invweight0,
stat.meaninertia follow the restatement in /root/reference/mujoco_warp/_src/set_const.py:35-59,
170-375.  Compiler semantics (inertia-from-geom, fromto, defaults) follow MuJoCo's documented
behaviour; they are NOT restated anywhere in the reference.

Supported: nested <default class>, childclass, <include>, bodies, <inertial>, free/ball/slide/hinge
joints, plane/sphere/capsule/cylinder/ellipsoid/box geoms (mesh geoms are kept as non-colliding
placeholders when they carry no mass), motor/position/velocity/general actuators on joints,
keyframes, <contact><exclude>, <option> + <flag>.  Unsupported features raise NotImplementedError.
"""

import copy
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import _npmath as nm

# ---- enum values (MuJoCo's mjt* enums) ------------------------------------------------------
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH, GEOM_SDF = range(9)
_GEOM_NAMES = {
  "plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
  "ellipsoid": GEOM_ELLIPSOID, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH, "sdf": GEOM_SDF,
}
_JNT_NAMES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
INT_EULER, INT_RK4, INT_IMPLICIT, INT_IMPLICITFAST = 0, 1, 2, 3
_INT_NAMES = {"euler": INT_EULER, "rk4": INT_RK4, "implicit": INT_IMPLICIT, "implicitfast": INT_IMPLICITFAST}
SOL_PGS, SOL_CG, SOL_NEWTON = 0, 1, 2
_SOL_NAMES = {"pgs": SOL_PGS, "cg": SOL_CG, "newton": SOL_NEWTON}
CONE_PYRAMIDAL, CONE_ELLIPTIC = 0, 1
JAC_DENSE, JAC_SPARSE, JAC_AUTO = 0, 1, 2
DYN_NONE, DYN_INTEGRATOR, DYN_FILTER, DYN_FILTEREXACT, DYN_MUSCLE, DYN_USER = range(6)
GAIN_FIXED, GAIN_AFFINE, GAIN_MUSCLE, GAIN_USER = range(4)
BIAS_NONE, BIAS_AFFINE, BIAS_MUSCLE, BIAS_USER = range(4)
TRN_JOINT, TRN_JOINTINPARENT, TRN_SLIDERCRANK, TRN_TENDON, TRN_SITE, TRN_BODY = range(6)

DSBL = {
  "constraint": 1 << 0, "equality": 1 << 1, "frictionloss": 1 << 2, "limit": 1 << 3, "contact": 1 << 4,
  "spring": 1 << 5, "damper": 1 << 6, "gravity": 1 << 7, "clampctrl": 1 << 8, "warmstart": 1 << 9,
  "filterparent": 1 << 10, "actuation": 1 << 11, "refsafe": 1 << 12, "sensor": 1 << 13, "midphase": 1 << 14,
  "eulerdamp": 1 << 15, "autoreset": 1 << 16, "nativeccd": 1 << 17, "island": 1 << 18, "multiccd": 1 << 19,
}
ENBL = {"override": 1 << 0, "energy": 1 << 1, "fwdinv": 1 << 2, "invdiscrete": 1 << 3, "sleep": 1 << 5}

MJ_MINVAL = 1e-15


class MjOption:
  def __init__(self):
    self.timestep = 0.002
    self.tolerance = 1e-8
    self.ls_tolerance = 0.01
    self.ccd_tolerance = 1e-6
    self.sleep_tolerance = 1e-4
    self.noslip_tolerance = 1e-6
    self.gravity = np.array([0.0, 0.0, -9.81])
    self.wind = np.zeros(3)
    self.magnetic = np.array([0.0, -0.5, 0.0])
    self.density = 0.0
    self.viscosity = 0.0
    self.impratio = 1.0
    self.integrator = INT_EULER
    self.cone = CONE_PYRAMIDAL
    self.jacobian = JAC_AUTO
    self.solver = SOL_NEWTON
    self.iterations = 100
    self.ls_iterations = 50
    self.noslip_iterations = 0
    self.ccd_iterations = 35
    self.disableflags = 0
    self.enableflags = 0
    self.sdf_initpoints = 40
    self.sdf_iterations = 10


class MjStatistic:
  def __init__(self):
    self.meaninertia = 1.0
    self.meanmass = 1.0
    self.meansize = 1.0
    self.extent = 1.0
    self.center = np.zeros(3)


class MjModel:
  """numpy stand-in for mujoco.MjModel (only fields on the mj_step hot path)."""

  @staticmethod
  def from_xml_path(path):
    return load_xml(path)

  @staticmethod
  def from_xml_string(xml, assets_dir=None):
    return from_xml_string(xml, assets_dir)

  def body(self, name):
    return self.body_names.index(name)


class MjData:
  """numpy stand-in for mujoco.MjData: state + the derived arrays get_data_into fills."""

  def __init__(self, m):
    self.time = 0.0
    self.qpos = np.array(m.qpos0, dtype=np.float64).copy()
    self.qvel = np.zeros(m.nv)
    self.act = np.zeros(m.na)
    self.ctrl = np.zeros(m.nu)
    self.qacc_warmstart = np.zeros(m.nv)
    self.qfrc_applied = np.zeros(m.nv)
    self.xfrc_applied = np.zeros((m.nbody, 6))
    mb = [int(np.flatnonzero(m.body_mocapid == k)[0]) for k in range(m.nmocap)]
    self.mocap_pos = np.array(m.body_pos[mb], dtype=np.float64).reshape(m.nmocap, 3)
    self.mocap_quat = np.array(m.body_quat[mb], dtype=np.float64).reshape(m.nmocap, 4)
    self.qacc = np.zeros(m.nv)
    self.ncon = 0
    self.nefc = 0
    self.ne = 0
    self.nf = 0
    self.nl = 0
    self.solver_niter = np.zeros(1, dtype=np.int32)
    # sleep state (MjData.tree_asleep): < 0 awake (a countdown towards sleep), >= 0 asleep = next tree of the sleep cycle; the
    # reference's `--init_asleep` sets tree_asleep[:] = arange(ntree) before put_data (cli.py:167-168): every tree asleep on its own
    self.tree_asleep = np.full(int(getattr(m, "ntree", 0)), -(1 + 10), dtype=np.int32)


def mj_resetDataKeyframe(m, d, key):
  """Reset state to keyframe `key` (reference call site: cli.py:159-160)."""
  d.time = float(m.key_time[key]) if m.nkey > key else 0.0
  d.qpos[:] = m.key_qpos[key] if m.nkey > key else m.qpos0
  d.qvel[:] = m.key_qvel[key] if m.nkey > key else 0.0
  d.act[:] = m.key_act[key] if m.nkey > key else 0.0
  d.ctrl[:] = m.key_ctrl[key] if m.nkey > key else 0.0
  if m.nmocap and m.nkey > key:
    d.mocap_pos[:] = m.key_mpos[key].reshape(-1, 3)
    d.mocap_quat[:] = m.key_mquat[key].reshape(-1, 4)
  d.qacc_warmstart[:] = 0.0
  d.qfrc_applied[:] = 0.0
  d.xfrc_applied[:] = 0.0


# ---- parsing helpers ---------------------------------------------------------------------------
def _floats(s):
  return [float(x) for x in s.replace(",", " ").split()]


def _bool(s):
  s = s.strip().lower()
  if s in ("true", "1"):
    return True
  if s in ("false", "0"):
    return False
  if s == "auto":
    return None
  raise ValueError(f"bad boolean {s}")


def _vec(attrs, key, default):
  """Attribute vector; a shorter attribute overrides only the leading entries (MuJoCo semantics)."""
  out = np.array(default, dtype=np.float64).copy()
  if key in attrs:
    v = _floats(attrs[key])
    if len(v) > len(out):
      raise ValueError(f"attribute {key} has too many values")
    out[: len(v)] = v
  return out


_ACT_TAGS = ("general", "motor", "position", "velocity", "intvelocity", "damper", "cylinder", "muscle", "adhesion")
_DEFAULT_TAGS = ("geom", "joint", "site", "mesh", "material", "camera", "light", "pair", "equality", "tendon") + _ACT_TAGS


class _Defaults:
  def __init__(self, name, parent=None):
    self.name = name
    self.parent = parent
    self.attrs = {}  # tag -> dict
    if parent is not None:
      for k, v in parent.attrs.items():
        self.attrs[k] = dict(v)

  def update(self, tag, attrs):
    key = "actuator" if tag in _ACT_TAGS else tag
    d = self.attrs.setdefault(key, {})
    if key == "actuator":
      # shortcut tags inside <default> carry their implied settings with them
      d.update(_actuator_shortcut(tag, dict(attrs), in_default=True))
    else:
      d.update(attrs)


def _parse_defaults(elem, parent, table):
  name = elem.get("class", "main" if parent is None else None)
  if name is None:
    raise ValueError("nested <default> needs a class name")
  d = _Defaults(name, parent)
  if parent is None and "main" in table:
    d = table["main"]
  table[name] = d
  for child in elem:
    if child.tag == "default":
      continue
    if child.tag in _DEFAULT_TAGS:
      d.update(child.tag, child.attrib)
  for child in elem:
    if child.tag == "default":
      _parse_defaults(child, d, table)


def _resolve(tag, elem, table, childclass):
  cls = elem.get("class", childclass)
  key = "actuator" if tag in _ACT_TAGS else tag
  base = dict(table[cls if cls is not None else "main"].attrs.get(key, {})) if table else {}
  explicit = {k: v for k, v in elem.attrib.items() if k != "class"}
  return base, explicit


def _actuator_shortcut(tag, attrs, in_default=False, base=None):
  """Expand motor/position/velocity shortcuts to `general` attributes (MuJoCo XML reference)."""
  out = dict(attrs)
  base = base or {}
  if tag == "motor":
    out.setdefault("gaintype", "fixed")
    out.setdefault("biastype", "none")
    out.setdefault("dyntype", "none")
    if not in_default or "gainprm" not in out:
      out.setdefault("gainprm", "1")
  elif tag == "position":
    kp = float(out.pop("kp", 1.0)) if ("kp" in out or not (in_default or "_kp" in base)) else None
    kv = float(out.pop("kv", 0.0)) if ("kv" in out or not (in_default or "_kv" in base)) else None
    out.setdefault("gaintype", "fixed")
    out.setdefault("biastype", "affine")
    out.setdefault("dyntype", "none")
    if kp is not None:
      out["_kp"] = str(kp)
    if kv is not None:
      out["_kv"] = str(kv)
    if "dampratio" in out or "timeconst" in out or "inheritrange" in out:
      raise NotImplementedError("position actuator dampratio/timeconst/inheritrange")
  elif tag == "velocity":
    kv = float(out.pop("kv", 1.0)) if ("kv" in out or not (in_default or "_vkv" in base)) else None
    out.setdefault("gaintype", "fixed")
    out.setdefault("biastype", "affine")
    out.setdefault("dyntype", "none")
    if kv is not None:
      out["_vkv"] = str(kv)
  elif tag == "general":
    pass
  else:
    raise NotImplementedError(f"actuator shortcut <{tag}>")
  return out


def _orientation(attrs, compiler):
  """quat from quat/axisangle/euler/xyaxes/zaxis attributes."""
  deg = compiler["angle"] == "degree"
  if "quat" in attrs:
    return nm.quat_normalize(_floats(attrs["quat"]))
  if "axisangle" in attrs:
    v = _floats(attrs["axisangle"])
    ang = np.deg2rad(v[3]) if deg else v[3]
    ax = np.array(v[:3]) / max(np.linalg.norm(v[:3]), MJ_MINVAL)
    return nm.axis_angle_to_quat(ax, ang)
  if "euler" in attrs:
    v = np.array(_floats(attrs["euler"]))
    if deg:
      v = np.deg2rad(v)
    q = np.array([1.0, 0, 0, 0])
    for i, ch in enumerate(compiler["eulerseq"]):
      ax = np.zeros(3)
      ax["xyz".index(ch.lower())] = 1.0
      r = nm.axis_angle_to_quat(ax, v[i])
      q = nm.quat_mul(q, r) if ch.islower() else nm.quat_mul(r, q)
    return nm.quat_normalize(q)
  if "xyaxes" in attrs:
    v = np.array(_floats(attrs["xyaxes"]))
    x = v[:3] / np.linalg.norm(v[:3])
    y = v[3:] - x * np.dot(x, v[3:])
    y = y / np.linalg.norm(y)
    z = np.cross(x, y)
    return nm.mat_to_quat(np.stack([x, y, z], axis=1))
  if "zaxis" in attrs:
    return nm.quat_z2vec(_floats(attrs["zaxis"]))
  return np.array([1.0, 0, 0, 0])


def _expand_includes(elem, base_dir):
  i = 0
  children = list(elem)
  for child in children:
    if child.tag == "include":
      path = os.path.join(base_dir, child.get("file"))
      sub = ET.parse(path).getroot()
      _expand_includes(sub, os.path.dirname(path))
      idx = list(elem).index(child)
      elem.remove(child)
      for k, sc in enumerate(list(sub)):
        elem.insert(idx + k, sc)
    else:
      _expand_includes(child, base_dir)
    i += 1


# ---- structural elements: <frame>, <replicate>, <attach> are expanded into plain bodies before compilation ------------------
_ORIENT_KEYS = ("quat", "axisangle", "euler", "xyaxes", "zaxis")
_POSED_TAGS = ("body", "geom", "site", "camera", "light")


def _fmt(v):
  return " ".join(repr(float(x)) for x in v)


def _compiler_of(root):
  comp = {"angle": "degree", "eulerseq": "xyz"}
  for c in root.findall("compiler"):
    for k in comp:
      if k in c.attrib:
        comp[k] = c.get(k)
  return comp


def _place(elem, fpos, fquat, comp):
  """Re-express a posed child of a frame in the frame's parent: p' = fpos + R(fquat) p, q' = fquat * q."""
  if elem.tag not in ("body", "geom", "site"):
    return  # cameras and lights do not enter the dynamics
  a = elem.attrib
  if "fromto" in a:
    ft = _floats(a["fromto"])
    p0 = fpos + nm.rot_vec_quat(np.array(ft[:3]), fquat)
    p1 = fpos + nm.rot_vec_quat(np.array(ft[3:]), fquat)
    a["fromto"] = _fmt(np.concatenate([p0, p1]))
    return
  pos = _vec(a, "pos", [0, 0, 0])
  quat = _orientation(a, comp)
  for k in _ORIENT_KEYS:
    a.pop(k, None)
  a["pos"] = _fmt(fpos + nm.rot_vec_quat(pos, fquat))
  a["quat"] = _fmt(nm.quat_normalize(nm.quat_mul(fquat, quat)))


class _Structure:
  """State of one expansion pass: the document root (attach appends actuators / excludes / defaults to it), the sub-models
  named by <asset><model>, and the name suffix of the enclosing <replicate> copies."""

  def __init__(self, root, base_dir):
    self.root, self.base_dir, self.comp = root, base_dir, _compiler_of(root)
    self.models, self.defaults_done = {}, set()
    for asset in root.findall("asset"):
      for mdl in asset.findall("model"):
        path = os.path.join(base_dir, mdl.get("file"))
        sub = ET.parse(path).getroot()
        _expand_includes(sub, os.path.dirname(path))
        _expand_structure(sub, os.path.dirname(path))
        if _compiler_of(sub) != self.comp:
          raise NotImplementedError("<attach> of a model with different <compiler> angle/eulerseq settings")
        self.models[mdl.get("name", os.path.splitext(os.path.basename(path))[0])] = sub

  def section(self, tag):
    sec = self.root.find(tag)
    if sec is None:
      sec = ET.SubElement(self.root, tag)
    return sec

  def children(self, parent, suffix):
    """Expanded child list of `parent` (a body, worldbody, frame or replicate)."""
    out = []
    for child in list(parent):
      if child.tag == "frame":
        fpos, fquat = _vec(child.attrib, "pos", [0, 0, 0]), _orientation(child.attrib, self.comp)
        cc = child.get("childclass")
        for c in self.children(child, suffix):
          _place(c, fpos, fquat, self.comp)
          if cc is not None:
            key = "childclass" if c.tag == "body" else "class"
            if c.tag in ("body", "geom", "site", "joint") and key not in c.attrib:
              c.set(key, cc)
          out.append(c)
      elif child.tag == "replicate":
        count = int(child.get("count"))
        step_pos = _vec(child.attrib, "offset", [0, 0, 0])
        step_quat = _orientation({"euler": child.get("euler")} if "euler" in child.attrib else {}, self.comp)
        sep = child.get("sep", "")
        pos, quat = np.zeros(3), np.array([1.0, 0, 0, 0])
        for i in range(count):  # copy i carries the replicate transform applied i times
          sfx = f"{sep}{i:0{len(str(count - 1))}d}"
          holder = ET.Element("frame")
          for c in child:
            cp = copy.deepcopy(c)
            for e in cp.iter():
              if e.tag not in ("attach",) and e.get("name"):
                e.set("name", e.get("name") + sfx)
            holder.append(cp)
          for c in self.children(holder, suffix + sfx):
            _place(c, pos, quat, self.comp)
            out.append(c)
          pos = pos + nm.rot_vec_quat(step_pos, quat)
          quat = nm.quat_normalize(nm.quat_mul(quat, step_quat))
      elif child.tag == "attach":
        out.append(self.attach(child, suffix))
      else:
        if child.tag == "body":
          new = self.children(child, suffix)
          for c in list(child):
            child.remove(c)
          child.extend(new)
        out.append(child)
    return out

  def attach(self, elem, suffix):
    """<attach model= body= prefix=>: the named subtree of a sub-model with prefixed names, plus the sub-model's defaults
    (as classes of the parent document) and the actuators / contact excludes / joint equalities that refer to the subtree."""
    name, prefix = elem.get("model"), elem.get("prefix", "")
    if name not in self.models:
      raise ValueError(f"<attach>: unknown model '{name}'")
    sub = self.models[name]
    src = next((b for b in sub.iter("body") if b.get("name") == elem.get("body")), None)
    if src is None:
      raise ValueError(f"<attach>: model '{name}' has no body '{elem.get('body')}'")
    main = prefix + "main"
    if (name, prefix) not in self.defaults_done:  # the sub-model's default tree becomes the class `prefix + main`
      self.defaults_done.add((name, prefix))
      holder = ET.SubElement(self.section("default"), "default", {"class": main})
      for dsec in sub.findall("default"):
        for c in dsec:
          holder.append(self._prefixed(c, prefix, ""))
    body = self._prefixed(src, prefix, suffix)
    if "childclass" not in body.attrib:
      body.set("childclass", main)
    names = {tag: {e.get("name") for e in src.iter(tag) if e.get("name")} for tag in ("body", "joint")}
    names["body"].add(src.get("name"))
    for sec in sub.findall("actuator"):
      for a in sec:
        if a.get("joint") in names["joint"]:
          c = self._prefixed(a, prefix, suffix)
          c.set("joint", prefix + a.get("joint") + suffix)
          if "class" not in c.attrib:
            c.set("class", main)
          self.section("actuator").append(c)
        elif any(k in a.attrib for k in ("tendon", "site", "body", "slidersite")):
          raise NotImplementedError("<attach>: actuators with non-joint transmissions")
    for sec in sub.findall("contact"):
      for ex in sec.findall("exclude"):
        if ex.get("body1") in names["body"] and ex.get("body2") in names["body"]:
          c = self._prefixed(ex, prefix, suffix)
          c.set("body1", prefix + ex.get("body1") + suffix)
          c.set("body2", prefix + ex.get("body2") + suffix)
          self.section("contact").append(c)
    for sec in sub.findall("equality"):
      for eq in sec:
        if eq.tag == "joint" and eq.get("joint1") in names["joint"]:
          c = self._prefixed(eq, prefix, suffix)
          for k in ("joint1", "joint2"):
            if k in c.attrib:
              c.set(k, prefix + eq.get(k) + suffix)
          if "class" not in c.attrib:
            c.set("class", main)
          self.section("equality").append(c)
    return body

  @staticmethod
  def _prefixed(elem, prefix, suffix):
    cp = copy.deepcopy(elem)
    for e in cp.iter():
      if e.get("name"):
        e.set("name", prefix + e.get("name") + suffix)
      for k in ("class", "childclass"):
        if k in e.attrib:
          e.set(k, prefix + e.get(k))
    return cp


def _expand_structure(root, base_dir):
  if not any(True for tag in ("frame", "replicate", "attach") for _ in root.iter(tag)):
    return
  st = _Structure(root, base_dir)
  for wb in root.findall("worldbody"):
    new = st.children(wb, "")
    for c in list(wb):
      wb.remove(c)
    wb.extend(new)


def load_xml(path):
  root = ET.parse(path).getroot()
  base = os.path.dirname(os.path.abspath(path))
  _expand_includes(root, base)
  _expand_structure(root, base)
  return _compile(root, base)


def from_xml_string(xml, assets_dir=None):
  root = ET.fromstring(xml)
  base = assets_dir or os.getcwd()
  _expand_includes(root, base)
  _expand_structure(root, base)
  return _compile(root, base)


# ---- geometry helpers ---------------------------------------------------------------------------
def _geom_volume_inertia(gtype, size):
  """Volume and unit-density-normalised diagonal inertia (per unit mass) in the geom frame."""
  if gtype == GEOM_SPHERE:
    r = size[0]
    vol = 4.0 / 3.0 * np.pi * r**3
    unit = np.full(3, 0.4 * r * r)
  elif gtype == GEOM_CAPSULE:
    r, h = size[0], 2.0 * size[1]
    vol = np.pi * r * r * h + 4.0 / 3.0 * np.pi * r**3
    ms = 4.0 * r / (4.0 * r + 3.0 * h) if (4.0 * r + 3.0 * h) > 0 else 1.0  # sphere mass fraction
    mc = 1.0 - ms
    ixy = mc * (3 * r * r + h * h) / 12.0 + 0.4 * ms * r * r + ms * h * (3 * r + 2 * h) / 8.0
    iz = mc * r * r / 2.0 + 0.4 * ms * r * r
    unit = np.array([ixy, ixy, iz])
  elif gtype == GEOM_CYLINDER:
    r, h = size[0], 2.0 * size[1]
    vol = np.pi * r * r * h
    unit = np.array([(3 * r * r + h * h) / 12.0, (3 * r * r + h * h) / 12.0, r * r / 2.0])
  elif gtype == GEOM_BOX:
    a, b, c = size
    vol = 8.0 * a * b * c
    unit = np.array([b * b + c * c, a * a + c * c, a * a + b * b]) / 3.0
  elif gtype == GEOM_ELLIPSOID:
    a, b, c = size
    vol = 4.0 / 3.0 * np.pi * a * b * c
    unit = np.array([b * b + c * c, a * a + c * c, a * a + b * b]) / 5.0
  else:
    vol, unit = 0.0, np.zeros(3)
  return vol, unit


def _geom_rbound_aabb(gtype, size):
  if gtype == GEOM_SPHERE:
    return size[0], np.array([size[0]] * 3)
  if gtype == GEOM_CAPSULE:
    return size[0] + size[1], np.array([size[0], size[0], size[0] + size[1]])
  if gtype == GEOM_CYLINDER:
    return float(np.hypot(size[0], size[1])), np.array([size[0], size[0], size[1]])
  if gtype == GEOM_BOX:
    return float(np.linalg.norm(size)), np.array(size)
  if gtype == GEOM_ELLIPSOID:
    return float(np.max(size)), np.array(size)
  if gtype == GEOM_PLANE:
    return 0.0, np.array([1e10, 1e10, 1e10])
  return 0.0, np.zeros(3)


def read_stl(path):
  """Vertices [n, 3] and triangles [m, 3] of an STL file, binary or ASCII; repeated vertices are merged (as MuJoCo's compiler does)."""
  raw = open(path, "rb").read()
  tri = None
  if len(raw) >= 84:
    n = int(np.frombuffer(raw, dtype="<u4", count=1, offset=80)[0])
    if len(raw) == 84 + 50 * n:  # binary: 80-byte header, count, then 50-byte records (normal, 3 vertices, attribute word)
      rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
      tri = rec["v"].astype(np.float64).reshape(-1, 3)
  if tri is None:
    txt = raw.decode("ascii", errors="replace")
    if not txt.lstrip().lower().startswith("solid"):
      raise ValueError(f"{path}: neither a binary STL (size does not match its triangle count) nor an ASCII one")
    pts = [[float(x) for x in line.split()[1:4]] for line in txt.splitlines() if line.strip().lower().startswith("vertex")]
    if not pts or len(pts) % 3:
      raise ValueError(f"{path}: ASCII STL with {len(pts)} vertex lines")
    tri = np.array(pts, dtype=np.float64)
  verts, inv = np.unique(tri, axis=0, return_inverse=True)
  return verts, np.asarray(inv).reshape(-1, 3).astype(np.int32)


def read_obj(path):
  """Vertices and (fan-triangulated) faces of a Wavefront OBJ file: `v x y z` and `f a b c ...` lines (texture / normal indices after
  a slash are ignored, negative indices count from the end)."""
  verts, faces = [], []
  for line in open(path, "r", errors="replace"):
    t = line.split()
    if not t:
      continue
    if t[0] == "v":
      verts.append([float(x) for x in t[1:4]])
    elif t[0] == "f":
      ids = [int(x.split("/")[0]) for x in t[1:]]
      ids = [i - 1 if i > 0 else len(verts) + i for i in ids]
      faces.extend([ids[0], ids[k], ids[k + 1]] for k in range(1, len(ids) - 1))
  if not verts:
    raise ValueError(f"{path}: no vertices")
  return np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int32).reshape(-1, 3)


def read_mesh_file(path):
  ext = os.path.splitext(path)[1].lower()
  if ext == ".stl":
    return read_stl(path)
  if ext == ".obj":
    return read_obj(path)
  raise NotImplementedError(f"mesh file format {ext!r} (STL and OBJ are supported)")


def read_hfield_file(path):
  """(nrow, ncol, elevation [nrow, ncol] with row 0 at -y) of a height-field file: a PNG image (grey levels, the image's top row is the
  +y edge) or MuJoCo's binary format (int32 nrow, int32 ncol, float32 data).  UNPINNED like the rest of this loader."""
  if os.path.splitext(path)[1].lower() == ".png":
    try:
      from PIL import Image
    except ImportError as e:  # pragma: no cover
      raise NotImplementedError("PNG height fields need the PIL package") from e
    img = np.asarray(Image.open(path).convert("F"), dtype=np.float64)
    return img.shape[0], img.shape[1], img[::-1].copy()
  raw = open(path, "rb").read()
  nrow, ncol = (int(x) for x in np.frombuffer(raw, dtype="<i4", count=2))
  if nrow <= 0 or ncol <= 0 or len(raw) != 8 + 4 * nrow * ncol:
    raise ValueError(f"{path}: not a height-field file (int32 nrow, int32 ncol, float32 data[nrow * ncol])")
  return nrow, ncol, np.frombuffer(raw, dtype="<f4", count=nrow * ncol, offset=8).astype(np.float64).reshape(nrow, ncol)


def _compile_mesh(verts, maxhullvert=-1, faces=None, inertia="legacy"):
  """Convex collision asset from inline vertices (MJCF <mesh vertex="...">), following what MuJoCo's compiler does with a mesh:
  convex hull, volume / centre of mass / inertia of the hull (uniform density), vertices re-expressed in the frame centred at the
  centre of mass and aligned with the principal axes (the geom frame is composed with that offset).  UNPINNED like the rest of this
  loader (MuJoCo's compiler is not in the reference tree); the axis order / signs of the principal frame are this module's own
  (descending moments, right handed): the physics does not depend on them.  The hull's vertex adjacency is stored as a hill-climbing
  graph in MuJoCo's layout (meshes of 10 or more vertices are then searched by hill climbing, smaller ones exhaustively:
  collision_gjk.py:156).

  Mass properties (<mesh inertia=...>): "convex" -- and meshes without faces -- take them from the hull; "exact" from the signed
  tetrahedra of the file's own triangles, "legacy" (MuJoCo's default) from the same tetrahedra with absolute volumes about the
  area-weighted centroid of the faces (over-counts where a non-convex surface folds back).  A mesh whose faces enclose its hull's
  volume IS convex: it keeps the hull path, so files and inline vertices of the same convex shape compile to identical models."""
  from scipy.spatial import ConvexHull

  verts = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
  if len(verts) < 4:
    raise ValueError("a mesh needs at least 4 vertices")
  # maxhullvert: MuJoCo stops Qhull after maxhullvert - 4 added vertices (`TA` option): an inner approximation of the hull
  if maxhullvert != -1 and maxhullvert < 4:
    raise ValueError("maxhullvert must be -1 or at least 4")
  hull = ConvexHull(verts, qhull_options=f"Qt TA{maxhullvert - 4}" if maxhullvert >= 4 else None)
  vol, com, second = 0.0, np.zeros(3), np.zeros((3, 3))
  # mass properties come from the whole mesh (its exact hull), the truncated hull only feeds the collision tables
  hull_in = ConvexHull(verts) if maxhullvert >= 4 else hull
  centre = verts[hull_in.vertices].mean(axis=0)
  canon = np.array([[2, 1, 1], [1, 2, 1], [1, 1, 2]]) / 120.0  # integral of x x^T over the unit tetrahedron
  # signed tetrahedra (centre, a, b, c) of all hull triangles at once, triangles oriented outwards
  T = verts[hull_in.simplices] - centre  # [nf, 3 (vertex), 3 (xyz)]
  flip = np.einsum("fk,fk->f", np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]), hull_in.equations[:, :3]) < 0
  T[flip] = T[flip][:, [0, 2, 1]]
  A = np.transpose(T, (0, 2, 1))  # columns a, b, c
  det = np.linalg.det(A)  # 6 x signed volume
  vol = float(det.sum() / 6.0)
  com = (det[:, None] / 24.0 * T.sum(axis=1)).sum(axis=0)
  second = np.einsum("f,fij,jk,flk->il", det, A, canon, A)
  com /= vol
  second -= vol * np.outer(com, com)  # second moment about the centre of mass
  if inertia not in ("convex", "exact", "legacy"):
    raise NotImplementedError(f"<mesh inertia={inertia!r}> (convex, exact and legacy are implemented)")
  if faces is not None and len(faces) and inertia != "convex":
    tri = verts[np.asarray(faces).reshape(-1, 3)]
    areas = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    ref = (areas[:, None] * tri.mean(axis=1)).sum(axis=0) / max(areas.sum(), MJ_MINVAL)
    Tf = tri - ref
    Af = np.transpose(Tf, (0, 2, 1))
    detf = np.linalg.det(Af)
    if inertia == "legacy":  # absolute pyramid volumes: orientation-independent, over-counts folds
      Af = np.where((detf < 0)[:, None, None], Af[:, :, [0, 2, 1]], Af)
      Tf = np.transpose(Af, (0, 2, 1))
      detf = np.abs(detf)
    volf = float(detf.sum() / 6.0)
    if volf <= 0:
      raise ValueError("mesh faces enclose no volume (inconsistent orientation?): use <mesh inertia=\"convex\"/> or \"legacy\"")
    if abs(volf - vol) > 1e-6 * vol:  # a genuinely non-convex surface (float32 STL vertices leave 1e-8 between a convex file and its hull): mass properties from the faces, hull only for collision
      comf = (detf[:, None] / 24.0 * Tf.sum(axis=1)).sum(axis=0) / volf
      secondf = np.einsum("f,fij,jk,flk->il", detf, Af, canon, Af) - volf * np.outer(comf, comf)
      vol, com, second = volf, comf + (ref - centre), secondf
  inertia = np.trace(second) * np.eye(3) - second
  w, v = np.linalg.eigh(inertia)
  order = np.argsort(-w)
  w, v = w[order], v[:, order]
  if np.linalg.det(v) < 0:
    v[:, 2] = -v[:, 2]
  vlocal = (verts - centre - com) @ v
  lo, hi = vlocal.min(axis=0), vlocal.max(axis=0)
  # polygon tables for multi-contact recovery (Model.mesh_poly*, types.py:1707-1733): coplanar hull triangles merged into convex
  # polygons, vertices counter-clockwise seen from outside; per vertex the polygons it belongs to.  A triangle joins the first group
  # whose plane (that of the group's first triangle) it shares, else it opens a new group
  eqs = hull.equations
  gn, gd = np.zeros((len(eqs), 3)), np.zeros(len(eqs))
  groups = []
  for tri, eq in zip(hull.simplices, eqs):
    ng = len(groups)
    hit = np.flatnonzero((gn[:ng] @ eq[:3] > 1.0 - 1e-9) & (np.abs(gd[:ng] - eq[3]) < 1e-9 * max(1.0, abs(eq[3]))))
    if len(hit):
      groups[int(hit[0])]["verts"].update(int(i) for i in tri)
    else:
      gn[ng], gd[ng] = eq[:3], eq[3]
      groups.append(dict(n=eq[:3].copy(), d=float(eq[3]), verts=set(int(i) for i in tri)))
  polys, normals = [], []
  for gr in groups:
    ids = sorted(gr["verts"])
    pts = verts[ids]
    c = pts.mean(axis=0)
    e1 = pts[0] - c
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(gr["n"], e1)
    ang = np.arctan2((pts - c) @ e2, (pts - c) @ e1)
    polys.append([ids[i] for i in np.argsort(ang)])
    normals.append(gr["n"] @ v)  # in the mesh frame
  polymap = [[] for _ in range(len(verts))]
  for p, poly in enumerate(polys):
    for i in poly:
      polymap[i].append(p)
  # hill-climbing graph in MuJoCo's mesh_graph layout (numvert, numface, vert_edgeadr[numvert], vert_globalid[numvert],
  # edge_localid[numvert + 3 numface] = neighbours of each hull vertex as local ids, -1 terminated, face_globalid[3 numface]); consumed by
  # reference collision_gjk.py:170-196 and collision_primitive.py:131-243 for meshes of 10 or more vertices
  hv = [int(i) for i in hull.vertices]
  local = {g: l for l, g in enumerate(hv)}
  nbr = [set() for _ in hv]
  for tri in hull.simplices:
    for a_, b_ in ((0, 1), (1, 2), (2, 0)):
      nbr[local[int(tri[a_])]].add(local[int(tri[b_])])
      nbr[local[int(tri[b_])]].add(local[int(tri[a_])])
  edgeadr, edges = [], []
  for l in range(len(hv)):
    edgeadr.append(len(edges))
    edges.extend(sorted(nbr[l]))
    edges.append(-1)
  graph = [len(hv), len(hull.simplices)] + edgeadr + hv + edges + [int(i) for tri in hull.simplices for i in tri]
  return dict(vert=vlocal, pos=centre + com, quat=nm.mat_to_quat(v), vol=vol, unit=w / vol, aabb=np.concatenate([(lo + hi) / 2, (hi - lo) / 2]),
              rbound=float(np.max(np.linalg.norm(vlocal, axis=1))), polys=polys, polynormal=np.array(normals), polymap=polymap, graph=graph)


def _site_size(a):
  """Site size with MuJoCo's fill rule for short size attributes (missing entries repeat the first; default 0.005)."""
  v = _floats(a["size"]) if "size" in a else [0.005]
  return np.array((list(v) + [v[0]] * 3)[:3], dtype=np.float64)


class _Body:
  pass


def _compile(root, base_dir):
  if root.tag != "mujoco":
    raise ValueError("root element must be <mujoco>")
  compiler = {"angle": "degree", "eulerseq": "xyz", "autolimits": True, "inertiafromgeom": "auto",
              "boundmass": 0.0, "boundinertia": 0.0, "balanceinertia": False, "settotalmass": -1.0, "meshdir": None, "assetdir": None}
  opt = MjOption()
  stat = MjStatistic()
  table = {}
  for elem in root:
    if elem.tag == "compiler":
      for k, v in elem.attrib.items():
        if k in ("angle", "eulerseq", "inertiafromgeom", "meshdir", "assetdir"):
          compiler[k] = v
        elif k in ("autolimits", "balanceinertia"):
          compiler[k] = _bool(v)
        elif k in ("boundmass", "boundinertia", "settotalmass"):
          compiler[k] = float(v)
        elif k == "coordinate" and v != "local":
          raise NotImplementedError("global coordinates")
    elif elem.tag == "option":
      _parse_option(elem, opt)
    elif elem.tag == "default":
      _parse_defaults(elem, None, table)
    elif elem.tag in ("deformable", "extension"):
      if len(list(elem)):
        raise NotImplementedError(f"<{elem.tag}> is outside the hot-path scope")
    elif elem.tag not in ("worldbody", "actuator", "contact", "keyframe", "equality", "tendon", "asset", "visual", "statistic",
                          "size", "custom", "sensor"):
      # (<sensor> and <custom> do not enter the dynamics; <include> etc. would change the model: never skipped silently)
      raise NotImplementedError(f"<{elem.tag}>")
  if "main" not in table:
    table["main"] = _Defaults("main")

  deg = compiler["angle"] == "degree"
  mesh_assets = {}  # name -> <mesh> element (compiled on first use by a colliding geom)
  for asset in root.findall("asset"):
    for me in asset.findall("mesh"):
      base, explicit = _resolve("mesh", me, table, None)
      a = dict(base)
      a.update(explicit)
      name = a.get("name") or os.path.splitext(os.path.basename(a.get("file", "")))[0]
      mesh_assets[name] = a
  mesh_compiled = {}
  hfield_assets, hfield_names = {}, []  # inline height fields (<hfield nrow= ncol= size= elevation=>); files are not in this tree
  for asset in root.findall("asset"):
    for he in asset.findall("hfield"):
      hfield_assets[he.get("name")] = he.attrib

  mat_names, mat_rgba = [], []  # materials: only their colour is kept (rays skip geoms whose material is fully transparent)
  for asset in root.findall("asset"):
    for ma in asset.findall("material"):
      base, explicit = _resolve("material", ma, table, None)
      a = dict(base)
      a.update(explicit)
      mat_names.append(a.get("name", ""))
      mat_rgba.append(_vec(a, "rgba", [1.0, 1.0, 1.0, 1.0]))

  bodies, joints, geoms, sites = [], [], [], []
  world = _Body()
  world.name, world.parent, world.pos, world.quat = "world", 0, np.zeros(3), np.array([1.0, 0, 0, 0])
  world.inertial, world.joints, world.geoms, world.gravcomp, world.mocap = None, [], [], 0.0, False
  bodies.append(world)

  def attach_mesh(g):
    """Compiles the geom's mesh asset (once per asset) and moves the geom frame onto the mesh's inertial frame (MuJoCo's convention).
    Called for colliding mesh geoms, and for non-colliding ones whose body takes its mass from its geoms."""
    pos, quat = g["pos"], g["quat"]
    asset = mesh_assets.get(g["mesh"])
    if asset is None:
      raise ValueError(f"geom refers to unknown mesh {g['mesh']!r}")
    if g["mesh"] not in mesh_compiled:
      faces = None
      if "vertex" in asset:
        v = np.array(_floats(asset["vertex"])).reshape(-1, 3)
        if "face" in asset:
          faces = np.array(_floats(asset["face"]), dtype=np.int64).reshape(-1, 3)
      elif "file" in asset:  # STL / OBJ, relative to <compiler meshdir> (or assetdir), itself relative to the model file
        for k in ("refpos", "refquat"):
          if k in asset:
            raise NotImplementedError(f"<mesh {k}=...>")
        mdir = compiler["meshdir"] if compiler["meshdir"] is not None else (compiler["assetdir"] or "")
        path = asset["file"] if os.path.isabs(asset["file"]) else os.path.join(base_dir or "", mdir, asset["file"])
        if not os.path.exists(path):
          raise FileNotFoundError(f"mesh {g['mesh']!r}: file {path!r} not found (colliding mesh geoms need their asset; non-colliding ones do not)")
        v, faces = read_mesh_file(path)
      else:
        raise ValueError(f"mesh {g['mesh']!r} has neither vertex data nor a file")
      sc = _vec(asset, "scale", [1, 1, 1])
      v = v * sc
      if faces is not None and np.prod(sc) < 0:  # a mirroring scale flips the triangles' orientation
        faces = np.asarray(faces)[:, [0, 2, 1]]
      mesh_compiled[g["mesh"]] = _compile_mesh(v, int(asset.get("maxhullvert", -1)), faces=faces, inertia=asset.get("inertia", "legacy"))
    md = mesh_compiled[g["mesh"]]
    g["meshdata"] = md
    g["pos"] = pos + nm.rot_vec_quat(md["pos"], quat)  # the geom frame is the mesh's inertial frame (MuJoCo's convention)
    g["quat"] = nm.quat_mul(quat, md["quat"])
    g["size"] = md["aabb"][3:].copy()


  def parse_geom(elem, childclass, bodyid):
    base, explicit = _resolve("geom", elem, table, childclass)
    a = dict(base)
    a.update(explicit)
    g = {}
    g["name"] = a.get("name", "")
    g["type"] = _GEOM_NAMES[a.get("type", "sphere")]
    size = _vec(a, "size", [0, 0, 0])
    g["contype"] = int(a.get("contype", 1))
    g["conaffinity"] = int(a.get("conaffinity", 1))
    g["condim"] = int(a.get("condim", 3))
    g["group"] = int(a.get("group", 0))
    g["rgba"] = _vec(a, "rgba", [0.5, 0.5, 0.5, 1.0])
    g["matid"] = mat_names.index(a["material"]) if a.get("material") in mat_names else -1
    g["priority"] = int(a.get("priority", 0))
    g["friction"] = _vec(a, "friction", [1.0, 0.005, 0.0001])
    g["solmix"] = float(a.get("solmix", 1.0))
    g["solref"] = _vec(a, "solref", [0.02, 1.0])
    g["solimp"] = _vec(a, "solimp", [0.9, 0.95, 0.001, 0.5, 2.0])
    g["margin"] = float(a.get("margin", 0.0))
    g["gap"] = float(a.get("gap", 0.0))
    g["density"] = float(a.get("density", 1000.0))
    g["mass"] = float(a["mass"]) if "mass" in a else None
    g["mesh"] = a.get("mesh")
    pos = _vec(a, "pos", [0, 0, 0])
    quat = _orientation(a, compiler)
    if "fromto" in a:
      ft = np.array(_floats(a["fromto"]))
      vec = ft[0:3] - ft[3:6]
      length = np.linalg.norm(vec)
      pos = 0.5 * (ft[0:3] + ft[3:6])
      quat = nm.quat_z2vec(vec)
      if g["type"] in (GEOM_CAPSULE, GEOM_CYLINDER):
        size = np.array([size[0], length / 2.0, 0.0])
      elif g["type"] in (GEOM_BOX, GEOM_ELLIPSOID):
        size = np.array([size[0], size[1] if size[1] > 0 else size[0], length / 2.0])
      else:
        raise ValueError("fromto only for capsule/cylinder/box/ellipsoid")
    if g["type"] == GEOM_SPHERE:
      size = np.array([size[0], 0.0, 0.0])
    elif g["type"] in (GEOM_CAPSULE, GEOM_CYLINDER):
      size = np.array([size[0], size[1], 0.0])
    g["size"], g["pos"], g["quat"], g["body"] = size, pos, quat, bodyid
    if g["type"] == GEOM_SDF:
      raise NotImplementedError("sdf geoms")
    g["hfield"] = None
    if g["type"] == GEOM_HFIELD:
      name = a.get("hfield")
      if name not in hfield_assets:
        raise ValueError(f"geom refers to unknown hfield {name!r}")
      if "elevation" not in hfield_assets[name] and "file" not in hfield_assets[name]:
        raise ValueError(f"hfield {name!r} has neither elevation data nor a file")
      if name not in hfield_names:
        hfield_names.append(name)
      g["hfield"] = name
      hs = np.array(_floats(hfield_assets[name]["size"]))
      g["size"] = np.array([hs[0], hs[1], 0.25 * hs[2] + 0.5 * hs[3]])  # (unpinned: only rbound / aabb below enter the engine)
    g["meshdata"] = None
    if g["type"] == GEOM_MESH and (g["contype"] or g["conaffinity"]):
      attach_mesh(g)
    return g

  def parse_joint(elem, childclass, bodyid, free=False):
    if free:
      a = {"type": "free", "name": elem.get("name", "")}
      for k in ("group", "align"):
        pass
    else:
      base, explicit = _resolve("joint", elem, table, childclass)
      a = dict(base)
      a.update(explicit)
    j = {}
    j["name"] = a.get("name", "")
    j["type"] = _JNT_NAMES[a.get("type", "hinge")]
    j["pos"] = _vec(a, "pos", [0, 0, 0])
    ax = _vec(a, "axis", [0, 0, 1])
    j["axis"] = ax / max(np.linalg.norm(ax), MJ_MINVAL)
    rng = _vec(a, "range", [0, 0])
    has_range = "range" in a
    lim = _bool(a["limited"]) if "limited" in a else None
    if lim is None:
      lim = has_range and compiler["autolimits"]
    j["limited"] = bool(lim)
    if deg and j["type"] in (JNT_HINGE, JNT_BALL):
      rng = np.deg2rad(rng)
    j["range"] = rng
    j["stiffness"] = float(a.get("stiffness", 0.0))
    j["damping"] = float(a.get("damping", 0.0))
    j["armature"] = float(a.get("armature", 0.0))
    j["frictionloss"] = float(a.get("frictionloss", 0.0))
    j["margin"] = float(a.get("margin", 0.0))
    ref = float(a.get("ref", 0.0))
    sref = float(a.get("springref", 0.0))
    if deg and j["type"] == JNT_HINGE:
      ref, sref = np.deg2rad(ref), np.deg2rad(sref)
    j["ref"], j["springref"] = ref, sref
    j["solreflimit"] = _vec(a, "solreflimit", [0.02, 1.0])
    j["solimplimit"] = _vec(a, "solimplimit", [0.9, 0.95, 0.001, 0.5, 2.0])
    j["solreffriction"] = _vec(a, "solreffriction", [0.02, 1.0])
    j["solimpfriction"] = _vec(a, "solimpfriction", [0.9, 0.95, 0.001, 0.5, 2.0])
    afr = _vec(a, "actuatorfrcrange", [0, 0])
    afl = _bool(a["actuatorfrclimited"]) if "actuatorfrclimited" in a else None
    if afl is None:
      afl = ("actuatorfrcrange" in a) and compiler["autolimits"]
    j["actfrclimited"], j["actfrcrange"] = bool(afl), afr
    j["actgravcomp"] = int(_bool(a.get("actuatorgravcomp", "false")) or 0)
    j["body"] = bodyid
    if j["type"] == JNT_FREE:
      j["pos"], j["axis"] = np.zeros(3), np.array([0.0, 0, 1])
      j["limited"] = False
    return j

  def parse_body(elem, parentid, childclass):
    b = _Body()
    cc = elem.get("childclass", childclass)
    b.name = elem.get("name", "")
    b.parent = parentid
    b.pos = _vec(elem.attrib, "pos", [0, 0, 0])
    b.quat = _orientation(elem.attrib, compiler)
    b.gravcomp = float(elem.get("gravcomp", 0.0))
    if elem.get("sleep", "auto") != "auto":
      raise NotImplementedError("body sleep policies other than auto (reference types.py:308: NEVER, ALLOWED, INIT unsupported)")
    b.mocap = _bool(elem.get("mocap", "false")) or False
    if b.mocap and parentid != 0:
      raise ValueError("mocap bodies must be children of the world")
    b.inertial, b.joints, b.geoms = None, [], []
    bodies.append(b)
    bid = len(bodies) - 1
    for child in elem:
      if child.tag == "inertial":
        ia = child.attrib
        ine = {"pos": _vec(ia, "pos", [0, 0, 0]), "quat": _orientation(ia, compiler), "mass": float(ia["mass"])}
        if "diaginertia" in ia:
          ine["diag"] = np.array(_floats(ia["diaginertia"]))
        elif "fullinertia" in ia:
          f = _floats(ia["fullinertia"])
          full = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
          w, v = np.linalg.eigh(full)
          order = np.argsort(-w)
          w, v = w[order], v[:, order]
          if np.linalg.det(v) < 0:
            v[:, 2] = -v[:, 2]
          ine["diag"] = w
          ine["quat"] = nm.quat_mul(ine["quat"], nm.mat_to_quat(v))
        else:
          raise ValueError("inertial needs diaginertia or fullinertia")
        b.inertial = ine
      elif child.tag == "freejoint":
        b.joints.append(parse_joint(child, cc, bid, free=True))
      elif child.tag == "joint":
        b.joints.append(parse_joint(child, cc, bid))
      elif child.tag == "geom":
        b.geoms.append(parse_geom(child, cc, bid))
      elif child.tag == "site":
        base, explicit = _resolve("site", child, table, cc)
        a = dict(base)
        a.update(explicit)
        sites.append({"name": a.get("name", ""), "body": bid, "pos": _vec(a, "pos", [0, 0, 0]),
                      "quat": _orientation(a, compiler), "size": _site_size(a), "type": _GEOM_NAMES[a.get("type", "sphere")]})
      elif child.tag not in ("camera", "light", "body"):
        # <frame>, <replicate>, <attach>, <composite>, <flexcomp>, <plugin>, ...: never skipped silently
        raise NotImplementedError(f"<{child.tag}> inside <body>")
    for child in elem:
      if child.tag == "body":
        parse_body(child, bid, cc)

  # a model may hold several <worldbody> (and <actuator>, <contact>, <keyframe>) sections, typically one per included
  # file; MuJoCo merges them in document order: all world geoms/sites first, then the bodies
  wbs = root.findall("worldbody")
  if not wbs:
    raise ValueError("no <worldbody>")
  wb = [child for sec in wbs for child in sec]
  for child in wb:
    if child.tag == "geom":
      world.geoms.append(parse_geom(child, None, 0))
    elif child.tag == "site":
      base, explicit = _resolve("site", child, table, None)
      a = dict(base)
      a.update(explicit)
      sites.append({"name": a.get("name", ""), "body": 0, "pos": _vec(a, "pos", [0, 0, 0]),
                    "quat": _orientation(a, compiler), "size": _site_size(a), "type": _GEOM_NAMES[a.get("type", "sphere")]})
  for child in wb:
    if child.tag == "body":
      parse_body(child, 0, None)
    elif child.tag not in ("geom", "site", "camera", "light"):
      raise NotImplementedError(f"<{child.tag}> inside <worldbody>")  # <frame>, <replicate>, <attach>, <composite>, ...

  if any(len(list(e)) > 0 for e in root.findall("tendon")):
    raise NotImplementedError("<tendon> is outside the hot-path scope (SURVEY §2 OUT rows)")

  m = MjModel()
  m.opt, m.stat = opt, stat
  nbody = len(bodies)
  m.nbody = nbody
  m.body_names = [b.name for b in bodies]
  m.body_parentid = np.array([b.parent for b in bodies], dtype=np.int32)
  m.body_pos = np.array([b.pos for b in bodies])
  m.body_quat = np.array([b.quat for b in bodies])
  m.body_gravcomp = np.array([b.gravcomp for b in bodies])
  m.body_mocapid = np.full(nbody, -1, dtype=np.int32)
  m.nmocap = 0
  for i, b in enumerate(bodies):
    if b.mocap:
      if b.joints:
        raise ValueError("mocap bodies cannot have joints")
      m.body_mocapid[i] = m.nmocap
      m.nmocap += 1

  # joints / dofs
  jl = []
  m.body_jntnum = np.zeros(nbody, dtype=np.int32)
  m.body_jntadr = np.full(nbody, -1, dtype=np.int32)
  m.body_dofnum = np.zeros(nbody, dtype=np.int32)
  m.body_dofadr = np.full(nbody, -1, dtype=np.int32)
  nq = nv = 0
  qpos0, qpos_spring = [], []
  jq, jd = [], []
  dof_body, dof_jnt, dof_parent = [], [], []
  last_dof_of_body = np.full(nbody, -1, dtype=np.int64)
  for bid, b in enumerate(bodies):
    if b.joints:
      m.body_jntadr[bid] = len(jl)
      m.body_jntnum[bid] = len(b.joints)
      m.body_dofadr[bid] = nv
    # last dof among ancestors
    p = b.parent if bid > 0 else -1
    prev = last_dof_of_body[p] if p >= 0 else -1
    for j in b.joints:
      if j["type"] == JNT_FREE and (b.parent != 0 or len(b.joints) != 1):
        raise ValueError("free joint must be alone in a top-level body")
      jid = len(jl)
      jl.append(j)
      jq.append(nq)
      jd.append(nv)
      t = j["type"]
      if t == JNT_FREE:
        q0 = np.concatenate([b.pos, b.quat])
        qs = q0.copy()
        ndof, nqj = 6, 7
      elif t == JNT_BALL:
        q0 = np.array([1.0, 0, 0, 0])
        qs = q0.copy()
        ndof, nqj = 3, 4
      else:
        q0 = np.array([j["ref"]])
        qs = np.array([j["springref"]])
        ndof, nqj = 1, 1
      qpos0.append(q0)
      qpos_spring.append(qs)
      for k in range(ndof):
        dof_body.append(bid)
        dof_jnt.append(jid)
        dof_parent.append(prev)
        prev = nv
        nv += 1
      nq += nqj
    m.body_dofnum[bid] = nv - (m.body_dofadr[bid] if b.joints else nv)
    last_dof_of_body[bid] = prev
  m.nq, m.nv, m.njnt = nq, nv, len(jl)
  m.qpos0 = np.concatenate(qpos0) if qpos0 else np.zeros(0)
  m.qpos_spring = np.concatenate(qpos_spring) if qpos_spring else np.zeros(0)
  m.jnt_names = [j["name"] for j in jl]
  m.jnt_type = np.array([j["type"] for j in jl], dtype=np.int32)
  m.jnt_qposadr = np.array(jq, dtype=np.int32)
  m.jnt_dofadr = np.array(jd, dtype=np.int32)
  m.jnt_bodyid = np.array([j["body"] for j in jl], dtype=np.int32)
  m.jnt_limited = np.array([j["limited"] for j in jl], dtype=np.int32)
  m.jnt_actfrclimited = np.array([j["actfrclimited"] for j in jl], dtype=bool)
  m.jnt_actgravcomp = np.array([j["actgravcomp"] for j in jl], dtype=np.int32)
  m.jnt_solref = np.array([j["solreflimit"] for j in jl]).reshape(-1, 2)
  m.jnt_solimp = np.array([j["solimplimit"] for j in jl]).reshape(-1, 5)
  m.jnt_pos = np.array([j["pos"] for j in jl]).reshape(-1, 3)
  m.jnt_axis = np.array([j["axis"] for j in jl]).reshape(-1, 3)
  m.jnt_stiffness = np.array([j["stiffness"] for j in jl], dtype=np.float64)
  m.jnt_stiffnesspoly = np.zeros((m.njnt, 2))
  m.jnt_range = np.array([j["range"] for j in jl]).reshape(-1, 2)
  m.jnt_actfrcrange = np.array([j["actfrcrange"] for j in jl]).reshape(-1, 2)
  m.jnt_margin = np.array([j["margin"] for j in jl], dtype=np.float64)
  m.dof_bodyid = np.array(dof_body, dtype=np.int32)
  m.dof_jntid = np.array(dof_jnt, dtype=np.int32)
  m.dof_parentid = np.array(dof_parent, dtype=np.int32)
  m.dof_solref = np.array([jl[j]["solreffriction"] for j in dof_jnt]).reshape(-1, 2)
  m.dof_solimp = np.array([jl[j]["solimpfriction"] for j in dof_jnt]).reshape(-1, 5)
  m.dof_frictionloss = np.array([jl[j]["frictionloss"] for j in dof_jnt], dtype=np.float64)
  m.dof_armature = np.array([jl[j]["armature"] for j in dof_jnt], dtype=np.float64)
  m.dof_damping = np.array([jl[j]["damping"] for j in dof_jnt], dtype=np.float64)
  m.dof_dampingpoly = np.zeros((nv, 2))

  # weld / root / tree ids
  m.body_weldid = np.zeros(nbody, dtype=np.int32)
  m.body_rootid = np.zeros(nbody, dtype=np.int32)
  m.body_treeid = np.full(nbody, -1, dtype=np.int32)
  ntree = 0
  for bid in range(1, nbody):
    p = m.body_parentid[bid]
    m.body_weldid[bid] = bid if m.body_jntnum[bid] > 0 else m.body_weldid[p]
    m.body_rootid[bid] = bid if p == 0 else m.body_rootid[p]
    if m.body_jntnum[bid] > 0 and m.body_treeid[p] < 0:
      m.body_treeid[bid] = ntree
      ntree += 1
    else:
      m.body_treeid[bid] = m.body_treeid[p]
  m.ntree = ntree
  m.dof_treeid = m.body_treeid[m.dof_bodyid] if nv else np.zeros(0, dtype=np.int32)
  m.tree_dofadr = np.array([int(np.argmax(m.dof_treeid == t)) for t in range(ntree)], dtype=np.int32)
  m.tree_dofnum = np.array([int(np.sum(m.dof_treeid == t)) for t in range(ntree)], dtype=np.int32)
  m.tree_bodynum = np.array([int(np.sum(m.body_treeid == t)) for t in range(ntree)], dtype=np.int32)

  # geoms (ordered by body)
  gl = []
  m.body_geomnum = np.zeros(nbody, dtype=np.int32)
  m.body_geomadr = np.full(nbody, -1, dtype=np.int32)
  for bid, b in enumerate(bodies):
    if b.geoms:
      m.body_geomadr[bid] = len(gl)
      m.body_geomnum[bid] = len(b.geoms)
    gl.extend(b.geoms)
  ng = len(gl)
  m.ngeom = ng
  m.geom_names = [g["name"] for g in gl]
  m.geom_type = np.array([g["type"] for g in gl], dtype=np.int32)
  m.geom_contype = np.array([g["contype"] for g in gl], dtype=np.int32)
  m.geom_conaffinity = np.array([g["conaffinity"] for g in gl], dtype=np.int32)
  m.geom_condim = np.array([g["condim"] for g in gl], dtype=np.int32)
  m.geom_bodyid = np.array([g["body"] for g in gl], dtype=np.int32)
  m.geom_dataid = np.full(ng, -1, dtype=np.int32)
  # height fields (MjModel.hfield_*): elevation normalised to [0, 1] like MuJoCo's compiler (raw = z_top * data)
  m.nhfield = len(hfield_names)
  m.hfield_size = np.zeros((m.nhfield, 4))
  m.hfield_nrow, m.hfield_ncol, m.hfield_adr = (np.zeros(m.nhfield, dtype=np.int32) for _ in range(3))
  hdata = []
  for i, name in enumerate(hfield_names):
    ha = hfield_assets[name]
    if "elevation" in ha:
      nrow, ncol = int(ha["nrow"]), int(ha["ncol"])
      e = np.array(_floats(ha["elevation"]), dtype=np.float64)
      if e.size != nrow * ncol:
        raise ValueError(f"hfield {name}: elevation has {e.size} values, expected {nrow * ncol}")
      e = e.reshape(nrow, ncol)[::-1]  # (MJCF lists the rows from the far edge (+y) first; MuJoCo stores row 0 at -y)
    else:  # PNG or MuJoCo's binary height-field format: like meshes, relative to <compiler meshdir>, which overrides assetdir
      adir = compiler["meshdir"] if compiler["meshdir"] is not None else (compiler["assetdir"] or "")
      path = ha["file"] if os.path.isabs(ha["file"]) else os.path.join(base_dir or "", adir, ha["file"])
      if not os.path.exists(path):
        raise FileNotFoundError(f"hfield {name!r}: file {path!r} not found")
      nrow, ncol, e = read_hfield_file(path)
    lo, hi = float(e.min()), float(e.max())
    e = (e - lo) / (hi - lo) if hi > lo else np.zeros_like(e)
    m.hfield_size[i] = _floats(ha["size"])
    m.hfield_nrow[i], m.hfield_ncol[i], m.hfield_adr[i] = nrow, ncol, sum(len(x) for x in hdata)
    hdata.append(e.reshape(-1))
  m.hfield_data = np.concatenate(hdata) if hdata else np.zeros(0)
  for i, g in enumerate(gl):
    if g["hfield"] is not None:
      m.geom_dataid[i] = hfield_names.index(g["hfield"])
  mesh_names = [n for n in mesh_compiled]
  m.nmesh = len(mesh_names)
  m.mesh_vertnum = np.array([len(mesh_compiled[n]["vert"]) for n in mesh_names], dtype=np.int32)
  m.mesh_vertadr = np.concatenate([[0], np.cumsum(m.mesh_vertnum)[:-1]]).astype(np.int32) if mesh_names else np.zeros(0, dtype=np.int32)
  m.mesh_vert = np.concatenate([mesh_compiled[n]["vert"] for n in mesh_names]).reshape(-1, 3) if mesh_names else np.zeros((0, 3))
  m.nmeshvert = len(m.mesh_vert)
  mds = [mesh_compiled[n] for n in mesh_names]
  glen = [len(md["graph"]) for md in mds]
  m.mesh_graphadr = np.concatenate([[0], np.cumsum(glen)[:-1]]).astype(np.int32) if mds else np.zeros(0, dtype=np.int32)
  m.mesh_graph = np.array([x for md in mds for x in md["graph"]], dtype=np.int32)
  m.mesh_polynum = np.array([len(md["polys"]) for md in mds], dtype=np.int32)
  m.mesh_polyadr = np.concatenate([[0], np.cumsum(m.mesh_polynum)[:-1]]).astype(np.int32) if mds else np.zeros(0, dtype=np.int32)
  m.mesh_polynormal = np.concatenate([md["polynormal"] for md in mds]).reshape(-1, 3) if mds else np.zeros((0, 3))
  m.mesh_polyvertnum = np.array([len(p) for md in mds for p in md["polys"]], dtype=np.int32)
  m.mesh_polyvertadr = np.concatenate([[0], np.cumsum(m.mesh_polyvertnum)[:-1]]).astype(np.int32) if mds else np.zeros(0, dtype=np.int32)
  m.mesh_polyvert = np.array([i for md in mds for p in md["polys"] for i in p], dtype=np.int32)
  m.mesh_polymapnum = np.array([len(pm) for md in mds for pm in md["polymap"]], dtype=np.int32)  # per vertex (global vertex index)
  m.mesh_polymapadr = np.concatenate([[0], np.cumsum(m.mesh_polymapnum)[:-1]]).astype(np.int32) if mds else np.zeros(0, dtype=np.int32)
  m.mesh_polymap = np.array([p for md in mds for pm in md["polymap"] for p in pm], dtype=np.int32)
  for i, g in enumerate(gl):
    if g["meshdata"] is not None:
      m.geom_dataid[i] = mesh_names.index(g["mesh"])
  m.geom_group = np.array([g["group"] for g in gl], dtype=np.int32)
  m.geom_rgba = np.array([g["rgba"] for g in gl], dtype=np.float64).reshape(-1, 4)
  m.geom_matid = np.array([g["matid"] for g in gl], dtype=np.int32)
  m.nmat = len(mat_names)
  m.mat_rgba = np.array(mat_rgba, dtype=np.float64).reshape(-1, 4)
  m.geom_priority = np.array([g["priority"] for g in gl], dtype=np.int32)
  m.geom_solmix = np.array([g["solmix"] for g in gl], dtype=np.float64)
  m.geom_solref = np.array([g["solref"] for g in gl]).reshape(-1, 2)
  m.geom_solimp = np.array([g["solimp"] for g in gl]).reshape(-1, 5)
  m.geom_size = np.array([g["size"] for g in gl]).reshape(-1, 3)
  m.geom_pos = np.array([g["pos"] for g in gl]).reshape(-1, 3)
  m.geom_quat = np.array([g["quat"] for g in gl]).reshape(-1, 4)
  m.geom_friction = np.array([g["friction"] for g in gl]).reshape(-1, 3)
  m.geom_margin = np.array([g["margin"] for g in gl], dtype=np.float64)
  m.geom_gap = np.array([g["gap"] for g in gl], dtype=np.float64)
  rb = [_geom_rbound_aabb(g["type"], g["size"]) for g in gl]
  m.geom_rbound = np.array([r[0] for r in rb], dtype=np.float64)
  m.geom_aabb = np.zeros((ng, 6))
  for i, r in enumerate(rb):
    m.geom_aabb[i, 3:] = r[1]
    if gl[i]["meshdata"] is not None:
      m.geom_rbound[i] = gl[i]["meshdata"]["rbound"]
      m.geom_aabb[i] = gl[i]["meshdata"]["aabb"]
    if gl[i]["hfield"] is not None:  # box from -base to +top around the grid
      hs = m.hfield_size[hfield_names.index(gl[i]["hfield"])]
      m.geom_aabb[i] = [0.0, 0.0, 0.5 * (hs[2] - hs[3]), hs[0], hs[1], 0.5 * (hs[2] + hs[3])]
      m.geom_rbound[i] = float(np.sqrt(hs[0] ** 2 + hs[1] ** 2 + max(hs[2], hs[3]) ** 2))

  # sites
  m.nsite = len(sites)
  m.site_bodyid = np.array([s["body"] for s in sites], dtype=np.int32)
  m.site_pos = np.array([s["pos"] for s in sites]).reshape(-1, 3)
  m.site_quat = np.array([s["quat"] for s in sites]).reshape(-1, 4)
  m.site_size = np.array([s["size"] for s in sites]).reshape(-1, 3)
  m.site_type = np.array([s["type"] for s in sites], dtype=np.int32)

  # body inertial properties
  m.body_mass = np.zeros(nbody)
  m.body_ipos = np.zeros((nbody, 3))
  m.body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
  m.body_inertia = np.zeros((nbody, 3))
  for bid, b in enumerate(bodies):
    use_geoms = compiler["inertiafromgeom"] == "true" or (compiler["inertiafromgeom"] == "auto" and b.inertial is None)
    if b.inertial is not None and not use_geoms:
      m.body_mass[bid] = b.inertial["mass"]
      m.body_ipos[bid] = b.inertial["pos"]
      m.body_iquat[bid] = b.inertial["quat"]
      m.body_inertia[bid] = b.inertial["diag"]
      continue
    if bid == 0:
      continue
    tot, com = 0.0, np.zeros(3)
    parts = []
    for g in b.geoms:
      if g["type"] == GEOM_MESH and g["meshdata"] is None and ((g["mass"] or 0.0) > 0 or (g["mass"] is None and g["density"] > 0)):
        attach_mesh(g)  # a non-colliding mesh geom that carries mass: its asset is needed after all (raises when the file is absent)
      vol, unit = _geom_volume_inertia(g["type"], g["size"])
      if g["meshdata"] is not None:
        vol, unit = g["meshdata"]["vol"], g["meshdata"]["unit"]
      mass = g["mass"] if g["mass"] is not None else g["density"] * vol
      if g["type"] == GEOM_PLANE or (g["type"] == GEOM_MESH and g["meshdata"] is None):
        mass = 0.0
      if mass <= 0:
        continue
      parts.append((mass, g["pos"], nm.quat_to_mat(g["quat"]), unit * mass))
      tot += mass
      com += mass * g["pos"]
    if tot <= 0:
      continue
    com /= tot
    inertia = np.zeros((3, 3))
    for mass, p, R, diag in parts:
      d = p - com
      inertia += R @ np.diag(diag) @ R.T + mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    w, v = np.linalg.eigh(inertia)
    order = np.argsort(-w)
    w, v = w[order], v[:, order]
    if np.linalg.det(v) < 0:
      v[:, 2] = -v[:, 2]
    m.body_mass[bid] = tot
    m.body_ipos[bid] = com
    m.body_iquat[bid] = nm.mat_to_quat(v)
    m.body_inertia[bid] = w
  if compiler["boundmass"] > 0:
    m.body_mass[1:] = np.maximum(m.body_mass[1:], compiler["boundmass"])
  if compiler["boundinertia"] > 0:
    m.body_inertia[1:] = np.maximum(m.body_inertia[1:], compiler["boundinertia"])
  for bid in range(1, nbody):
    if m.body_weldid[bid] != 0 and m.body_dofnum[bid] > 0:
      # a moving body (or one of its welded children) must carry mass
      pass

  # actuators
  acts = []
  ae = [child for sec in root.findall("actuator") for child in sec]
  if ae:
    for child in ae:
      if child.tag not in _ACT_TAGS:
        continue
      base, explicit = _resolve(child.tag, child, table, None)
      a = dict(base)
      a.update(_actuator_shortcut(child.tag, explicit, base=base))
      acts.append(a)
  nu = len(acts)
  m.nu = nu
  m.actuator_names = [a.get("name", "") for a in acts]
  m.actuator_trntype = np.zeros(nu, dtype=np.int32)
  m.actuator_dyntype = np.zeros(nu, dtype=np.int32)
  m.actuator_gaintype = np.zeros(nu, dtype=np.int32)
  m.actuator_biastype = np.zeros(nu, dtype=np.int32)
  m.actuator_trnid = np.full((nu, 2), -1, dtype=np.int32)
  m.actuator_actadr = np.full(nu, -1, dtype=np.int32)
  m.actuator_actnum = np.zeros(nu, dtype=np.int32)
  m.actuator_dynprm = np.zeros((nu, 10))
  m.actuator_gainprm = np.zeros((nu, 10))
  m.actuator_biasprm = np.zeros((nu, 10))
  m.actuator_ctrllimited = np.zeros(nu, dtype=bool)
  m.actuator_forcelimited = np.zeros(nu, dtype=bool)
  m.actuator_actlimited = np.zeros(nu, dtype=bool)
  m.actuator_actearly = np.zeros(nu, dtype=bool)
  m.actuator_ctrlrange = np.zeros((nu, 2))
  m.actuator_forcerange = np.zeros((nu, 2))
  m.actuator_actrange = np.zeros((nu, 2))
  m.actuator_gear = np.zeros((nu, 6))
  m.actuator_cranklength = np.zeros(nu)
  m.actuator_acc0 = np.zeros(nu)
  m.actuator_lengthrange = np.zeros((nu, 2))
  na = 0
  dynmap = {"none": DYN_NONE, "integrator": DYN_INTEGRATOR, "filter": DYN_FILTER, "filterexact": DYN_FILTEREXACT}
  gainmap = {"fixed": GAIN_FIXED, "affine": GAIN_AFFINE}
  biasmap = {"none": BIAS_NONE, "affine": BIAS_AFFINE}
  for i, a in enumerate(acts):
    if "joint" not in a:
      raise NotImplementedError("only joint transmissions are supported")
    jid = m.jnt_names.index(a["joint"])
    if m.jnt_type[jid] not in (JNT_HINGE, JNT_SLIDE):
      raise NotImplementedError("actuators on ball/free joints")
    m.actuator_trntype[i] = TRN_JOINT
    m.actuator_trnid[i, 0] = jid
    for key, table_, dflt in (("dyntype", dynmap, "none"), ("gaintype", gainmap, "fixed"), ("biastype", biasmap, "none")):
      if a.get(key, dflt) not in table_:
        raise NotImplementedError(f"actuator {key} '{a[key]}'")
    m.actuator_dyntype[i] = dynmap[a.get("dyntype", "none")]
    m.actuator_gaintype[i] = gainmap[a.get("gaintype", "fixed")]
    m.actuator_biastype[i] = biasmap[a.get("biastype", "none")]
    m.actuator_dynprm[i] = _vec(a, "dynprm", [1.0] + [0.0] * 9)
    m.actuator_gainprm[i] = _vec(a, "gainprm", [1.0] + [0.0] * 9)
    m.actuator_biasprm[i] = _vec(a, "biasprm", [0.0] * 10)
    if "_kp" in a:
      kp = float(a["_kp"])
      kv = float(a.get("_kv", 0.0))
      m.actuator_gainprm[i, 0] = kp
      m.actuator_biasprm[i, 0:3] = [0.0, -kp, -kv]
    if "_vkv" in a:
      kv = float(a["_vkv"])
      m.actuator_gainprm[i, 0] = kv
      m.actuator_biasprm[i, 0:3] = [0.0, 0.0, -kv]
    g = _vec(a, "gear", [1.0, 0, 0, 0, 0, 0])
    m.actuator_gear[i] = g
    for key, lim, rngname in (("ctrl", m.actuator_ctrllimited, m.actuator_ctrlrange),
                              ("force", m.actuator_forcelimited, m.actuator_forcerange),
                              ("act", m.actuator_actlimited, m.actuator_actrange)):
      has = (key + "range") in a
      l = _bool(a[key + "limited"]) if (key + "limited") in a else None
      if l is None:
        l = has and compiler["autolimits"]
      lim[i] = bool(l)
      rngname[i] = _vec(a, key + "range", [0, 0])
    m.actuator_actearly[i] = bool(_bool(a.get("actearly", "false")))
    if m.actuator_dyntype[i] != DYN_NONE:
      m.actuator_actadr[i] = na
      m.actuator_actnum[i] = 1
      na += 1
  m.na = na

  # equality constraints: joint couplings only (constraint.py:500-640; Panda's finger coupling)
  eqs = []
  for child in (c for sec in root.findall("equality") for c in sec):
    if child.tag != "joint":
      raise NotImplementedError(f"<equality><{child.tag}> is outside the hot-path scope (only joint equalities are built)")
    base, explicit = _resolve("equality", child, table, None)
    a = dict(base)
    a.update(explicit)
    j2 = a.get("joint2")
    data = np.zeros(11)
    data[:5] = _vec(a, "polycoef", [0.0, 1.0, 0.0, 0.0, 0.0])
    eqs.append({"obj1": m.jnt_names.index(a["joint1"]), "obj2": m.jnt_names.index(j2) if j2 else -1,
                "active": _bool(a.get("active", "true")), "solref": _vec(a, "solref", [0.02, 1.0]),
                "solimp": _vec(a, "solimp", [0.9, 0.95, 0.001, 0.5, 2.0]), "data": data})
  for e in eqs:
    for jid in (e["obj1"], e["obj2"]):
      if jid >= 0 and m.jnt_type[jid] not in (2, 3):
        raise ValueError("joint equality needs slide or hinge joints")
  m.neq = len(eqs)
  m.eq_type = np.full(m.neq, 2, dtype=np.int32)  # mjEQ_JOINT
  m.eq_obj1id = np.array([e["obj1"] for e in eqs], dtype=np.int32)
  m.eq_obj2id = np.array([e["obj2"] for e in eqs], dtype=np.int32)
  m.eq_active0 = np.array([int(e["active"]) for e in eqs], dtype=np.int32)
  m.eq_solref = np.array([e["solref"] for e in eqs]).reshape(-1, 2)
  m.eq_solimp = np.array([e["solimp"] for e in eqs]).reshape(-1, 5)
  m.eq_data = np.array([e["data"] for e in eqs]).reshape(-1, 11)

  # contact excludes / pairs
  m.exclude_signature = np.zeros(0, dtype=np.int32)
  m.npair = 0
  m.pair_geom1 = np.zeros(0, dtype=np.int32)
  m.pair_geom2 = np.zeros(0, dtype=np.int32)
  ce = [child for sec in root.findall("contact") for child in sec]
  if ce:
    sig = []
    for child in ce:
      if child.tag == "exclude":
        b1 = m.body_names.index(child.get("body1"))
        b2 = m.body_names.index(child.get("body2"))
        lo, hi = min(b1, b2), max(b1, b2)
        sig.append((lo << 16) + hi)
    m.exclude_signature = np.array(sig, dtype=np.int32)
    # explicit pairs: unspecified attributes take the values the two geoms would mix to (MuJoCo compiler semantics)
    prs = [child for child in ce if child.tag == "pair"]
    m.npair = len(prs)
    m.pair_geom1 = np.zeros(m.npair, dtype=np.int32)
    m.pair_geom2 = np.zeros(m.npair, dtype=np.int32)
    m.pair_dim = np.zeros(m.npair, dtype=np.int32)
    m.pair_friction = np.zeros((m.npair, 5))
    m.pair_solref = np.zeros((m.npair, 2))
    m.pair_solreffriction = np.zeros((m.npair, 2))
    m.pair_solimp = np.zeros((m.npair, 5))
    m.pair_margin = np.zeros(m.npair)
    m.pair_gap = np.zeros(m.npair)
    for i, child in enumerate(prs):
      base, explicit = _resolve("pair", child, table, None)
      a = dict(base)
      a.update(explicit)
      g1, g2 = m.geom_names.index(a["geom1"]), m.geom_names.index(a["geom2"])
      mix = mixed_contact_params(m, g1, g2)
      m.pair_geom1[i], m.pair_geom2[i] = g1, g2
      m.pair_dim[i] = int(a.get("condim", mix["condim"]))
      fr = _floats(a["friction"]) if "friction" in a else []
      m.pair_friction[i] = np.maximum(1e-5, np.concatenate([fr, mix["friction"][len(fr):]]))
      m.pair_solref[i] = _vec(a, "solref", mix["solref"])
      m.pair_solreffriction[i] = _vec(a, "solreffriction", [0.0, 0.0])
      m.pair_solimp[i] = _vec(a, "solimp", mix["solimp"])
      m.pair_margin[i] = float(a.get("margin", mix["margin"]))
      m.pair_gap[i] = float(a.get("gap", mix["gap"]))
  m.nexclude = len(m.exclude_signature)

  # keyframes
  keys = []
  keys = [k for sec in root.findall("keyframe") for k in sec if k.tag == "key"]
  m.nkey = len(keys)
  m.key_names = [k.get("name", "") for k in keys]
  m.key_time = np.zeros(m.nkey)
  m.key_qpos = np.tile(m.qpos0, (m.nkey, 1)) if m.nkey else np.zeros((0, nq))
  m.key_qvel = np.zeros((m.nkey, nv))
  m.key_act = np.zeros((m.nkey, na))
  m.key_ctrl = np.zeros((m.nkey, nu))
  mb = [int(np.flatnonzero(m.body_mocapid == j)[0]) for j in range(m.nmocap)]
  m.key_mpos = np.tile(m.body_pos[mb].reshape(1, -1), (m.nkey, 1)) if m.nkey else np.zeros((0, 3 * m.nmocap))
  m.key_mquat = np.tile(m.body_quat[mb].reshape(1, -1), (m.nkey, 1)) if m.nkey else np.zeros((0, 4 * m.nmocap))
  for i, k in enumerate(keys):
    if "time" in k.attrib:
      m.key_time[i] = float(k.get("time"))
    for name, arr in (("qpos", m.key_qpos), ("qvel", m.key_qvel), ("act", m.key_act), ("ctrl", m.key_ctrl), ("mpos", m.key_mpos),
                      ("mquat", m.key_mquat)):
      if name in k.attrib:
        v = _floats(k.get(name))
        if len(v) != arr.shape[1]:
          raise ValueError(f"keyframe {name} size {len(v)} != {arr.shape[1]}")
        arr[i] = v
  # MuJoCo's zero-quaternion rule (mju_normalize4: a quaternion of norm < mjMINVAL becomes the identity): test_data/aloha_pot stores
  # the pot's free joint as 0 0 0 0 in two keys.  Non-zero quaternions stay as written (every consumer normalises them).
  for i in range(m.nkey):
    for j in range(m.njnt):
      if m.jnt_type[j] in (0, 1):
        a = int(m.jnt_qposadr[j]) + (3 if m.jnt_type[j] == 0 else 0)
        if np.linalg.norm(m.key_qpos[i, a:a + 4]) < MJ_MINVAL:
          m.key_qpos[i, a:a + 4] = [1.0, 0.0, 0.0, 0.0]
    for j in range(m.nmocap):
      if np.linalg.norm(m.key_mquat[i, 4 * j:4 * j + 4]) < MJ_MINVAL:
        m.key_mquat[i, 4 * j:4 * j + 4] = [1.0, 0.0, 0.0, 0.0]

  # sizes not on the hot path
  m.ntendon = m.nflex = m.nplugin = 0
  m.ncam = m.nlight = 0
  m.nuserdata = 0
  _compile_sensors(m, root, [s_["name"] for s_ in sites])

  _sparse_structure(m)
  set_const(m)
  for st in root.findall("statistic"):  # explicit overrides of the compiled statistics (the solver scales its tolerances by it)
    if "meaninertia" in st.attrib:
      m.stat.meaninertia = float(st.get("meaninertia"))
  return m


# mjtSensor / mjtObj / mjtDataType / mjtStage values used below (MuJoCo's enums; UNPINNED here -- the mujoco package is absent: a real
# MjModel carries its own numbers in sensor_type, which put_model compares with these)
SENS = {"touch": 0, "accelerometer": 1, "force": 4, "torque": 5, "magnetometer": 6, "jointactuatorfrc": 16, "jointlimitpos": 20, "jointlimitvel": 21, "jointlimitfrc": 22, "e_potential": 43, "e_kinetic": 44, "framelinacc": 33, "frameangacc": 34, "velocimeter": 2, "gyro": 3, "jointpos": 9, "jointvel": 10, "actuatorpos": 13, "actuatorvel": 14, "actuatorfrc": 15, "ballquat": 18, "ballangvel": 19,
        "framepos": 26, "framequat": 27, "framexaxis": 28, "frameyaxis": 29, "framezaxis": 30, "framelinvel": 31, "frameangvel": 32, "subtreecom": 35, "subtreelinvel": 36, "subtreeangmom": 37, "clock": 45, "rangefinder": 7}
# sensors that keep their slot in sensordata (the reference's layout) but are not computed: the engine writes zeros and put_model warns
SENS_UNSUPPORTED = {}
_SENS_DIM = {"touch": 1, "rangefinder": 1, "ballquat": 4, "framequat": 4, "jointactuatorfrc": 1, "jointlimitpos": 1, "jointlimitvel": 1, "jointlimitfrc": 1, "e_potential": 1, "e_kinetic": 1, "jointpos": 1, "jointvel": 1, "actuatorpos": 1, "actuatorvel": 1, "actuatorfrc": 1, "clock": 1}
_SENS_STAGE = {"velocimeter": 2, "gyro": 2, "jointvel": 2, "actuatorvel": 2, "ballangvel": 2, "framelinvel": 2, "frameangvel": 2, "subtreelinvel": 2, "subtreeangmom": 2, "jointlimitvel": 2, "e_kinetic": 2, "touch": 3, "jointlimitfrc": 3, "jointactuatorfrc": 3, "actuatorfrc": 3, "accelerometer": 3, "force": 3, "torque": 3, "framelinacc": 3, "frameangacc": 3}  # default: POS (1)
_OBJ = {"body": 1, "xbody": 2, "geom": 5, "site": 6, "camera": 7}


def _compile_sensors(m, root, site_names):
  """<sensor> section, the subset csrc/sensor.hpp computes (reference sensor.py: joint / actuator / ball / frame / IMU-style site sensors,
  subtree centre of mass, clock); anything else raises."""
  rows = []
  lookup = {1: m.body_names, 2: m.body_names, 5: m.geom_names, 6: site_names}
  for sec in root.findall("sensor"):
    for e in sec:
      a = e.attrib
      if e.tag in SENS_UNSUPPORTED:
        rows.append(dict(type=SENS_UNSUPPORTED[e.tag][0], datatype=0, needstage=3, objtype=0, objid=-1, reftype=0, refid=-1, dim=SENS_UNSUPPORTED[e.tag][1], cutoff=0.0,
                         name=a.get("name", "")))
        continue
      if e.tag not in SENS:
        raise NotImplementedError(f"sensor <{e.tag}> is not implemented")
      objtype, objid, reftype, refid = 0, -1, 0, -1
      if e.tag in ("jointactuatorfrc", "jointlimitpos", "jointlimitvel", "jointlimitfrc"):
        objtype, objid = 3, m.jnt_names.index(a["joint"])
      elif e.tag in ("jointpos", "jointvel", "ballquat", "ballangvel"):
        objtype, objid = 3, m.jnt_names.index(a["joint"])  # mjOBJ_JOINT
        ball = m.jnt_type[objid] == JNT_BALL
        if ball != (e.tag in ("ballquat", "ballangvel")) or m.jnt_type[objid] == JNT_FREE:
          raise ValueError(f"sensor <{e.tag}> on a joint of the wrong type")
      elif e.tag in ("actuatorpos", "actuatorvel", "actuatorfrc"):
        objtype, objid = 19, m.actuator_names.index(a["actuator"])  # mjOBJ_ACTUATOR
      elif e.tag in ("velocimeter", "gyro", "accelerometer", "force", "torque", "touch", "magnetometer", "rangefinder"):
        objtype, objid = 6, site_names.index(a["site"])
      elif e.tag in ("subtreecom", "subtreelinvel", "subtreeangmom"):
        objtype, objid = 1, m.body_names.index(a["body"])
      elif e.tag not in ("clock", "e_potential", "e_kinetic"):
        objtype = _OBJ[a["objtype"]]
        objid = lookup[objtype].index(a["objname"])
        if "reftype" in a:
          reftype = _OBJ[a["reftype"]]
          refid = lookup[reftype].index(a["refname"])
      dim = _SENS_DIM.get(e.tag, 3)
      rows.append(dict(type=SENS[e.tag], datatype=1 if e.tag == "touch" else 3 if e.tag in ("ballquat", "framequat") else (2 if e.tag.startswith("frame") and e.tag.endswith("axis") else 0),
                       needstage=_SENS_STAGE.get(e.tag, 1), objtype=objtype, objid=objid, reftype=reftype, refid=refid, dim=dim,
                       cutoff=float(a.get("cutoff", 0.0)), name=a.get("name", "")))
  m.nsensor = len(rows)
  for k in ("type", "datatype", "needstage", "objtype", "objid", "reftype", "refid", "dim"):
    setattr(m, "sensor_" + k, np.array([r[k] for r in rows], dtype=np.int32))
  m.sensor_cutoff = np.array([r["cutoff"] for r in rows], dtype=np.float64)
  m.sensor_adr = np.concatenate([[0], np.cumsum(m.sensor_dim)[:-1]]).astype(np.int32) if rows else np.zeros(0, dtype=np.int32)
  m.nsensordata = int(m.sensor_dim.sum()) if rows else 0
  m.sensor_names = [r["name"] for r in rows]


def mixed_contact_params(m, g1, g2):
  """Contact parameters two geoms mix to (reference collision_core.py:297-414): priority, then solmix-weighted."""
  s1, s2 = float(m.geom_solmix[g1]), float(m.geom_solmix[g2])
  p1, p2 = int(m.geom_priority[g1]), int(m.geom_priority[g2])
  f1, f2 = np.asarray(m.geom_friction[g1], dtype=np.float64), np.asarray(m.geom_friction[g2], dtype=np.float64)
  if p1 > p2:
    mix, condim, f = 1.0, int(m.geom_condim[g1]), f1
  elif p2 > p1:
    mix, condim, f = 0.0, int(m.geom_condim[g2]), f2
  else:
    small1, small2 = s1 < MJ_MINVAL, s2 < MJ_MINVAL
    mix = 0.5 if (small1 and small2) else 0.0 if small1 else 1.0 if small2 else s1 / (s1 + s2)
    condim, f = max(int(m.geom_condim[g1]), int(m.geom_condim[g2])), np.maximum(f1, f2)
  r1, r2 = np.asarray(m.geom_solref[g1]), np.asarray(m.geom_solref[g2])
  solref = mix * r1 + (1 - mix) * r2 if (r1[0] > 0 and r2[0] > 0) else np.minimum(r1, r2)
  return {"condim": condim, "friction": np.array([f[0], f[0], f[1], f[2], f[2]]), "solref": solref,
          "solimp": mix * np.asarray(m.geom_solimp[g1]) + (1 - mix) * np.asarray(m.geom_solimp[g2]),
          "margin": float(m.geom_margin[g1] + m.geom_margin[g2]), "gap": float(m.geom_gap[g1] + m.geom_gap[g2])}


def _parse_option(elem, opt):
  for k, v in elem.attrib.items():
    if k in ("timestep", "tolerance", "ls_tolerance", "impratio", "density", "viscosity", "noslip_tolerance", "ccd_tolerance", "sleep_tolerance"):
      setattr(opt, k, float(v))
    elif k in ("iterations", "ls_iterations", "noslip_iterations", "ccd_iterations", "sdf_iterations", "sdf_initpoints"):
      setattr(opt, k, int(v))
    elif k in ("gravity", "wind", "magnetic"):
      setattr(opt, k, np.array(_floats(v)))
    elif k == "integrator":
      opt.integrator = _INT_NAMES[v.lower()]
    elif k == "solver":
      opt.solver = _SOL_NAMES[v.lower()]
    elif k == "cone":
      opt.cone = CONE_ELLIPTIC if v.lower() == "elliptic" else CONE_PYRAMIDAL
    elif k == "jacobian":
      opt.jacobian = {"dense": JAC_DENSE, "sparse": JAC_SPARSE, "auto": JAC_AUTO}[v.lower()]
  for f in elem.findall("flag"):
    for k, v in f.attrib.items():
      on = v.lower() == "enable"
      if k in DSBL:
        if on:
          opt.disableflags &= ~DSBL[k]
        else:
          opt.disableflags |= DSBL[k]
      elif k in ENBL:
        if on:
          opt.enableflags |= ENBL[k]
        else:
          opt.enableflags &= ~ENBL[k]
      elif k == "passive":
        bits = DSBL["spring"] | DSBL["damper"]
        opt.disableflags = (opt.disableflags & ~bits) if on else (opt.disableflags | bits)


def _sparse_structure(m):
  """CSR "M-structure": row i = dof-ancestor chain in ascending order, diagonal last
  (reference smooth.py:1064-1076, set_const.py:186-188)."""
  nv = m.nv
  rownnz = np.zeros(nv, dtype=np.int32)
  rows = []
  for i in range(nv):
    chain = []
    j = i
    while j >= 0:
      chain.append(j)
      j = m.dof_parentid[j]
    chain.reverse()
    rows.append(chain)
    rownnz[i] = len(chain)
  rowadr = np.zeros(nv, dtype=np.int32)
  if nv:
    rowadr[1:] = np.cumsum(rownnz)[:-1]
  m.M_rownnz = rownnz
  m.M_rowadr = rowadr
  m.M_colind = np.array([c for r in rows for c in r], dtype=np.int32)
  m.nC = m.nM = int(rownnz.sum())
  m.dof_Madr = (rowadr + rownnz - 1).astype(np.int32)  # address of the diagonal in CSR order
  # body-level helpers used by host code
  m.body_subtreemass = np.zeros(m.nbody)


# ---- host-side float64 smooth dynamics at one configuration (used by set_const only) ------------
def _host_kinematics(m, qpos):
  nb = m.nbody
  xpos, xquat = np.zeros((nb, 3)), np.zeros((nb, 4))
  xquat[0] = [1, 0, 0, 0]
  xanchor, xaxis = np.zeros((m.njnt, 3)), np.zeros((m.njnt, 3))
  for b in range(1, nb):
    p = m.body_parentid[b]
    ja, jn = m.body_jntadr[b], m.body_jntnum[b]
    if jn == 1 and m.jnt_type[ja] == JNT_FREE:
      qa = m.jnt_qposadr[ja]
      xpos[b] = qpos[qa : qa + 3]
      xquat[b] = nm.quat_normalize(qpos[qa + 3 : qa + 7])
      xanchor[ja] = xpos[b]
      xaxis[ja] = m.jnt_axis[ja]
      continue
    pos = nm.rot_vec_quat(m.body_pos[b], xquat[p]) + xpos[p]
    quat = nm.quat_mul(xquat[p], m.body_quat[b])
    for j in range(ja, ja + jn):
      qa = m.jnt_qposadr[j]
      anchor = nm.rot_vec_quat(m.jnt_pos[j], quat) + pos
      axis = nm.rot_vec_quat(m.jnt_axis[j], quat)
      t = m.jnt_type[j]
      if t == JNT_BALL:
        quat = nm.quat_mul(quat, nm.quat_normalize(qpos[qa : qa + 4]))
        pos = anchor - nm.rot_vec_quat(m.jnt_pos[j], quat)
      elif t == JNT_SLIDE:
        pos = pos + axis * (qpos[qa] - m.qpos0[qa])
      elif t == JNT_HINGE:
        quat = nm.quat_mul(quat, nm.axis_angle_to_quat(m.jnt_axis[j], qpos[qa] - m.qpos0[qa]))
        pos = anchor - nm.rot_vec_quat(m.jnt_pos[j], quat)
      xanchor[j], xaxis[j] = anchor, axis
    xpos[b], xquat[b] = pos, nm.quat_normalize(quat)
  return xpos, xquat, xanchor, xaxis


def host_mass_matrix(m, qpos):
  """Dense M(q) (float64) plus intermediate fields; mirrors smooth.py com_pos/crb."""
  nb, nv = m.nbody, m.nv
  xpos, xquat, xanchor, xaxis = _host_kinematics(m, qpos)
  xmat = np.array([nm.quat_to_mat(q) for q in xquat])
  xipos = np.array([xpos[b] + xmat[b] @ m.body_ipos[b] for b in range(nb)])
  ximat = np.array([nm.quat_to_mat(nm.quat_mul(xquat[b], m.body_iquat[b])) for b in range(nb)])
  # subtree com
  sub = xipos * m.body_mass[:, None]
  for b in range(nb - 1, 0, -1):
    sub[m.body_parentid[b]] += sub[b]
  stm = m.body_mass.copy()
  for b in range(nb - 1, 0, -1):
    stm[m.body_parentid[b]] += stm[b]
  for b in range(nb):
    if stm[b] != 0:
      sub[b] /= stm[b]
    else:
      sub[b] = xipos[b]
  cinert = np.zeros((nb, 10))
  for b in range(nb):
    mat, inert, mass = ximat[b], m.body_inertia[b], m.body_mass[b]
    dif = xipos[b] - sub[m.body_rootid[b]]
    tmp = mat @ np.diag(inert) @ mat.T
    r = np.zeros(10)
    r[0:3] = [tmp[0, 0], tmp[1, 1], tmp[2, 2]]
    r[3:6] = [tmp[0, 1], tmp[0, 2], tmp[1, 2]]
    r[0] += mass * (dif[1] ** 2 + dif[2] ** 2)
    r[1] += mass * (dif[0] ** 2 + dif[2] ** 2)
    r[2] += mass * (dif[0] ** 2 + dif[1] ** 2)
    r[3] -= mass * dif[0] * dif[1]
    r[4] -= mass * dif[0] * dif[2]
    r[5] -= mass * dif[1] * dif[2]
    r[6:9] = mass * dif
    r[9] = mass
    cinert[b] = r
  cdof = np.zeros((nv, 6))
  for j in range(m.njnt):
    b, d, t = m.jnt_bodyid[j], m.jnt_dofadr[j], m.jnt_type[j]
    off = sub[m.body_rootid[b]] - xanchor[j]
    if t == JNT_FREE:
      cdof[d + 0, 3] = cdof[d + 1, 4] = cdof[d + 2, 5] = 1.0
      for k in range(3):
        ax = xmat[b][:, k]
        cdof[d + 3 + k] = np.concatenate([ax, np.cross(ax, off)])
    elif t == JNT_BALL:
      for k in range(3):
        ax = xmat[b][:, k]
        cdof[d + k] = np.concatenate([ax, np.cross(ax, off)])
    elif t == JNT_SLIDE:
      cdof[d] = np.concatenate([np.zeros(3), xaxis[j]])
    else:
      cdof[d] = np.concatenate([xaxis[j], np.cross(xaxis[j], off)])
  crb = cinert.copy()
  for b in range(nb - 1, 0, -1):
    p = m.body_parentid[b]
    if p != 0:
      crb[p] += crb[b]
  M = np.zeros((nv, nv))
  for i in range(nv):
    buf = nm.inert_vec(crb[m.dof_bodyid[i]], cdof[i])
    j = i
    while j >= 0:
      M[i, j] = M[j, i] = np.dot(cdof[j], buf)
      j = m.dof_parentid[j]
    M[i, i] += m.dof_armature[i]
  return dict(M=M, xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat, subtree_com=sub,
              cinert=cinert, cdof=cdof, crb=crb, xanchor=xanchor, xaxis=xaxis)


def set_const(m):
  """mj_setConst restatement (reference set_const.py:35-59, 170-190, 208-375)."""
  nb, nv = m.nbody, m.nv
  stm = m.body_mass.copy()
  for b in range(nb - 1, 0, -1):
    stm[m.body_parentid[b]] += stm[b]
  m.body_subtreemass = stm
  m.dof_invweight0 = np.zeros(nv)
  m.body_invweight0 = np.zeros((nb, 2))
  if nv == 0:
    m.stat.meaninertia = 1.0
    m.tree_sleep_policy, m.dof_length = np.zeros(0, dtype=np.int32), np.zeros(0)
    return
  h = host_mass_matrix(m, m.qpos0)
  M = h["M"]
  m.stat.meaninertia = float(np.mean(np.diag(M)))
  Minv = np.linalg.inv(M)
  A = np.diag(Minv)
  for j in range(m.njnt):
    d, t = m.jnt_dofadr[j], m.jnt_type[j]
    if t == JNT_FREE:
      m.dof_invweight0[d : d + 3] = np.mean(A[d : d + 3])
      m.dof_invweight0[d + 3 : d + 6] = np.mean(A[d + 3 : d + 6])
    elif t == JNT_BALL:
      m.dof_invweight0[d : d + 3] = np.mean(A[d : d + 3])
    else:
      m.dof_invweight0[d] = A[d]
  for b in range(1, nb):
    if m.body_weldid[b] == 0:
      continue
    bb = b
    while bb > 0 and m.body_dofnum[bb] == 0:
      bb = m.body_parentid[bb]
    if bb == 0:
      continue
    J = np.zeros((6, nv))
    off = h["xipos"][b] - h["subtree_com"][m.body_rootid[b]]
    d = m.body_dofadr[bb] + m.body_dofnum[bb] - 1
    while d >= 0:
      ang, lin = h["cdof"][d, :3], h["cdof"][d, 3:]
      J[0:3, d] = lin + np.cross(ang, off)
      J[3:6, d] = ang
      d = m.dof_parentid[d]
    Ad = np.einsum("ri,ij,rj->r", J, Minv, J)
    tr, ro = float(np.mean(Ad[:3])), float(np.mean(Ad[3:]))
    if tr < MJ_MINVAL and ro > MJ_MINVAL:
      tr = ro
    elif ro < MJ_MINVAL and tr > MJ_MINVAL:
      ro = tr
    m.body_invweight0[b] = [tr, ro]
  # actuator_acc0 = |M^-1 moment| for joint transmissions (set_const.py:494-506)
  for i in range(m.nu):
    mom = np.zeros(nv)
    mom[m.jnt_dofadr[m.actuator_trnid[i, 0]]] = m.actuator_gear[i, 0]
    m.actuator_acc0[i] = float(np.linalg.norm(Minv @ mom))
  _sleep_tables(m, h)


def _sleep_tables(m, h):
  """Per-tree sleep policy and per-dof velocity weights (MjModel.tree_sleep_policy / dof_length; consumed by reference sleep.py:273-322).

  MuJoCo's compiler derives both; its source is not in the reference tree, so this is an UNPINNED stand-in for models loaded
  through this module (a real mujoco.MjModel carries its own values): a tree some actuator drives never sleeps (AUTO_NEVER, the one
  case the reference's own test holds: sleep_test.py:761-792), every other tree may (AUTO_ALLOWED); translational dofs weigh 1,
  rotational dofs weigh the reach of their body's subtree about the joint anchor at qpos0 (farthest geom bounding sphere, at least
  the body's equivalent-inertia-box half extent), which makes |dof_length * qvel| a linear speed as sleep_tolerance expects."""
  nv = m.nv
  m.tree_sleep_policy = np.full(m.ntree, 2, dtype=np.int32)  # SleepPolicy.AUTO_ALLOWED
  for i in range(m.nu):
    m.tree_sleep_policy[m.dof_treeid[m.jnt_dofadr[m.actuator_trnid[i, 0]]]] = 1  # SleepPolicy.AUTO_NEVER
  m.dof_length = np.ones(nv)
  if nv == 0:
    return
  xpos, xquat, xanchor, _ = _host_kinematics(m, m.qpos0)
  inbox = np.zeros(m.nbody)
  for b in range(1, m.nbody):
    if m.body_mass[b] > 0:
      I = np.asarray(m.body_inertia[b])
      ext = np.sqrt(np.maximum(6.0 * (I.sum() / 2.0 - I) / m.body_mass[b], 0.0)) * 0.5  # half sizes of the equivalent box
      inbox[b] = float(np.linalg.norm(ext))
  for i in range(nv):
    j = m.dof_jntid[i]
    t, b = m.jnt_type[j], m.dof_bodyid[i]
    rotational = (t == JNT_HINGE) or (t == JNT_BALL) or (t == JNT_FREE and i - m.jnt_dofadr[j] >= 3)
    if not rotational:
      continue
    reach = 0.0
    for bb in range(b, m.nbody):
      p = bb  # bodies of the subtree of b
      while p > b:
        p = m.body_parentid[p]
      if p != b:
        continue
      ipos = nm.rot_vec_quat(m.body_ipos[bb], xquat[bb]) + xpos[bb]
      reach = max(reach, float(np.linalg.norm(ipos - xanchor[j])) + inbox[bb])
      for g in range(m.ngeom):
        if m.geom_bodyid[g] == bb and m.geom_type[g] != 0:
          gp = nm.rot_vec_quat(m.geom_pos[g], xquat[bb]) + xpos[bb]
          reach = max(reach, float(np.linalg.norm(gp - xanchor[j])) + float(m.geom_rbound[g]))
    m.dof_length[i] = reach if reach > 0 else 1.0
