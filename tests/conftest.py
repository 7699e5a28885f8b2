import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HUMANOID_XML = os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml")
G1_XML = os.path.join(ROOT, "benchmarks", "unitree_g1", "scene_flat.xml")
PANDA_XML = os.path.join(ROOT, "benchmarks", "franka_emika_panda", "scene.xml")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
  try:
    import torch

    return torch.cuda.is_available()
  except Exception:
    return False


def pytest_collection_modifyitems(config, items):
  if _has_gpu():
    return
  skip = pytest.mark.skip(reason="no HIP device visible")
  for item in items:
    if "gpu" in item.keywords:
      item.add_marker(skip)


@pytest.fixture(scope="session")
def humanoid():
  import mujoco_warp_amd as mjw

  return mjw.mjcf.load_xml(HUMANOID_XML)


def relerr(a, b):
  """MAX-NORM relative error max|a - b| / max|b|: the right measure for a field whose entries share one scale (a Jacobian, a
  force vector); for state vectors use relerr_elem."""
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  if a.size == 0:
    return 0.0
  return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-12))


def relerr_elem(a, b, floor):
  """PER-ELEMENT relative error max_i |a_i - b_i| / max(|b_i|, floor); `floor` is an absolute value in the field's unit below
  which an entry is compared absolutely (a joint angle of 1e-3 rad must not hide behind a root position of 1 m)."""
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  if a.size == 0:
    return 0.0
  return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


# small inline models shared by CPU and GPU tests (same role as the reference's test_data/*.xml)
PENDULA_XML = """
<mujoco>
  <option timestep="0.002" gravity="0 0 -9.81"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05" pos="0 0 -1.2"/>
    <body name="p1" pos="0 0 0">
      <joint name="h1" type="hinge" axis="0 1 0" damping="0.05" armature="0.01" range="-60 60" limited="true"/>
      <geom type="capsule" fromto="0 0 0 0 0 -.3" size=".03"/>
      <body name="p2" pos="0 0 -.3">
        <joint name="h2" type="hinge" axis="1 0 0" stiffness="2" springref="10" frictionloss="0.05"/>
        <geom type="capsule" fromto="0 0 0 0 0 -.3" size=".025"/>
        <body name="p3" pos="0 0 -.3">
          <joint name="s3" type="slide" axis="0 0 1" range="-.1 .1" limited="true" damping=".5"/>
          <geom type="sphere" size=".05" pos="0 0 -.1"/>
        </body>
      </body>
    </body>
    <body name="ballbody" pos=".5 0 0">
      <joint name="b1" type="ball" damping="0.02" range="0 40" limited="true"/>
      <geom type="capsule" fromto="0 0 0 .2 0 -.2" size=".03"/>
    </body>
  </worldbody>
  <actuator>
    <motor joint="h1" gear="2" ctrlrange="-1 1" ctrllimited="true"/>
    <position joint="s3" kp="20" kv="1"/>
  </actuator>
  <keyframe>
    <key name="k0" qpos="0.9 0.2 0.05 0.9 0.3 0.3 0.1" qvel="0.5 -0.3 0.1 0.2 -0.4 0.3" ctrl="0.3 0.02"/>
  </keyframe>
</mujoco>
"""

FREE_BODIES_XML = """
<mujoco>
  <option timestep="0.004"/>
  <default>
    <geom contype="0" conaffinity="1"/>
  </default>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05" condim="3" contype="1" conaffinity="0"/>
    <body name="box" pos="0 0 .099">
      <freejoint/>
      <geom type="box" size=".1 .15 .1" friction=".8"/>
    </body>
    <body name="ball" pos=".5 0 .079">
      <freejoint/>
      <geom type="sphere" size=".08" condim="1"/>
    </body>
    <body name="cyl" pos="-.5 0 .099">
      <freejoint/>
      <geom type="cylinder" size=".07 .1" condim="3"/>
    </body>
    <body name="ell" pos="0 .6 .049">
      <freejoint/>
      <geom type="ellipsoid" size=".1 .08 .05"/>
    </body>
    <body name="cap" pos="0 -.6 .049" euler="0 80 20">
      <freejoint/>
      <geom type="capsule" size=".05 .1"/>
    </body>
  </worldbody>
  <keyframe>
    <key name="drop" qvel="0.3 0 0 0 1 0   0 0.2 -0.1 1 0 0   0 0 0 0 0 2   0.1 0 0 0 0 0   0 0 0 0.5 0 0"/>
  </keyframe>
</mujoco>
"""

# spheres / capsules / a box hitting each other and the floor (sphere-sphere, sphere-capsule, capsule-capsule, sphere-box)
PILE_XML = """
<mujoco>
  <option timestep="0.003" solver="CG"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="s1" pos="0 0 .1"><freejoint/><geom type="sphere" size=".1"/></body>
    <body name="s2" pos=".02 .01 .29"><freejoint/><geom type="sphere" size=".09" condim="1"/></body>
    <body name="c1" pos=".5 0 .06" euler="90 0 0"><freejoint/><geom type="capsule" size=".06 .15"/></body>
    <body name="c2" pos=".5 0.02 .17" euler="0 90 0"><freejoint/><geom type="capsule" size=".05 .12"/></body>
    <body name="s3" pos=".5 -.13 .2"><freejoint/><geom type="sphere" size=".06"/></body>
  </worldbody>
  <keyframe>
    <key name="k" qvel="0 0 0 0 0 0  0.1 0 -0.5 0 0 0  0 0 0 0 0 0  0 0 -0.3 0 0 0  0 0.2 -0.2 0 0 0"/>
  </keyframe>
</mujoco>
"""

# spheres resting on / pushed against cylinders: side, cap and rim regimes of sphere_cylinder (the upright cylinder is static: a
# free upright cylinder loaded off-centre tilts by ~1e-4 rad, where plane_cylinder senses the tilt through 1 - cos(tilt) and
# float32 cannot; contype 2: the cylinders do not collide with each other -- cylinder-cylinder needs the convex path)
SPHERE_CYLINDER_XML = """
<mujoco>
  <option timestep="0.003"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="cyl_up" pos="0 0 .15"><geom type="cylinder" size=".12 .15" contype="2"/></body>
    <body name="on_cap" pos=".03 .02 .379"><freejoint/><geom type="sphere" size=".08"/></body>
    <body name="on_rim" pos=".16 0 .33"><freejoint/><geom type="sphere" size=".06" condim="1"/></body>
    <body name="cyl_side" pos=".6 0 .1" euler="90 0 0"><freejoint/><geom type="cylinder" size=".1 .2" contype="2"/></body>
    <body name="on_side" pos=".62 .05 .279"><freejoint/><geom type="sphere" size=".08"/></body>
  </worldbody>
  <keyframe>
    <key name="k" qvel="0 0 -0.2 0 0 0  -0.3 0 -0.2 0 0 0  0 0 0 0 0 0  0 0 -0.3 0 0 0"/>
  </keyframe>
</mujoco>
"""


def multi_humanoid_xml(n=3, spacing=1.5):
  """n copies of the benchmark humanoid side by side in one model (nv = 27 n), built by cloning the torso subtree and the
  actuators with suffixed names -- the structure of the reference's three_humanoids.xml (which uses <replicate>/<attach>,
  elements the subset compiler does not expand)."""
  import copy
  import xml.etree.ElementTree as ET

  root = ET.parse(HUMANOID_XML).getroot()
  wb = root.find("worldbody")
  torso = next(b for b in wb.findall("body") if b.get("name") == "torso")
  acts = root.find("actuator")
  act0 = list(acts)
  for k in range(1, n):
    t = copy.deepcopy(torso)
    for e in t.iter():
      if e.get("name"):
        e.set("name", f"{e.get('name')}_{k}")
    p = [float(x) for x in torso.get("pos").split()]
    t.set("pos", f"{p[0]} {p[1] + k * spacing} {p[2]}")
    wb.append(t)
    for a in act0:
      c = copy.deepcopy(a)
      c.set("name", f"{a.get('name')}_{k}")
      c.set("joint", f"{a.get('joint')}_{k}")
      acts.append(c)
  for sec in root.findall("keyframe") + root.findall("contact") + root.findall("sensor"):
    root.remove(sec)
  return ET.tostring(root, encoding="unicode")


# capsules against boxes in every capsule_box regime: lying on a face, standing on it, overhanging an edge, leaning on an edge,
# next to a corner (the static box keeps the scene simple; the capsules also rest on the floor or on each other)
CAPSULE_BOX_XML = """
<mujoco>
  <option timestep="0.003"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="table" pos="0 0 .2"><geom type="box" size=".4 .3 .2" contype="2"/></body>
    <body name="lying" pos="0 0 .449" euler="0 90 0"><freejoint/><geom type="capsule" size=".05 .15"/></body>
    <body name="standing" pos=".2 .15 .599"><freejoint/><geom type="capsule" size=".04 .16" condim="1"/></body>
    <body name="overhang" pos=".4 -.15 .449" euler="0 90 20"><freejoint/><geom type="capsule" size=".05 .2"/></body>
    <body name="leaning" pos="-.52 0 .23" euler="0 35 0"><freejoint/><geom type="capsule" size=".04 .3"/></body>
    <body name="box2" pos="1 0 .0699"><freejoint/><geom type="box" size=".06 .05 .07" contype="2"/></body>
    <body name="onbox" pos="1 0 .1789" euler="90 0 0"><freejoint/><geom type="capsule" size=".04 .1"/></body>
  </worldbody>
  <keyframe>
    <key name="k" qvel="0 0 -.1 0 0 0  0 0 -.1 0 0 0  0 0 -.1 0 0.5 0  0 0 0 0 0 0  0 0 -.1 0 0 0  0 0 -.2 0 0 0"/>
  </keyframe>
</mujoco>
"""


# boxes on boxes: flat stacking, a rotated and an overhanging box, a two-box tower, a box balancing on an edge and one tumbling
# onto the table's rim (edge-edge and vertex-face configurations appear along the rollout)
BOX_BOX_XML = """
<mujoco>
  <option timestep="0.003"><flag nativeccd="disable"/></option>  <!-- the primitive box-box collider (the default is CCD + multi-contact: BOX_CCD_XML) -->
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="table" pos="0 0 .2"><geom type="box" size=".5 .4 .2"/></body>
    <body name="flat" pos="-.3 -.2 .4495"><freejoint/><geom type="box" size=".08 .06 .05"/></body>
    <body name="rot" pos="-.3 .15 .4495" euler="0 0 40"><freejoint/><geom type="box" size=".07 .07 .05"/></body>
    <body name="over" pos=".5 -.2 .4495" euler="0 0 15"><freejoint/><geom type="box" size=".1 .05 .05"/></body>
    <body name="base" pos=".1 .15 .4595"><freejoint/><geom type="box" size=".09 .08 .06"/></body>
    <body name="top" pos=".12 .16 .559" euler="0 0 25"><freejoint/><geom type="box" size=".05 .05 .04" condim="1"/></body>
    <body name="edge" pos=".1 -.2 .47" euler="45 0 10"><freejoint/><geom type="box" size=".06 .05 .05"/></body>
    <body name="tumble" pos=".55 .25 .52" euler="30 40 50"><freejoint/><geom type="box" size=".05 .04 .06"/></body>
  </worldbody>
  <keyframe>
    <key name="k" qvel="0 0 0 0 0 0  0 0 0 0 0 0  0 0 0 0 0 0  0 0 0 0 0 0  0 0 -.1 0 0 0  0 0 0 1 0 0  -.3 0 0 2 1 0"/>
  </keyframe>
</mujoco>
"""


# a mocap platform (posed through Data.mocap_pos / mocap_quat) carrying a ball and tilting under a box
MOCAP_XML = """
<mujoco>
  <option timestep="0.003"><flag nativeccd="disable"/></option>  <!-- (the crate on the tilting tray sits at the 1.6 mrad face-alignment threshold of the CCD multi-contact path; this scene is about mocap) -->
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="tray" mocap="true" pos="0 0 .3"><geom type="box" size=".25 .2 .02"/></body>
    <body name="ball" pos=".05 0 .3699"><freejoint/><geom type="sphere" size=".05"/></body>
    <body name="crate" pos="-.1 .05 .3599"><freejoint/><geom type="box" size=".05 .04 .04"/></body>
    <body name="pole" mocap="true" pos=".6 0 .2" euler="0 20 0"><geom type="capsule" size=".03 .2"/></body>
    <body name="arm" pos=".6 0 .6"><joint type="hinge" axis="0 1 0" damping=".05"/><geom type="capsule" fromto="0 0 0 0 0 -.25" size=".03"/></body>
  </worldbody>
</mujoco>
"""


# explicit <contact><pair>: a pair the type/affinity filter would drop (ghost sphere on the floor), a pair whose parameters
# override the geoms' (soft, frictionless capsule on the floor, with margin and gap) next to ordinary filtered pairs
EXPLICIT_PAIR_XML = """
<mujoco>
  <option timestep="0.003"/>
  <default><pair solref="0.01 1"/></default>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="ghost" pos="0 0 .099"><freejoint/><geom name="ghost" type="sphere" size=".1" contype="0" conaffinity="0"/></body>
    <body name="soft" pos=".5 0 .06" euler="0 90 0"><freejoint/><geom name="soft" type="capsule" size=".05 .15"/></body>
    <body name="plain" pos="-.5 0 .079"><freejoint/><geom name="plain" type="sphere" size=".08"/></body>
    <body name="rider" pos="-.5 .02 .235"><freejoint/><geom name="rider" type="sphere" size=".08"/></body>
  </worldbody>
  <contact>
    <pair geom1="floor" geom2="ghost" condim="1"/>
    <pair geom1="soft" geom2="floor" condim="4" margin="0.02" gap="0.005" friction="0.3 0.3 0.02 0.001 0.001" solref="0.03 0.8" solimp="0.8 0.9 0.01 0.5 2"/>
    <pair geom1="plain" geom2="rider" friction="2 2 0.01 0.0005 0.0005"/>
  </contact>
</mujoco>
"""
