"""Writes tests/models/clutter_synth.xml: a stand-in for BASELINE configs[4] (aloha_clutter), whose YCB / GSO mesh assets are not in the
reference tree (216 files referenced by scene_clutter.xml, none present).  Same problem class, every asset inline:
two 8-dof arms (6 hinges + 2 finger slides coupled by a joint equality, position actuators, armature / damping / frictionloss as the
aloha joints have), a table, 20 free convex-mesh objects (6 shapes x scales, 4 to 12 vertices -- both support-function branches)
-> nv = 2 * 8 + 20 * 6 = 136, elliptic cones, impratio 10, timestep 0.002, sleeping enabled (the arms are actuated -> never sleep).
python tools/make_clutter_synth.py
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shapes():
  s = {}
  s["tetra"] = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], float) * 0.6
  s["wedge"] = np.array([[-1, -1, -0.5], [1, -1, -0.5], [1, 1, -0.5], [-1, 1, -0.5], [-1, -1, 0.5], [-1, 1, 0.5]], float)
  s["octa"] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 0.8], [0, 0, -0.8]], float)
  s["brick"] = np.array([[x, y, z] for x in (-1, 1) for y in (-0.7, 0.7) for z in (-0.5, 0.5)], float)
  # cuboctahedron: 12 vertices (the hill-climbing branch of the mesh support function) and only triangles / squares as faces -- a clipped
  # polygon with more than four vertices is pruned by a start-vertex dependent greedy search in the reference (collision_gjk.py:1463),
  # which float32 and float64 resolve differently (tests/test_mesh.py); a parity scene keeps its resting faces at <= 4 vertices
  s["cubocta"] = np.array([[x, y, 0] for x in (-1, 1) for y in (-1, 1)] + [[x, 0, z] for x in (-1, 1) for z in (-1, 1)] + [[0, y, z] for y in (-1, 1) for z in (-1, 1)], float) * 0.8
  s["frustum"] = np.array([[x, y, -0.5] for x in (-1, 1) for y in (-1, 1)] + [[0.5 * x, 0.5 * y, 0.5] for x in (-1, 1) for y in (-1, 1)], float)
  return s


def arm(prefix, x, yaw):
  p = prefix
  return f"""
    <body name="{p}_base" pos="{x} 0 0.02" euler="0 0 {yaw}">
      <geom type="cylinder" size=".05 .02" class="arm" contype="0" conaffinity="0"/>
      <body name="{p}_l1" pos="0 0 .04">
        <joint name="{p}_waist" axis="0 0 1" range="-3.1 3.1" damping="5.76" armature=".1"/>
        <geom type="capsule" fromto="0 0 0 0 0 .08" size=".035" class="arm" contype="0" conaffinity="0"/>
        <body name="{p}_l2" pos="0 0 .08">
          <joint name="{p}_shoulder" axis="0 1 0" range="-1.85 1.25" damping="20" armature=".395" frictionloss="2"/>
          <geom type="capsule" fromto="0 0 0 0 0 .26" size=".03" class="arm"/>
          <body name="{p}_l3" pos="0 0 .26">
            <joint name="{p}_elbow" axis="0 1 0" range="-1.76 1.6" damping="18.5" armature=".383" frictionloss="1.15"/>
            <geom type="capsule" fromto="0 0 0 .22 0 0" size=".025" class="arm"/>
            <body name="{p}_l4" pos=".22 0 0">
              <joint name="{p}_forearm_roll" axis="1 0 0" range="-3.1 3.1" damping="6.78" armature=".14"/>
              <geom type="capsule" fromto="0 0 0 .08 0 0" size=".022" class="arm"/>
              <body name="{p}_l5" pos=".08 0 0">
                <joint name="{p}_wrist_angle" axis="0 1 0" range="-1.87 2.23" damping="6.28" armature=".008"/>
                <geom type="capsule" fromto="0 0 0 .06 0 0" size=".02" class="arm"/>
                <body name="{p}_l6" pos=".06 0 0">
                  <joint name="{p}_wrist_rotate" axis="1 0 0" range="-3.1 3.1" damping="1.2" armature=".0035"/>
                  <geom type="box" size=".02 .045 .02" pos=".02 0 0" class="arm"/>
                  <body name="{p}_fa" pos=".05 .012 0">
                    <joint name="{p}_finger_a" type="slide" axis="0 1 0" range="0 .03" damping="60" armature=".25" frictionloss="20"/>
                    <geom type="box" size=".03 .006 .012" pos=".03 0 0" class="finger"/>
                  </body>
                  <body name="{p}_fb" pos=".05 -.012 0">
                    <joint name="{p}_finger_b" type="slide" axis="0 -1 0" range="0 .03" damping="60" armature=".25" frictionloss="20"/>
                    <geom type="box" size=".03 .006 .012" pos=".03 0 0" class="finger"/>
                  </body>
                </body>
              </body>
            </body>
          </body>
        </body>
      </body>
    </body>"""


def main():
  rng = np.random.default_rng(7)
  sh = shapes()
  names = list(sh)
  assets, bodies = [], []
  # 20 objects on a 5 x 4 grid of the table (0.9 x 0.6), apart at the start: every object is its own constraint island until the arms
  # or a neighbour reach it; two of them start stacked on another one
  k = 0
  for iy in range(4):
    for ix in range(5):
      name = names[k % len(names)]
      scale = np.array([0.035, 0.035, 0.035]) * (1.0 + 0.15 * ((k * 7) % 5)) * np.array([1.0, 0.9 + 0.05 * (k % 3), 1.0 + 0.1 * (k % 2)])
      assets.append(f'    <mesh name="m{k}" vertex="{" ".join(f"{v:.6g}" for v in sh[name].reshape(-1))}" scale="{scale[0]:.4g} {scale[1]:.4g} {scale[2]:.4g}"/>')
      x, y = -0.32 + 0.16 * ix + 0.01 * rng.uniform(-1, 1), -0.21 + 0.14 * iy + 0.01 * rng.uniform(-1, 1)
      z = 0.04 + 0.05 + float(np.max(np.abs(sh[name][:, 2])) * scale[2]) + 0.002
      if k in (7, 13):  # stacked on the object below it in the grid
        x, y, z = bodies_xy[k - 5][0], bodies_xy[k - 5][1], z + 0.09
      condim = 4 if k % 9 == 4 else (6 if k % 11 == 10 else 3)
      bodies.append((x, y, z, rng.uniform(0, 2 * np.pi), k, condim))
      k += 1
      bodies_xy = [(b[0], b[1]) for b in bodies]
  body_xml = "\n".join(
    f'    <body name="obj{k}" pos="{x:.4f} {y:.4f} {z:.4f}" euler="0 0 {yaw:.4f}"><freejoint/><geom type="mesh" mesh="m{k}" condim="{cd}" class="obj"/></body>'
    for x, y, z, yaw, k, cd in bodies)
  act = "\n".join(
    f'    <position joint="{p}_{j}" kp="{kp}" ctrlrange="{lo} {hi}"/>'
    for p in ("left", "right")
    for j, kp, lo, hi in (("waist", 43, -3.1, 3.1), ("shoulder", 265, -1.85, 1.25), ("elbow", 227, -1.76, 1.6), ("forearm_roll", 78, -3.1, 3.1),
                          ("wrist_angle", 37, -1.87, 2.23), ("wrist_rotate", 10.4, -3.1, 3.1), ("finger_a", 450, 0, 0.03)))
  armq = [0, -0.6, 0.9, 0, 0.6, 0, 0.02, 0.02]  # the pose the position targets hold: elbows bent, hands above the table, apart
  objq = []
  for x, y, z, yaw, k, cd in bodies:
    objq += [x, y, z, np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
  key_qpos = " ".join(f"{v:.6g}" for v in armq + armq + objq)
  xml = f"""<mujoco model="clutter synth">
  <!-- generated by tools/make_clutter_synth.py: stand-in for BASELINE configs[4] (aloha_clutter); see that script -->
  <compiler angle="radian" autolimits="true"/>
  <option timestep="0.002" impratio="10" cone="elliptic">
    <flag sleep="enable" multiccd="enable"/>
  </option>
  <default>
    <default class="arm"><geom density="600" friction="1 .005 .0001" rgba=".2 .2 .25 1"/></default>
    <default class="finger"><geom density="600" friction="1.5 .005 .0001" condim="4" rgba=".6 .6 .6 1"/></default>
    <default class="obj"><geom density="1500" friction=".8 .005 .0001" rgba=".8 .5 .2 1"/></default>
  </default>
  <asset>
{os.linesep.join(assets)}
  </asset>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05" pos="0 0 -.5"/>
    <geom name="table" type="box" size=".5 .35 .02" pos="0 0 .02" friction="1 .005 .0001"/>
{arm("left", -0.45, 0.0)}
{arm("right", 0.45, 3.14159265)}
{body_xml}
  </worldbody>
  <equality>
    <joint joint1="left_finger_a" joint2="left_finger_b" polycoef="0 1 0 0 0"/>
    <joint joint1="right_finger_a" joint2="right_finger_b" polycoef="0 1 0 0 0"/>
  </equality>
  <actuator>
{act}
  </actuator>
  <keyframe>
    <key name="reach" qpos="{key_qpos}" ctrl="0 0 .7 0 .25 0 .02  0.3 0 .7 0 .25 0 .02"/>
  </keyframe>
</mujoco>
"""
  for out in (os.path.join(ROOT, "tests", "models", "clutter_synth.xml"), os.path.join(ROOT, "benchmarks", "clutter_synth", "scene_clutter_synth.xml")):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write(xml)
    print("wrote", out)


if __name__ == "__main__":
  main()
