// Sensors (reference sensor.py), the subset of RL-style consumers: joint / actuator / ball-joint readings, frame position / axes /
// quaternion / linear and angular velocity (optionally relative to a reference frame), the IMU-style site sensors velocimeter and gyro,
// subtree centre of mass, clock.  One thread per (world, sensor); everything is read from the public Data arrays, which are final for all
// of these once fwd_position / fwd_velocity / fwd_actuation have run -- the launch sits before the solver (whose fused epilogue may
// already integrate the state: Data.qpos / qvel / time must still be the step's inputs when JOINTPOS / JOINTVEL / CLOCK are read).
// Not built: sensors that need rne_postconstraint (accelerometer, force, torque, touch, frame accelerations), tendons, cameras, rays,
// collision sensors, energies (put_model / the loader raise).
#pragma once
#include "dev_common.hpp"
#include "ray.hpp"

enum { SENS_TOUCH = 0, SENS_ACCELEROMETER = 1, SENS_FORCE = 4, SENS_TORQUE = 5, SENS_MAGNETOMETER = 6, SENS_VELOCIMETER = 2, SENS_GYRO = 3, SENS_JOINTPOS = 9, SENS_JOINTVEL = 10, SENS_ACTUATORPOS = 13, SENS_ACTUATORVEL = 14, SENS_ACTUATORFRC = 15,
       SENS_JOINTACTFRC = 16, SENS_BALLQUAT = 18, SENS_BALLANGVEL = 19, SENS_JOINTLIMITPOS = 20, SENS_JOINTLIMITVEL = 21, SENS_JOINTLIMITFRC = 22, SENS_FRAMEPOS = 26, SENS_FRAMEQUAT = 27, SENS_FRAMEXAXIS = 28, SENS_FRAMEYAXIS = 29, SENS_FRAMEZAXIS = 30,
       SENS_FRAMELINVEL = 31, SENS_FRAMEANGVEL = 32, SENS_FRAMELINACC = 33, SENS_FRAMEANGACC = 34, SENS_SUBTREECOM = 35, SENS_SUBTREELINVEL = 36, SENS_SUBTREEANGMOM = 37, SENS_E_POTENTIAL = 43, SENS_E_KINETIC = 44, SENS_CLOCK = 45, SENS_RANGEFINDER = 7 };
enum { OBJ_BODY = 1, OBJ_XBODY = 2, OBJ_GEOM = 5, OBJ_SITE = 6 };

struct SensFrame {
  V3 pos;
  const float* mat;
  int body;
};
// sensor.py:266-340 _get_pos / _get_mat / _get_body_id
DEV SensFrame sens_frame(const MjhModel& m, const MjhData& d, int w, int objtype, int id) {
  SensFrame f{V3{0, 0, 0}, nullptr, 0};
  if (objtype == OBJ_BODY) {
    f.body = id;
    f.pos = ld3(d.xipos + ((size_t)w * m.nbody + id) * 3);
    f.mat = d.ximat + ((size_t)w * m.nbody + id) * 9;
  } else if (objtype == OBJ_XBODY) {
    f.body = id;
    f.pos = ld3(d.xpos + ((size_t)w * m.nbody + id) * 3);
    f.mat = d.xmat + ((size_t)w * m.nbody + id) * 9;
  } else if (objtype == OBJ_GEOM) {
    f.body = m.geom_bodyid[id];
    f.pos = ld3(d.geom_xpos + ((size_t)w * m.ngeom + id) * 3);
    f.mat = d.geom_xmat + ((size_t)w * m.ngeom + id) * 9;
  } else if (objtype == OBJ_SITE) {
    f.body = m.site_bodyid[id];
    f.pos = ld3(d.site_xpos + ((size_t)w * m.nsite + id) * 3);
    f.mat = d.site_xmat + ((size_t)w * m.nsite + id) * 9;
  }
  return f;
}
// sensor.py:342-374 _get_quat
DEV Q4 sens_quat(const MjhModel& m, const MjhData& d, int w, int objtype, int id) {
  const float* xquat = d.xquat + (size_t)w * m.nbody * 4;
  if (objtype == OBJ_BODY) return mul_quat(ld4(xquat + 4 * id), ld4(bf(m.body_iquat, m.body_iquat_nb, w, 4 * m.nbody) + 4 * id));
  if (objtype == OBJ_XBODY) return ld4(xquat + 4 * id);
  if (objtype == OBJ_GEOM) return mul_quat(ld4(xquat + 4 * m.geom_bodyid[id]), ld4(bf(m.geom_quat, m.geom_quat_nb, w, 4 * m.ngeom) + 4 * id));
  if (objtype == OBJ_SITE) return mul_quat(ld4(xquat + 4 * m.site_bodyid[id]), ld4(bf(m.site_quat, m.site_quat_nb, w, 4 * m.nsite) + 4 * id));
  return Q4{1, 0, 0, 0};
}
// sensor.py:1066-1105 _cvel_offset and the velocity of the frame's origin
DEV void sens_vel(const MjhModel& m, const MjhData& d, int w, const SensFrame& f, V3& lin, V3& ang) {
  const float* cv = d.cvel + ((size_t)w * m.nbody + f.body) * 6;
  const V3 off = f.pos - ld3(d.subtree_com + ((size_t)w * m.nbody + m.body_rootid[f.body]) * 3);
  ang = ld3(cv);
  lin = ld3(cv + 3) - cross(off, ang);
}

// acceleration of a body's frame at the tree's centre of mass (smooth.py:1354-1426 rne_postconstraint's cacc with flg_acc): the world
// accelerates at -gravity, every dof on the way down adds cdof_dot qvel + cdof qacc; one thread walks the body's dof chain
DEV void sens_cacc(const MjhModel& m, const MjhData& d, int w, int body, V3& ang, V3& lin) {
  ang = V3{0, 0, 0};
  lin = (m.disableflags & DSBL_GRAVITY) ? V3{0, 0, 0} : V3{0, 0, 0} - ld3(bf(m.opt_gravity, m.opt_gravity_nb, w, 3));
  int dof = m.body_lastdof[body];  // last dof of the body or of its nearest moving ancestor, -1 for static bodies
  const float* cdof = d.cdof + (size_t)w * m.nv * 6;
  const float* cdd = d.cdof_dot + (size_t)w * m.nv * 6;
  const float* qvel = d.qvel + (size_t)w * m.nv;
  const float* qacc = d.qacc + (size_t)w * m.nv;
  while (dof >= 0) {
    ang = ang + qvel[dof] * ld3(cdd + 6 * dof) + qacc[dof] * ld3(cdof + 6 * dof);
    lin = lin + qvel[dof] * ld3(cdd + 6 * dof + 3) + qacc[dof] * ld3(cdof + 6 * dof + 3);
    dof = m.dof_parentid[dof];
  }
}

// does the ray pnt + t vec, t >= 0, meet the site's shape (sensor.py:2063-2139 asks ray.ray_geom for a non-negative distance; only the
// yes / no matters, so the conventions of the reference's ray routines do not enter): sphere, ellipsoid, box, cylinder, capsule
DEV bool ray_interval_cyl(V3 p, V3 v, float r, float hh, float& t0, float& t1) {  // finite cylinder about z: entry / exit parameters
  const float a = v.x * v.x + v.y * v.y, b = p.x * v.x + p.y * v.y, c = p.x * p.x + p.y * p.y - r * r;
  t0 = -3.0e38f;
  t1 = 3.0e38f;
  if (a < 1e-20f) {
    if (c > 0.0f) return false;
  } else {
    const float disc = b * b - a * c;
    if (disc < 0.0f) return false;
    const float sq = sqrtf(disc);
    t0 = (-b - sq) / a;
    t1 = (-b + sq) / a;
  }
  if (fabsf(v.z) < 1e-20f) {
    if (fabsf(p.z) > hh) return false;
  } else {
    const float ta = (-hh - p.z) / v.z, tb = (hh - p.z) / v.z;
    t0 = fmaxf(t0, fminf(ta, tb));
    t1 = fminf(t1, fmaxf(ta, tb));
  }
  return t1 >= fmaxf(t0, 0.0f);
}
DEV bool ray_hits_sphere(V3 p, V3 v, float r) {
  const float a = dot(v, v), b = dot(p, v), c = dot(p, p) - r * r, disc = b * b - a * c;
  return a > 0.0f && disc >= 0.0f && (-b + sqrtf(disc)) >= 0.0f;
}
DEV bool ray_hits_zone(int type, V3 size, V3 pos, const float* mat, V3 pnt, V3 vec) {
  const V3 p = matT_mul(mat, pnt - pos), v = matT_mul(mat, vec);
  float t0, t1;
  if (type == G_SPHERE) return ray_hits_sphere(p, v, size.x);
  if (type == G_ELLIPSOID) return ray_hits_sphere(V3{p.x / size.x, p.y / size.y, p.z / size.z}, V3{v.x / size.x, v.y / size.y, v.z / size.z}, 1.0f);
  if (type == G_CYLINDER) return ray_interval_cyl(p, v, size.x, size.y, t0, t1);
  if (type == G_CAPSULE)
    return ray_interval_cyl(p, v, size.x, size.y, t0, t1) || ray_hits_sphere(p - V3{0, 0, size.y}, v, size.x) || ray_hits_sphere(p + V3{0, 0, size.y}, v, size.x);
  if (type == G_BOX) {
    t0 = -3.0e38f;
    t1 = 3.0e38f;
    const float pp[3] = {p.x, p.y, p.z}, vv[3] = {v.x, v.y, v.z}, ss[3] = {size.x, size.y, size.z};
    for (int k = 0; k < 3; ++k) {
      if (fabsf(vv[k]) < 1e-20f) {
        if (fabsf(pp[k]) > ss[k]) return false;
      } else {
        const float ta = (-ss[k] - pp[k]) / vv[k], tb = (ss[k] - pp[k]) / vv[k];
        t0 = fmaxf(t0, fminf(ta, tb));
        t1 = fminf(t1, fmaxf(ta, tb));
      }
    }
    return t1 >= fmaxf(t0, 0.0f);
  }
  return false;
}

// stage 0: position / velocity stage sensors and actuator forces (before the solver); stage 1: acceleration stage (after it, before the integrator)
__global__ void __launch_bounds__(256) k_sensor(MjhModel m, MjhData d, int stage) {
  const int idx = blockIdx.x * 256 + threadIdx.x, ns = m.nsensor;
  if (idx >= d.nworld * ns) return;
  const int w = idx / ns, i = idx - w * ns;
  const int t = m.sensor_type[i], id = m.sensor_objid[i], ot = m.sensor_objtype[i], rid = m.sensor_refid[i], rt = m.sensor_reftype[i];
  const bool acc_type = t == SENS_ACCELEROMETER || t == SENS_FRAMELINACC || t == SENS_FRAMEANGACC || t == SENS_FORCE || t == SENS_TORQUE || t == SENS_JOINTLIMITFRC || t == SENS_TOUCH;
  if (acc_type != (stage == 1)) return;
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto put3 = [&](V3 a) {
    v[0] = a.x;
    v[1] = a.y;
    v[2] = a.z;
  };
  if (t == SENS_JOINTPOS) v[0] = d.qpos[(size_t)w * m.nq + m.jnt_qposadr[id]];
  else if (t == SENS_JOINTVEL) v[0] = d.qvel[(size_t)w * m.nv + m.jnt_dofadr[id]];
  else if (t == SENS_ACTUATORPOS) v[0] = d.actuator_length[(size_t)w * m.nu + id];
  else if (t == SENS_ACTUATORVEL) v[0] = d.actuator_velocity[(size_t)w * m.nu + id];
  else if (t == SENS_ACTUATORFRC) v[0] = d.actuator_force[(size_t)w * m.nu + id];
  else if (t == SENS_MAGNETOMETER) put3(matT_mul(sens_frame(m, d, w, OBJ_SITE, id).mat, ld3(bf(m.opt_magnetic, m.opt_magnetic_nb, w, 3))));  // sensor.py:117-128
  else if (t == SENS_JOINTACTFRC) v[0] = d.qfrc_actuator[(size_t)w * m.nv + m.jnt_dofadr[id]];
  else if (t == SENS_E_POTENTIAL) v[0] = d.energy[2 * w];  // (k_energy ran just before)
  else if (t == SENS_E_KINETIC) v[0] = d.energy[2 * w + 1];
  else if (t == SENS_JOINTLIMITPOS || t == SENS_JOINTLIMITVEL || t == SENS_JOINTLIMITFRC) {
    // sensor.py:228-263, 1028-1063, 1640-1675: the joint's limit row if it is active (0 otherwise)
    const int r0 = d.ne[w] + d.nf[w], r1 = min(r0 + d.nl[w], d.njmax);
    const size_t eo = (size_t)w * d.njmax;
    for (int r = r0; r < r1; ++r)
      if (d.efc_type[eo + r] == CT_LIMIT_JOINT && d.efc_id[eo + r] == id)
        v[0] = t == SENS_JOINTLIMITPOS ? d.efc_pos[eo + r] - d.efc_margin[eo + r] : (t == SENS_JOINTLIMITVEL ? d.efc_vel[eo + r] : d.efc_force[eo + r]);
  } else if (t == SENS_BALLQUAT) {
    const Q4 q = quat_normalize(ld4(d.qpos + (size_t)w * m.nq + m.jnt_qposadr[id]));
    v[0] = q.w; v[1] = q.x; v[2] = q.y; v[3] = q.z;
  } else if (t == SENS_BALLANGVEL) put3(ld3(d.qvel + (size_t)w * m.nv + m.jnt_dofadr[id]));
  else if (t == SENS_CLOCK) v[0] = d.time[w];
  else if (t == SENS_RANGEFINDER) {  // sensor.py:179-196, 815-845: along the site's z axis, past the site's own body, every group, static geoms too
    const float* xm = d.site_xmat + ((size_t)w * m.nsite + id) * 9;
    const RayGroup all = {{MJ_MAXVAL, MJ_MAXVAL, MJ_MAXVAL, MJ_MAXVAL, MJ_MAXVAL, MJ_MAXVAL}};
    int g;
    V3 n;
    v[0] = ray_world(m, d, w, ld3(d.site_xpos + ((size_t)w * m.nsite + id) * 3), V3{xm[2], xm[5], xm[8]}, all, 1, m.site_bodyid[id], g, n);
  }
  else if (t == SENS_SUBTREECOM) put3(ld3(d.subtree_com + ((size_t)w * m.nbody + id) * 3));
  else if (t == SENS_SUBTREELINVEL) put3(ld3(d.subtree_linvel + ((size_t)w * m.nbody + id) * 3));  // (k_subtree_vel ran just before)
  else if (t == SENS_SUBTREEANGMOM) put3(ld3(d.subtree_angmom + ((size_t)w * m.nbody + id) * 3));
  else if (t == SENS_FRAMEPOS) {
    const SensFrame f = sens_frame(m, d, w, ot, id);
    if (rid == -1) put3(f.pos);
    else {
      const SensFrame r = sens_frame(m, d, w, rt, rid);
      put3(matT_mul(r.mat, f.pos - r.pos));
    }
  } else if (t == SENS_FRAMEXAXIS || t == SENS_FRAMEYAXIS || t == SENS_FRAMEZAXIS) {
    const int ax = t - SENS_FRAMEXAXIS;
    const SensFrame f = sens_frame(m, d, w, ot, id);
    const V3 a = V3{f.mat[ax], f.mat[3 + ax], f.mat[6 + ax]};
    if (rid == -1) put3(a);
    else put3(matT_mul(sens_frame(m, d, w, rt, rid).mat, a));
  } else if (t == SENS_FRAMEQUAT) {
    Q4 q = sens_quat(m, d, w, ot, id);
    if (rid != -1) {
      const Q4 r = sens_quat(m, d, w, rt, rid);
      q = mul_quat(Q4{r.w, -r.x, -r.y, -r.z}, q);
    }
    v[0] = q.w; v[1] = q.x; v[2] = q.y; v[3] = q.z;
  } else if (t == SENS_VELOCIMETER || t == SENS_GYRO) {
    const SensFrame f = sens_frame(m, d, w, OBJ_SITE, id);
    V3 lin, ang;
    sens_vel(m, d, w, f, lin, ang);
    put3(matT_mul(f.mat, t == SENS_GYRO ? ang : lin));
  } else if (t == SENS_FRAMELINVEL || t == SENS_FRAMEANGVEL) {
    const SensFrame f = sens_frame(m, d, w, ot, id);
    V3 lin, ang;
    sens_vel(m, d, w, f, lin, ang);
    if (rid == -1) put3(t == SENS_FRAMELINVEL ? lin : ang);
    else {
      const SensFrame r = sens_frame(m, d, w, rt, rid);
      V3 rlin, rang;
      sens_vel(m, d, w, r, rlin, rang);
      if (t == SENS_FRAMELINVEL) put3(matT_mul(r.mat, lin - rlin + cross(f.pos - r.pos, rang)));
      else put3(matT_mul(r.mat, ang - rang));
    }
  }
  else if (t == SENS_TOUCH) {
    // sensor.py:2063-2139: sum of the normal forces of the contacts of the site's body whose normal ray meets the site's zone (from the
    // world's contact records: the public contact arrays are published off the critical path)
    const SensFrame f = sens_frame(m, d, w, OBJ_SITE, id);
    const V3 zsize = ld3(m.site_size + 3 * id);
    const int ztype = m.site_type[id], ncon = min(d.ws_ncon[w], d.concap);
    const float* force = d.efc_force + (size_t)w * d.njmax;
    for (int c = 0; c < ncon; ++c) {
      const float* rec = d.ws_contact + ((size_t)w * d.concap + c) * CON_STRIDE;
      const int* reci = reinterpret_cast<const int*>(rec);
      const int b1 = m.geom_bodyid[reci[25]], b2 = m.geom_bodyid[reci[26]], adr0 = reci[28], nrow = reci[29];
      if (adr0 < 0 || (f.body != b1 && f.body != b2)) continue;
      float nf = 0.0f;
      const int rows = m.cone == CONE_PYRAMIDAL ? nrow : 1;  // pyramidal: the rows of a contact add up to its normal force
      for (int k = 0; k < rows; ++k)
        if (adr0 + k < d.njmax) nf += force[adr0 + k];
      if (nf <= 0.0f) continue;
      V3 ray = normalize(ld3(rec + 4) * nf);
      if (f.body == b2) ray = V3{0, 0, 0} - ray;
      if (ray_hits_zone(ztype, zsize, f.pos, f.mat, ld3(rec + 1), ray)) v[0] += nf;
    }
  } else if (t == SENS_FORCE || t == SENS_TORQUE) {  // sensor.py:1542-1577: the interaction force / torque at the site's body (k_rne_postconstraint ran just before)
    const SensFrame f = sens_frame(m, d, w, OBJ_SITE, id);
    const float* ci = d.cfrc_int + ((size_t)w * m.nbody + f.body) * 6;
    if (t == SENS_FORCE) put3(matT_mul(f.mat, ld3(ci + 3)));
    else {
      const V3 dif = f.pos - ld3(d.subtree_com + ((size_t)w * m.nbody + m.body_rootid[f.body]) * 3);
      put3(matT_mul(f.mat, ld3(ci) - cross(dif, ld3(ci + 3))));
    }
  } else if (acc_type) {  // sensor.py:1510-1539, 1678-1753
    const SensFrame f = sens_frame(m, d, w, t == SENS_ACCELEROMETER ? OBJ_SITE : ot, id);
    V3 aang, alin;
    sens_cacc(m, d, w, f.body, aang, alin);
    if (t == SENS_FRAMEANGACC) put3(aang);
    else {
      V3 lin, ang;
      sens_vel(m, d, w, f, lin, ang);
      const V3 off = f.pos - ld3(d.subtree_com + ((size_t)w * m.nbody + m.body_rootid[f.body]) * 3);
      const V3 a = alin - cross(off, aang) + cross(ang, lin);
      put3(t == SENS_ACCELEROMETER ? matT_mul(f.mat, a) : a);
    }
  }
  // sensor.py:57-114: cutoff for real (clamp) and positive (min) data
  const float cut = m.sensor_cutoff[i];
  const int dt = m.sensor_datatype[i], adr = m.sensor_adr[i], dim = m.sensor_dim[i];
  float* out = d.sensordata + (size_t)w * m.nsensordata + adr;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < dim) {
      float x = v[k];
      if (cut > 0.0f && dt == 0) x = fminf(fmaxf(x, -cut), cut);
      else if (cut > 0.0f && dt == 1) x = fminf(x, cut);
      out[k] = x;
    }
}
