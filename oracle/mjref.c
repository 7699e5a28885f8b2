/* oracle/mjref.c -- TEST INFRASTRUCTURE ONLY (pin status: mjref.h header -- externally pinned for the G1 chain since round 3, the
 * ALOHA lift behaviour the reference holds in unroll_test.py:40-58 since round 4; CG / PGS as algorithms, elliptic cones and sleep
 * policy stand-ins remain pinned through identities only).
 *
 * float64 single-world restatement of the reference's mj_step hot path.  Every function cites the
 * reference kernel(s) it follows (paths relative to /root/reference/mujoco_warp/_src/).
 * Deterministic ordering replaces the reference's atomic allocation: contacts are emitted in
 * geom-pair order, constraint rows in MuJoCo order (friction dofs, joint limits, contacts).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* float32 build of the SAME restatement (make libmjref32.so: -DREF_REAL_FLOAT): every `double` below -- storage and arithmetic -- becomes
   `float`; libm calls compute in double and round once, i.e. at least as accurately as their float versions.  It measures where the
   float32 floor of the reference's algorithm sits (tools/parity_report.py); the float64 build stays the oracle. */
#ifdef REF_REAL_FLOAT
#define double float
#endif

#include "mjref.h"

#define MINVAL 1e-15
#define MAXVAL 1e10
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define MINMU 1e-5

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { ST_SATISFIED = 0, ST_QUADRATIC = 1, ST_LINEARNEG = 2, ST_LINEARPOS = 3, ST_CONE = 4 };
enum { CT_EQUALITY = 0, CT_FRICTION_DOF = 1, CT_FRICTION_TENDON = 2, CT_LIMIT_JOINT = 3, CT_LIMIT_TENDON = 4,
       CT_CONTACT_FRICTIONLESS = 5, CT_CONTACT_PYRAMIDAL = 6, CT_CONTACT_ELLIPTIC = 7 };
enum { DSBL_CONSTRAINT = 1 << 0, DSBL_EQUALITY = 1 << 1, DSBL_FRICTIONLOSS = 1 << 2, DSBL_LIMIT = 1 << 3, DSBL_CONTACT = 1 << 4,
       DSBL_SPRING = 1 << 5, DSBL_DAMPER = 1 << 6, DSBL_GRAVITY = 1 << 7, DSBL_CLAMPCTRL = 1 << 8,
       DSBL_WARMSTART = 1 << 9, DSBL_ACTUATION = 1 << 11, DSBL_REFSAFE = 1 << 12, DSBL_EULERDAMP = 1 << 15, DSBL_NATIVECCD = 1 << 17, DSBL_MULTICCD = 1 << 19 };
enum { SOL_PGS = 0, SOL_CG = 1, SOL_NEWTON = 2 };
enum { INT_EULER = 0, INT_RK4 = 1, INT_IMPLICIT = 2, INT_IMPLICITFAST = 3 };
enum { ENBL_SLEEP = 1 << 5, DSBL_ISLAND = 1 << 18 };
enum { SLEEP_STATIC = -1, SLEEP_ASLEEP = 0, SLEEP_AWAKE = 1 };     /* SleepState (types.py:311; mjtSleepState) */
enum { POLICY_AUTO = 0, POLICY_AUTO_NEVER = 1, POLICY_AUTO_ALLOWED = 2 }; /* SleepPolicy (types.py:296) */
enum { MINAWAKE = 10 };                                              /* mjMINAWAKE (types.py:29) */
#define K_AWAKE_VAL (-(1 + MINAWAKE))                                /* sleep.py:28 */
static int sleep_enabled(const RefModel* m) { /* forward.py:345 */
  return (m->enableflags & ENBL_SLEEP) && !(m->disableflags & DSBL_ISLAND);
}
enum { OVF_NEFC = 1 << 0, OVF_NARROW = 1 << 3, OVF_ITER = 1 << 9, OVF_LS = 1 << 10 };

/* ---------------------------------------------------------------- math.py:24-333 */
static void v3set(double* r, double x, double y, double z) { r[0] = x; r[1] = y; r[2] = z; }
static void v3cpy(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static void v3sub(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void v3add(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static void v3addscl(double* r, const double* a, const double* b, double s) { r[0] = a[0] + s * b[0]; r[1] = a[1] + s * b[1]; r[2] = a[2] + s * b[2]; }
static double v3dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double v3len(const double* a) { return sqrt(v3dot(a, a)); }
static void v3cross(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double v3normalize(double* a) { /* wp.normalize: zero stays zero */
  double n = v3len(a);
  if (n > 0) { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static double safe_div(double x, double y) { return x / (y != 0.0 ? y : MINVAL); }
static double clampd(double x, double lo, double hi) { /* wp.clamp = min(max(lo, x), hi): with lo > hi the result is hi -- _efc_row clamps the impedance to [dmin, dmax] of a solimp whose dmin exceeds dmax (aloha fingers: solimp 2 1 0.01 mixed with a default geom) */
  double y = x > lo ? x : lo;
  return y < hi ? y : hi;
}

static void mul_quat(double* r, const double* u, const double* v) { /* math.py:24 */
  double w = u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3];
  double x = u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2];
  double y = u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1];
  double z = u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void rot_vec_quat(double* r, const double* v, const double* q) { /* math.py:46 */
  double s = q[0];
  const double* u = q + 1;
  double uv = v3dot(u, v), uu = v3dot(u, u), c[3];
  v3cross(c, u, v);
  for (int i = 0; i < 3; i++) r[i] = 2.0 * uv * u[i] + (s * s - uu) * v[i] + 2.0 * s * c[i];
}
static void axis_angle_to_quat(double* q, const double* axis, double angle) { /* math.py:54 */
  double s = sin(angle * 0.5), c = cos(angle * 0.5);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static void quat_normalize(double* q) { /* MuJoCo C mju_normalize4: a (near-)zero quaternion becomes the identity -- the rule the
  arithmetic of record applies to qpos in mj_kinematics and to keyframes at compile time (test_data/aloha_pot stores 0 0 0 0) */
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void quat_to_mat(double* m, const double* q) { /* math.py:61 */
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[1] = 2.0 * (q12 - q03); m[2] = 2.0 * (q13 + q02);
  m[3] = 2.0 * (q12 + q03); m[4] = q00 - q11 + q22 - q33; m[5] = 2.0 * (q23 - q01);
  m[6] = 2.0 * (q13 - q02); m[7] = 2.0 * (q23 + q01); m[8] = q00 - q11 - q22 + q33;
}
static void quat_integrate(double* q, const double* v, double dt) { /* math.py:189 */
  double vv[3] = {v[0], v[1], v[2]};
  double norm = v3normalize(vv);
  double qr[4], qn[4] = {q[0], q[1], q[2], q[3]}, out[4];
  axis_angle_to_quat(qr, vv, dt * norm);
  quat_normalize(qn);
  mul_quat(out, qn, qr);
  quat_normalize(out);
  memcpy(q, out, sizeof(out));
}
static void quat_sub(double* res, const double* qa, const double* qb) { /* math.py:176 + quat_to_vel:161 */
  double qneg[4] = {qb[0], -qb[1], -qb[2], -qb[3]}, qdif[4];
  mul_quat(qdif, qneg, qa);
  double axis[3] = {qdif[1], qdif[2], qdif[3]};
  double s = v3len(axis);
  if (s == 0.0) { v3set(res, 0, 0, 0); return; }
  double speed = 2.0 * atan2(s, qdif[0]);
  if (speed > M_PI) speed -= 2.0 * M_PI;
  for (int i = 0; i < 3; i++) res[i] = axis[i] * speed / s;
}
static void mat_mul_vec(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void matT_mul_vec(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  double y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  double z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void inert_vec(double* r, const double* i, const double* v) { /* math.py:121 */
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void motion_cross(double* r, const double* u, const double* v) { /* math.py:134 */
  double a[3], b[3], c[3];
  v3cross(a, u, v);
  v3cross(b, u + 3, v);
  v3cross(c, u, v + 3);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void motion_cross_force(double* r, const double* v, const double* f) { /* math.py:148 */
  double a[3], b[3], c[3];
  v3cross(a, v, f);
  v3cross(b, v + 3, f + 3);
  v3cross(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
static void orthogonals(const double* a, double* b, double* c) { /* math.py:203 */
  double y[3] = {0, 1, 0}, z[3] = {0, 0, 1};
  const double* s = (-0.5 < a[1] && a[1] < 0.5) ? y : z;
  double d = v3dot(a, s);
  for (int i = 0; i < 3; i++) b[i] = s[i] - a[i] * d;
  v3normalize(b);
  if (v3len(a) == 0.0) v3set(b, 0, 0, 0);
  v3cross(c, a, b);
}
static void make_frame(double* frame, const double* a_in) { /* math.py:247 */
  double a[3] = {a_in[0], a_in[1], a_in[2]}, b[3], c[3];
  v3normalize(a);
  orthogonals(a, b, c);
  v3cpy(frame, a); v3cpy(frame + 3, b); v3cpy(frame + 6, c);
}

/* ---------------------------------------------------------------- smooth.py:46-145, 148-200 */
void ref_kinematics(const RefModel* m, RefData* d) {
  v3set(d->xpos, 0, 0, 0);
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  for (int b = 1; b < m->nbody; b++) {
    int pid = m->body_parentid[b], jntadr = m->body_jntadr[b], jntnum = m->body_jntnum[b];
    double* xpos = d->xpos + 3 * b;
    double* xquat = d->xquat + 4 * b;
    if (jntnum == 1 && m->jnt_type[jntadr] == JNT_FREE) {
      int qa = m->jnt_qposadr[jntadr];
      v3cpy(xpos, d->qpos + qa);
      memcpy(xquat, d->qpos + qa + 3, 4 * sizeof(double));
      quat_normalize(xquat);
      v3cpy(d->xanchor + 3 * jntadr, xpos);
      v3cpy(d->xaxis + 3 * jntadr, m->jnt_axis + 3 * jntadr);
      continue;
    }
    double pos[3], quat[4];
    int mid = m->nmocap ? m->body_mocapid[b] : -1; /* mocap bodies are posed by Data (smooth.py:104-108) */
    const double* bp = mid >= 0 ? d->mocap_pos + 3 * mid : m->body_pos + 3 * b;
    const double* bq = mid >= 0 ? d->mocap_quat + 4 * mid : m->body_quat + 4 * b;
    rot_vec_quat(pos, bp, d->xquat + 4 * pid);
    v3add(pos, pos, d->xpos + 3 * pid);
    mul_quat(quat, d->xquat + 4 * pid, bq);
    for (int j = jntadr; j < jntadr + jntnum; j++) {
      int qa = m->jnt_qposadr[j], t = m->jnt_type[j];
      double anchor[3], axis[3], tmp[3];
      rot_vec_quat(anchor, m->jnt_pos + 3 * j, quat);
      v3add(anchor, anchor, pos);
      rot_vec_quat(axis, m->jnt_axis + 3 * j, quat);
      if (t == JNT_BALL) {
        double qloc[4] = {d->qpos[qa], d->qpos[qa + 1], d->qpos[qa + 2], d->qpos[qa + 3]}, q2[4];
        quat_normalize(qloc);
        mul_quat(q2, quat, qloc);
        memcpy(quat, q2, sizeof(q2));
        rot_vec_quat(tmp, m->jnt_pos + 3 * j, quat);
        v3sub(pos, anchor, tmp);
      } else if (t == JNT_SLIDE) {
        v3addscl(pos, pos, axis, d->qpos[qa] - m->qpos0[qa]);
      } else if (t == JNT_HINGE) {
        double qloc[4], q2[4];
        axis_angle_to_quat(qloc, m->jnt_axis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
        mul_quat(q2, quat, qloc);
        memcpy(quat, q2, sizeof(q2));
        rot_vec_quat(tmp, m->jnt_pos + 3 * j, quat);
        v3sub(pos, anchor, tmp);
      }
      v3cpy(d->xanchor + 3 * j, anchor);
      v3cpy(d->xaxis + 3 * j, axis);
    }
    quat_normalize(quat);
    v3cpy(xpos, pos);
    memcpy(xquat, quat, sizeof(quat));
  }
  for (int b = 0; b < m->nbody; b++) {
    double tmp[3], q[4];
    quat_to_mat(d->xmat + 9 * b, d->xquat + 4 * b);
    rot_vec_quat(tmp, m->body_ipos + 3 * b, d->xquat + 4 * b);
    v3add(d->xipos + 3 * b, d->xpos + 3 * b, tmp);
    mul_quat(q, d->xquat + 4 * b, m->body_iquat + 4 * b);
    quat_to_mat(d->ximat + 9 * b, q);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double tmp[3], q[4];
    rot_vec_quat(tmp, m->geom_pos + 3 * g, d->xquat + 4 * b);
    v3add(d->geom_xpos + 3 * g, d->xpos + 3 * b, tmp);
    mul_quat(q, d->xquat + 4 * b, m->geom_quat + 4 * g);
    quat_to_mat(d->geom_xmat + 9 * g, q);
  }
}

/* ---------------------------------------------------------------- smooth.py:686-822 */
void ref_com_pos(const RefModel* m, RefData* d) {
  int nb = m->nbody;
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = d->xipos[3 * b + k] * m->body_mass[b];
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int k = 0; k < 3; k++) d->subtree_com[3 * p + k] += d->subtree_com[3 * b + k];
  }
  for (int b = 0; b < nb; b++) {
    double mass = m->body_subtreemass[b];
    if (mass != 0.0)
      for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] /= mass;
  }
  for (int b = 0; b < nb; b++) { /* _cinert smooth.py:733 */
    const double* mat = d->ximat + 9 * b;
    const double* inert = m->body_inertia + 3 * b;
    double mass = m->body_mass[b], dif[3], tmp[9];
    v3sub(dif, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        tmp[3 * i + j] = mat[3 * i] * inert[0] * mat[3 * j] + mat[3 * i + 1] * inert[1] * mat[3 * j + 1] + mat[3 * i + 2] * inert[2] * mat[3 * j + 2];
    double* r = d->cinert + 10 * b;
    r[0] = tmp[0] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
    r[1] = tmp[4] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
    r[2] = tmp[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
    r[3] = tmp[1] - mass * dif[0] * dif[1];
    r[4] = tmp[2] - mass * dif[0] * dif[2];
    r[5] = tmp[5] - mass * dif[1] * dif[2];
    r[6] = mass * dif[0]; r[7] = mass * dif[1]; r[8] = mass * dif[2];
    r[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) { /* _cdof smooth.py:779 */
    int b = m->jnt_bodyid[j], dof = m->jnt_dofadr[j], t = m->jnt_type[j];
    const double* xmat = d->xmat + 9 * b;
    double off[3], c[3];
    v3sub(off, d->subtree_com + 3 * m->body_rootid[b], d->xanchor + 3 * j);
    if (t == JNT_FREE || t == JNT_BALL) {
      if (t == JNT_FREE) {
        for (int k = 0; k < 3; k++) {
          memset(d->cdof + 6 * (dof + k), 0, 6 * sizeof(double));
          d->cdof[6 * (dof + k) + 3 + k] = 1.0;
        }
        dof += 3;
      }
      for (int k = 0; k < 3; k++) {
        double ax[3] = {xmat[k], xmat[3 + k], xmat[6 + k]};
        v3cross(c, ax, off);
        v3cpy(d->cdof + 6 * (dof + k), ax);
        v3cpy(d->cdof + 6 * (dof + k) + 3, c);
      }
    } else if (t == JNT_SLIDE) {
      v3set(d->cdof + 6 * dof, 0, 0, 0);
      v3cpy(d->cdof + 6 * dof + 3, d->xaxis + 3 * j);
    } else {
      v3cross(c, d->xaxis + 3 * j, off);
      v3cpy(d->cdof + 6 * dof, d->xaxis + 3 * j);
      v3cpy(d->cdof + 6 * dof + 3, c);
    }
  }
}

/* ---------------------------------------------------------------- smooth.py:1029-1098 */
void ref_crb(const RefModel* m, RefData* d) {
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * m->nbody);
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p == 0) continue;
    for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k];
  }
  memset(d->M, 0, sizeof(double) * m->nC);
  for (int i = 0; i < m->nv; i++) {
    int adr = m->M_rowadr[i] + m->M_rownnz[i] - 1;
    double buf[6];
    d->M[adr] = m->dof_armature[i];
    inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    int j = i;
    while (j >= 0) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += d->cdof[6 * j + k] * buf[k];
      d->M[adr] += s;
      adr--;
      j = m->dof_parentid[j];
    }
  }
}

/* sparse L'DL: smooth.py:1183-1232 (_qLD_acc, _qLDiag_div), same recursion as MuJoCo's mj_factorI */
static void factor_sparse(const RefModel* m, const double* M, double* L, double* Dinv) {
  memcpy(L, M, sizeof(double) * m->nC);
  for (int k = m->nv - 1; k >= 0; k--) {
    int start = m->M_rowadr[k], diag = start + m->M_rownnz[k] - 1;
    for (int adr = diag - 1; adr >= start; adr--) {
      int i = m->M_colind[adr];
      double tmp = L[adr] / L[diag];
      int ai = m->M_rowadr[i];
      for (int j = 0; j < m->M_rownnz[i]; j++) L[ai + j] -= L[start + j] * tmp;
      L[adr] = tmp;
    }
    Dinv[k] = 1.0 / L[diag];
  }
}
static void solve_sparse(const RefModel* m, const double* L, const double* Dinv, double* x, const double* y) {
  int nv = m->nv;
  if (x != y) memcpy(x, y, sizeof(double) * nv);
  for (int k = nv - 1; k >= 0; k--) { /* x <- L^-T x */
    int start = m->M_rowadr[k], diag = start + m->M_rownnz[k] - 1;
    for (int adr = start; adr < diag; adr++) x[m->M_colind[adr]] -= L[adr] * x[k];
  }
  for (int k = 0; k < nv; k++) x[k] *= Dinv[k];
  for (int k = 0; k < nv; k++) { /* x <- L^-1 x */
    int start = m->M_rowadr[k], diag = start + m->M_rownnz[k] - 1;
    for (int adr = start; adr < diag; adr++) x[k] -= L[adr] * x[m->M_colind[adr]];
  }
}
void ref_factor_m(const RefModel* m, RefData* d) { factor_sparse(m, d->M, d->qLD, d->qLDiagInv); }
void ref_solve_m(const RefModel* m, const RefData* d, double* x, const double* y) { solve_sparse(m, d->qLD, d->qLDiagInv, x, y); }
void ref_mul_m(const RefModel* m, const RefData* d, double* res, const double* vec) { /* support.py:154 */
  for (int i = 0; i < m->nv; i++) res[i] = 0;
  for (int i = 0; i < m->nv; i++) {
    int start = m->M_rowadr[i], diag = start + m->M_rownnz[i] - 1;
    res[i] += d->M[diag] * vec[i];
    for (int adr = start; adr < diag; adr++) {
      int j = m->M_colind[adr];
      res[i] += d->M[adr] * vec[j];
      res[j] += d->M[adr] * vec[i];
    }
  }
}

/* ---------------------------------------------------------------- smooth.py:2179-2258 */
void ref_com_vel(const RefModel* m, RefData* d) {
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double cvel[6];
    memcpy(cvel, d->cvel + 6 * m->body_parentid[b], sizeof(cvel));
    int dof = m->body_dofadr[b];
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int t = m->jnt_type[j];
      if (t == JNT_FREE) {
        for (int k = 0; k < 3; k++) {
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (dof + k) + c] * d->qvel[dof + k];
          memset(d->cdof_dot + 6 * (dof + k), 0, 6 * sizeof(double));
        }
        for (int k = 3; k < 6; k++) motion_cross(d->cdof_dot + 6 * (dof + k), cvel, d->cdof + 6 * (dof + k));
        for (int k = 3; k < 6; k++)
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (dof + k) + c] * d->qvel[dof + k];
        dof += 6;
      } else if (t == JNT_BALL) {
        for (int k = 0; k < 3; k++) motion_cross(d->cdof_dot + 6 * (dof + k), cvel, d->cdof + 6 * (dof + k));
        for (int k = 0; k < 3; k++)
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (dof + k) + c] * d->qvel[dof + k];
        dof += 3;
      } else {
        motion_cross(d->cdof_dot + 6 * dof, cvel, d->cdof + 6 * dof);
        for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * dof + c] * d->qvel[dof];
        dof += 1;
      }
    }
    memcpy(d->cvel + 6 * b, cvel, sizeof(cvel));
  }
}

/* ---------------------------------------------------------------- passive.py:74-210, 275-306, 631-668 */
void ref_passive(const RefModel* m, RefData* d) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_spring[i] = d->qfrc_damper[i] = d->qfrc_gravcomp[i] = 0.0;
  int spring = !(m->disableflags & DSBL_SPRING), damper = !(m->disableflags & DSBL_DAMPER);
  for (int j = 0; j < m->njnt; j++) {
    int dof = m->jnt_dofadr[j], qa = m->jnt_qposadr[j], t = m->jnt_type[j];
    double k = m->jnt_stiffness[j];
    if (spring && k != 0.0) {
      if (t == JNT_FREE) {
        for (int c = 0; c < 3; c++) d->qfrc_spring[dof + c] = -k * (d->qpos[qa + c] - m->qpos_spring[qa + c]);
        double rot[4] = {d->qpos[qa + 3], d->qpos[qa + 4], d->qpos[qa + 5], d->qpos[qa + 6]}, dif[3];
        quat_normalize(rot);
        quat_sub(dif, rot, m->qpos_spring + qa + 3);
        for (int c = 0; c < 3; c++) d->qfrc_spring[dof + 3 + c] = -k * dif[c];
      } else if (t == JNT_BALL) {
        double rot[4] = {d->qpos[qa], d->qpos[qa + 1], d->qpos[qa + 2], d->qpos[qa + 3]}, dif[3];
        quat_normalize(rot);
        quat_sub(dif, rot, m->qpos_spring + qa);
        for (int c = 0; c < 3; c++) d->qfrc_spring[dof + c] = -k * dif[c];
      } else {
        d->qfrc_spring[dof] = -k * (d->qpos[qa] - m->qpos_spring[qa]);
      }
    }
  }
  if (damper)
    for (int i = 0; i < nv; i++) d->qfrc_damper[i] = -m->dof_damping[i] * d->qvel[i];
  /* gravity compensation passive.py:275-306 */
  if (!(m->disableflags & DSBL_GRAVITY)) {
    for (int b = 1; b < m->nbody; b++) {
      double gc = m->body_gravcomp[b];
      if (gc == 0.0) continue;
      double force[3];
      for (int c = 0; c < 3; c++) force[c] = -m->gravity[c] * m->body_mass[b] * gc;
      /* apply at xipos: J^T force */
      int bb = b;
      while (bb > 0 && m->body_dofnum[bb] == 0) bb = m->body_parentid[bb];
      if (bb == 0) continue;
      double off[3];
      v3sub(off, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
      int dof = m->body_dofadr[bb] + m->body_dofnum[bb] - 1;
      while (dof >= 0) {
        double jp[3];
        v3cross(jp, d->cdof + 6 * dof, off);
        v3add(jp, jp, d->cdof + 6 * dof + 3);
        d->qfrc_gravcomp[dof] += v3dot(jp, force);
        dof = m->dof_parentid[dof];
      }
    }
  }
  for (int i = 0; i < nv; i++) { /* passive.py:631-668: gravcomp is passive unless the joint routes it through its actuators */
    d->qfrc_passive[i] = d->qfrc_spring[i] + d->qfrc_damper[i];
    if (!(m->disableflags & DSBL_GRAVITY) && !m->jnt_actgravcomp[m->dof_jntid[i]]) d->qfrc_passive[i] += d->qfrc_gravcomp[i];
  }
}

/* ---------------------------------------------------------------- smooth.py:1353-1515 */
void ref_rne(const RefModel* m, RefData* d) {
  int nb = m->nbody;
  memset(d->cacc, 0, 6 * sizeof(double));
  if (!(m->disableflags & DSBL_GRAVITY))
    for (int c = 0; c < 3; c++) d->cacc[3 + c] = -m->gravity[c];
  for (int b = 1; b < nb; b++) {
    double cacc[6];
    memcpy(cacc, d->cacc + 6 * m->body_parentid[b], sizeof(cacc));
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int dof = m->body_dofadr[b] + k;
      for (int c = 0; c < 6; c++) cacc[c] += d->cdof_dot[6 * dof + c] * d->qvel[dof];
    }
    memcpy(d->cacc + 6 * b, cacc, sizeof(cacc));
  }
  memset(d->cfrc_int, 0, 6 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    double f1[6], iv[6], f2[6];
    inert_vec(f1, d->cinert + 10 * b, d->cacc + 6 * b);
    inert_vec(iv, d->cinert + 10 * b, d->cvel + 6 * b);
    motion_cross_force(f2, d->cvel + 6 * b, iv);
    for (int c = 0; c < 6; c++) d->cfrc_int[6 * b + c] = f1[c] + f2[c];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int c = 0; c < 6; c++) d->cfrc_int[6 * p + c] += d->cfrc_int[6 * b + c];
  }
  for (int i = 0; i < m->nv; i++) {
    double s = 0;
    for (int c = 0; c < 6; c++) s += d->cdof[6 * i + c] * d->cfrc_int[6 * m->dof_bodyid[i] + c];
    d->qfrc_bias[i] = s;
  }
}

/* ---------------------------------------------------------------- smooth.py:2288-2400 (joint branch), forward.py:680-702 */
void ref_transmission(const RefModel* m, RefData* d) {
  for (int i = 0; i < m->nu; i++) {
    int j = m->actuator_trnid[2 * i];
    double gear = m->actuator_gear[6 * i];
    d->actuator_length[i] = d->qpos[m->jnt_qposadr[j]] * gear; /* slide/hinge only */
  }
}

/* forward.py:756-1050 (NONE/INTEGRATOR/FILTER/FILTEREXACT dyn; FIXED/AFFINE gain; NONE/AFFINE bias), 1097-1149 */
void ref_fwd_actuation(const RefModel* m, RefData* d) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_actuator[i] = 0.0;
  if (m->nu == 0 || (m->disableflags & DSBL_ACTUATION)) { /* forward.py:1155-1159: no actuators or actuation disabled -- qfrc_actuator stays zero, joint-level actuator gravity compensation included */
    for (int i = 0; i < m->nu; i++) d->actuator_force[i] = 0.0;
    return;
  }
  for (int i = 0; i < m->nu; i++) {
    double ctrl = d->ctrl[i];
    if (m->actuator_ctrllimited[i] && !(m->disableflags & DSBL_CLAMPCTRL))
      ctrl = clampd(ctrl, m->actuator_ctrlrange[2 * i], m->actuator_ctrlrange[2 * i + 1]);
    double ctrl_act = ctrl;
    int dyn = m->actuator_dyntype[i];
    if (dyn != 0) {
      int adr = m->actuator_actadr[i];
      double act = d->act[adr], act_dot = 0.0;
      const double* prm = m->actuator_dynprm + 10 * i;
      if (dyn == 1) act_dot = ctrl;                                   /* INTEGRATOR */
      else if (dyn == 2 || dyn == 3) act_dot = (ctrl - act) / fmax(MINVAL, prm[0]); /* FILTER(EXACT) */
      d->act_dot[adr] = act_dot;
      ctrl_act = act;
    }
    double length = d->actuator_length[i], velocity = d->actuator_velocity[i];
    const double* gp = m->actuator_gainprm + 10 * i;
    const double* bp = m->actuator_biasprm + 10 * i;
    double gain = (m->actuator_gaintype[i] == 0) ? gp[0] : gp[0] + gp[1] * length + gp[2] * velocity;
    double bias = (m->actuator_biastype[i] == 0) ? 0.0 : bp[0] + bp[1] * length + bp[2] * velocity;
    double force = gain * ctrl_act + bias;
    if (m->actuator_forcelimited[i]) force = clampd(force, m->actuator_forcerange[2 * i], m->actuator_forcerange[2 * i + 1]);
    d->actuator_force[i] = force;
    int dof = m->jnt_dofadr[m->actuator_trnid[2 * i]];
    d->qfrc_actuator[dof] += m->actuator_gear[6 * i] * force;
  }
  for (int i = 0; i < nv; i++) { /* forward.py:1121-1150 _qfrc_actuator_gravcomp_limits */
    int j = m->dof_jntid[i];
    if (!(m->disableflags & DSBL_GRAVITY) && m->jnt_actgravcomp[j]) d->qfrc_actuator[i] += d->qfrc_gravcomp[i];
    if (m->jnt_actfrclimited[j]) d->qfrc_actuator[i] = clampd(d->qfrc_actuator[i], m->jnt_actfrcrange[2 * j], m->jnt_actfrcrange[2 * j + 1]);
  }
}

void ref_fwd_velocity(const RefModel* m, RefData* d) { /* forward.py:732-753 */
  for (int i = 0; i < m->nu; i++) {
    int dof = m->jnt_dofadr[m->actuator_trnid[2 * i]];
    d->actuator_velocity[i] = m->actuator_gear[6 * i] * d->qvel[dof];
  }
  ref_com_vel(m, d);
  ref_passive(m, d);
  ref_rne(m, d);
}

/* forward.py:1255-1324: qfrc_smooth, xfrc (support.py:259-322), qacc_smooth = M^-1 qfrc_smooth */
void ref_fwd_acceleration(const RefModel* m, RefData* d) {
  for (int i = 0; i < m->nv; i++)
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i] + d->qfrc_applied[i];
  if (sleep_enabled(m)) /* forward.py:1273-1278: no smooth force on the dofs of a sleeping tree */
    for (int i = 0; i < m->nv; i++)
      if (!d->tree_awake[m->dof_treeid[i]]) d->qfrc_smooth[i] = 0.0;
  for (int b = 1; b < m->nbody; b++) { /* xfrc_applied: (force[3], torque[3]) at xipos */
    const double* f = d->xfrc_applied + 6 * b;
    if (f[0] == 0 && f[1] == 0 && f[2] == 0 && f[3] == 0 && f[4] == 0 && f[5] == 0) continue;
    int bb = b;
    while (bb > 0 && m->body_dofnum[bb] == 0) bb = m->body_parentid[bb];
    if (bb == 0) continue;
    double off[3];
    v3sub(off, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
    int dof = m->body_dofadr[bb] + m->body_dofnum[bb] - 1;
    while (dof >= 0) {
      double jp[3];
      v3cross(jp, d->cdof + 6 * dof, off);
      v3add(jp, jp, d->cdof + 6 * dof + 3);
      d->qfrc_smooth[dof] += v3dot(jp, f) + v3dot(d->cdof + 6 * dof, f + 3);
      dof = m->dof_parentid[dof];
    }
  }
  ref_factor_m(m, d);
  ref_solve_m(m, d, d->qacc_smooth, d->qfrc_smooth);
  if (sleep_enabled(m)) /* solver.py:3899 smooth_solve_compact: M is block diagonal over trees, so the compacted solve is the full one
                           on the awake trees; sleeping dofs are frozen at 0 (_scatter_solution solver.py:3882) */
    for (int i = 0; i < m->nv; i++)
      if (!d->tree_awake[m->dof_treeid[i]]) d->qacc_smooth[i] = 0.0;
}

/* ================================================================ collision */
typedef struct { double dist; double pos[3]; double frame[9]; } Con;

/* collision_primitive_core.py:48 */
static void plane_sphere(const double* n, const double* ppos, const double* spos, double r, double* dist, double* pos) {
  double dif[3];
  v3sub(dif, spos, ppos);
  *dist = v3dot(dif, n) - r;
  v3addscl(pos, spos, n, -(r + 0.5 * (*dist)));
}
/* collision_primitive_core.py:56 */
static void sphere_sphere(const double* p1, double r1, const double* p2, double r2, double* dist, double* pos, double* n) {
  double dir[3];
  v3sub(dir, p2, p1);
  double dd = v3len(dir);
  if (dd == 0.0) v3set(n, 1, 0, 0);
  else { n[0] = dir[0] / dd; n[1] = dir[1] / dd; n[2] = dir[2] / dd; }
  *dist = dd - (r1 + r2);
  v3addscl(pos, p1, n, r1 + 0.5 * (*dist));
}
static void closest_segment_point(double* r, const double* a, const double* b, const double* pt) { /* math.py:270 */
  double ab[3], pa[3];
  v3sub(ab, b, a);
  v3sub(pa, pt, a);
  double t = v3dot(pa, ab) / (v3dot(ab, ab) + 1e-6);
  v3addscl(r, a, ab, clampd(t, 0.0, 1.0));
}

/* sphere_box collision_primitive_core.py:1044: contact of a sphere (centre sp, radius r) with a box (pos, R, half sizes) */
static void sphere_box(const double* sp, double r, const double* bp, const double* R, const double* bs, Con* out) {
  double dif[3], center[3], clamped[3], cdir[3], pos[3], nn[3];
  v3sub(dif, sp, bp);
  matT_mul_vec(center, R, dif);
  for (int k = 0; k < 3; k++) clamped[k] = fmax(-bs[k], fmin(bs[k], center[k]));
  v3sub(cdir, clamped, center);
  double dist = v3normalize(cdir);
  if (dist <= MINVAL) { /* centre inside the box: push out through the nearest face */
    double closest = 2.0 * (bs[0] + bs[1] + bs[2]);
    int kk = 0;
    for (int i = 0; i < 6; i++) {
      double fd = fabs(((i % 2) ? 1.0 : -1.0) * bs[i / 2] - center[i / 2]);
      if (closest > fd) { closest = fd; kk = i; }
    }
    double nearest[3] = {0, 0, 0};
    nearest[kk / 2] = (kk % 2) ? -1.0 : 1.0;
    for (int k = 0; k < 3; k++) pos[k] = center[k] + nearest[k] * (r - closest) / 2.0;
    mat_mul_vec(nn, R, nearest);
    out->dist = -closest - r;
  } else {
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (clamped[k] + center[k] + cdir[k] * r);
    mat_mul_vec(nn, R, cdir);
    out->dist = dist - r;
  }
  mat_mul_vec(out->pos, R, pos);
  v3add(out->pos, out->pos, bp);
  make_frame(out->frame, nn);
}

/* ---- small math.py functions that carry reference-held test vectors (math_test.py:27-130) ------------------------------ */
/* closest_segment_to_segment_points math.py:283-316 (not on the mj_step path of the primitive colliders restated here -- the
 * reference uses it for triangle / flex queries, collision_primitive_core.py:1939 -- restated so that the vectors the
 * reference's own test holds pin this file's vector conventions) */
void ref_closest_segment_to_segment_points(const double* a0, const double* a1, const double* b0, const double* b1, double* best_a_out,
                                           double* best_b_out) {
  double da[3], db[3], la = 0, lb = 0;
  for (int k = 0; k < 3; k++) { da[k] = a1[k] - a0[k]; db[k] = b1[k] - b0[k]; la += da[k] * da[k]; lb += db[k] * db[k]; }
  la = sqrt(la); lb = sqrt(lb);
  if (la != 0.0) for (int k = 0; k < 3; k++) da[k] /= la;  /* normalize_with_norm math.py:258: zero stays zero */
  if (lb != 0.0) for (int k = 0; k < 3; k++) db[k] /= lb;
  double ha = 0.5 * la, hb = 0.5 * lb, am[3], bm[3], tr[3];
  for (int k = 0; k < 3; k++) { am[k] = a0[k] + da[k] * ha; bm[k] = b0[k] + db[k] * hb; tr[k] = am[k] - bm[k]; }
  double dab = v3dot(da, db), dat = v3dot(da, tr), dbt = v3dot(db, tr), denom = 1.0 - dab * dab;
  double ota = (-dat + dab * dbt) / (denom + 1e-6), otb = dbt + ota * dab;
  double ta = clampd(ota, -ha, ha), tb = clampd(otb, -hb, hb);
  double ba[3], bb[3], na[3], nb[3];
  for (int k = 0; k < 3; k++) { ba[k] = am[k] + da[k] * ta; bb[k] = bm[k] + db[k] * tb; }
  closest_segment_point(na, a0, a1, bb);
  closest_segment_point(nb, b0, b1, ba);
  double d1 = 0, d2 = 0;
  for (int k = 0; k < 3; k++) { d1 += (bb[k] - na[k]) * (bb[k] - na[k]); d2 += (ba[k] - nb[k]) * (ba[k] - nb[k]); }
  if (d1 < d2) { v3cpy(best_a_out, na); v3cpy(best_b_out, bb); }
  else { v3cpy(best_a_out, ba); v3cpy(best_b_out, nb); }
}
/* math.py:323-333: index of a_ij = a_ji in a packed upper triangle without / with the diagonal */
int ref_upper_tri_index(int n, int i, int j) { return (i * (2 * n - i - 3)) / 2 + j - 1; }
int ref_upper_trid_index(int n, int i, int j) {
  if (j < i) { int t = i; i = j; j = t; }
  return (i * (2 * n - i - 1)) / 2 + j;
}

/* capsule_box collision_primitive_core.py:1099 (MuJoCo's mjc_CapsuleBox): the point of the capsule segment closest to the box
 * gives the first contact sphere; when the capsule lies along a face or an edge a second sphere is placed further along the
 * segment (as far as the capsule still is above the box).  Segment parameter t in [-1, 1]: point = pos + t * halfaxis. */
static int capsule_box(const double* cpos, const double* caxis, double r, double hl, const double* bpos, const double* R,
                       const double* bs, Con* out) {
  double dif0[3], pos[3], axis[3], ha[3];
  v3sub(dif0, cpos, bpos);
  matT_mul_vec(pos, R, dif0);
  matT_mul_vec(axis, R, caxis);
  for (int k = 0; k < 3; k++) ha[k] = axis[k] * hl;
  int axisdir = (ha[0] > 0.0) + 2 * (ha[1] > 0.0) + 4 * (ha[2] > 0.0);
  double bestdist = 1e32, bestseg = -12.0, bestbox = 0.0, second = -4.0;
  int cltype = -4, clface = -12, clcorner = -123, cledge = -123;
  /* (1) a capsule end over a face (or inside): clamped in at most one coordinate */
  for (int sgn = -1; sgn <= 1; sgn += 2) {
    double tip[3], d2 = 0.0;
    int nout = 0, axout = -1;
    for (int k = 0; k < 3; k++) {
      tip[k] = pos[k] + sgn * ha[k];
      double c = tip[k];
      if (c < -bs[k]) { nout++; axout = k; c = -bs[k]; }
      else if (c > bs[k]) { nout++; axout = k; c = bs[k]; }
      d2 += (c - tip[k]) * (c - tip[k]);
    }
    if (nout > 1) continue;
    if (d2 < bestdist) { bestdist = d2; bestseg = sgn; cltype = -2 + sgn; clface = axout; }
  }
  /* (2) the segment against each of the 12 box edges (corner i, edge direction j with bit j of i clear) */
  for (int i = 0; i < 8; i++) {
    for (int j = 0; j < 3; j++) {
      if (i & (1 << j)) continue;
      double bp[3], dif[3];
      for (int k = 0; k < 3; k++) bp[k] = ((i >> k) & 1 ? 1.0 : -1.0) * bs[k];
      bp[j] = 0.0;
      v3sub(dif, bp, pos);
      double u = -bs[j] * dif[j], v = v3dot(ha, dif);
      double ma = bs[j] * bs[j], mb = -bs[j] * ha[j], mc = hl * hl;
      double det = ma * mc - mb * mb;
      if (fabs(det) < MINVAL) continue;
      double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det; /* x1 along the edge, x2 along the capsule */
      int s1 = 1, s2 = 1;                                                /* 1: interior, 0 / 2: lower / upper end */
      if (x1 > 1.0) { x1 = 1.0; s1 = 2; x2 = safe_div(v - mb, mc); }
      else if (x1 < -1.0) { x1 = -1.0; s1 = 0; x2 = safe_div(v + mb, mc); }
      if (x2 > 1.0 || x2 < -1.0) {
        if (x2 > 1.0) { x2 = 1.0; s2 = 2; x1 = safe_div(u - mb, ma); }
        else { x2 = -1.0; s2 = 0; x1 = safe_div(u + mb, ma); }
        if (x1 > 1.0) { x1 = 1.0; s1 = 2; }
        else if (x1 < -1.0) { x1 = -1.0; s1 = 0; }
      }
      for (int k = 0; k < 3; k++) dif[k] -= ha[k] * x2;
      dif[j] += bs[j] * x1;
      double d2 = v3dot(dif, dif);
      if (d2 < bestdist - MINVAL) {
        int ct = s1 * 3 + s2;
        bestdist = d2; bestseg = x2; bestbox = x1;
        clcorner = i + (1 << j) * (ct / 6); /* the box corner nearest to the closest point */
        cledge = j;
        cltype = ct;
      }
    }
  }
  if (cltype == -4) return 0;
  /* (3) a second contact further along the capsule */
  if (cltype >= 0 && cltype / 3 != 1) { /* closest box point is a corner */
    int c1 = axisdir ^ clcorner;
    if (c1 != 0 && c1 != 7) { /* the capsule does not point straight at / away from the corner */
      int mul = 1;
      if (!(c1 == 1 || c1 == 2 || c1 == 4)) { mul = -1; c1 = 7 - c1; }
      int ax = c1 == 1 ? 0 : (c1 == 2 ? 1 : 2), ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (axis[ax] * axis[ax] > 0.5) { /* along the edge */
        double mlim = 2.0 * safe_div(bs[ax], fabs(ha[ax]));
        second = fmin(1.0 - mul * bestseg, mlim);
      } else { /* across a face */
        double mlim = 2.0 * fmin(safe_div(bs[ax1], fabs(ha[ax1])), safe_div(bs[ax2], fabs(ha[ax2])));
        second = -fmin(1.0 + mul * bestseg, mlim);
      }
      second *= mul;
    }
  } else if (cltype >= 0) { /* closest box point is inside an edge */
    int c1 = (axisdir ^ clcorner) & (7 - (1 << cledge));
    if (c1 == 1 || c1 == 2 || c1 == 4) { /* X configuration (a T configuration has no second contact) */
      int ax = cledge, ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (fabs(axis[ax1]) > fabs(axis[ax2])) ax1 = ax2; /* the face the capsule makes the smaller angle with */
      ax2 = 3 - ax - ax1;
      int mul;
      if (c1 & (1 << ax2)) { mul = 1; second = 1.0 - bestseg; }
      else { mul = -1; second = 1.0 + bestseg; }
      second = fmin(2.0 * safe_div(bs[ax2], fabs(ha[ax2])), second);
      double e2 = (((axisdir & (1 << ax)) != 0) == ((c1 & (1 << ax2)) != 0)) ? 1.0 - bestbox : 1.0 + bestbox;
      second = fmin(bs[ax] * safe_div(e2, fabs(ha[ax])), second);
      second *= mul;
    }
  } else if (clface != -1) { /* an end is closest to a face: follow the capsule until it leaves the box footprint */
    int mul = cltype == -3 ? 1 : -1;
    second = 2.0;
    for (int k = 0; k < 3; k++) {
      if (k == clface) continue;
      double tmp = pos[k] - ha[k] * mul, har = safe_div((double)mul, ha[k]);
      double e1 = (bs[k] - tmp) * har;
      if (0.0 < e1 && e1 < second) second = e1;
      e1 = (-bs[k] - tmp) * har;
      if (0.0 < e1 && e1 < second) second = e1;
    }
    second *= mul;
  }
  /* contact spheres at the chosen segment points, in world coordinates */
  int n = 0;
  for (int c = 0; c < 2; c++) {
    if (c == 1 && !(second > -3.0)) break;
    double t = c == 0 ? bestseg : second + bestseg, loc[3], sp[3];
    for (int k = 0; k < 3; k++) loc[k] = pos[k] + ha[k] * t;
    mat_mul_vec(sp, R, loc);
    v3add(sp, sp, bpos);
    sphere_box(sp, r, bpos, R, bs, &out[n]);
    n++;
  }
  return n;
}

/* rotation that takes face `f` (0..2: +x,+y,+z side; 3..5: -x,-y,-z) of a box to the +z direction (core:557) */
static void face_rot(int f, double* r) {
  static const double T[6][9] = {{0, 0, -1, 0, 1, 0, 1, 0, 0}, {1, 0, 0, 0, 0, -1, 0, 1, 0}, {1, 0, 0, 0, 1, 0, 0, 0, 1},
                                 {0, 0, 1, 0, 1, 0, -1, 0, 0}, {1, 0, 0, 0, 0, 1, 0, -1, 0}, {-1, 0, 0, 0, 1, 0, 0, 0, -1}};
  memcpy(r, T[f], sizeof(T[f]));
}
static void m3mul(double* r, const double* a, const double* b) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
static void m3T(double* r, const double* a) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r[3 * i + j] = a[3 * j + i];
}

/* box_box collision_primitive_core.py:589 (MuJoCo's mjc_BoxBox): separating-axis test over the 6 face normals and the 9
 * edge cross products; the axis of least penetration selects a face-vertex clipping (up to 8 contacts on the face plane) or
 * an edge-edge configuration.  Returns the number of contacts written to out (dist, pos, frame). */
static int box_box(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                   double margin, Con* out) {
  double d21[3], d12[3], pos21[3], pos12[3], R1T[9], R2T[9], rot21[9], rot12[9], a21[9], a12[9], plen1[3], plen2[3];
  v3sub(d21, p2, p1);
  v3sub(d12, p1, p2);
  matT_mul_vec(pos21, R1, d21);
  matT_mul_vec(pos12, R2, d12);
  m3T(R1T, R1);
  m3T(R2T, R2);
  m3mul(rot21, R1T, R2);
  m3T(rot12, rot21);
  for (int k = 0; k < 9; k++) { a21[k] = fabs(rot21[k]); a12[k] = fabs(rot12[k]); }
  mat_mul_vec(plen2, a21, s2);
  mat_mul_vec(plen1, a12, s1);
  double separation = margin + 3.0 * (s1[0] + s2[0]) + 3.0 * (s1[1] + s2[1]) + 3.0 * (s1[2] + s2[2]);
  int code = -1;
  const double tie = 1e-12 * (s1[0] + s1[1] + s1[2] + s2[0] + s2[1] + s2[2]);
  for (int i = 0; i < 3; i++) { /* face normals */
    double c1 = -fabs(pos21[i]) + s1[i] + plen2[i], c2 = -fabs(pos12[i]) + s2[i] + plen1[i];
    if (c1 < -margin || c2 < -margin) return 0;
    /* ties (flat face-face contact: the two boxes' faces overlap by the same amount) go to the earlier candidate instead
     * of to round-off; same rule, with a float32-sized guard, in csrc/collide.hpp */
    if (c1 < separation - tie) { separation = c1; code = i + 3 * (pos21[i] < 0.0); }
    if (c2 < separation - tie) { separation = c2; code = i + 3 * (pos12[i] < 0.0) + 6; }
  }
  double clnorm[3] = {0, 0, 0};
  int inv = 0, cle1 = 0, cle2 = 0;
  for (int i = 0; i < 3; i++) { /* edge i of box 1 x edge j of box 2 */
    for (int j = 0; j < 3; j++) {
      const double* a = rot12 + 3 * j; /* axis j of box 2 in the frame of box 1 */
      double cr[3];
      if (i == 0) v3set(cr, 0.0, -a[2], a[1]);
      else if (i == 1) v3set(cr, a[2], 0.0, -a[0]);
      else v3set(cr, -a[1], a[0], 0.0);
      double len = v3len(cr);
      if (len < MINVAL) continue;
      for (int k = 0; k < 3; k++) cr[k] /= len;
      double bd = v3dot(pos21, cr), c3 = 0.0;
      for (int k = 0; k < 3; k++) {
        if (k != i) c3 += s1[k] * fabs(cr[k]);
        if (k != j) c3 += s2[k] * a21[3 * i + (3 - k - j)] / len;
      }
      c3 -= fabs(bd);
      if (c3 < -margin) return 0;
      if (c3 < separation * (1.0 - 1e-12)) {
        separation = c3;
        cle1 = cle2 = 0;
        for (int k = 0; k < 3; k++) {
          if (k != i && ((cr[k] > 0.0) ^ (bd < 0.0))) cle1 += 1 << k;
          if (k != j && ((rot21[3 * i + (3 - k - j)] > 0.0) ^ (bd < 0.0) ^ (((k - j + 3) % 3) == 1))) cle2 += 1 << k;
        }
        code = 12 + 3 * i + j;
        v3cpy(clnorm, cr);
        inv = bd < 0.0;
      }
    }
  }
  if (code == -1) return 0;
  double pts[8][3], depth[8], rw[9], pw[3], normal[3], hz;
  int n = 0;
  if (code < 12) { /* a face of one box against vertices / edges of the other */
    int f = code % 6, bi = code / 6;
    double rm[9], rmT[9], r[9], rt[9], pp[3], ss[3], tmp[3];
    face_rot(f, rm);
    m3T(rmT, rm);
    m3mul(r, rm, bi ? rot12 : rot21);
    mat_mul_vec(pp, rm, bi ? pos12 : pos21);
    mat_mul_vec(tmp, rm, bi ? s2 : s1);
    for (int k = 0; k < 3; k++) ss[k] = fabs(tmp[k]);
    const double* so = bi ? s1 : s2; /* half sizes of the other box */
    m3T(rt, r);
    double lx = ss[0], ly = ss[1];
    hz = ss[2];
    pp[2] -= hz;
    int corner = 0;
    for (int i = 0; i < 3; i++)
      if (r[6 + i] < 0.0) corner += 1 << i;
    double lp[3], cn1[3] = {0, 0, 0}, cn2[3] = {0, 0, 0};
    v3cpy(lp, pp);
    for (int i = 0; i < 3; i++) v3addscl(lp, lp, rt + 3 * i, so[i] * ((corner >> i) & 1 ? 1.0 : -1.0));
    int dirs = 0;
    for (int i = 0; i < 3; i++) {
      if (fabs(r[6 + i]) < 0.5) {
        double sc = so[i] * ((corner >> i) & 1 ? -2.0 : 2.0);
        double* cn = dirs ? cn2 : cn1;
        for (int k = 0; k < 3; k++) cn[k] = rt[3 * i + k] * sc;
        dirs++;
      }
    }
    double cand[24][3];
    int nc = 0;
    for (int i = 0; i < dirs * dirs; i++) { /* edges of the other box's lowest face against the face rectangle */
      for (int q = 0; q < 2; q++) {
        double lav[3], lbv[3];
        for (int k = 0; k < 3; k++) {
          lav[k] = lp[k] + (i < 2 ? 0.0 : (i == 2 ? cn1[k] : cn2[k]));
          lbv[k] = (i == 0 || i == 3) ? cn1[k] : cn2[k];
        }
        if (fabs(lbv[q]) > MINVAL) {
          double br = 1.0 / lbv[q];
          for (int j = -1; j <= 1; j += 2) {
            double l = ss[q] * j, c1 = (l - lav[q]) * br;
            if (c1 < 0.0 || c1 > 1.0) continue;
            double c2 = lav[1 - q] + lbv[1 - q] * c1;
            if (fabs(c2) > ss[1 - q]) continue;
            for (int k = 0; k < 3; k++) cand[nc][k] = lav[k] + c1 * lbv[k];
            nc++;
          }
        }
      }
    }
    if (dirs == 2) { /* rectangle corners inside the other face's parallelogram */
      double ax = cn1[0], bx = cn2[0], ay = cn1[1], by = cn2[1], C = safe_div(1.0, ax * by - bx * ay);
      for (int i = 0; i < 4; i++) {
        double llx = (i / 2) ? lx : -lx, lly = (i % 2) ? ly : -ly, x = llx - lp[0], y = lly - lp[1];
        double u = (x * by - y * bx) * C, v = (y * ax - x * ay) * C;
        if (u > 0.0 && v > 0.0 && u < 1.0 && v < 1.0) {
          v3set(cand[nc], llx, lly, lp[2] + u * cn1[2] + v * cn2[2]);
          nc++;
        }
      }
    }
    for (int i = 0; i < (1 << dirs); i++) { /* the other box's corners above the rectangle */
      double t[3];
      for (int k = 0; k < 3; k++) t[k] = lp[k] + (double)(i & 1) * cn1[k] + (double)((i & 2) != 0) * cn2[k];
      if (t[0] > -lx && t[0] < lx && t[1] > -ly && t[1] < ly) { v3cpy(cand[nc], t); nc++; }
    }
    for (int i = 0; i < nc && n < 8; i++) {
      if (cand[i][2] > margin) continue;
      v3cpy(pts[n], cand[i]);
      depth[n] = pts[n][2];
      pts[n][2] *= 0.5;
      n++;
    }
    m3mul(rw, bi ? R2 : R1, rmT);
    v3cpy(pw, bi ? p2 : p1);
    for (int k = 0; k < 3; k++) normal[k] = (bi ? -1.0 : 1.0) * rw[3 * k + 2];
  } else { /* edge against edge */
    int e1 = (code - 12) / 3, e2 = (code - 12) % 3;
    int ax1 = 1 - (e2 & 1), ax2 = 2 - (e2 & 2), pax1 = 1 - (e1 & 1), pax2 = 2 - (e1 & 2);
    if (a21[3 * e1 + ax1] < a21[3 * e1 + ax2]) { int t = ax1; ax1 = ax2; ax2 = t; }
    if (a12[3 * e2 + pax1] < a12[3 * e2 + pax2]) { int t = pax1; pax1 = pax2; pax2 = t; }
    double rm[9], rmT[9], pp[3], rnorm[3], r[9], rt[9], tmp[3], s[3];
    face_rot((cle1 & (1 << pax2)) ? pax2 : pax2 + 3, rm);
    m3T(rmT, rm);
    mat_mul_vec(pp, rm, pos21);
    mat_mul_vec(rnorm, rm, clnorm);
    m3mul(r, rm, rot21);
    m3T(rt, r);
    mat_mul_vec(tmp, rmT, s1);
    for (int k = 0; k < 3; k++) s[k] = fabs(tmp[k]);
    double lx = s[0], ly = s[1];
    hz = s[2];
    pp[2] -= hz;
    double q4[4][3]; /* the face of box 2 nearest to box 1, as 4 points */
    for (int k = 0; k < 3; k++) {
      double b1 = rt[3 * ax1 + k] * s2[ax1], b2 = rt[3 * ax2 + k] * s2[ax2], be = rt[3 * e2 + k] * s2[e2];
      double sg1 = (cle2 & (1 << ax1)) ? 1.0 : -1.0, sg2 = (cle2 & (1 << ax2)) ? 1.0 : -1.0;
      double base0 = pp[k] + b1 * sg1 + b2 * sg2, base2 = pp[k] - b1 * sg1 + b2 * sg2;
      q4[0][k] = base0 + be; q4[1][k] = base0 - be;
      q4[2][k] = base2 + be; q4[3][k] = base2 - be;
    }
    double axi_lp[3], axi_cn1[3], axi_cn2[3];
    v3cpy(axi_lp, q4[0]);
    v3sub(axi_cn1, q4[1], q4[0]);
    v3sub(axi_cn2, q4[2], q4[0]);
    if (fabs(rnorm[2]) < MINVAL) return 0;
    double sgn = inv ? -1.0 : 1.0, innorm = sgn / rnorm[2];
    double pu[4][3];
    for (int i = 0; i < 4; i++) { /* project along the contact normal onto the plane z = 0 */
      v3cpy(pu[i], q4[i]);
      double c = q4[i][2] * sgn * innorm;
      for (int k = 0; k < 3; k++) q4[i][k] -= rnorm[k] * c;
    }
    double lp[3], cn1[3], cn2[3];
    v3cpy(lp, q4[0]);
    v3sub(cn1, q4[1], q4[0]);
    v3sub(cn2, q4[2], q4[0]);
    for (int i = 0; i < 4; i++) {
      for (int q = 0; q < 2; q++) {
        double la = lp[q] + (i < 2 ? 0.0 : (i == 2 ? cn1[q] : cn2[q])), lb = (i == 0 || i == 3) ? cn1[q] : cn2[q];
        double lc = lp[1 - q] + (i < 2 ? 0.0 : (i == 2 ? cn1[1 - q] : cn2[1 - q])), ld = (i == 0 || i == 3) ? cn1[1 - q] : cn2[1 - q];
        double lua[3], lub[3];
        for (int k = 0; k < 3; k++) {
          lua[k] = axi_lp[k] + (i < 2 ? 0.0 : (i == 2 ? axi_cn1[k] : axi_cn2[k]));
          lub[k] = (i == 0 || i == 3) ? axi_cn1[k] : axi_cn2[k];
        }
        if (fabs(lb) > MINVAL) {
          double br = 1.0 / lb;
          for (int j = -1; j <= 1; j += 2) {
            if (n == 8) break;
            double l = s[q] * j, c1 = (l - la) * br;
            if (c1 < 0.0 || c1 > 1.0) continue;
            double c2 = lc + ld * c1;
            if (fabs(c2) > s[1 - q]) continue;
            if ((lua[2] + lub[2] * c1) * innorm > margin) continue;
            for (int k = 0; k < 3; k++) pts[n][k] = lua[k] * 0.5 + c1 * lub[k] * 0.5;
            pts[n][q] += 0.5 * l;
            pts[n][1 - q] += 0.5 * c2;
            depth[n] = pts[n][2] * innorm * 2.0;
            n++;
          }
        }
      }
    }
    int nl = n;
    double ax = cn1[0], bx = cn2[0], ay = cn1[1], by = cn2[1], C = safe_div(1.0, ax * by - bx * ay);
    for (int i = 0; i < 4; i++) {
      if (n == 8) break;
      double llx = (i / 2) ? lx : -lx, lly = (i % 2) ? ly : -ly, x = llx - lp[0], y = lly - lp[1];
      double u = (x * by - y * bx) * C, v = (y * ax - x * ay) * C;
      if (nl == 0) {
        if ((u < 0.0 || u > 1.0) && (v < 0.0 || v > 1.0)) continue;
      } else if (u < 0.0 || v < 0.0 || u > 1.0 || v > 1.0) continue;
      u = clampd(u, 0.0, 1.0);
      v = clampd(v, 0.0, 1.0);
      double w = 1.0 - u - v, vt[3], dd[3];
      for (int k = 0; k < 3; k++) vt[k] = pu[0][k] * w + pu[1][k] * u + pu[2][k] * v;
      v3set(pts[n], llx, lly, 0.0);
      v3sub(dd, pts[n], vt);
      double tc1 = v3dot(dd, dd);
      if (vt[2] > 0.0 && tc1 > margin * margin) continue;
      for (int k = 0; k < 3; k++) pts[n][k] = 0.5 * (pts[n][k] + vt[k]);
      depth[n] = sqrt(tc1) * (vt[2] < 0.0 ? -1.0 : 1.0);
      n++;
    }
    int nf = n;
    for (int i = 0; i < 4; i++) {
      if (n >= 8) break;
      double x = pu[i][0], y = pu[i][1];
      if (nl == 0 && nf != 0) {
        if ((x < -lx || x > lx) && (y < -ly || y > ly)) continue;
      } else if (x < -lx || x > lx || y < -ly || y > ly) continue;
      double c1 = 0.0;
      for (int j = 0; j < 2; j++) {
        if (pu[i][j] < -s[j]) c1 += (pu[i][j] + s[j]) * (pu[i][j] + s[j]);
        else if (pu[i][j] > s[j]) c1 += (pu[i][j] - s[j]) * (pu[i][j] - s[j]);
      }
      c1 += pu[i][2] * innorm * pu[i][2] * innorm;
      if (pu[i][2] > 0.0 && c1 > margin * margin) continue;
      double tp[3] = {pu[i][0], pu[i][1], 0.0};
      for (int j = 0; j < 2; j++) {
        if (pu[i][j] < -s[j]) tp[j] = -s[j] * 0.5;
        else if (pu[i][j] > s[j]) tp[j] = s[j] * 0.5;
      }
      for (int k = 0; k < 3; k++) pts[n][k] = 0.5 * (tp[k] + pu[i][k]);
      depth[n] = sqrt(c1) * (pu[i][2] < 0.0 ? -1.0 : 1.0);
      n++;
    }
    m3mul(rw, R1, rmT);
    v3cpy(pw, p1);
    double nn[3];
    mat_mul_vec(nn, rw, rnorm);
    for (int k = 0; k < 3; k++) normal[k] = sgn * nn[k];
  }
  for (int i = 0; i < n; i++) {
    pts[i][2] += hz;
    mat_mul_vec(out[i].pos, rw, pts[i]);
    v3add(out[i].pos, out[i].pos, pw);
    out[i].dist = depth[i];
    make_frame(out[i].frame, normal);
  }
  return n;
}

#include "ccd.c"

/* ---- branch trace of the convex narrowphase (test infrastructure: which gate of gjk_phase every convex pair of a collision pass left through) ---- */
#define CCD_TRACE_CAP 4096
typedef struct CcdTrace { int g1, g2, branch, dim, sep, n; double gjk_dist, dist; } CcdTrace;
static CcdTrace g_trace[CCD_TRACE_CAP];
static int g_trace_on = 0, g_trace_n = 0;
void ref_ccd_trace_start(void) { g_trace_on = 1; g_trace_n = 0; }
/* ints: [n, 6] = g1 g2 branch dim separated contacts; reals: [n, 2] = GJK distance, final distance; returns the number of records */
int ref_ccd_trace_get(int* ints, double* reals, int cap) {
  int n = g_trace_n < cap ? g_trace_n : cap;
  for (int i = 0; i < n; i++) {
    ints[6 * i] = g_trace[i].g1; ints[6 * i + 1] = g_trace[i].g2; ints[6 * i + 2] = g_trace[i].branch;
    ints[6 * i + 3] = g_trace[i].dim; ints[6 * i + 4] = g_trace[i].sep; ints[6 * i + 5] = g_trace[i].n;
    reals[2 * i] = g_trace[i].gjk_dist; reals[2 * i + 1] = g_trace[i].dist;
  }
  g_trace_on = 0;
  return n;
}

/* contact of a convex pair through GJK / EPA (collision_convex.py:747-977 eval_ccd_write_contact): both geoms carry the pair's
 * margin (support points are inflated by half of it), the GJK cutoff is the gap, and the distance is reported un-inflated */
static void mesh_of(const RefModel* m, int g, const double** vert, int* nvert) {
  *vert = NULL;
  *nvert = 0;
  if (g >= 0 && m->geom_type[g] == G_MESH) {
    int id = m->geom_dataid[g];
    *vert = m->mesh_vert + 3 * m->mesh_vertadr[id];
    *nvert = m->mesh_vertnum[id];
  }
}
static const int* mesh_graph_of(const RefModel* m, int g) { /* the mesh's block of mesh_graph, NULL without one */
  if (m && g >= 0 && m->geom_type[g] == G_MESH && m->mesh_graphadr[m->geom_dataid[g]] >= 0) return m->mesh_graph + m->mesh_graphadr[m->geom_dataid[g]];
  return NULL;
}
static void mesh_poly_of(const RefModel* m, int g, CcdGeom* c) { /* polygon tables, offset to the geom's mesh */
  c->polynormal = NULL;
  c->polyvertadr = c->polyvertnum = c->polyvert = c->polymapadr = c->polymapnum = c->polymap = NULL;
  if (g >= 0 && m->geom_type[g] == G_MESH && m->nmeshpoly > 0) {
    int id = m->geom_dataid[g], pa = m->mesh_polyadr[id], va = m->mesh_vertadr[id];
    c->polynormal = m->mesh_polynormal + 3 * pa;
    c->polyvertadr = m->mesh_polyvertadr + pa;
    c->polyvertnum = m->mesh_polyvertnum + pa;
    c->polyvert = m->mesh_polyvert;
    c->polymapadr = m->mesh_polymapadr + va;
    c->polymapnum = m->mesh_polymapnum + va;
    c->polymap = m->mesh_polymap;
  }
}
/* multi-contact recovery applies to (collision_convex.py:875-889): box-box always; box-mesh / mesh-mesh unless DisableBit.MULTICCD;
   meshes need their polygon tables */
static int multiccd_pair(const RefModel* m, const CcdGeom* a, const CcdGeom* b) {
  int bb = a->type == G_BOX && b->type == G_BOX;
  if (!bb && (m->disableflags & DSBL_MULTICCD)) return 0;
  if ((a->type != G_BOX && a->type != G_MESH) || (b->type != G_BOX && b->type != G_MESH)) return 0;
  if ((a->type == G_MESH && !a->polynormal) || (b->type == G_MESH && !b->polynormal)) return 0;
  return 1;
}
static int ccd_contact(const RefModel* m, int g1, int g2, int t1, const double* p1, const double* R1, const double* s1, int t2, const double* p2,
                       const double* R2, const double* s2, double margin, double gap, Con* out, int* overflow) {
  CcdGeom a, b;
  a.type = t1; b.type = t2;
  mesh_of(m, g1, &a.vert, &a.nvert);
  mesh_of(m, g2, &b.vert, &b.nvert);
  mesh_poly_of(m, g1, &a);
  mesh_poly_of(m, g2, &b);
  g_mc_cap = 2 * m->npolygonmax;
  a.index = b.index = a.cache = b.cache = -1;
  a.graph = mesh_graph_of(m, g1);
  b.graph = mesh_graph_of(m, g2);
  v3cpy(a.pos, p1); v3cpy(b.pos, p2);
  memcpy(a.rot, R1, sizeof(a.rot)); memcpy(b.rot, R2, sizeof(b.rot));
  v3cpy(a.size, s1); v3cpy(b.size, s2);
  a.margin = b.margin = margin;
  static Polytope pt; /* (the oracle is single threaded) */
  double dist, w1[4][3], w2[4][3], nrm[3];
  int face;
  int n = ccd_run(m->ccd_tolerance, gap, m->ccd_iterations, m->epa_iterations, a, b, &dist, w1[0], w2[0], overflow, &face, &pt);
  if (g_trace_on && g_trace_n < CCD_TRACE_CAP) { /* (test infrastructure: tools/ccd_branch_report.py) */
    CcdTrace* t = &g_trace[g_trace_n++];
    t->g1 = g1; t->g2 = g2; t->branch = g_ccd_branch; t->dim = g_ccd_gjk_dim; t->sep = g_ccd_gjk_sep; t->n = n;
    t->gjk_dist = g_ccd_gjk_dist; t->dist = dist;
  }
  if (n == 0 || dist >= gap) return 0;
  dist += margin;
  if (face >= 0 && multiccd_pair(m, &a, &b)) { /* zero margin: recover up to four contacts from the EPA face (collision_convex.py:875-917) */
    double x1[3], x2[3];
    v3cpy(x1, w1[0]);
    v3cpy(x2, w2[0]);
    n = ccd_multicontact_box(&pt, face, x1, x2, &a, &b, w1, w2);
    if (n == 0) return 0;
  }
  if (dist <= margin) v3sub(nrm, w1[0], w2[0]); /* overlapping: witness 1 has crossed witness 2 */
  else v3sub(nrm, w2[0], w1[0]);
  for (int i = 0; i < n; i++) {
    out[i].dist = dist;
    for (int k = 0; k < 3; k++) out[i].pos[k] = 0.5 * (w1[i][k] + w2[i][k]);
    make_frame(out[i].frame, nrm);
  }
  return n;
}
/* height field against a convex geom (collision_convex.py:60-161 _hfield_filter, 164-730 ccd_hfield kernel; MuJoCo mjc_ConvexHField):
   in the height field's frame, every triangular prism of the cells under the geom's bounding box runs GJK / EPA against the geom; of the
   (at most MJ_MAXCONPAIR = 50) results up to four are kept: the deepest, the one furthest from it, the one furthest from that line, the
   one furthest from the other two edges */
#define MAXCONPAIR 50
enum { OVF_HFIELD = 1 << 5 }; /* types.py:171 */
static int collide_hfield(const RefModel* m, RefData* d, int g1, int g2, double margin, double gap, Con* out) {
  const double *pos1 = d->geom_xpos + 3 * g1, *mat1 = d->geom_xmat + 9 * g1, *pos2 = d->geom_xpos + 3 * g2, *mat2 = d->geom_xmat + 9 * g2;
  int hid = m->geom_dataid[g1], t2 = m->geom_type[g2];
  const double* size1 = m->hfield_size + 4 * hid;
  double dif[3], pos[3], R[9];
  v3sub(dif, pos2, pos1);
  matT_mul_vec(pos, mat1, dif);
  double r2 = m->geom_rbound[g2], fmargin = m->geom_margin[g1] + m->geom_margin[g2];
  for (int i = 0; i < 2; i++)
    if (size1[i] < pos[i] - r2 - fmargin || -size1[i] > pos[i] + r2 + fmargin) return 0;
  if (size1[2] < pos[2] - r2 - fmargin) return 0;
  if (-size1[3] > pos[2] + r2 + fmargin) return 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = mat1[i] * mat2[j] + mat1[3 + i] * mat2[3 + j] + mat1[6 + i] * mat2[6 + j]; /* mat1^T mat2 */
  CcdGeom b, a;
  memset(&a, 0, sizeof(a));
  memset(&b, 0, sizeof(b));
  b.type = t2;
  v3cpy(b.pos, pos);
  memcpy(b.rot, R, sizeof(R));
  v3cpy(b.size, m->geom_size + 3 * g2);
  b.margin = 0.0;
  b.index = b.cache = -1;
  mesh_of(m, g2, &b.vert, &b.nvert);
  b.graph = mesh_graph_of(m, g2);
  mesh_poly_of(m, g2, &b);
  double ext[6], p[3];
  static const double AX[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  for (int k = 0; k < 6; k++) { /* tight bounds through the support function: xmax, xmin, ymax, ymin, zmax, zmin */
    ccd_support(&b, AX[k], p);
    ext[k] = p[k / 2];
  }
  double xmax = ext[0], xmin = ext[1], ymax = ext[2], ymin = ext[3], zmax = ext[4], zmin = ext[5];
  if (xmin - fmargin > size1[0] || xmax + fmargin < -size1[0] || ymin - fmargin > size1[1] || ymax + fmargin < -size1[1] || zmin - fmargin > size1[2] ||
      zmax + fmargin < -size1[3])
    return 0;
  int nrow = m->hfield_nrow[hid], ncol = m->hfield_ncol[hid], adr = m->hfield_adr[hid];
  double x_scale = 0.5 * (double)(ncol - 1) / size1[0], y_scale = 0.5 * (double)(nrow - 1) / size1[1];
  int cmin = (int)floor((xmin + size1[0]) * x_scale), cmax = (int)ceil((xmax + size1[0]) * x_scale);
  int rmin = (int)floor((ymin + size1[1]) * y_scale), rmax = (int)ceil((ymax + size1[1]) * y_scale);
  if (cmin < 0) cmin = 0;
  if (cmax > ncol - 1) cmax = ncol - 1;
  if (rmin < 0) rmin = 0;
  if (rmax > nrow - 1) rmax = nrow - 1;
  double dx = 2.0 * size1[0] / (double)(ncol - 1), dy = 2.0 * size1[1] / (double)(nrow - 1);
  a.type = G_HFIELD;
  a.rot[0] = a.rot[4] = a.rot[8] = 1.0;
  a.margin = 0.0;
  a.index = a.cache = -1;
  b.margin = margin; /* the geom is inflated by half the margin, the prism tops are raised by the whole of it */
  double prism[6][3];
  memset(prism, 0, sizeof(prism));
  prism[0][2] = prism[1][2] = prism[2][2] = -size1[3];
  double cdist[MAXCONPAIR], cpos[MAXCONPAIR][3], cnrm[MAXCONPAIR][3];
  double min_dist = 1e10, min_pos[3] = {1e10, 1e10, 1e10}, min_nrm[3] = {1e10, 1e10, 1e10};
  int min_id = -1, count = 0;
  static Polytope pt;
  for (int r = rmin; r < rmax; r++) {
    for (int k = 0; k < 2; k++) { /* the first two vertices of the row strip */
      double x = dx * (double)cmin - size1[0], y = dy * (double)(r + (k == 0 ? 1 : 0)) - size1[1];
      double z = m->hfield_data[adr + (r + (k == 0 ? 1 : 0)) * ncol + cmin] * size1[2] + margin;
      v3cpy(prism[0], prism[1]); v3cpy(prism[1], prism[2]); v3cpy(prism[3], prism[4]); v3cpy(prism[4], prism[5]);
      prism[2][0] = prism[5][0] = x;
      prism[2][1] = prism[5][1] = y;
      prism[5][2] = z;
    }
    for (int c = cmin + 1; c <= cmax; c++)
      for (int k = 0; k < 2; k++) { /* both triangles of the cell */
        if (count >= MAXCONPAIR) { d->overflow |= OVF_HFIELD; continue; }
        int rr = r + (k == 0 ? 1 : 0);
        double x = dx * (double)c - size1[0], y = dy * (double)rr - size1[1], z = m->hfield_data[adr + rr * ncol + c] * size1[2] + margin;
        v3cpy(prism[0], prism[1]); v3cpy(prism[1], prism[2]); v3cpy(prism[3], prism[4]); v3cpy(prism[4], prism[5]);
        prism[2][0] = prism[5][0] = x;
        prism[2][1] = prism[5][1] = y;
        prism[5][2] = z;
        if (prism[3][2] < zmin && prism[4][2] < zmin && prism[5][2] < zmin) continue;
        memcpy(a.prism, prism, sizeof(prism));
        for (int q = 0; q < 3; q++) a.pos[q] = (prism[0][q] + prism[1][q] + prism[2][q] + prism[3][q] + prism[4][q] + prism[5][q]) * (1.0 / 6.0);
        double dist, w1[3], w2[3];
        int face;
        b.index = -1; /* (the reference passes its geom structs by value: every prism starts from the uncached geom) */
        a.index = -1;
        int n = ccd_run(m->ccd_tolerance, 0.0, m->ccd_iterations, m->epa_iterations, a, b, &dist, w1, w2, &d->overflow, &face, &pt);
        if (n == 0) continue;
        cdist[count] = dist;
        double pl[3], fr[9], wd[3];
        for (int q = 0; q < 3; q++) pl[q] = 0.5 * (w1[q] + w2[q]);
        mat_mul_vec(cpos[count], mat1, pl);
        v3add(cpos[count], cpos[count], pos1);
        v3sub(wd, w1, w2);
        make_frame(fr, wd);
        mat_mul_vec(cnrm[count], mat1, fr);
        if (dist < min_dist) { min_dist = dist; v3cpy(min_nrm, cnrm[count]); v3cpy(min_pos, cpos[count]); min_id = count; }
        count++;
      }
  }
  int nout = 0;
#define HF_EMIT(dist_, pos_, nrm_) do { out[nout].dist = (dist_); v3cpy(out[nout].pos, (pos_)); make_frame(out[nout].frame, (nrm_)); nout++; } while (0)
  HF_EMIT(min_dist, min_pos, min_nrm); /* (contact 0 is written unconditionally; the caller's margin test drops an empty one) */
  const double MIN_NEXT = 1.0e-3;
  int id1 = -1, id2 = -1, id3 = -1;
  double best = -1e10, t[3], u[3];
  for (int i = 0; i < count; i++) { /* furthest from the deepest */
    if (i == min_id) continue;
    v3sub(t, cpos[i], min_pos);
    double dd = v3len(t);
    if (dd > best) { id1 = i; best = dd; }
  }
  if (id1 == -1 || (0.0 < best && best < MIN_NEXT)) return nout;
  HF_EMIT(cdist[id1], cpos[id1], cnrm[id1]);
  double dmin1[3];
  v3sub(t, min_pos, cpos[id1]);
  v3cross(dmin1, min_nrm, t);
  best = -1e10;
  for (int i = 0; i < count; i++) { /* furthest from the line deepest - contact 1 */
    if (i == min_id || i == id1) continue;
    v3sub(t, cpos[i], min_pos);
    double dd = fabs(v3dot(t, dmin1));
    if (dd > best) { id2 = i; best = dd; }
  }
  if (id2 == -1 || (0.0 < best && best < MIN_NEXT)) return nout;
  HF_EMIT(cdist[id2], cpos[id2], cnrm[id2]);
  double vmin2[3], v12[3];
  v3sub(t, min_pos, cpos[id2]);
  v3cross(vmin2, min_nrm, t);
  v3sub(t, cpos[id1], cpos[id2]);
  v3cross(v12, min_nrm, t);
  best = -1e10;
  for (int i = 0; i < count; i++) { /* furthest from the other two edges */
    if (i == min_id || i == id1 || i == id2) continue;
    v3sub(t, cpos[i], min_pos);
    v3sub(u, cpos[id1], cpos[i]);
    double dd = fabs(v3dot(t, vmin2)) + fabs(v3dot(u, v12));
    if (dd > best) { id3 = i; best = dd; }
  }
  if (id3 == -1 || (0.0 < best && best < MIN_NEXT)) return nout;
  HF_EMIT(cdist[id3], cpos[id3], cnrm[id3]);
  return nout;
}
static int is_convex_pair(int t1, int t2) { /* MJ_COLLISION_TABLE collision_driver.py:47-80 (box-box: below; plane-mesh is primitive) */
  if (t2 == G_MESH && t1 >= G_SPHERE) return 1;
  return (t1 == G_SPHERE && t2 == G_ELLIPSOID) || (t1 == G_CAPSULE && (t2 == G_ELLIPSOID || t2 == G_CYLINDER)) ||
         (t1 == G_ELLIPSOID && (t2 == G_ELLIPSOID || t2 == G_CYLINDER || t2 == G_BOX)) || (t1 == G_CYLINDER && (t2 == G_CYLINDER || t2 == G_BOX));
}

static int collide_pair(const RefModel* m, RefData* d, int g1, int g2, double margin, double gap, Con* out) {
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  const double *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2;
  const double *R1 = d->geom_xmat + 9 * g1, *R2 = d->geom_xmat + 9 * g2;
  const double *s1 = m->geom_size + 3 * g1, *s2 = m->geom_size + 3 * g2;
  double ax1[3] = {R1[2], R1[5], R1[8]}, ax2[3] = {R2[2], R2[5], R2[8]};
  int n = 0;
  if (t1 == G_HFIELD) return t2 >= G_SPHERE ? collide_hfield(m, d, g1, g2, margin, gap, out) : 0; /* collision_driver.py:54-59 */
  /* box-box is a convex pair unless DisableBit.NATIVECCD asks for the primitive collider (collision_driver.py:867-870) */
  if (is_convex_pair(t1, t2) || (t1 == G_BOX && t2 == G_BOX && !(m->disableflags & DSBL_NATIVECCD)))
    return ccd_contact(m, g1, g2, t1, p1, R1, s1, t2, p2, R2, s2, margin, gap, out, &d->overflow);
  if (t1 == G_PLANE && t2 == G_MESH) { /* collision_primitive.py:52-274 plane_convex, exhaustive branch (no graph or < 10 vertices) */
    const double* vert;
    int nvert, idx[4] = {-1, -1, -1, -1};
    mesh_of(m, g2, &vert, &nvert);
    const double HUGE_V = 1e6;
    double pl[3], nl[3], dif[3];
    v3sub(dif, p1, p2);
    matT_mul_vec(pl, R2, dif);
    matT_mul_vec(nl, R2, ax1);
    double max_support = -HUGE_V, a[3] = {0, 0, 0}, b[3] = {0, 0, 0}, c[3] = {0, 0, 0}, t[3];
    const int* graph = mesh_graph_of(m, g2);
    if (graph && nvert >= 10) {
      /* collision_primitive.py:131-243: the same four picks by hill climbing on the hull's vertex graph, each climb starting where the
         previous one ended (the results are local maxima along the graph, not the exhaustive branch's global ones) */
      int numvert = graph[0];
      const int *edgeadr = graph + 2, *globalid = graph + 2 + numvert, *edge = graph + 2 + 2 * numvert;
      int imax = 0, prev;
      do { /* deepest vertex */
        prev = imax;
        for (int i = edgeadr[imax]; edge[i] >= 0; i++) {
          v3sub(t, pl, vert + 3 * globalid[edge[i]]);
          double sup = v3dot(t, nl);
          if (sup > max_support) { max_support = sup; imax = edge[i]; }
        }
      } while (imax != prev);
      double threshold = fmax(0.0, max_support - 1e-3), best = -HUGE_V;
      do { /* a: deepest among the vertices above the threshold (the climb may start on a vertex that was never scored) */
        prev = imax;
        for (int i = edgeadr[imax]; edge[i] >= 0; i++) {
          v3sub(t, pl, vert + 3 * globalid[edge[i]]);
          double sup = v3dot(t, nl), dd = sup > threshold ? sup : -HUGE_V;
          if (dd > best) { best = dd; imax = edge[i]; }
        }
      } while (imax != prev);
      idx[0] = globalid[imax];
      v3cpy(a, vert + 3 * idx[0]);
      best = -HUGE_V;
      do { /* b: furthest from a */
        prev = imax;
        for (int i = edgeadr[imax]; edge[i] >= 0; i++) {
          const double* v = vert + 3 * globalid[edge[i]];
          v3sub(t, pl, v);
          double mask = v3dot(t, nl) > threshold ? 0.0 : -HUGE_V;
          v3sub(t, a, v);
          double dd = v3dot(t, t) + mask;
          if (dd > best) { best = dd; imax = edge[i]; }
        }
      } while (imax != prev);
      idx[1] = globalid[imax];
      v3cpy(b, vert + 3 * idx[1]);
      double ab[3], ac[3], bc[3];
      v3sub(t, a, b);
      v3cross(ab, nl, t);
      best = -HUGE_V;
      do { /* c: furthest from the line a-b */
        prev = imax;
        for (int i = edgeadr[imax]; edge[i] >= 0; i++) {
          const double* v = vert + 3 * globalid[edge[i]];
          v3sub(t, pl, v);
          double mask = v3dot(t, nl) > threshold ? 0.0 : -HUGE_V;
          v3sub(t, a, v);
          double dd = fabs(v3dot(t, ab)) + mask;
          if (dd > best) { best = dd; imax = edge[i]; }
        }
      } while (imax != prev);
      idx[2] = globalid[imax];
      v3cpy(c, vert + 3 * idx[2]);
      v3sub(t, a, c);
      v3cross(ac, nl, t);
      v3sub(t, b, c);
      v3cross(bc, nl, t);
      best = -HUGE_V;
      do { /* d: furthest from the other two edges */
        prev = imax;
        for (int i = edgeadr[imax]; edge[i] >= 0; i++) {
          const double* v = vert + 3 * globalid[edge[i]];
          v3sub(t, pl, v);
          double mask = v3dot(t, nl) > threshold ? 0.0 : -HUGE_V, ap[3], bp[3];
          v3sub(ap, a, v);
          v3sub(bp, b, v);
          double dd = fabs(v3dot(ap, ac)) + mask + fabs(v3dot(bp, bc)) + mask;
          if (dd > best) { best = dd; imax = edge[i]; }
        }
      } while (imax != prev);
      idx[3] = globalid[imax];
    } else {
    for (int i = 0; i < nvert; i++) {
      v3sub(t, pl, vert + 3 * i);
      double sup = v3dot(t, nl);
      if (sup > max_support) { max_support = sup; idx[0] = i; v3cpy(a, vert + 3 * i); }
    }
    if (max_support < 0) return 0;
    double threshold = max_support - 1e-3, best = -HUGE_V;
    for (int i = 0; i < nvert; i++) { /* b: furthest from a among the vertices within 1 mm of the deepest */
      v3sub(t, pl, vert + 3 * i);
      double mask = v3dot(t, nl) > threshold ? 0.0 : -HUGE_V;
      v3sub(t, a, vert + 3 * i);
      double dd = v3dot(t, t) + mask;
      if (dd > best) { idx[1] = i; best = dd; v3cpy(b, vert + 3 * i); }
    }
    double ab[3], ac[3], bc[3];
    v3sub(t, a, b);
    v3cross(ab, nl, t);
    best = -HUGE_V;
    for (int i = 0; i < nvert; i++) { /* c: furthest from the line a-b */
      v3sub(t, pl, vert + 3 * i);
      double mask = v3dot(t, nl) > threshold ? 0.0 : -HUGE_V;
      v3sub(t, a, vert + 3 * i);
      double dd = fabs(v3dot(t, ab)) + mask;
      if (dd > best) { idx[2] = i; best = dd; v3cpy(c, vert + 3 * i); }
    }
    v3sub(t, a, c);
    v3cross(ac, nl, t);
    v3sub(t, b, c);
    v3cross(bc, nl, t);
    best = -HUGE_V;
    for (int i = 0; i < nvert; i++) { /* d: furthest from the other two edges */
      v3sub(t, pl, vert + 3 * i);
      double mask = v3dot(t, nl) > threshold ? 0.0 : -HUGE_V, ap[3], bp[3];
      v3sub(ap, a, vert + 3 * i);
      v3sub(bp, b, vert + 3 * i);
      double dd = fabs(v3dot(ap, ac)) + mask + fabs(v3dot(bp, bc)) + mask;
      if (dd > best) { idx[3] = i; best = dd; }
    }
    }
    for (int i = 3; i >= 0; i--) { /* vertices that appear once among indices[0..i] */
      int count = 0;
      for (int j = 0; j <= i; j++) count += idx[j] == idx[i];
      if (count != 1) continue;
      double pw[3];
      mat_mul_vec(pw, R2, vert + 3 * idx[i]);
      v3sub(t, pl, vert + 3 * idx[i]);
      double dist = -v3dot(t, nl);
      for (int k = 0; k < 3; k++) out[n].pos[k] = p2[k] + pw[k] - 0.5 * dist * ax1[k];
      out[n].dist = dist;
      make_frame(out[n].frame, ax1);
      n++;
    }
    return n;
  }
  if (t1 == G_PLANE && t2 == G_SPHERE) { /* collision_primitive.py:281 */
    plane_sphere(ax1, p1, p2, s2[0], &out[0].dist, out[0].pos);
    make_frame(out[0].frame, ax1);
    n = 1;
  } else if (t1 == G_PLANE && t2 == G_CAPSULE) { /* core:253 */
    double b[3], c[3], seg[3], e[3];
    double dn = v3dot(ax1, ax2);
    for (int i = 0; i < 3; i++) b[i] = ax2[i] - ax1[i] * dn;
    double bn = v3len(b);
    if (bn > 0) { b[0] /= bn; b[1] /= bn; b[2] /= bn; }
    if (bn < 0.5) {
      if (-0.5 < ax1[1] && ax1[1] < 0.5) v3set(b, 0, 1, 0);
      else v3set(b, 0, 0, 1);
    }
    v3cross(c, ax1, b);
    for (int i = 0; i < 3; i++) seg[i] = ax2[i] * s2[1];
    for (int k = 0; k < 2; k++) {
      v3addscl(e, p2, seg, k == 0 ? 1.0 : -1.0);
      plane_sphere(ax1, p1, e, s2[0], &out[k].dist, out[k].pos);
      v3cpy(out[k].frame, ax1); v3cpy(out[k].frame + 3, b); v3cpy(out[k].frame + 6, c);
    }
    n = 2;
  } else if (t1 == G_PLANE && t2 == G_BOX) { /* core:337 */
    double dif[3];
    v3sub(dif, p2, p1);
    double cd = v3dot(dif, ax1);
    for (int i = 0; i < 8; i++) {
      double corner[3] = {(i & 1) ? s2[0] : -s2[0], (i & 2) ? s2[1] : -s2[1], (i & 4) ? s2[2] : -s2[2]}, cw[3];
      mat_mul_vec(cw, R2, corner);
      double cdist = cd + v3dot(ax1, cw);
      out[i].dist = cdist;
      for (int k = 0; k < 3; k++) out[i].pos[k] = cw[k] + p2[k] - 0.5 * ax1[k] * cdist;
      make_frame(out[i].frame, ax1);
    }
    n = 8;
  } else if (t1 == G_PLANE && t2 == G_ELLIPSOID) { /* core:306 */
    double loc[3], sup[3], pw[3], dif[3];
    matT_mul_vec(loc, R2, ax1);
    for (int k = 0; k < 3; k++) sup[k] = loc[k] * s2[k];
    v3normalize(sup);
    for (int k = 0; k < 3; k++) sup[k] = -sup[k] * s2[k];
    mat_mul_vec(pw, R2, sup);
    v3add(pw, pw, p2);
    v3sub(dif, pw, p1);
    double dist = v3dot(ax1, dif);
    out[0].dist = dist;
    v3addscl(out[0].pos, pw, ax1, -0.5 * dist);
    make_frame(out[0].frame, ax1);
    n = 1;
  } else if (t1 == G_PLANE && t2 == G_CYLINDER) { /* core:460 */
    double axis[3] = {ax2[0], ax2[1], ax2[2]}, vec[3], dif[3], vec1[3];
    double r = s2[0], hh = s2[1];
    double prjaxis = v3dot(ax1, axis);
    if (prjaxis > 0) { axis[0] = -axis[0]; axis[1] = -axis[1]; axis[2] = -axis[2]; prjaxis = -prjaxis; }
    v3sub(dif, p2, p1);
    double dist0 = v3dot(dif, ax1);
    for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - ax1[k];
    double len_sqr = v3dot(vec, vec);
    if (len_sqr >= 1e-12) { double s = safe_div(r, sqrt(len_sqr)); vec[0] *= s; vec[1] *= s; vec[2] *= s; }
    else v3set(vec, r, 0, 0);
    double prjvec = v3dot(vec, ax1);
    for (int k = 0; k < 3; k++) axis[k] *= hh;
    prjaxis *= hh;
    double d1 = dist0 + prjaxis + prjvec, d2 = dist0 - prjaxis + prjvec;
    for (int k = 0; k < 3; k++) out[0].pos[k] = p2[k] + vec[k] + axis[k] - ax1[k] * d1 * 0.5;
    for (int k = 0; k < 3; k++) out[1].pos[k] = p2[k] + vec[k] - axis[k] - ax1[k] * d2 * 0.5;
    out[0].dist = d1; out[1].dist = d2;
    double d3 = dist0 + prjaxis - 0.5 * prjvec;
    v3cross(vec1, vec, axis);
    v3normalize(vec1);
    for (int k = 0; k < 3; k++) vec1[k] *= r * sqrt(3.0) * 0.5;
    for (int k = 0; k < 3; k++) out[2].pos[k] = p2[k] + vec1[k] + axis[k] - vec[k] * 0.5 - ax1[k] * d3 * 0.5;
    for (int k = 0; k < 3; k++) out[3].pos[k] = p2[k] - vec1[k] + axis[k] - vec[k] * 0.5 - ax1[k] * d3 * 0.5;
    out[2].dist = out[3].dist = d3;
    for (int i = 0; i < 4; i++) make_frame(out[i].frame, ax1);
    n = 4;
  } else if (t1 == G_SPHERE && t2 == G_SPHERE) {
    double nn[3];
    sphere_sphere(p1, s1[0], p2, s2[0], &out[0].dist, out[0].pos, nn);
    make_frame(out[0].frame, nn);
    n = 1;
  } else if (t1 == G_SPHERE && t2 == G_CAPSULE) { /* core:88 */
    double seg[3], a[3], b[3], pt[3], nn[3];
    for (int k = 0; k < 3; k++) seg[k] = ax2[k] * s2[1];
    v3sub(a, p2, seg);
    v3add(b, p2, seg);
    closest_segment_point(pt, a, b, p1);
    sphere_sphere(p1, s1[0], pt, s2[0], &out[0].dist, out[0].pos, nn);
    make_frame(out[0].frame, nn);
    n = 1;
  } else if (t1 == G_CAPSULE && t2 == G_CAPSULE) { /* core:123 */
    double axis1[3], axis2[3], dif[3], v1[3], v2[3], nn[3];
    for (int k = 0; k < 3; k++) { axis1[k] = ax1[k] * s1[1]; axis2[k] = ax2[k] * s2[1]; }
    v3sub(dif, p1, p2);
    double ma = v3dot(axis1, axis1), mb = -v3dot(axis1, axis2), mc = v3dot(axis2, axis2);
    double u = -v3dot(axis1, dif), v = v3dot(axis2, dif);
    double det = ma * mc - mb * mb;
    if (fabs(det) >= MINVAL) {
      double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
      if (x1 > 1.0) { x1 = 1.0; x2 = (v - mb) / mc; }
      else if (x1 < -1.0) { x1 = -1.0; x2 = (v + mb) / mc; }
      if (x2 > 1.0) { x2 = 1.0; x1 = clampd((u - mb) / ma, -1.0, 1.0); }
      else if (x2 < -1.0) { x2 = -1.0; x1 = clampd((u + mb) / ma, -1.0, 1.0); }
      v3addscl(v1, p1, axis1, x1);
      v3addscl(v2, p2, axis2, x2);
      double dist, pos[3];
      sphere_sphere(v1, s1[0], v2, s2[0], &dist, pos, nn);
      if (dist <= margin) { out[0].dist = dist; v3cpy(out[0].pos, pos); make_frame(out[0].frame, nn); n = 1; }
    } else {
      double dist, pos[3], x1, x2;
      v3add(v1, p1, axis1);
      x2 = clampd((v - mb) / mc, -1.0, 1.0);
      v3addscl(v2, p2, axis2, x2);
      sphere_sphere(v1, s1[0], v2, s2[0], &dist, pos, nn);
      if (dist <= margin) { out[n].dist = dist; v3cpy(out[n].pos, pos); make_frame(out[n].frame, nn); n++; }
      v3sub(v1, p1, axis1);
      x2 = clampd((v + mb) / mc, -1.0, 1.0);
      v3addscl(v2, p2, axis2, x2);
      sphere_sphere(v1, s1[0], v2, s2[0], &dist, pos, nn);
      if (dist <= margin) { out[n].dist = dist; v3cpy(out[n].pos, pos); make_frame(out[n].frame, nn); n++; }
      if (n < 2) {
        v3add(v2, p2, axis2);
        x1 = clampd((u - mb) / ma, -1.0, 1.0);
        v3addscl(v1, p1, axis1, x1);
        sphere_sphere(v1, s1[0], v2, s2[0], &dist, pos, nn);
        if (dist <= margin) { out[n].dist = dist; v3cpy(out[n].pos, pos); make_frame(out[n].frame, nn); n++; }
      }
      if (n < 2) {
        v3sub(v2, p2, axis2);
        x1 = clampd((u + mb) / ma, -1.0, 1.0);
        v3addscl(v1, p1, axis1, x1);
        sphere_sphere(v1, s1[0], v2, s2[0], &dist, pos, nn);
        if (dist <= margin) { out[n].dist = dist; v3cpy(out[n].pos, pos); make_frame(out[n].frame, nn); n++; }
      }
    }
  } else if (t1 == G_SPHERE && t2 == G_BOX) { /* core:1044 */
    sphere_box(p1, s1[0], p2, R2, s2, &out[0]);
    n = 1;
  } else if (t1 == G_CAPSULE && t2 == G_BOX) { /* core:1099 */
    n = capsule_box(p1, ax1, s1[0], s1[1], p2, R2, s2, out);
  } else if (t1 == G_BOX && t2 == G_BOX) { /* core:589 */
    n = box_box(p1, R1, s1, p2, R2, s2, margin, out);
  } else if (t1 == G_SPHERE && t2 == G_CYLINDER) { /* core:388 */
    double vec[3], aproj[3], pproj[3], nn[3], pos[3], target[3], dist;
    double r = s2[0], hh = s2[1];
    v3sub(vec, p1, p2);
    double x = v3dot(vec, ax2);
    for (int k = 0; k < 3; k++) { aproj[k] = ax2[k] * x; pproj[k] = vec[k] - aproj[k]; }
    double psq = v3dot(pproj, pproj);
    int side = fabs(x) < hh, cap = psq < r * r;
    if (side && cap) { /* centre inside the cylinder: the nearer surface wins */
      if (hh - fabs(x) < r - sqrt(psq)) side = 0; else cap = 0;
    }
    if (side) {
      v3add(target, p2, aproj);
      sphere_sphere(p1, s1[0], target, r, &dist, pos, nn);
    } else if (cap) {
      double sg = x > 0.0 ? 1.0 : -1.0, pn[3], pc[3];
      for (int k = 0; k < 3; k++) { pn[k] = sg * ax2[k]; pc[k] = p2[k] + pn[k] * hh; }
      plane_sphere(pn, pc, p1, s1[0], &dist, pos);
      for (int k = 0; k < 3; k++) nn[k] = -pn[k];
    } else { /* rim */
      double inv = safe_div(1.0, sqrt(psq)), sg = x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0);
      for (int k = 0; k < 3; k++) target[k] = p2[k] + ax2[k] * (sg * hh) + pproj[k] * (r * inv);
      sphere_sphere(p1, s1[0], target, 0.0, &dist, pos, nn);
    }
    out[0].dist = dist;
    v3cpy(out[0].pos, pos);
    make_frame(out[0].frame, nn);
    n = 1;
  }
  return n;
}

/* collision_core.py:321-414 (priority/solmix mixing) + 297-318 (margin/gap) */
static void contact_params(const RefModel* m, int g1, int g2, int pid, int* condim, double* friction, double* solref,
                           double* solreffriction, double* solimp, double* margin, double* gap) {
  if (pid >= 0) { /* explicit pair: its own parameters */
    *condim = m->xpair_dim[pid];
    for (int k = 0; k < 5; k++) { friction[k] = fmax(MINMU, m->xpair_friction[5 * pid + k]); solimp[k] = m->xpair_solimp[5 * pid + k]; }
    for (int k = 0; k < 2; k++) { solref[k] = m->xpair_solref[2 * pid + k]; solreffriction[k] = m->xpair_solreffriction[2 * pid + k]; }
    *margin = m->xpair_margin[pid];
    *gap = m->xpair_gap[pid];
    return;
  }
  *margin = m->geom_margin[g1] + m->geom_margin[g2];
  *gap = m->geom_gap[g1] + m->geom_gap[g2];
  double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
  int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
  double f[3];
  if (p1 > p2) { mix = 1.0; *condim = m->geom_condim[g1]; v3cpy(f, m->geom_friction + 3 * g1); }
  else if (p2 > p1) { mix = 0.0; *condim = m->geom_condim[g2]; v3cpy(f, m->geom_friction + 3 * g2); }
  else {
    mix = safe_div(s1, s1 + s2);
    if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
    if (s1 < MINVAL && s2 >= MINVAL) mix = 0.0;
    if (s1 >= MINVAL && s2 < MINVAL) mix = 1.0;
    *condim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
    for (int k = 0; k < 3; k++) f[k] = fmax(m->geom_friction[3 * g1 + k], m->geom_friction[3 * g2 + k]);
  }
  friction[0] = friction[1] = f[0]; friction[2] = f[1]; friction[3] = friction[4] = f[2];
  const double *r1 = m->geom_solref + 2 * g1, *r2 = m->geom_solref + 2 * g2;
  if (r1[0] > 0.0 && r2[0] > 0.0) { solref[0] = mix * r1[0] + (1 - mix) * r2[0]; solref[1] = mix * r1[1] + (1 - mix) * r2[1]; }
  else { solref[0] = fmin(r1[0], r2[0]); solref[1] = fmin(r1[1], r2[1]); }
  solreffriction[0] = solreffriction[1] = 0.0;
  for (int k = 0; k < 5; k++) solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
  for (int k = 0; k < 5; k++) friction[k] = fmax(MINMU, friction[k]);
}

/* _aabb_filter collision_driver.py:124-223: world-aligned boxes around the two rotated local boxes */
static int aabb_filter(const double* c1, const double* c2, const double* s1, const double* s2, double margin, const double* x1,
                       const double* x2, const double* R1, const double* R2) {
  double ctr[2][3], mx[2][3], mn[2][3];
  const double* cs[2] = {c1, c2};
  const double* ss[2] = {s1, s2};
  const double* xs[2] = {x1, x2};
  const double* Rs[2] = {R1, R2};
  for (int g = 0; g < 2; g++) {
    for (int k = 0; k < 3; k++) {
      ctr[g][k] = Rs[g][3 * k] * cs[g][0] + Rs[g][3 * k + 1] * cs[g][1] + Rs[g][3 * k + 2] * cs[g][2] + xs[g][k];
      mx[g][k] = -MAXVAL;
      mn[g][k] = MAXVAL;
    }
    for (int i = 0; i < 8; i++) {
      double corner[3] = {(i & 4 ? 1 : -1) * ss[g][0], (i & 2 ? 1 : -1) * ss[g][1], (i & 1 ? 1 : -1) * ss[g][2]};
      for (int k = 0; k < 3; k++) {
        double pk = Rs[g][3 * k] * corner[0] + Rs[g][3 * k + 1] * corner[1] + Rs[g][3 * k + 2] * corner[2];
        if (pk > mx[g][k]) mx[g][k] = pk;
        if (pk < mn[g][k]) mn[g][k] = pk;
      }
    }
  }
  for (int k = 0; k < 3; k++) {
    if (ctr[0][k] + mx[0][k] + margin < ctr[1][k] + mn[1][k]) return 0;
    if (ctr[1][k] + mx[1][k] + margin < ctr[0][k] + mn[0][k]) return 0;
  }
  return 1;
}
/* _obb_filter collision_driver.py:226-275 (mj_collideOBB): separating axis test on the six face normals */
static int obb_filter(const double* c1, const double* c2, const double* s1, const double* s2, double margin, const double* x1,
                      const double* x2, const double* R1, const double* R2) {
  double xc[2][3], nrm[6][3];
  const double* ss[2] = {s1, s2};
  for (int k = 0; k < 3; k++) {
    xc[0][k] = R1[3 * k] * c1[0] + R1[3 * k + 1] * c1[1] + R1[3 * k + 2] * c1[2] + x1[k];
    xc[1][k] = R2[3 * k] * c2[0] + R2[3 * k + 1] * c2[1] + R2[3 * k + 2] * c2[2] + x2[k];
  }
  for (int a = 0; a < 3; a++)
    for (int k = 0; k < 3; k++) { nrm[a][k] = R1[3 * k + a]; nrm[3 + a][k] = R2[3 * k + a]; }
  for (int j = 0; j < 2; j++)
    for (int k = 0; k < 3; k++) {
      double proj[2], radius[2];
      for (int i = 0; i < 2; i++) {
        proj[i] = v3dot(xc[i], nrm[3 * j + k]);
        radius[i] = fabs(ss[i][0] * v3dot(nrm[3 * i], nrm[3 * j + k])) + fabs(ss[i][1] * v3dot(nrm[3 * i + 1], nrm[3 * j + k])) +
                    fabs(ss[i][2] * v3dot(nrm[3 * i + 2], nrm[3 * j + k]));
      }
      if (radius[0] + radius[1] + margin < fabs(proj[1] - proj[0])) return 0;
    }
  return 1;
}
/* _broadphase_filter collision_driver.py:278-334.  (Explicit <contact><pair> entries use the pair's margin + gap here, as
 * MuJoCo C does; the reference uses the geoms' own.) */
static int broadphase_filter(const RefModel* m, const RefData* d, int p, int g1, int g2) {
  double rb1 = m->geom_rbound[g1], rb2 = m->geom_rbound[g2];
  int pid = m->nexplicit ? m->nxn_pairid[p] : -1;
  double mg = pid >= 0 ? m->xpair_margin[pid] + m->xpair_gap[pid]
                       : m->geom_margin[g1] + m->geom_gap[g1] + m->geom_margin[g2] + m->geom_gap[g2];
  const double *x1 = d->geom_xpos + 3 * g1, *x2 = d->geom_xpos + 3 * g2;
  const double *R1 = d->geom_xmat + 9 * g1, *R2 = d->geom_xmat + 9 * g2;
  double dif[3];
  int filt = m->broadphase_filter;
  if (rb1 == 0.0 || rb2 == 0.0) {
    if (!(filt & 1)) return 1;
    if (rb1 == 0.0) {
      double nrm[3] = {R1[2], R1[5], R1[8]};
      v3sub(dif, x2, x1);
      return v3dot(dif, nrm) <= rb2 + mg;
    }
    double nrm[3] = {R2[2], R2[5], R2[8]};
    v3sub(dif, x1, x2);
    return v3dot(dif, nrm) <= rb1 + mg;
  }
  if (filt & 2) {
    double bound = rb1 + rb2 + mg;
    v3sub(dif, x2, x1);
    if (!(v3dot(dif, dif) <= bound * bound)) return 0;
  }
  const double *a1 = m->geom_aabb + 6 * g1, *a2 = m->geom_aabb + 6 * g2;
  if ((filt & 4) && !aabb_filter(a1, a2, a1 + 3, a2 + 3, mg, x1, x2, R1, R2)) return 0;
  if ((filt & 8) && !obb_filter(a1, a2, a1 + 3, a2 + 3, mg, x1, x2, R1, R2)) return 0;
  return 1;
}

/* sap_broadphase collision_driver.py:567-682: candidates = pairs whose bounding-sphere projections on a fixed direction overlap
 * (_sap_project 375, sort, sap_range collision_core.py:501 incl. its "limit" element, sweep 424).  The reference appends pairs
 * in atomic order; here the surviving pairs are reported in the canonical pair order so that contacts do not depend on the
 * broadphase.  mark[p] = 1 for every filtered pair index p the sweep visits. */
static void sap_candidates(const RefModel* m, const RefData* d, unsigned char* mark) {
  int ng = m->ngeom;
  double dir[3] = {0.5935, 0.7790, 0.1235};
  double dn = sqrt(v3dot(dir, dir));
  for (int k = 0; k < 3; k++) dir[k] /= dn;
  double* lo = (double*)malloc(sizeof(double) * 2 * (size_t)(ng > 0 ? ng : 1));
  double* hi = lo + ng;
  int* idx = (int*)malloc(sizeof(int) * (size_t)(ng > 0 ? ng : 1));
  for (int g = 0; g < ng; g++) {
    double rb = m->geom_rbound[g];
    if (rb == 0.0) rb = MAXVAL;
    double radius = rb + m->geom_margin[g] + m->geom_gap[g], center = v3dot(dir, d->geom_xpos + 3 * g);
    idx[g] = g;
    if (center == center) { lo[g] = center - radius; hi[g] = center + radius; }
    else { lo[g] = MAXVAL; hi[g] = MAXVAL; }
  }
  for (int i = 1; i < ng; i++) {  /* stable insertion sort by the lower bound */
    int gi = idx[i];
    int j = i - 1;
    while (j >= 0 && lo[idx[j]] > lo[gi]) { idx[j + 1] = idx[j]; j--; }
    idx[j + 1] = gi;
  }
  /* pair index of an unordered geom pair in the filtered list (-1: filtered out) */
  for (int i = 0; i < ng; i++) {
    double upper = hi[idx[i]];
    int lower_i = i + 1, upper_i = ng;  /* sap_binary_search: first sorted element with lower bound > upper */
    while (lower_i < upper_i) {
      int mid = (lower_i + upper_i) >> 1;
      if (lo[idx[mid]] > upper) upper_i = mid; else lower_i = mid + 1;
    }
    int limit = upper_i < ng - 1 ? upper_i : ng - 1;
    for (int j = i + 1; j <= limit; j++) {
      int g1 = idx[i], g2 = idx[j];
      if (g2 < g1) { int t = g1; g1 = g2; g2 = t; }
      for (int p = 0; p < m->npair; p++)  /* (the oracle scans; the engine looks the pair up in a table) */
        if (m->pair_geom[2 * p] == g1 && m->pair_geom[2 * p + 1] == g2) { mark[p] = 1; break; }
    }
  }
  free(lo);
  free(idx);
}

/* collision_driver.py:98-334 (filters), 567-682 (SAP), 684-770 (nxn), collision_core.py:214-294 (write_contact) */
void ref_collision(const RefModel* m, RefData* d) {
  d->ncon = 0;
  d->ncollision = 0;
  if (m->disableflags & (DSBL_CONSTRAINT | DSBL_CONTACT)) return;
  unsigned char* mark = NULL;
  if (m->broadphase != 0) {
    mark = (unsigned char*)calloc((size_t)(m->npair > 0 ? m->npair : 1), 1);
    sap_candidates(m, d, mark);
  }
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom[2 * p], g2 = m->pair_geom[2 * p + 1];
    int pid = m->nexplicit ? m->nxn_pairid[p] : -1;
    if (mark && !mark[p]) continue;
    if (sleep_enabled(m)) { /* collision_driver.py:494-503: no pair of two sleeping bodies, or of a sleeping and a static one */
      int s1 = d->body_awake[m->geom_bodyid[g1]], s2 = d->body_awake[m->geom_bodyid[g2]];
      if (s1 == SLEEP_ASLEEP && s2 == SLEEP_ASLEEP) continue;
      if ((s1 == SLEEP_ASLEEP && s2 == SLEEP_STATIC) || (s2 == SLEEP_ASLEEP && s1 == SLEEP_STATIC)) continue;
    }
    int pass = broadphase_filter(m, d, p, g1, g2);
    if (!pass) continue;
    d->ncollision++;
    if (m->geom_type[g1] > m->geom_type[g2]) { int t = g1; g1 = g2; g2 = t; }
    int condim;
    double friction[5], solref[2], solreffriction[2], solimp[5], margin, gap;
    contact_params(m, g1, g2, pid, &condim, friction, solref, solreffriction, solimp, &margin, &gap);
    Con out[8];
    int n = collide_pair(m, d, g1, g2, margin, gap, out);
    for (int k = 0; k < n; k++) {
      if (!(out[k].dist < margin + gap)) continue;
      int c = d->ncon;
      if (c >= m->nconmax) { d->overflow |= OVF_NARROW; d->ncon++; continue; }
      d->con_dist[c] = out[k].dist;
      v3cpy(d->con_pos + 3 * c, out[k].pos);
      memcpy(d->con_frame + 9 * c, out[k].frame, 9 * sizeof(double));
      d->con_includemargin[c] = margin;
      memcpy(d->con_friction + 5 * c, friction, sizeof(friction));
      memcpy(d->con_solref + 2 * c, solref, sizeof(solref));
      memcpy(d->con_solreffriction + 2 * c, solreffriction, sizeof(solreffriction));
      memcpy(d->con_solimp + 5 * c, solimp, sizeof(solimp));
      d->con_dim[c] = condim;
      d->con_geom[2 * c] = g1;
      d->con_geom[2 * c + 1] = g2;
      for (int j = 0; j < 10; j++) d->con_efc_address[10 * c + j] = -1;
      d->ncon++;
    }
  }
  if (d->ncon > m->nconmax) d->ncon = m->nconmax;
  free(mark);
}

int ref_ccd_mesh(int type1, const double* pos1, const double* mat1, const double* size1, const double* vert1, int nvert1, int type2,
                 const double* pos2, const double* mat2, const double* size2, const double* vert2, int nvert2, double margin, double tolerance,
                 double cutoff, int iterations, int multiccd, double* out, double* wit);
int ref_ccd(int type1, const double* pos1, const double* mat1, const double* size1, int type2, const double* pos2, const double* mat2,
            const double* size2, double margin, double tolerance, double cutoff, int iterations, int multiccd, double* out, double* wit) {
  return ref_ccd_mesh(type1, pos1, mat1, size1, NULL, 0, type2, pos2, mat2, size2, NULL, 0, margin, tolerance, cutoff, iterations, multiccd, out, wit);
}
int ref_ccd_mesh(int type1, const double* pos1, const double* mat1, const double* size1, const double* vert1, int nvert1, int type2,
                 const double* pos2, const double* mat2, const double* size2, const double* vert2, int nvert2, double margin, double tolerance,
                 double cutoff, int iterations, int multiccd, double* out, double* wit) {
  CcdGeom a, b;
  a.type = type1; b.type = type2;
  v3cpy(a.pos, pos1); v3cpy(b.pos, pos2);
  memcpy(a.rot, mat1, sizeof(a.rot)); memcpy(b.rot, mat2, sizeof(b.rot));
  v3cpy(a.size, size1); v3cpy(b.size, size2);
  a.margin = b.margin = margin;
  a.vert = vert1; a.nvert = nvert1; b.vert = vert2; b.nvert = nvert2;
  a.index = b.index = a.cache = b.cache = -1;
  a.graph = b.graph = NULL;
  mesh_poly_of(NULL, -1, &a);
  mesh_poly_of(NULL, -1, &b);
  g_mc_cap = 8;
  static Polytope pt;
  int face, overflow = 0;
  int n = ccd_run(tolerance, cutoff, iterations, iterations, a, b, out, out + 1, out + 4, &overflow, &face, &pt);
  if (type1 == G_MESH || type2 == G_MESH) face = -1; /* (no polygon tables on this entry: see ref_ccd_geoms) */
  out[7] = (double)overflow;
  out[8] = (double)face;
  for (int k = 0; k < 3; k++) { wit[k] = out[1 + k]; wit[3 + k] = out[4 + k]; }
  if (multiccd && n > 0 && face >= 0) {
    double w1[4][3], w2[4][3];
    n = ccd_multicontact_box(&pt, face, out + 1, out + 4, &a, &b, w1, w2);
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) { wit[6 * i + k] = w1[i][k]; wit[6 * i + 3 + k] = w2[i][k]; }
  }
  return n;
}

/* ccd on two geoms of a model at given poses, with multi-contact recovery when `multiccd` (collision_gjk_test.py:_geom_dist) */
int ref_ccd_geoms(const RefModel* m, int g1, int g2, const double* pos1, const double* mat1, const double* pos2, const double* mat2, double margin,
                  double tolerance, double cutoff, int iterations, int multiccd, double* out, double* wit) {
  CcdGeom a, b;
  a.type = m->geom_type[g1]; b.type = m->geom_type[g2];
  v3cpy(a.pos, pos1); v3cpy(b.pos, pos2);
  memcpy(a.rot, mat1, sizeof(a.rot)); memcpy(b.rot, mat2, sizeof(b.rot));
  v3cpy(a.size, m->geom_size + 3 * g1); v3cpy(b.size, m->geom_size + 3 * g2);
  a.margin = b.margin = margin;
  mesh_of(m, g1, &a.vert, &a.nvert);
  mesh_of(m, g2, &b.vert, &b.nvert);
  mesh_poly_of(m, g1, &a);
  mesh_poly_of(m, g2, &b);
  g_mc_cap = 8; /* (posed pairs need not be colliding pairs of the model: size the clip buffers from the two geoms, like the reference's harness) */
  for (int p = 0; p < m->nmeshpoly; p++)
    if (2 * m->mesh_polyvertnum[p] > g_mc_cap) g_mc_cap = 2 * m->mesh_polyvertnum[p];
  a.index = b.index = a.cache = b.cache = -1;
  a.graph = mesh_graph_of(m, g1);
  b.graph = mesh_graph_of(m, g2);
  static Polytope pt;
  int face, overflow = 0;
  int n = ccd_run(tolerance, cutoff, iterations, iterations, a, b, out, out + 1, out + 4, &overflow, &face, &pt);
  out[7] = (double)overflow;
  out[8] = (double)face;
  for (int k = 0; k < 3; k++) { wit[k] = out[1 + k]; wit[3 + k] = out[4 + k]; }
  if (multiccd && n > 0 && face >= 0) {
    double w1[4][3], w2[4][3];
    n = ccd_multicontact_box(&pt, face, out + 1, out + 4, &a, &b, w1, w2);
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) { wit[6 * i + k] = w1[i][k]; wit[6 * i + 3 + k] = w2[i][k]; }
  }
  return n;
}

/* ================================================================ constraint.py */
/* _efc_row constraint.py:84-153 */
static void efc_row(const RefModel* m, RefData* d, int r, double pos_aref, double pos_imp, double invweight,
                    const double* solref, const double* solimp, double margin, double vel, double frictionloss,
                    int type, int id) {
  double timeconst = solref[0], dampratio = solref[1];
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (!(m->disableflags & DSBL_REFSAFE)) timeconst = fmax(timeconst, 2.0 * m->timestep);
  dmin = clampd(dmin, MINIMP, MAXIMP);
  dmax = clampd(dmax, MINIMP, MAXIMP);
  width = fmax(MINVAL, width);
  mid = clampd(mid, MINIMP, MAXIMP);
  power = fmax(1.0, power);
  double dmax_sq = dmax * dmax;
  double k = 1.0 / (dmax_sq * timeconst * timeconst * dampratio * dampratio);
  double b = 2.0 / (dmax * timeconst);
  if (solref[0] <= 0) k = -solref[0] / dmax_sq;
  if (solref[1] <= 0) b = -solref[1] / dmax;
  double imp_x = fabs(pos_imp) / width;
  double imp_a = (1.0 / pow(mid, power - 1.0)) * pow(imp_x, power);
  double imp_b = 1.0 - (1.0 / pow(1.0 - mid, power - 1.0)) * pow(1.0 - imp_x, power);
  double imp_y = imp_x < mid ? imp_a : imp_b;
  double imp = dmin + imp_y * (dmax - dmin);
  imp = clampd(imp, dmin, dmax);
  if (imp_x > 1.0) imp = dmax;
  d->efc_D[r] = 1.0 / fmax(invweight * (1.0 - imp) / imp, MINVAL);
  d->efc_vel[r] = vel;
  d->efc_aref[r] = -k * imp * pos_aref - b * vel;
  d->efc_pos[r] = pos_aref + margin;
  d->efc_margin[r] = margin;
  d->efc_frictionloss[r] = frictionloss;
  d->efc_type[r] = type;
  d->efc_id[r] = id;
}

void ref_make_constraint(const RefModel* m, RefData* d) {
  int nv = m->nv, njmax = m->njmax;
  d->ne = d->nf = d->nl = d->nefc = 0;
  if (m->disableflags & DSBL_CONSTRAINT) return;
  int nefc = 0;
  /* equality constraints, joint couplings only: constraint.py:500-640 (_equality_joint) */
  if (!(m->disableflags & DSBL_EQUALITY)) {
    for (int e = 0; e < m->neq; e++) {
      if (!m->eq_active0[e]) continue;
      d->ne++;
      int r = nefc++;
      if (r >= njmax) continue;
      int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
      const double* data = m->eq_data + 11 * e;
      int dof1 = m->jnt_dofadr[j1], qa1 = m->jnt_qposadr[j1];
      memset(d->efc_J + (size_t)r * nv, 0, sizeof(double) * nv);
      d->efc_J[(size_t)r * nv + dof1] = 1.0;
      double pos, vel, invweight;
      if (j2 >= 0) {
        int dof2 = m->jnt_dofadr[j2], qa2 = m->jnt_qposadr[j2];
        double dif = d->qpos[qa2] - m->qpos0[qa2];
        double rhs = data[0] + dif * (data[1] + dif * (data[2] + dif * (data[3] + dif * data[4])));
        double deriv = data[1] + dif * (2.0 * data[2] + dif * (3.0 * data[3] + dif * 4.0 * data[4]));
        pos = d->qpos[qa1] - m->qpos0[qa1] - rhs;
        vel = d->qvel[dof1] - d->qvel[dof2] * deriv;
        invweight = m->dof_invweight0[dof1] + m->dof_invweight0[dof2];
        d->efc_J[(size_t)r * nv + dof2] = -deriv;
      } else {
        pos = d->qpos[qa1] - m->qpos0[qa1] - data[0];
        vel = d->qvel[dof1];
        invweight = m->dof_invweight0[dof1];
      }
      efc_row(m, d, r, pos, pos, invweight, m->eq_solref + 2 * e, m->eq_solimp + 5 * e, 0.0, vel, 0.0, CT_EQUALITY, e);
    }
  }
  /* dof friction: constraint.py:1766-1865 */
  if (!(m->disableflags & DSBL_FRICTIONLOSS)) {
    for (int i = 0; i < nv; i++) {
      if (m->dof_frictionloss[i] <= 0.0) continue;
      d->nf++;
      int r = nefc++;
      if (r >= njmax) continue;
      memset(d->efc_J + (size_t)r * nv, 0, sizeof(double) * nv);
      d->efc_J[(size_t)r * nv + i] = 1.0;
      efc_row(m, d, r, 0.0, 0.0, m->dof_invweight0[i], m->dof_solref + 2 * i, m->dof_solimp + 5 * i, 0.0, d->qvel[i],
              m->dof_frictionloss[i], CT_FRICTION_DOF, i);
    }
  }
  /* joint limits: constraint.py:1991-2105 (slide/hinge), 2107-2240 (ball) */
  if (!(m->disableflags & DSBL_LIMIT)) {
    for (int j = 0; j < m->njnt; j++) {
      if (!m->jnt_limited[j]) continue;
      int t = m->jnt_type[j], dof = m->jnt_dofadr[j], qa = m->jnt_qposadr[j];
      double margin = m->jnt_margin[j];
      if (t == JNT_SLIDE || t == JNT_HINGE) {
        double q = d->qpos[qa];
        double dmin = q - m->jnt_range[2 * j], dmax = m->jnt_range[2 * j + 1] - q;
        double pos = fmin(dmin, dmax) - margin;
        if (!(pos < 0)) continue;
        d->nl++;
        int r = nefc++;
        if (r >= njmax) continue;
        double J = (dmin < dmax) ? 1.0 : -1.0;
        memset(d->efc_J + (size_t)r * nv, 0, sizeof(double) * nv);
        d->efc_J[(size_t)r * nv + dof] = J;
        efc_row(m, d, r, pos, pos, m->dof_invweight0[dof], m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, margin,
                J * d->qvel[dof], 0.0, CT_LIMIT_JOINT, j);
      } else if (t == JNT_BALL) {
        double q[4] = {d->qpos[qa], d->qpos[qa + 1], d->qpos[qa + 2], d->qpos[qa + 3]}, aa[3];
        quat_normalize(q);
        double unit[4] = {1, 0, 0, 0};
        quat_sub(aa, q, unit); /* quat_to_vel(q) */
        double angle = v3normalize(aa);
        double pos = fmax(m->jnt_range[2 * j], m->jnt_range[2 * j + 1]) - angle - margin;
        if (!(pos < 0)) continue;
        d->nl++;
        int r = nefc++;
        if (r >= njmax) continue;
        memset(d->efc_J + (size_t)r * nv, 0, sizeof(double) * nv);
        double vel = 0;
        for (int k = 0; k < 3; k++) { d->efc_J[(size_t)r * nv + dof + k] = -aa[k]; vel -= aa[k] * d->qvel[dof + k]; }
        efc_row(m, d, r, pos, pos, m->dof_invweight0[dof], m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, margin, vel, 0.0,
                CT_LIMIT_JOINT, j);
      }
    }
  }
  /* contacts (pyramidal / frictionless): constraint.py:2641-2757, 3751-3879, 4197-4343 */
  if (!(m->disableflags & DSBL_CONTACT)) {
    double* jacp = (double*)malloc(sizeof(double) * 6 * nv);
    double* jacr = jacp + 3 * nv;
    for (int c = 0; c < d->ncon; c++) {
      double includemargin = d->con_includemargin[c];
      double pos = d->con_dist[c] - includemargin;
      if (!(pos < 0.0)) continue;
      int condim = d->con_dim[c];
      int elliptic = m->cone == 1 && condim > 1;
      int ndim = condim == 1 ? 1 : (elliptic ? condim : 2 * (condim - 1)); /* constraint.py:2698-2704 */
      int base = nefc;
      nefc += ndim;
      int g1 = d->con_geom[2 * c], g2 = d->con_geom[2 * c + 1];
      int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
      /* jac difference: support.py:488-533 */
      memset(jacp, 0, sizeof(double) * 6 * nv);
      for (int side = 0; side < 2; side++) {
        int b = side ? b2 : b1;
        double sgn = side ? 1.0 : -1.0;
        int bb = b;
        while (bb > 0 && m->body_dofnum[bb] == 0) bb = m->body_parentid[bb];
        if (bb == 0) continue;
        double off[3];
        v3sub(off, d->con_pos + 3 * c, d->subtree_com + 3 * m->body_rootid[b]);
        int dof = m->body_dofadr[bb] + m->body_dofnum[bb] - 1;
        while (dof >= 0) {
          double jp[3];
          v3cross(jp, d->cdof + 6 * dof, off);
          v3add(jp, jp, d->cdof + 6 * dof + 3);
          for (int k = 0; k < 3; k++) {
            jacp[k * nv + dof] += sgn * jp[k];
            jacr[k * nv + dof] += sgn * d->cdof[6 * dof + k];
          }
          dof = m->dof_parentid[dof];
        }
      }
      const double* frame = d->con_frame + 9 * c;
      const double* fri = d->con_friction + 5 * c;
      double invweight = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
      if (condim > 1 && !elliptic) {
        double fri0 = fri[0];
        invweight = invweight + fri0 * fri0 * invweight;
        invweight = invweight * 2.0 * fri0 * fri0 / m->impratio;
      }
      for (int dimid = 0; dimid < ndim; dimid++) {
        int r = base + dimid;
        if (r >= njmax) { d->con_efc_address[10 * c + dimid] = -1; continue; }
        d->con_efc_address[10 * c + dimid] = r;
        double* J = d->efc_J + (size_t)r * nv;
        double vel = 0;
        if (elliptic) {
          /* elliptic cone: one row per contact dimension (constraint.py:3836-3847): normal, two tangents (translational Jacobian
           * along the frame axes), then torsion and the two rolling directions (rotational Jacobian); parameters of the
           * friction rows: constraint.py:4277-4294 */
          const double* fr = frame + 3 * (dimid < 3 ? dimid : dimid - 3);
          const double* jac = dimid < 3 ? jacp : jacr;
          for (int i = 0; i < nv; i++) {
            J[i] = fr[0] * jac[i] + fr[1] * jac[nv + i] + fr[2] * jac[2 * nv + i];
            vel += J[i] * d->qvel[i];
          }
          double iw = invweight;
          const double* ref = d->con_solref + 2 * c;
          double pos_aref = pos;
          if (dimid > 0) {
            const double* srf = d->con_solreffriction + 2 * c;
            if (srf[0] != 0.0 || srf[1] != 0.0) ref = srf;
            iw = iw / m->impratio;
            if (dimid > 1) iw *= fri[0] * fri[0] / (fri[dimid - 1] * fri[dimid - 1]);
            pos_aref = 0.0;
          }
          efc_row(m, d, r, pos_aref, pos, iw, ref, d->con_solimp + 5 * c, includemargin, vel, 0.0, CT_CONTACT_ELLIPTIC, c);
          continue;
        }
        for (int i = 0; i < nv; i++) {
          double j0 = frame[0] * jacp[i] + frame[1] * jacp[nv + i] + frame[2] * jacp[2 * nv + i];
          double val = j0;
          if (condim > 1) {
            int dimid2 = dimid / 2 + 1;
            double frii = fri[dimid2 - 1] * (1.0 - 2.0 * (double)(dimid & 1));
            double ji;
            if (dimid2 < 3) {
              const double* fr = frame + 3 * dimid2;
              ji = fr[0] * jacp[i] + fr[1] * jacp[nv + i] + fr[2] * jacp[2 * nv + i];
            } else {
              const double* fr = frame + 3 * (dimid2 - 3);
              ji = fr[0] * jacr[i] + fr[1] * jacr[nv + i] + fr[2] * jacr[2 * nv + i];
            }
            val += ji * frii;
          }
          J[i] = val;
          vel += val * d->qvel[i];
        }
        efc_row(m, d, r, pos, pos, invweight, d->con_solref + 2 * c, d->con_solimp + 5 * c, includemargin, vel, 0.0,
                condim == 1 ? CT_CONTACT_FRICTIONLESS : CT_CONTACT_PYRAMIDAL, c);
      }
    }
    free(jacp);
  }
  d->nefc = nefc;
  if (nefc > njmax) d->overflow |= OVF_NEFC;
}

/* ================================================================ solver.py */
typedef struct {
  int nv, nefc;
  double *Jaref, *jv, *search, *mv, *grad, *Mgrad, *prev_grad, *prev_Mgrad, *H;
} Ctx;

/* _update_constraint_efc solver.py:1698-1822 + _eval_constraint 424-517 (pyramidal), qfrc_constraint 1912-1947 */
static void update_constraint(const RefModel* m, RefData* d, Ctx* c) {
  int nv = c->nv, nefc = c->nefc, ne = d->ne, nf = d->nf;
  for (int r = 0; r < nefc; r++) {
    double jaref = c->Jaref[r], D = d->efc_D[r];
    if (r < ne) { d->efc_force[r] = -D * jaref; d->efc_state[r] = ST_QUADRATIC; }
    else if (r < ne + nf) {
      double f = d->efc_frictionloss[r], rf = safe_div(f, D);
      if (jaref <= -rf) { d->efc_force[r] = f; d->efc_state[r] = ST_LINEARNEG; }
      else if (jaref >= rf) { d->efc_force[r] = -f; d->efc_state[r] = ST_LINEARPOS; }
      else { d->efc_force[r] = -D * jaref; d->efc_state[r] = ST_QUADRATIC; }
    } else if (d->efc_type[r] == CT_CONTACT_ELLIPTIC) {
      /* _eval_constraint solver.py:455-472 + _eval_elliptic_middle 406-421: the rows of one contact decide together */
      int con = d->efc_id[r], r0 = d->con_efc_address[10 * con], dim = d->con_dim[con];
      const double* fri = d->con_friction + 5 * con;
      double mu = fri[0] / sqrt(m->impratio), N = c->Jaref[r0] * mu, TT = 0.0, ufrictionj = 0.0;
      for (int j = 1; j < dim; j++) {
        double uj = c->Jaref[r0 + j] * fri[j - 1];
        TT += uj * uj;
        if (r == r0 + j) ufrictionj = uj * fri[j - 1];
      }
      double T = TT <= 0.0 ? 0.0 : sqrt(TT);
      if (N >= mu * T || (T <= 0.0 && N >= 0.0)) { d->efc_force[r] = 0.0; d->efc_state[r] = ST_SATISFIED; }
      else if (mu * N + T <= 0.0 || (T <= 0.0 && N < 0.0)) { d->efc_force[r] = -D * jaref; d->efc_state[r] = ST_QUADRATIC; }
      else {
        double dm = safe_div(d->efc_D[r0], mu * mu * (1.0 + mu * mu)), fn = -dm * (N - mu * T) * mu;
        d->efc_force[r] = r == r0 ? fn : -safe_div(fn, T) * ufrictionj;
        d->efc_state[r] = ST_CONE;
      }
    } else {
      if (jaref >= 0.0) { d->efc_force[r] = 0.0; d->efc_state[r] = ST_SATISFIED; }
      else { d->efc_force[r] = -D * jaref; d->efc_state[r] = ST_QUADRATIC; }
    }
  }
  for (int i = 0; i < nv; i++) {
    double s = 0;
    for (int r = 0; r < nefc; r++) s += d->efc_J[(size_t)r * nv + i] * d->efc_force[r];
    d->qfrc_constraint[i] = s;
  }
}

static int chol_factor(double* A, int n) { /* lower Cholesky in place; returns rank deficiency flag */
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
    if (s < MINVAL) s = MINVAL;
    double l = sqrt(s);
    A[j * n + j] = l;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / l;
    }
  }
  return 0;
}
static void chol_solve(const double* L, int n, double* x, const double* b) {
  for (int i = 0; i < n; i++) {
    double t = b[i];
    for (int k = 0; k < i; k++) t -= L[i * n + k] * x[k];
    x[i] = t / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double t = x[i];
    for (int k = i + 1; k < n; k++) t -= L[k * n + i] * x[k];
    x[i] = t / L[i * n + i];
  }
}

/* _update_gradient solver.py:3061-3220; Newton H = M + J^T diag(D*active) J (2365-2440), Cholesky (2567-2603) */
static void update_gradient(const RefModel* m, RefData* d, Ctx* c, double* grad_dot, double* search_dot, double* decrement) {
  int nv = c->nv, nefc = c->nefc;
  double gd = 0;
  for (int i = 0; i < nv; i++) {
    c->grad[i] = d->Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
    gd += c->grad[i] * c->grad[i];
  }
  *grad_dot = gd;
  if (m->solver == SOL_CG) {
    ref_solve_m(m, d, c->Mgrad, c->grad);
  } else {
    double* H = c->H;
    memset(H, 0, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) {
      int start = m->M_rowadr[i], diag = start + m->M_rownnz[i] - 1;
      H[i * nv + i] = d->M[diag];
      for (int adr = start; adr < diag; adr++) {
        int j = m->M_colind[adr];
        H[i * nv + j] = H[j * nv + i] = d->M[adr];
      }
    }
    for (int r = 0; r < nefc; r++) {
      if (d->efc_state[r] != ST_QUADRATIC) continue;
      const double* J = d->efc_J + (size_t)r * nv;
      double D = d->efc_D[r];
      for (int i = 0; i < nv; i++) {
        if (J[i] == 0.0) continue;
        double ji = J[i] * D;
        for (int j = 0; j <= i; j++) H[i * nv + j] += ji * J[j];
      }
    }
    /* cone Hessian of the contacts in the middle zone (_update_gradient_JTCJ_dense solver.py:2466-2564) */
    for (int r = 0; r < nefc; r++) {
      if (d->efc_type[r] != CT_CONTACT_ELLIPTIC || d->efc_state[r] != ST_CONE) continue;
      int con = d->efc_id[r], r0 = d->con_efc_address[10 * con], dim = d->con_dim[con];
      if (r != r0) continue;
      const double* fri = d->con_friction + 5 * con;
      double mu = fri[0] / sqrt(m->impratio), mu2 = mu * mu, dm = safe_div(d->efc_D[r0], mu2 * (1.0 + mu2));
      if (dm == 0.0) continue;
      double n = c->Jaref[r0] * mu, tt = 0.0;
      for (int j = 1; j < dim; j++) { double u = c->Jaref[r0 + j] * fri[j - 1]; tt += u * u; }
      double t = fmax(sqrt(tt), MINVAL), ttt = fmax(t * t * t, MINVAL), mu_tinv = safe_div(mu, t);
      double mu_n_ttt = mu * safe_div(n, ttt), tdiag = mu2 - n * mu_tinv;
      for (int i = 0; i < nv; i++)
        for (int k = 0; k <= i; k++) {
          double z01 = mu * d->efc_J[(size_t)r0 * nv + i], z02 = mu * d->efc_J[(size_t)r0 * nv + k], p1 = 0, p2 = 0, td = 0;
          for (int j = 1; j < dim; j++) {
            double sc = fri[j - 1], u = c->Jaref[r0 + j] * sc;
            double z1 = sc * d->efc_J[(size_t)(r0 + j) * nv + i], z2 = sc * d->efc_J[(size_t)(r0 + j) * nv + k];
            p1 += u * z1; p2 += u * z2; td += z1 * z2;
          }
          H[i * nv + k] += dm * (z01 * z02 - mu_tinv * (z01 * p2 + z02 * p1) + mu_n_ttt * p1 * p2 + tdiag * td);
        }
    }
    for (int i = 0; i < nv; i++)
      for (int j = i + 1; j < nv; j++) H[i * nv + j] = H[j * nv + i];
    chol_factor(H, nv);
    chol_solve(H, nv, c->Mgrad, c->grad);
    double sd = 0, dec = 0;
    for (int i = 0; i < nv; i++) { sd += c->Mgrad[i] * c->Mgrad[i]; dec += c->grad[i] * c->Mgrad[i]; }
    for (int i = 0; i < nv; i++) c->search[i] = -c->Mgrad[i];
    *search_dot = sd;
    *decrement = dec;
  }
}

/* per-row (cost - cost0, grad, hess) at alpha: _compute_efc_eval_pt_pyramidal solver.py:518-556 */
static double m_impratio = 1.0; /* set by ref_solve (eval_pt has no model argument) */
static void eval_pt(const RefData* d, const Ctx* c, double alpha, double* out) {
  int ne = d->ne, nf = d->nf;
  double s0 = 0, s1 = 0, s2 = 0;
  for (int r = 0; r < c->nefc; r++) {
    double ja = c->Jaref[r], jv = c->jv[r], D = d->efc_D[r];
    double x = ja + alpha * jv, jvD = jv * D, hess = jv * jvD;
    if (d->efc_type[r] == CT_CONTACT_ELLIPTIC) {
      /* cost(alpha) - cost(0), derivative and curvature of one elliptic contact on the ray (solver.py:272-403; float64 can
       * afford the plain difference of costs where the reference uses shifted forms): evaluated at the contact's first row */
      int con = d->efc_id[r], r0 = d->con_efc_address[10 * con], dim = d->con_dim[con];
      if (r != r0) continue;
      const double* fri = d->con_friction + 5 * con;
      double mu = fri[0] / sqrt(m_impratio), dm = safe_div(d->efc_D[r0], mu * mu * (1.0 + mu * mu));
      double cost[2], grad = 0.0, hs = 0.0;
      for (int pass = 0; pass < 2; pass++) { /* pass 0: alpha = 0 (the reference point), pass 1: alpha */
        double a = pass ? alpha : 0.0;
        double N = (c->Jaref[r0] + a * c->jv[r0]) * mu, v0 = c->jv[r0] * mu, uu = 0, uv = 0, vv = 0, quad = 0, quadg = 0, quadh = 0;
        for (int j = 0; j < dim; j++) {
          double xj = c->Jaref[r0 + j] + a * c->jv[r0 + j], Dj = d->efc_D[r0 + j];
          quad += 0.5 * Dj * xj * xj; quadg += Dj * xj * c->jv[r0 + j]; quadh += Dj * c->jv[r0 + j] * c->jv[r0 + j];
          if (j > 0) { double uj = xj * fri[j - 1], vj = c->jv[r0 + j] * fri[j - 1]; uu += uj * uj; uv += uj * vj; vv += vj * vj; }
        }
        double T = uu <= 0.0 ? 0.0 : sqrt(uu), cst, g = 0.0, h = 0.0;
        if (N >= mu * T || (T <= 0.0 && N >= 0.0)) cst = 0.0;
        else if (mu * N + T <= 0.0 || (T <= 0.0 && N < 0.0)) { cst = quad; g = quadg; h = quadh; }
        else {
          double rr = N - mu * T, T1 = uv / T, T2 = (vv - T1 * T1) / T, r1 = v0 - mu * T1;
          cst = 0.5 * dm * rr * rr; g = dm * rr * r1; h = dm * (r1 * r1 + rr * (-mu * T2));
        }
        cost[pass] = cst;
        if (pass) { grad = g; hs = h; }
      }
      s0 += cost[1] - cost[0]; s1 += grad; s2 += hs;
      continue;
    }
    if (r >= ne + nf) {
      double quad0 = 0.5 * D * ja * ja, cost0 = ja < 0.0 ? quad0 : 0.0;
      if (x < 0.0) { s0 += alpha * (jvD * ja + 0.5 * alpha * hess) + (quad0 - cost0); s1 += jvD * ja + alpha * hess; s2 += hess; }
      else s0 += -cost0;
    } else if (r >= ne) {
      double f = d->efc_frictionloss[r], rf = safe_div(f, D);
      double cost0 = (-rf < ja && ja < rf) ? 0.5 * D * ja * ja : (ja <= -rf ? f * (-0.5 * rf - ja) : f * (-0.5 * rf + ja));
      if (-rf < x && x < rf) { s0 += 0.5 * D * x * x - cost0; s1 += jvD * x; s2 += hess; }
      else if (x <= -rf) { s0 += f * (-0.5 * rf - x) - cost0; s1 += -f * jv; }
      else { s0 += f * (-0.5 * rf + x) - cost0; s1 += f * jv; }
    } else {
      s0 += alpha * (jvD * ja + 0.5 * alpha * hess); s1 += jvD * ja + alpha * hess; s2 += hess;
    }
  }
  out[0] = s0; out[1] = s1; out[2] = s2;
}
static int in_bracket(const double* x, const double* y) { return (x[1] < y[1] && y[1] < 0.0) || (x[1] > y[1] && y[1] > 0.0); }

/* _linesearch_iterative_kernel solver.py:835-1347 (pyramidal, non-incremental) */
static double linesearch(const RefModel* m, RefData* d, Ctx* c, double search_dot, double* improvement_out) {
  int nv = c->nv;
  double gauss1 = 0, gauss2 = 0;
  for (int i = 0; i < nv; i++) {
    gauss1 += c->search[i] * (d->Ma[i] - d->qfrc_smooth[i]);
    gauss2 += 0.5 * c->search[i] * c->mv[i];
  }
  double snorm = sqrt(search_dot), scale = m->meaninertia * (double)nv;
  double gtol = fmax(m->tolerance * m->ls_tolerance * snorm * scale, 1e-6);
  double p0s[3], p0[3], lo_in[3], tmp[3];
  /* alpha = 0 (_compute_efc_eval_pt_alpha_zero_pyramidal 620-647): same as eval_pt(0) but with unshifted cost */
  eval_pt(d, c, 0.0, p0s);
  p0[0] = 0.0; p0[1] = gauss1 + p0s[1]; p0[2] = 2.0 * gauss2 + p0s[2];
  double p0_delta[3] = {0.0, p0[1], p0[2]};
  double lo_alpha_in = -safe_div(p0[1], p0[2]);
  eval_pt(d, c, lo_alpha_in, tmp);
  lo_in[0] = lo_alpha_in * lo_alpha_in * gauss2 + lo_alpha_in * gauss1 + tmp[0];
  lo_in[1] = 2.0 * lo_alpha_in * gauss2 + gauss1 + tmp[1];
  lo_in[2] = 2.0 * gauss2 + tmp[2];
  double alpha = 0.0, improvement = 0.0;
  int converged = fabs(lo_in[1]) < gtol && lo_in[0] < 0.0;
  if (converged) { alpha = lo_alpha_in; improvement = -lo_in[0]; }
  else {
    int lo_less = lo_in[1] < p0[1];
    double lo[3], hi[3], lo_alpha, hi_alpha;
    memcpy(lo, lo_less ? lo_in : p0_delta, sizeof(lo));
    memcpy(hi, lo_less ? p0_delta : lo_in, sizeof(hi));
    lo_alpha = lo_less ? lo_alpha_in : 0.0;
    hi_alpha = lo_less ? 0.0 : lo_alpha_in;
    for (int it = 0; it < m->ls_iterations; it++) {
      double a3[3] = {lo_alpha - safe_div(lo[1], lo[2]), hi_alpha - safe_div(hi[1], hi[2]), 0.5 * (lo_alpha + hi_alpha)};
      double pt[3][3];
      for (int k = 0; k < 3; k++) {
        eval_pt(d, c, a3[k], tmp);
        pt[k][0] = a3[k] * a3[k] * gauss2 + a3[k] * gauss1 + tmp[0];
        pt[k][1] = 2.0 * a3[k] * gauss2 + gauss1 + tmp[1];
        pt[k][2] = 2.0 * gauss2 + tmp[2];
      }
      double *lo_next = pt[0], *hi_next = pt[1], *mid = pt[2];
      int s1 = in_bracket(lo, lo_next);
      if (s1) { memcpy(lo, lo_next, sizeof(lo)); lo_alpha = a3[0]; }
      int s2 = in_bracket(lo, mid);
      if (s2) { memcpy(lo, mid, sizeof(lo)); lo_alpha = a3[2]; }
      int s3 = in_bracket(lo, hi_next);
      if (s3) { memcpy(lo, hi_next, sizeof(lo)); lo_alpha = a3[1]; }
      int swap_lo = s1 || s2 || s3;
      int h1 = in_bracket(hi, hi_next);
      if (h1) { memcpy(hi, hi_next, sizeof(hi)); hi_alpha = a3[1]; }
      int h2 = in_bracket(hi, mid);
      if (h2) { memcpy(hi, mid, sizeof(hi)); hi_alpha = a3[2]; }
      int h3 = in_bracket(hi, lo_next);
      if (h3) { memcpy(hi, lo_next, sizeof(hi)); hi_alpha = a3[0]; }
      int swap_hi = h1 || h2 || h3;
      int ls_done = (!swap_lo && !swap_hi) || (lo[0] < 0.0 && lo[1] < 0.0 && lo[1] > -gtol) || (hi[0] < 0.0 && hi[1] > 0.0 && hi[1] < gtol);
      int improved = lo[0] < 0.0 || hi[0] < 0.0;
      int lo_better = lo[0] < hi[0];
      if (improved) { alpha = lo_better ? lo_alpha : hi_alpha; improvement = -(lo_better ? lo[0] : hi[0]); }
      if (ls_done) { converged = 1; break; }
    }
  }
  if (!converged) d->overflow |= OVF_LS;
  *improvement_out = improvement;
  return alpha;
}

/* Projected Gauss-Seidel on the dual problem  min_f 0.5 f'(A+R)f + f'b,  A = J M^-1 J', R = 1/D, b = J qacc_smooth - aref,
 * with f unbounded on equality rows, |f| <= frictionloss on friction-loss rows and f >= 0 on limit / (pyramidal or
 * frictionless) contact rows.
 *
 * The reference has NO PGS (types.py:502 "unsupported", io.py solver check), so there is no file:line to follow and no
 * reference test: this restates the algorithm MuJoCo C documents and implements (mujoco 3.x, engine_solver.c mj_solPGS with
 * its helpers residual / costChange / dualState / dualFinish, and the PGS branch of engine_forward.c warmstart):
 *   warm start : force = primal constraint update at qacc_warmstart; kept only if its dual cost is <= 0 (the cost of f = 0)
 *   sweep      : for every row i in order: res = b_i + (A+R)_i. f;  f_i <- project(f_i - res / (A+R)_ii);
 *                change = 0.5 delta^2 (A+R)_ii + delta res;  a row whose change is > 1e-10 is restored
 *   stop       : -(sum of changes) / (meaninertia * max(1, nv)) < tolerance, or `iterations` sweeps
 *   finish     : qfrc_constraint = J' f,  qacc = qacc_smooth + M^-1 qfrc_constraint, states from the forces
 * Parity unpinned (like the rest of this oracle); tests/test_oracle.py additionally checks that the PGS fixed point agrees
 * with the Newton solution of the primal problem, which pins the restatement against an independent algorithm. */
/* min 0.5 v'A v + v'b  s.t.  sum_j (v_j / mu_j)^2 <= r^2   (n <= 5; MuJoCo's mju_QCQP family, used by mj_solPGS for the friction part of an
 * elliptic contact): in the scaled variable y = v / mu the constraint is a ball; the unconstrained minimiser if it is inside, otherwise
 * Newton on the multiplier lambda of |y(lambda)|^2 = r^2, y(lambda) = -(As + lambda I)^-1 bs.  Returns 1 when the constraint is active. */
static int qcqp(int n, const double* A, const double* b, const double* mu, double r, double* v) {
  double As[25], bs[5], L[25], y[5], t[5];
  for (int i = 0; i < n; i++) {
    bs[i] = b[i] * mu[i];
    for (int j = 0; j < n; j++) As[i * n + j] = A[i * n + j] * mu[i] * mu[j];
  }
  double la = 0.0;
  int active = 0;
  for (int it = 0; it < 20; it++) {
    memcpy(L, As, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) L[i * n + i] += la;
    chol_factor(L, n);
    for (int i = 0; i < n; i++) t[i] = -bs[i];
    chol_solve(L, n, y, t);
    double val = -r * r;
    for (int i = 0; i < n; i++) val += y[i] * y[i];
    if (val < 1e-10) break; /* inside (or on) the ball */
    active = 1;
    chol_solve(L, n, t, y); /* (As + la I)^-1 y */
    double deriv = 0.0;
    for (int i = 0; i < n; i++) deriv -= 2.0 * y[i] * t[i];
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  for (int i = 0; i < n; i++) v[i] = y[i] * mu[i];
  if (active) { /* round-off: land exactly on the boundary */
    double s = 0.0;
    for (int i = 0; i < n; i++) s += y[i] * y[i];
    if (s > r * r && s > 0.0) for (int i = 0; i < n; i++) v[i] *= r / sqrt(s);
  }
  return active;
}

static void solve_pgs(const RefModel* m, RefData* d, int nefc) {
  int nv = m->nv, ne = d->ne, nf = d->nf;
  double* buf = (double*)calloc((size_t)nefc * nv + (size_t)nefc * nefc + 3 * (size_t)nefc + 2 * (size_t)nv, sizeof(double));
  double *B = buf, *AR = B + (size_t)nefc * nv, *b = AR + (size_t)nefc * nefc, *jar = b + nefc, *ARf = jar + nefc, *y = ARf + nefc, *z = y + nv;
  for (int r = 0; r < nefc; r++) solve_sparse(m, d->qLD, d->qLDiagInv, B + (size_t)r * nv, d->efc_J + (size_t)r * nv);
  for (int r = 0; r < nefc; r++) {
    for (int c = 0; c < nefc; c++) {
      double s = 0;
      for (int i = 0; i < nv; i++) s += d->efc_J[(size_t)r * nv + i] * B[(size_t)c * nv + i];
      AR[(size_t)r * nefc + c] = s;
    }
    AR[(size_t)r * nefc + r] += 1.0 / d->efc_D[r];
    double s = 0, sw = 0;
    for (int i = 0; i < nv; i++) {
      s += d->efc_J[(size_t)r * nv + i] * d->qacc_smooth[i];
      sw += d->efc_J[(size_t)r * nv + i] * d->qacc_warmstart[i];
    }
    b[r] = s - d->efc_aref[r];
    jar[r] = sw - d->efc_aref[r];
  }
  double* f = d->efc_force;
  int warm = !(m->disableflags & DSBL_WARMSTART);
  if (warm) {
    Ctx c;
    c.nv = nv; c.nefc = nefc; c.Jaref = jar;
    update_constraint(m, d, &c);
    double cost = 0;
    for (int r = 0; r < nefc; r++) {
      double s = 0;
      for (int k = 0; k < nefc; k++) s += AR[(size_t)r * nefc + k] * f[k];
      cost += f[r] * (b[r] + 0.5 * s);
    }
    if (cost > 0.0) warm = 0;
  }
  if (!warm) for (int r = 0; r < nefc; r++) f[r] = 0.0;
  double scl = 1.0 / (m->meaninertia * (double)(nv > 1 ? nv : 1));
  for (int iter = 0; iter < m->iterations; iter++) {
    double improvement = 0;
    for (int i = 0; i < nefc; i++) {
      if (i >= ne + nf && d->efc_type[i] == CT_CONTACT_ELLIPTIC) {
        /* elliptic contact: its dim rows are updated together (MuJoCo mj_solPGS): a projected step on the normal force with the friction
         * forces held, then the friction forces as the QCQP  min 0.5 v'A_ff v + v'b_c  inside the cone section  sum (v_j / mu_j)^2 <= f_n^2 */
        int con = d->efc_id[i], dim = d->con_dim[con];
        if (i + dim > nefc) dim = nefc - i;
        const double* fri = d->con_friction + 5 * con;
        double res[6], oldf[6] = {0, 0, 0, 0, 0, 0}, Ac[25], bc[5], v[5], mu[5];
        for (int a = 0; a < dim; a++) {
          res[a] = b[i + a];
          for (int k = 0; k < nefc; k++) res[a] += AR[(size_t)(i + a) * nefc + k] * f[k];
          oldf[a] = f[i + a];
        }
        double fn;
        if (oldf[0] < MINVAL) {
          /* at the apex: leave it along the steepest feasible direction of the cone -- v = (1, -mu_j^2 res_j / s), s = |mu o res_f| (the
           * boundary ray whose friction opposes the friction residual; the normal alone when there is none) -- if the cost decreases there:
           * slope res_0 - s < 0; exact minimisation along v */
          double sres = 0.0, vdir[6], vAv = 0.0;
          for (int a = 1; a < dim; a++) sres += fri[a - 1] * fri[a - 1] * res[a] * res[a];
          sres = sqrt(sres);
          vdir[0] = 1.0;
          for (int a = 1; a < dim; a++) vdir[a] = sres > MINVAL ? -fri[a - 1] * fri[a - 1] * res[a] / sres : 0.0;
          for (int a = 0; a < dim; a++)
            for (int c2 = 0; c2 < dim; c2++) vAv += vdir[a] * AR[(size_t)(i + a) * nefc + i + c2] * vdir[c2];
          double slope = res[0] - sres, t = (slope < 0.0 && vAv >= MINVAL) ? -slope / vAv : 0.0;
          for (int a = 0; a < dim; a++) f[i + a] = t * vdir[a];
          fn = f[i];
        } else { /* ray update: exact minimisation along the current force direction (scales normal and friction together) */
          double vAv = 0.0, vr = 0.0, Av[6];
          for (int a = 0; a < dim; a++) {
            Av[a] = 0.0;
            for (int c2 = 0; c2 < dim; c2++) Av[a] += AR[(size_t)(i + a) * nefc + i + c2] * oldf[c2];
            vAv += oldf[a] * Av[a];
            vr += oldf[a] * res[a];
          }
          double x = vAv >= MINVAL ? -vr / vAv : 0.0;
          if (x < -1.0) x = -1.0; /* (the normal force stays non-negative) */
          for (int a = 0; a < dim; a++) f[i + a] = oldf[a] + x * oldf[a];
          fn = f[i];
        }
        double cur[6]; /* the force reached so far: the friction step below starts from it */
        for (int a = 0; a < dim; a++) cur[a] = f[i + a];
        if (fn < MINVAL) {
          for (int a = 1; a < dim; a++) f[i + a] = 0.0;
        } else {
          for (int a = 1; a < dim; a++) { /* gradient wrt the friction forces = A_ff v + bc, with the normal force fixed at fn */
            mu[a - 1] = fri[a - 1];
            bc[a - 1] = res[a] + AR[(size_t)(i + a) * nefc + i] * (fn - oldf[0]);
            for (int c2 = 1; c2 < dim; c2++) {
              Ac[(a - 1) * (dim - 1) + (c2 - 1)] = AR[(size_t)(i + a) * nefc + i + c2];
              bc[a - 1] -= AR[(size_t)(i + a) * nefc + i + c2] * oldf[c2];
            }
            (void)cur;
          }
          qcqp(dim - 1, Ac, bc, mu, fn, v);
          for (int a = 1; a < dim; a++) f[i + a] = v[a - 1];
        }
        double change = 0.0;
        for (int a = 0; a < dim; a++) {
          double da = f[i + a] - oldf[a], s2 = 0.0;
          for (int c2 = 0; c2 < dim; c2++) s2 += AR[(size_t)(i + a) * nefc + i + c2] * (f[i + c2] - oldf[c2]);
          change += da * (0.5 * s2 + res[a]);
        }
        if (change > 1e-10) {
          for (int a = 0; a < dim; a++) f[i + a] = oldf[a];
          change = 0.0;
        }
        improvement -= change;
        i += dim - 1;
        continue;
      }
      double res = b[i], Aii = AR[(size_t)i * nefc + i], old = f[i];
      for (int k = 0; k < nefc; k++) res += AR[(size_t)i * nefc + k] * f[k];
      double fn = old - res / Aii;
      if (i >= ne && i < ne + nf) {
        double fl = d->efc_frictionloss[i];
        fn = fn < -fl ? -fl : (fn > fl ? fl : fn);
      } else if (i >= ne + nf) {
        if (fn < 0.0) fn = 0.0;
      }
      double delta = fn - old, change = 0.5 * delta * delta * Aii + delta * res;
      if (change > 1e-10) { fn = old; change = 0.0; }
      f[i] = fn;
      improvement -= change;
    }
    d->solver_niter++;
    if (improvement * scl < m->tolerance) break;
  }
  for (int r = 0; r < nefc; r++) {  /* dualState */
    if (r < ne) d->efc_state[r] = ST_QUADRATIC;
    else if (r < ne + nf) {
      double fl = d->efc_frictionloss[r];
      d->efc_state[r] = f[r] <= -fl ? ST_LINEARPOS : (f[r] >= fl ? ST_LINEARNEG : ST_QUADRATIC);
    } else if (d->efc_type[r] == CT_CONTACT_ELLIPTIC) {
      /* per contact: no normal force -> satisfied; friction on the cone's surface -> cone; strictly inside -> quadratic */
      int con = d->efc_id[r], r0 = d->con_efc_address[10 * con], dim = d->con_dim[con];
      const double* fri = d->con_friction + 5 * con;
      double tt = 0.0;
      for (int a = 1; a < dim && r0 + a < nefc; a++) tt += (f[r0 + a] / fri[a - 1]) * (f[r0 + a] / fri[a - 1]);
      d->efc_state[r] = f[r0] <= 0.0 ? ST_SATISFIED : (tt >= f[r0] * f[r0] * (1.0 - 1e-9) ? ST_CONE : ST_QUADRATIC);
    } else d->efc_state[r] = f[r] <= 0.0 ? ST_SATISFIED : ST_QUADRATIC;
  }
  for (int i = 0; i < nv; i++) {  /* dualFinish */
    double s = 0;
    for (int r = 0; r < nefc; r++) s += d->efc_J[(size_t)r * nv + i] * f[r];
    d->qfrc_constraint[i] = y[i] = s;
  }
  solve_sparse(m, d->qLD, d->qLDiagInv, z, y);
  for (int i = 0; i < nv; i++) d->qacc[i] = d->qacc_smooth[i] + z[i];
  ref_mul_m(m, d, d->Ma, d->qacc);
  free(buf);
}

/* solve solver.py:3671-3743, _solver_iteration 3525-3620, init_context 3622-3668 */
static void solve_full(const RefModel* m, RefData* d) {
  int nv = m->nv, nefc = d->nefc < m->njmax ? d->nefc : m->njmax;
  d->solver_niter = 0;
  if (nefc == 0 || nv == 0 || m->njmax == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0.0;
    ref_mul_m(m, d, d->Ma, d->qacc);
    return;
  }
  if (m->solver == SOL_PGS) {
    solve_pgs(m, d, nefc);
    return;
  }
  m_impratio = m->impratio;
  Ctx c;
  c.nv = nv; c.nefc = nefc;
  double* buf = (double*)calloc((size_t)2 * m->njmax + 7 * nv + (size_t)nv * nv, sizeof(double));
  c.Jaref = buf; c.jv = c.Jaref + m->njmax; c.search = c.jv + m->njmax; c.mv = c.search + nv; c.grad = c.mv + nv;
  c.Mgrad = c.grad + nv; c.prev_grad = c.Mgrad + nv; c.prev_Mgrad = c.prev_grad + nv; c.H = c.prev_Mgrad + nv;
  int warm = !(m->disableflags & DSBL_WARMSTART);
  memcpy(d->qacc, warm ? d->qacc_warmstart : d->qacc_smooth, sizeof(double) * nv);
  for (int r = 0; r < nefc; r++) {
    double s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[(size_t)r * nv + i] * d->qacc[i];
    c.Jaref[r] = s - d->efc_aref[r];
  }
  ref_mul_m(m, d, d->Ma, d->qacc);
  double grad_dot = 0, search_dot = 0, decrement = 0;
  update_constraint(m, d, &c);
  update_gradient(m, d, &c, &grad_dot, &search_dot, &decrement);
  if (m->solver == SOL_CG) {
    search_dot = 0;
    for (int i = 0; i < nv; i++) {
      c.search[i] = -c.Mgrad[i];
      search_dot += c.search[i] * c.search[i];
      c.prev_grad[i] = c.grad[i];
      c.prev_Mgrad[i] = c.Mgrad[i];
    }
  }
  double scl = 1.0 / (m->meaninertia * (double)nv);
  for (int iter = 0; iter < m->iterations; iter++) {
    ref_mul_m(m, d, c.mv, c.search);
    for (int r = 0; r < nefc; r++) {
      double s = 0;
      for (int i = 0; i < nv; i++) s += d->efc_J[(size_t)r * nv + i] * c.search[i];
      c.jv[r] = s;
    }
    double improvement;
    double alpha = linesearch(m, d, &c, search_dot, &improvement);
    for (int i = 0; i < nv; i++) { d->qacc[i] += alpha * c.search[i]; d->Ma[i] += alpha * c.mv[i]; }
    for (int r = 0; r < nefc; r++) c.Jaref[r] += alpha * c.jv[r];
    update_constraint(m, d, &c);
    update_gradient(m, d, &c, &grad_dot, &search_dot, &decrement);
    d->solver_niter++;
    double imp = scl * improvement, gradient = scl * sqrt(grad_dot);
    int done;
    if (m->solver == SOL_CG) {
      double num = 0, den = 0;
      for (int i = 0; i < nv; i++) { num += c.grad[i] * (c.Mgrad[i] - c.prev_Mgrad[i]); den += c.prev_grad[i] * c.prev_Mgrad[i]; }
      double beta = fmax(0.0, num / fmax(MINVAL, den));
      done = (imp < m->tolerance) || (gradient < m->tolerance);
      if (!done) {
        search_dot = 0;
        for (int i = 0; i < nv; i++) {
          c.search[i] = -c.Mgrad[i] + beta * c.search[i];
          search_dot += c.search[i] * c.search[i];
          c.prev_grad[i] = c.grad[i];
          c.prev_Mgrad[i] = c.Mgrad[i];
        }
      }
    } else {
      double model_imp = scl * 0.5 * decrement;
      done = (imp < m->tolerance) || (gradient < m->tolerance) || (model_imp < m->tolerance);
    }
    if (done) break;
    if (d->solver_niter == m->iterations) d->overflow |= OVF_ITER;
  }
  free(buf);
}

/* ================================================================ forward.py */
/* solver.py:3671 solve; with sleeping: solver.py:4022 solve_compact.  The reference gathers the awake dofs into dense nvmax-wide
   arrays (M block, J columns, warm start, smooth terms), runs the stock solver there -- with the tolerances rescaled by nv / nvmax
   so that the nv-normalised termination test is the full model's (io.py:1630) -- and scatters back with the sleeping dofs frozen
   at 0.  Because M is block diagonal over kinematic trees, that is the full-size solve with the sleeping dofs' columns of J and
   their warm start, qacc_smooth and qfrc_smooth zeroed (the gradient then vanishes identically on those dofs, so they never
   move); the restatement does it that way on a masked copy and restores the public J afterwards. */
void ref_solve(const RefModel* m, RefData* d) {
  if (!sleep_enabled(m)) {
    solve_full(m, d);
    return;
  }
  int nv = m->nv, nefc = d->nefc < m->njmax ? d->nefc : m->njmax;
  double* save = (double*)malloc(sizeof(double) * ((size_t)(nefc > 0 ? nefc : 1) * nv + nv));
  double* wsave = save + (size_t)(nefc > 0 ? nefc : 1) * nv;
  memcpy(save, d->efc_J, sizeof(double) * (size_t)nefc * nv);
  memcpy(wsave, d->qacc_warmstart, sizeof(double) * nv);
  for (int i = 0; i < nv; i++)
    if (!d->tree_awake[m->dof_treeid[i]]) {
      for (int r = 0; r < nefc; r++) d->efc_J[(size_t)r * nv + i] = 0.0;
      d->qacc_warmstart[i] = 0.0;
    }
  solve_full(m, d);
  memcpy(d->efc_J, save, sizeof(double) * (size_t)nefc * nv);
  memcpy(d->qacc_warmstart, wsave, sizeof(double) * nv);
  for (int i = 0; i < nv; i++)
    if (!d->tree_awake[m->dof_treeid[i]]) { d->qacc[i] = 0.0; d->qfrc_constraint[i] = 0.0; }
  ref_mul_m(m, d, d->Ma, d->qacc); /* _compact_scatter solver.py:4108 */
  free(save);
}

/* ================================================================ constraint islands + sleeping */
/* sleep.py:171-215 update_sleep (flg_staticawake = 0): tree / body / dof awake tables from tree_asleep.  (The reference fills the
   index lists through atomics, in no particular order; here ascending.) */
void ref_update_sleep(const RefModel* m, RefData* d) {
  d->ntree_awake = d->nbody_awake = d->nv_awake = 0;
  for (int t = 0; t < m->ntree; t++) {
    d->tree_awake[t] = d->tree_asleep[t] < 0;
    d->ntree_awake += d->tree_awake[t];
  }
  for (int b = 0; b < m->nbody; b++) {
    int tree = m->body_treeid[b], state;
    if (tree < 0) state = m->body_mocapid[m->body_rootid[b]] >= 0 ? SLEEP_AWAKE : SLEEP_STATIC;
    else state = d->tree_awake[tree] ? SLEEP_AWAKE : SLEEP_ASLEEP;
    d->body_awake[b] = state;
    if (state != SLEEP_ASLEEP) d->body_awake_ind[d->nbody_awake++] = b;
  }
  for (int i = 0; i < m->nv; i++) {
    int b = m->dof_bodyid[i];
    if (m->body_treeid[b] >= 0 && d->body_awake[b] == SLEEP_AWAKE) d->dof_awake_ind[d->nv_awake++] = i;
  }
}
static int sleep_cycle(const RefModel* m, const RefData* d, int treeid) { /* sleep.py:33: smallest tree of the cycle, -1 if awake */
  if (treeid < 0 || treeid >= m->ntree) return -1;
  int smallest = treeid, current = treeid;
  for (int step = 0; step < m->ntree + 1; step++) {
    int next = d->tree_asleep[current];
    if (next < 0 || next >= m->ntree) return -1;
    if (next < smallest) smallest = next;
    current = next;
    if (current == treeid) break;
  }
  return smallest;
}
static int wake_tree(const RefModel* m, RefData* d, int treeid, int wakeval) { /* sleep.py:236: wakes the tree and its cycle */
  if (treeid < 0 || treeid >= m->ntree) return 0;
  int val = d->tree_asleep[treeid];
  if (val < 0) {
    if (wakeval < val) d->tree_asleep[treeid] = wakeval;
    return 0;
  }
  int nwoke = 0, current = treeid;
  for (int step = 0; step < m->ntree + 1; step++) {
    int next = d->tree_asleep[current];
    if (next < 0 || next >= m->ntree) break;
    d->tree_asleep[current] = wakeval;
    nwoke++;
    current = next;
    if (current == treeid) break;
  }
  return nwoke;
}
static int tree_can_sleep(const RefModel* m, const RefData* d, int t, double tol) { /* sleep.py:273 */
  if (m->tree_sleep_policy[t] == POLICY_AUTO_NEVER) return 0;
  for (int b = 0; b < m->nbody; b++)
    if (m->body_treeid[b] == t)
      for (int i = 0; i < 6; i++)
        if (d->xfrc_applied[6 * b + i] != 0.0) return 0;
  int adr = m->tree_dofadr[t], num = m->tree_dofnum[t];
  for (int k = 0; k < num; k++)
    if (d->qfrc_applied[adr + k] != 0.0) return 0;
  for (int k = 0; k < num; k++) {
    double v = d->qvel[adr + k];
    if (tol > 0.0) {
      if (fabs(m->dof_length[adr + k] * v) >= tol) return 0;
    } else if (v != 0.0) return 0;
  }
  return 1;
}
void ref_wake(const RefModel* m, RefData* d) { /* sleep.py:325, 721: user changes (velocity, applied forces) wake a sleeping tree */
  for (int t = 0; t < m->ntree; t++) {
    if (d->tree_asleep[t] < 0) continue;
    if (d->tree_awake[t] == 1 || !tree_can_sleep(m, d, t, 0.0)) wake_tree(m, d, t, K_AWAKE_VAL);
  }
}
void ref_wake_collision(const RefModel* m, RefData* d) { /* sleep.py:367, 744: a contact between an awake and a sleeping tree */
  for (int c = 0; c < d->ncon; c++) {
    int g1 = d->con_geom[2 * c], g2 = d->con_geom[2 * c + 1];
    if (g1 < 0 || g2 < 0) continue;
    int t1 = m->body_treeid[m->geom_bodyid[g1]], t2 = m->body_treeid[m->geom_bodyid[g2]];
    if (t1 < 0 || t2 < 0) continue;
    int a1 = d->tree_awake[t1], a2 = d->tree_awake[t2];
    if (a1 == a2) continue;
    wake_tree(m, d, a1 == 1 ? t2 : t1, a1 == 1 ? d->tree_asleep[t1] : d->tree_asleep[t2]);
  }
}
void ref_wake_equality(const RefModel* m, RefData* d) { /* sleep.py:579, 793 (joint equalities: the only type on this path) */
  for (int e = 0; e < m->neq; e++) {
    if (!m->eq_active0[e]) continue;
    int id1 = m->eq_obj1id[e], id2 = m->eq_obj2id[e];
    int t1 = id1 >= 0 ? m->body_treeid[m->jnt_bodyid[id1]] : -1, t2 = id2 >= 0 ? m->body_treeid[m->jnt_bodyid[id2]] : -1;
    int s1 = t1 >= 0 ? d->tree_awake[t1] : SLEEP_STATIC, s2 = t2 >= 0 ? d->tree_awake[t2] : SLEEP_STATIC;
    if (s1 != SLEEP_ASLEEP && s2 != SLEEP_ASLEEP) continue;
    if (s1 == SLEEP_STATIC || s2 == SLEEP_STATIC) continue;
    if (t1 == t2) continue;
    if (s1 == SLEEP_ASLEEP && s2 == SLEEP_ASLEEP) {
      if (sleep_cycle(m, d, t1) != sleep_cycle(m, d, t2)) {
        wake_tree(m, d, t1, K_AWAKE_VAL);
        wake_tree(m, d, t2, K_AWAKE_VAL);
      }
    } else wake_tree(m, d, s1 == SLEEP_ASLEEP ? t1 : t2, K_AWAKE_VAL);
  }
}
/* island.py:28-310: tree-tree edges from the constraint rows, flood fill in the order of the smallest tree; trees without rows
   keep island -1 */
void ref_island(const RefModel* m, RefData* d) {
  int nt = m->ntree, nv = m->nv;
  d->nisland = 0;
  if (nt == 0) return;
  unsigned char* tt = (unsigned char*)calloc((size_t)nt * nt, 1);
  int nefc = d->nefc < m->njmax ? d->nefc : m->njmax;
  for (int r = 0; r < nefc; r++) {
    int type = d->efc_type[r], id = d->efc_id[r], t0 = -1, t1 = -1, generic = 0;
    if (type == CT_EQUALITY) generic = 1; /* (connect / weld would use their bodies; joint equalities scan the row) */
    else if (type == CT_FRICTION_DOF) t0 = m->dof_treeid[id];
    else if (type == CT_LIMIT_JOINT) t0 = m->dof_treeid[m->jnt_dofadr[id]];
    else if (type == CT_CONTACT_FRICTIONLESS || type == CT_CONTACT_PYRAMIDAL || type == CT_CONTACT_ELLIPTIC) {
      t0 = m->body_treeid[m->geom_bodyid[d->con_geom[2 * id]]];
      t1 = m->body_treeid[m->geom_bodyid[d->con_geom[2 * id + 1]]];
    } else generic = 1;
    if (!generic) {
      if (t0 < 0 && t1 >= 0) { t0 = t1; t1 = -1; }
      if (t0 >= 0) {
        if (t1 < 0 || t0 == t1) tt[t0 * nt + t0] = 1;
        else tt[t0 * nt + t1] = tt[t1 * nt + t0] = 1;
      }
      continue;
    }
    int first = -1, cross = 0;
    for (int i = 0; i < nv; i++) {
      if (d->efc_J[(size_t)r * nv + i] == 0.0) continue;
      int t = m->dof_treeid[i];
      if (t < 0) continue;
      if (first == -1) first = t;
      else if (t != first) { tt[first * nt + t] = tt[t * nt + first] = 1; cross = 1; }
    }
    if (first >= 0 && !cross) tt[first * nt + first] = 1;
  }
  int* stack = (int*)malloc(sizeof(int) * ((size_t)nt * nt + 1));
  for (int t = 0; t < nt; t++) d->tree_island[t] = -1;
  for (int i = 0; i < nt; i++) {
    if (d->tree_island[i] != -1) continue;
    int has = 0;
    for (int j = 0; j < nt; j++) has |= tt[i * nt + j];
    if (!has) continue;
    int ns = 0;
    stack[ns++] = i;
    while (ns > 0) {
      int v = stack[--ns];
      if (d->tree_island[v] != -1) continue;
      d->tree_island[v] = d->nisland;
      for (int n = 0; n < nt; n++)
        if (tt[v * nt + n] && d->tree_island[n] == -1) stack[ns++] = n;
    }
    d->nisland++;
  }
  free(stack);
  free(tt);
}
void ref_sleep(const RefModel* m, RefData* d) { /* sleep.py:824-999 */
  int nt = m->ntree;
  for (int t = 0; t < nt; t++) { /* 1. awake trees count towards sleep while they could sleep */
    int val = d->tree_asleep[t];
    if (val >= 0) continue;
    if (tree_can_sleep(m, d, t, m->sleep_tolerance)) {
      if (val < -1) d->tree_asleep[t] = val + 1;
    } else d->tree_asleep[t] = K_AWAKE_VAL;
  }
  int* can = (int*)malloc(sizeof(int) * (nt + 1)); /* 2. an island sleeps only when every tree in it is ready */
  for (int k = 0; k < nt; k++) can[k] = 1;
  for (int t = 0; t < nt; t++) {
    int isl = d->tree_island[t];
    if (isl >= 0 && isl < d->nisland && d->tree_asleep[t] < -1) can[isl] = 0;
  }
  for (int isl = 0; isl < d->nisland; isl++) { /* 3. sleep cycles of sleeping islands */
    if (!can[isl]) continue;
    int first = -1, prev = -1;
    for (int t = 0; t < nt; t++) {
      if (d->tree_island[t] != isl) continue;
      if (first == -1) first = t;
      if (prev != -1) d->tree_asleep[prev] = t;
      prev = t;
      for (int k = 0; k < m->tree_dofnum[t]; k++) d->qvel[m->tree_dofadr[t] + k] = d->qacc[m->tree_dofadr[t] + k] = 0.0;
    }
    if (first != -1) d->tree_asleep[prev] = first;
  }
  for (int t = 0; t < nt; t++) { /* unconstrained trees sleep on their own */
    int isl = d->tree_island[t];
    if (isl < 0 || isl >= d->nisland) {
      if (d->tree_asleep[t] == -1) d->tree_asleep[t] = t;
      if (d->tree_asleep[t] >= 0)
        for (int k = 0; k < m->tree_dofnum[t]; k++) d->qvel[m->tree_dofadr[t] + k] = d->qacc[m->tree_dofadr[t] + k] = 0.0;
    }
  }
  free(can);
}

/* ================================================================ rays (ray.py) */
/* a x^2 + 2 b x + c = 0 (ray.py:105-125): smallest non-negative root, both roots in xx */
static double ray_quad(double a, double b, double c, double* xx) {
  xx[0] = xx[1] = -1.0;
  double det = b * b - a * c;
  if (det < MINVAL) return -1.0;
  det = sqrt(det);
  double den = 1.0 / (a != 0.0 ? a : MINVAL);
  xx[0] = (-b - det) * den;
  xx[1] = (-b + det) * den;
  if (xx[0] >= 0.0) return xx[0];
  if (xx[1] >= 0.0) return xx[1];
  return -1.0;
}
static void ray_map(const double* pos, const double* mat, const double* pnt, const double* vec, double* lpnt, double* lvec) { /* ray.py:32-49 */
  double dif[3];
  v3sub(dif, pnt, pos);
  matT_mul_vec(lpnt, mat, dif);
  matT_mul_vec(lvec, mat, vec);
}
static double ray_sphere(const double* pos, double dist_sqr, const double* pnt, const double* vec, double* normal) { /* ray.py:237-251 */
  double dif[3], xx[2];
  v3sub(dif, pnt, pos);
  double sol = ray_quad(v3dot(vec, vec), v3dot(vec, dif), v3dot(dif, dif) - dist_sqr, xx);
  normal[0] = normal[1] = normal[2] = 0.0;
  if (sol >= 0.0) {
    for (int k = 0; k < 3; k++) normal[k] = pnt[k] + vec[k] * sol - pos[k];
    v3normalize(normal);
  }
  return sol;
}
static double ray_geom(int type, const double* pos, const double* mat, const double* size, const double* pnt, const double* vec, double* normal) {
  double lp[3], lv[3], xx[2], n[3] = {0, 0, 0}, x = -1.0;
  normal[0] = normal[1] = normal[2] = 0.0;
  if (type == G_SPHERE) return ray_sphere(pos, size[0] * size[0], pnt, vec, normal);
  if (type != G_PLANE && type != G_CAPSULE && type != G_ELLIPSOID && type != G_CYLINDER && type != G_BOX) return -1.0;
  /* bounding spheres (ray.py:257-261, 362-366, 425-429) */
  double bound = type == G_CAPSULE ? (size[0] + size[1]) * (size[0] + size[1]) : (type == G_CYLINDER ? size[0] * size[0] + size[1] * size[1] : (type == G_BOX ? v3dot(size, size) : -1.0));
  if (bound >= 0.0 && ray_sphere(pos, bound, pnt, vec, n) < 0.0) return -1.0;
  ray_map(pos, mat, pnt, vec, lp, lv);
  if (type == G_PLANE) { /* ray.py:213-234 */
    if (lv[2] > -MINVAL) return -1.0;
    x = -lp[2] / lv[2];
    if (x < 0.0) return -1.0;
    double p0 = lp[0] + x * lv[0], p1 = lp[1] + x * lv[1];
    if ((size[0] <= 0.0 || fabs(p0) <= size[0]) && (size[1] <= 0.0 || fabs(p1) <= size[1])) {
      normal[0] = mat[2], normal[1] = mat[5], normal[2] = mat[8];
      return x;
    }
    return -1.0;
  }
  if (type == G_ELLIPSOID) { /* ray.py:328-356 */
    double s[3], slv[3], slp[3];
    for (int k = 0; k < 3; k++) {
      double q = size[k] * size[k];
      s[k] = 1.0 / (q != 0.0 ? q : MINVAL);
      slv[k] = s[k] * lv[k];
      slp[k] = s[k] * lp[k];
    }
    x = ray_quad(v3dot(slv, lv), v3dot(slv, lp), v3dot(slp, lp) - 1.0, xx);
    if (x >= 0.0) {
      for (int k = 0; k < 3; k++) n[k] = s[k] * (lp[k] + lv[k] * x);
      v3normalize(n);
      mat_mul_vec(normal, mat, n);
    }
    return x;
  }
  if (type == G_CAPSULE) { /* ray.py:254-325 */
    int part = 0;
    double r2 = size[0] * size[0], a = lv[0] * lv[0] + lv[1] * lv[1];
    double sol = ray_quad(a, lv[0] * lp[0] + lv[1] * lp[1], lp[0] * lp[0] + lp[1] * lp[1] - r2, xx);
    if (sol >= 0.0 && fabs(lp[2] + sol * lv[2]) <= size[1]) x = sol;
    a += lv[2] * lv[2];
    for (int cap = 1; cap >= -1; cap -= 2) {
      double ld[3] = {lp[0], lp[1], lp[2] - cap * size[1]};
      ray_quad(a, v3dot(lv, ld), v3dot(ld, ld) - r2, xx);
      for (int i = 0; i < 2; i++) {
        int outer = cap > 0 ? lp[2] + xx[i] * lv[2] >= size[1] : lp[2] + xx[i] * lv[2] <= -size[1];
        if (xx[i] >= 0.0 && outer && (x < 0.0 || xx[i] < x)) x = xx[i], part = cap;
      }
    }
    if (x >= 0.0) {
      n[0] = lp[0] + lv[0] * x, n[1] = lp[1] + lv[1] * x, n[2] = part == 0 ? 0.0 : lp[2] + lv[2] * x - size[1] * part;
      v3normalize(n);
      mat_mul_vec(normal, mat, n);
    }
    return x;
  }
  if (type == G_CYLINDER) { /* ray.py:359-417 */
    int part = 0;
    if (fabs(lv[2]) > MINVAL)
      for (int side = -1; side <= 1; side += 2) {
        double sol = (side * size[1] - lp[2]) / lv[2];
        if (sol >= 0.0) {
          double p0 = lp[0] + sol * lv[0], p1 = lp[1] + sol * lv[1];
          if (p0 * p0 + p1 * p1 <= size[0] * size[0] && (x < 0.0 || sol < x)) x = sol, part = side;
        }
      }
    double sol = ray_quad(lv[0] * lv[0] + lv[1] * lv[1], lv[0] * lp[0] + lv[1] * lp[1], lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0], xx);
    if (sol >= 0.0 && fabs(lp[2] + sol * lv[2]) <= size[1] && (x < 0.0 || sol < x)) x = sol, part = 0;
    if (x >= 0.0) {
      if (part == 0) {
        n[0] = lp[0] + lv[0] * x, n[1] = lp[1] + lv[1] * x, n[2] = 0.0;
        v3normalize(n);
      } else n[0] = n[1] = 0.0, n[2] = part;
      mat_mul_vec(normal, mat, n);
    }
    return x;
  }
  /* box (ray.py:420-471) */
  int face_axis = -1, face_side = -1;
  for (int i = 0; i < 3; i++) {
    if (!(fabs(lv[i]) > MINVAL)) continue;
    for (int side = -1; side <= 1; side += 2) {
      double sol = (side * size[i] - lp[i]) / lv[i];
      if (sol < 0.0) continue;
      int id0 = i == 0 ? 1 : 0, id1 = i == 2 ? 1 : 2;
      if (fabs(lp[id0] + sol * lv[id0]) <= size[id0] && fabs(lp[id1] + sol * lv[id1]) <= size[id1] && (x < 0.0 || sol < x)) x = sol, face_axis = i, face_side = side;
    }
  }
  if (x >= 0.0) {
    n[0] = n[1] = n[2] = 0.0;
    n[face_axis] = face_side;
    mat_mul_vec(normal, mat, n);
  }
  return x;
}
static int ray_eliminate(const RefModel* m, int g, const double* gg, int flg_static, int bodyexclude) { /* ray.py:52-102 */
  int b = m->geom_bodyid[g], mat = m->geom_matid[g];
  if (b == bodyexclude) return 1;
  if (mat < 0 && m->geom_rgba[4 * g + 3] == 0.0) return 1;
  if (mat >= 0 && m->mat_rgba[4 * mat + 3] == 0.0) return 1;
  if (!flg_static && m->body_weldid[b] == 0) return 1;
  if (!gg || (gg[0] == -1 && gg[1] == -1 && gg[2] == -1 && gg[3] == -1 && gg[4] == -1 && gg[5] == -1)) return 0;
  int grp = m->geom_group[g] < 0 ? 0 : (m->geom_group[g] > 5 ? 5 : m->geom_group[g]);
  return gg[grp] == 0.0;
}
double ref_ray(const RefModel* m, const RefData* d, const double* pnt, const double* vec, const double* geomgroup, int flg_static, int bodyexclude, int* geomid,
               double* normal) {
  double best = MAXVAL, n[3], nb[3] = {0, 0, 0};
  int gid = -1;
  for (int g = 0; g < m->ngeom; g++) {
    if (ray_eliminate(m, g, geomgroup, flg_static, bodyexclude)) continue;
    double dist = ray_geom(m->geom_type[g], d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_size + 3 * g, pnt, vec, n);
    if (dist >= 0.0 && dist < best) best = dist, gid = g, v3cpy(nb, n);
  }
  if (geomid) *geomid = gid;
  if (normal) v3cpy(normal, nb);
  return best >= MAXVAL ? -1.0 : best;
}

/* ================================================================ sensors (sensor.py, subset) */
enum { SENS_TOUCH = 0, SENS_MAGNETOMETER = 6, SENS_ACCELEROMETER = 1, SENS_FORCE = 4, SENS_TORQUE = 5, SENS_FRAMELINACC = 33, SENS_FRAMEANGACC = 34, SENS_VELOCIMETER = 2, SENS_GYRO = 3, SENS_JOINTPOS = 9, SENS_JOINTVEL = 10, SENS_ACTUATORPOS = 13, SENS_ACTUATORVEL = 14, SENS_ACTUATORFRC = 15,
       SENS_BALLQUAT = 18, SENS_BALLANGVEL = 19, SENS_FRAMEPOS = 26, SENS_FRAMEQUAT = 27, SENS_FRAMEXAXIS = 28, SENS_FRAMEYAXIS = 29, SENS_FRAMEZAXIS = 30,
       SENS_FRAMELINVEL = 31, SENS_FRAMEANGVEL = 32, SENS_SUBTREECOM = 35, SENS_SUBTREELINVEL = 36, SENS_SUBTREEANGMOM = 37, SENS_CLOCK = 45,
       SENS_JOINTACTFRC = 16, SENS_JOINTLIMITPOS = 20, SENS_JOINTLIMITVEL = 21, SENS_JOINTLIMITFRC = 22, SENS_E_POTENTIAL = 43, SENS_E_KINETIC = 44, SENS_RANGEFINDER = 7 };
enum { OBJ_BODY = 1, OBJ_XBODY = 2, OBJ_GEOM = 5, OBJ_SITE = 6 };
/* pose, quaternion and body of a frame object (sensor.py:266-374 _get_pos / _get_mat / _get_quat / _get_body_id; sites are posed here: the
   oracle keeps no site arrays) */
static int frame_of(const RefModel* m, const RefData* d, int objtype, int id, double* pos, double* mat, double* quat) {
  double q[4] = {1, 0, 0, 0}, p[3] = {0, 0, 0};
  int body = 0;
  if (objtype == OBJ_BODY) {
    body = id;
    v3cpy(p, d->xipos + 3 * id);
    mul_quat(q, d->xquat + 4 * id, m->body_iquat + 4 * id);
  } else if (objtype == OBJ_XBODY) {
    body = id;
    v3cpy(p, d->xpos + 3 * id);
    memcpy(q, d->xquat + 4 * id, sizeof(q));
  } else if (objtype == OBJ_GEOM) {
    body = m->geom_bodyid[id];
    v3cpy(p, d->geom_xpos + 3 * id);
    mul_quat(q, d->xquat + 4 * body, m->geom_quat + 4 * id);
  } else if (objtype == OBJ_SITE) {
    body = m->site_bodyid[id];
    rot_vec_quat(p, m->site_pos + 3 * id, d->xquat + 4 * body);
    v3add(p, p, d->xpos + 3 * body);
    mul_quat(q, d->xquat + 4 * body, m->site_quat + 4 * id);
  }
  if (pos) v3cpy(pos, p);
  if (quat) memcpy(quat, q, sizeof(q));
  if (mat) {
    if (objtype == OBJ_BODY) memcpy(mat, d->ximat + 9 * id, 9 * sizeof(double));
    else if (objtype == OBJ_XBODY) memcpy(mat, d->xmat + 9 * id, 9 * sizeof(double));
    else if (objtype == OBJ_GEOM) memcpy(mat, d->geom_xmat + 9 * id, 9 * sizeof(double));
    else { quat_normalize(q); quat_to_mat(mat, q); }
  }
  return body;
}
/* linear and angular velocity of a frame object in world coordinates (sensor.py:1066-1105 _cvel_offset + 1196-1200) */
static void frame_vel(const RefModel* m, const RefData* d, int objtype, int id, double* lin, double* ang) {
  double pos[3], off[3], c[3];
  int body = frame_of(m, d, objtype, id, pos, NULL, NULL);
  const double* cv = d->cvel + 6 * body;
  v3sub(off, pos, d->subtree_com + 3 * m->body_rootid[body]);
  v3cpy(ang, cv);
  v3cross(c, off, cv);
  v3sub(lin, cv + 3, c);
}
static void sensor_write(const RefModel* m, RefData* d, int i, const double* v) { /* sensor.py:57-114: cutoff for real / positive data */
  int adr = m->sensor_adr[i], dt = m->sensor_datatype[i];
  double cut = m->sensor_cutoff[i];
  for (int k = 0; k < m->sensor_dim[i]; k++) {
    double x = v[k];
    if (cut > 0.0 && dt == 0) x = clampd(x, -cut, cut);
    else if (cut > 0.0 && dt == 1) x = x < cut ? x : cut;
    d->sensordata[adr + k] = x;
  }
}
/* smooth.py:1519-1826 rne_postconstraint: cacc, cfrc_ext, cfrc_int with the constraint forces in (spatial vectors: torque first).
   Joint equalities -- the only equality type on this path -- put nothing into cfrc_ext (only connect / weld do, smooth.py:1570). */
void ref_rne_postconstraint(const RefModel* m, RefData* d) {
  int nb = m->nbody, nefc = d->nefc < m->njmax ? d->nefc : m->njmax;
  for (int k = 0; k < 6 * nb; k++) d->cfrc_ext[k] = 0.0;
  for (int b = 1; b < nb; b++) { /* support.transform_force(xfrc_applied, subtree_com - xipos) */
    const double* xf = d->xfrc_applied + 6 * b;
    double off[3], c[3];
    v3sub(off, d->subtree_com + 3 * m->body_rootid[b], d->xipos + 3 * b);
    v3cross(c, off, xf);
    for (int k = 0; k < 3; k++) { d->cfrc_ext[6 * b + k] = xf[3 + k] - c[k]; d->cfrc_ext[6 * b + 3 + k] = xf[k]; }
  }
  for (int ci = 0; ci < d->ncon; ci++) { /* _cfrc_ext_contact with support.contact_force_fn (support.py:311-397) */
    int b1 = m->geom_bodyid[d->con_geom[2 * ci]], b2 = m->geom_bodyid[d->con_geom[2 * ci + 1]], adr0 = d->con_efc_address[10 * ci], condim = d->con_dim[ci];
    if ((b1 == 0 && b2 == 0) || adr0 < 0) continue;
    double f[6] = {0, 0, 0, 0, 0, 0};
    if (m->cone == 0) {
      if (condim == 1) f[0] = adr0 < nefc ? d->efc_force[adr0] : 0.0;
      else
        for (int i = 0; i < condim - 1; i++) {
          int a = adr0 + 2 * i;
          double d1 = a < m->njmax ? d->efc_force[a] : 0.0, d2 = a + 1 < m->njmax ? d->efc_force[a + 1] : 0.0;
          f[0] += d1 + d2;
          f[i + 1] = (d1 - d2) * d->con_friction[5 * ci + i];
        }
    } else {
      for (int i = 0; i < condim; i++)
        if (adr0 + i < m->njmax) f[i] = d->efc_force[adr0 + i];
    }
    const double* R = d->con_frame + 9 * ci;
    double force[3], torque[3];
    for (int k = 0; k < 3; k++) {
      force[k] = f[0] * R[k] + f[1] * R[3 + k] + f[2] * R[6 + k];
      torque[k] = f[3] * R[k] + f[4] * R[3 + k] + f[5] * R[6 + k];
    }
    for (int side = 0; side < 2; side++) {
      int b = side ? b2 : b1;
      if (!b) continue;
      double off[3], c[3], sg = side ? 1.0 : -1.0;
      v3sub(off, d->subtree_com + 3 * m->body_rootid[b], d->con_pos + 3 * ci);
      v3cross(c, off, force);
      for (int k = 0; k < 3; k++) { d->cfrc_ext[6 * b + k] += sg * (torque[k] - c[k]); d->cfrc_ext[6 * b + 3 + k] += sg * force[k]; }
    }
  }
  for (int b = 0; b < nb; b++) {
    double* a = d->cacc + 6 * b;
    if (b == 0) {
      for (int k = 0; k < 6; k++) a[k] = 0.0;
      if (!(m->disableflags & DSBL_GRAVITY))
        for (int k = 0; k < 3; k++) a[3 + k] = -m->gravity[k];
      for (int k = 0; k < 6; k++) d->cfrc_int[k] = 0.0;
      continue;
    }
    memcpy(a, d->cacc + 6 * m->body_parentid[b], 6 * sizeof(double));
    for (int j = 0; j < m->body_dofnum[b]; j++) {
      int dof = m->body_dofadr[b] + j;
      for (int k = 0; k < 6; k++) a[k] += d->cdof_dot[6 * dof + k] * d->qvel[dof] + d->cdof[6 * dof + k] * d->qacc[dof];
    }
    double f1[6], iv[6], f2[6];
    inert_vec(f1, d->cinert + 10 * b, a);
    inert_vec(iv, d->cinert + 10 * b, d->cvel + 6 * b);
    motion_cross_force(f2, d->cvel + 6 * b, iv);
    for (int k = 0; k < 6; k++) d->cfrc_int[6 * b + k] = f1[k] + f2[k] - d->cfrc_ext[6 * b + k];
  }
  for (int b = nb - 1; b > 0; b--)
    for (int k = 0; k < 6; k++) d->cfrc_int[6 * m->body_parentid[b] + k] += d->cfrc_int[6 * b + k];
}
/* acceleration of a body's frame at the tree's centre of mass: smooth.py:1354-1426 (rne_postconstraint's cacc, flg_acc) */
static void body_cacc(const RefModel* m, const RefData* d, int body, double* cacc) {
  for (int k = 0; k < 6; k++) cacc[k] = 0.0;
  if (!(m->disableflags & DSBL_GRAVITY))
    for (int k = 0; k < 3; k++) cacc[3 + k] = -m->gravity[k];
  int bb = body;
  while (bb > 0 && m->body_dofnum[bb] == 0) bb = m->body_parentid[bb];
  if (bb == 0) return;
  for (int dof = m->body_dofadr[bb] + m->body_dofnum[bb] - 1; dof >= 0; dof = m->dof_parentid[dof])
    for (int k = 0; k < 6; k++) cacc[k] += d->cdof_dot[6 * dof + k] * d->qvel[dof] + d->cdof[6 * dof + k] * d->qacc[dof];
}
/* does the ray pnt + t vec, t >= 0, meet a site's shape (sensor.py:2063-2139 asks ray.ray_geom for a non-negative distance; only the
   yes / no matters, so the conventions of the reference's ray routines do not enter) */
static int ray_cyl(const double* p, const double* v, double r, double hh) {
  double a = v[0] * v[0] + v[1] * v[1], b = p[0] * v[0] + p[1] * v[1], c = p[0] * p[0] + p[1] * p[1] - r * r, t0 = -1e300, t1 = 1e300;
  if (a < 1e-300) { if (c > 0.0) return 0; }
  else {
    double disc = b * b - a * c;
    if (disc < 0.0) return 0;
    t0 = (-b - sqrt(disc)) / a;
    t1 = (-b + sqrt(disc)) / a;
  }
  if (fabs(v[2]) < 1e-300) { if (fabs(p[2]) > hh) return 0; }
  else {
    double ta = (-hh - p[2]) / v[2], tb = (hh - p[2]) / v[2];
    t0 = fmax(t0, fmin(ta, tb));
    t1 = fmin(t1, fmax(ta, tb));
  }
  return t1 >= fmax(t0, 0.0);
}
static int ray_sph(const double* p, const double* v, double r) {
  double a = v3dot(v, v), b = v3dot(p, v), c = v3dot(p, p) - r * r, disc = b * b - a * c;
  return a > 0.0 && disc >= 0.0 && (-b + sqrt(disc)) >= 0.0;
}
static int ray_hits_zone(int type, const double* size, const double* pos, const double* mat, const double* pnt, const double* vec) {
  double dif[3], p[3], v[3];
  v3sub(dif, pnt, pos);
  matT_mul_vec(p, mat, dif);
  matT_mul_vec(v, mat, vec);
  if (type == G_SPHERE) return ray_sph(p, v, size[0]);
  if (type == G_ELLIPSOID) {
    double ps[3] = {p[0] / size[0], p[1] / size[1], p[2] / size[2]}, vs[3] = {v[0] / size[0], v[1] / size[1], v[2] / size[2]};
    return ray_sph(ps, vs, 1.0);
  }
  if (type == G_CYLINDER) return ray_cyl(p, v, size[0], size[1]);
  if (type == G_CAPSULE) {
    double pa[3] = {p[0], p[1], p[2] - size[1]}, pb[3] = {p[0], p[1], p[2] + size[1]};
    return ray_cyl(p, v, size[0], size[1]) || ray_sph(pa, v, size[0]) || ray_sph(pb, v, size[0]);
  }
  if (type == G_BOX) {
    double t0 = -1e300, t1 = 1e300;
    for (int k = 0; k < 3; k++) {
      if (fabs(v[k]) < 1e-300) { if (fabs(p[k]) > size[k]) return 0; }
      else {
        double ta = (-size[k] - p[k]) / v[k], tb = (size[k] - p[k]) / v[k];
        t0 = fmax(t0, fmin(ta, tb));
        t1 = fmin(t1, fmax(ta, tb));
      }
    }
    return t1 >= fmax(t0, 0.0);
  }
  return 0;
}
/* smooth.py:3502-3662 subtree_vel: velocity of every subtree's centre of mass, angular momentum of every subtree about it */
void ref_subtree_vel(const RefModel* m, RefData* d) {
  int nb = m->nbody;
  double* lin = (double*)malloc(sizeof(double) * 3 * nb); /* velocity of each body's own centre of mass */
  for (int b = 0; b < nb; b++) {
    const double* cv = d->cvel + 6 * b;
    double off[3], c[3], dv[3];
    v3sub(off, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
    v3cross(c, off, cv);
    v3sub(lin + 3 * b, cv + 3, c);
    for (int k = 0; k < 3; k++) d->subtree_linvel[3 * b + k] = m->body_mass[b] * lin[3 * b + k];
    matT_mul_vec(dv, d->ximat + 9 * b, cv);
    for (int k = 0; k < 3; k++) dv[k] *= m->body_inertia[3 * b + k];
    mat_mul_vec(d->subtree_angmom + 3 * b, d->ximat + 9 * b, dv);
  }
  for (int b = nb - 1; b >= 0; b--) {
    if (b > 0) v3add(d->subtree_linvel + 3 * m->body_parentid[b], d->subtree_linvel + 3 * m->body_parentid[b], d->subtree_linvel + 3 * b);
    double s = 1.0 / fmax(MINVAL, m->body_subtreemass[b]);
    for (int k = 0; k < 3; k++) d->subtree_linvel[3 * b + k] *= s;
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    double dx[3], dv[3], dL[3];
    v3sub(dx, d->xipos + 3 * b, d->subtree_com + 3 * b);
    for (int k = 0; k < 3; k++) dv[k] = (lin[3 * b + k] - d->subtree_linvel[3 * b + k]) * m->body_mass[b];
    v3cross(dL, dx, dv);
    v3add(d->subtree_angmom + 3 * b, d->subtree_angmom + 3 * b, dL);
    v3add(d->subtree_angmom + 3 * p, d->subtree_angmom + 3 * p, d->subtree_angmom + 3 * b);
    v3sub(dx, d->subtree_com + 3 * b, d->subtree_com + 3 * p);
    for (int k = 0; k < 3; k++) dv[k] = (d->subtree_linvel[3 * b + k] - d->subtree_linvel[3 * p + k]) * m->body_subtreemass[b];
    v3cross(dL, dx, dv);
    v3add(d->subtree_angmom + 3 * p, d->subtree_angmom + 3 * p, dL);
  }
  free(lin);
}
/* stage 0: position / velocity stage sensors and actuator forces; stage 1: acceleration stage (accelerometer, frame accelerations) */
static void sensor_stage(const RefModel* m, RefData* d, int stage) {
  if (m->disableflags & (1 << 13)) return; /* DisableBit.SENSOR */
  if (stage == 1)
    for (int i = 0; i < m->nsensor; i++)
      if (m->sensor_type[i] == SENS_FORCE || m->sensor_type[i] == SENS_TORQUE) { ref_rne_postconstraint(m, d); break; }
  if (stage == 0)
    for (int i = 0; i < m->nsensor; i++)
      if (m->sensor_type[i] == SENS_SUBTREELINVEL || m->sensor_type[i] == SENS_SUBTREEANGMOM) { ref_subtree_vel(m, d); break; }
  for (int i = 0; i < m->nsensor; i++) {
    int acc_type = m->sensor_type[i] == SENS_ACCELEROMETER || m->sensor_type[i] == SENS_FRAMELINACC || m->sensor_type[i] == SENS_FRAMEANGACC ||
                   m->sensor_type[i] == SENS_FORCE || m->sensor_type[i] == SENS_TORQUE || m->sensor_type[i] == SENS_JOINTLIMITFRC || m->sensor_type[i] == SENS_TOUCH;
    if (acc_type != (stage == 1)) continue;
    int t = m->sensor_type[i], id = m->sensor_objid[i], ot = m->sensor_objtype[i], rid = m->sensor_refid[i], rt = m->sensor_reftype[i];
    double v[4] = {0, 0, 0, 0}, pos[3], mat[9], q[4], rpos[3], rmat[9], rq[4], dif[3];
    if (t == SENS_JOINTPOS) v[0] = d->qpos[m->jnt_qposadr[id]];
    else if (t == SENS_JOINTVEL) v[0] = d->qvel[m->jnt_dofadr[id]];
    else if (t == SENS_ACTUATORPOS) v[0] = d->actuator_length[id];
    else if (t == SENS_ACTUATORVEL) v[0] = d->actuator_velocity[id];
    else if (t == SENS_ACTUATORFRC) v[0] = d->actuator_force[id];
    else if (t == SENS_MAGNETOMETER) { /* 117-128 */
      frame_of(m, d, OBJ_SITE, id, NULL, mat, NULL);
      matT_mul_vec(v, mat, m->magnetic);
    }
    else if (t == SENS_JOINTACTFRC) v[0] = d->qfrc_actuator[m->jnt_dofadr[id]];
    else if (t == SENS_E_POTENTIAL || t == SENS_E_KINETIC) { /* sensor.py:2773-3018 energy_pos / energy_vel */
      if (t == SENS_E_KINETIC) {
        double* mv = (double*)malloc(sizeof(double) * (m->nv > 0 ? m->nv : 1));
        ref_mul_m(m, d, mv, d->qvel);
        for (int k = 0; k < m->nv; k++) v[0] += 0.5 * d->qvel[k] * mv[k];
        free(mv);
      } else {
        if (!(m->disableflags & DSBL_GRAVITY))
          for (int b = 1; b < m->nbody; b++) v[0] -= m->body_mass[b] * v3dot(m->gravity, d->xipos + 3 * b);
        if (!(m->disableflags & DSBL_SPRING))
          for (int j = 0; j < m->njnt; j++) {
            double kk = m->jnt_stiffness[j];
            if (kk == 0.0) continue;
            int qa = m->jnt_qposadr[j], jt = m->jnt_type[j];
            if (jt == JNT_FREE || jt == JNT_BALL) {
              int o = jt == JNT_FREE ? 3 : 0;
              double qn[4], dq[3];
              memcpy(qn, d->qpos + qa + o, sizeof(qn));
              quat_normalize(qn);
              quat_sub(dq, qn, m->qpos_spring + qa + o);
              v[0] += 0.5 * kk * v3dot(dq, dq);
              if (jt == JNT_FREE)
                for (int k = 0; k < 3; k++) v[0] += 0.5 * kk * (d->qpos[qa + k] - m->qpos_spring[qa + k]) * (d->qpos[qa + k] - m->qpos_spring[qa + k]);
            } else v[0] += 0.5 * kk * (d->qpos[qa] - m->qpos_spring[qa]) * (d->qpos[qa] - m->qpos_spring[qa]);
          }
      }
    } else if (t == SENS_JOINTLIMITPOS || t == SENS_JOINTLIMITVEL || t == SENS_JOINTLIMITFRC) { /* 228-263, 1028-1063, 1640-1675 */
      int r1 = d->ne + d->nf + d->nl < m->njmax ? d->ne + d->nf + d->nl : m->njmax;
      for (int r = d->ne + d->nf; r < r1; r++)
        if (d->efc_type[r] == CT_LIMIT_JOINT && d->efc_id[r] == id)
          v[0] = t == SENS_JOINTLIMITPOS ? d->efc_pos[r] - d->efc_margin[r] : (t == SENS_JOINTLIMITVEL ? d->efc_vel[r] : d->efc_force[r]);
    } else if (t == SENS_BALLQUAT) { /* 216-225 */
      memcpy(v, d->qpos + m->jnt_qposadr[id], 4 * sizeof(double));
      quat_normalize(v);
    } else if (t == SENS_BALLANGVEL) memcpy(v, d->qvel + m->jnt_dofadr[id], 3 * sizeof(double));
    else if (t == SENS_CLOCK) v[0] = d->time;
    else if (t == SENS_RANGEFINDER) { /* sensor.py:179-196, 815-845 */
      double all[6] = {MAXVAL, MAXVAL, MAXVAL, MAXVAL, MAXVAL, MAXVAL}, z[3];
      int body = frame_of(m, d, OBJ_SITE, id, pos, mat, q);
      z[0] = mat[2], z[1] = mat[5], z[2] = mat[8];
      v[0] = ref_ray(m, d, pos, z, all, 1, body, NULL, NULL);
    }
    else if (t == SENS_SUBTREECOM) v3cpy(v, d->subtree_com + 3 * id);
    else if (t == SENS_SUBTREELINVEL) v3cpy(v, d->subtree_linvel + 3 * id);
    else if (t == SENS_SUBTREEANGMOM) v3cpy(v, d->subtree_angmom + 3 * id);
    else if (t == SENS_FRAMEPOS) { /* 377-403 */
      frame_of(m, d, ot, id, pos, NULL, NULL);
      if (rid == -1) v3cpy(v, pos);
      else {
        frame_of(m, d, rt, rid, rpos, rmat, NULL);
        v3sub(dif, pos, rpos);
        matT_mul_vec(v, rmat, dif);
      }
    } else if (t == SENS_FRAMEXAXIS || t == SENS_FRAMEYAXIS || t == SENS_FRAMEZAXIS) { /* 406-429 */
      int ax = t - SENS_FRAMEXAXIS;
      frame_of(m, d, ot, id, NULL, mat, NULL);
      double a[3] = {mat[ax], mat[3 + ax], mat[6 + ax]};
      if (rid == -1) v3cpy(v, a);
      else {
        frame_of(m, d, rt, rid, NULL, rmat, NULL);
        matT_mul_vec(v, rmat, a);
      }
    } else if (t == SENS_FRAMEQUAT) { /* 432-482 */
      frame_of(m, d, ot, id, NULL, NULL, q);
      if (rid == -1) memcpy(v, q, sizeof(q));
      else {
        frame_of(m, d, rt, rid, NULL, NULL, rq);
        rq[1] = -rq[1]; rq[2] = -rq[2]; rq[3] = -rq[3];
        mul_quat(v, rq, q);
      }
    } else if (t == SENS_VELOCIMETER || t == SENS_GYRO) { /* 964-1004: site frame */
      double lin[3], ang[3];
      frame_vel(m, d, OBJ_SITE, id, lin, ang);
      frame_of(m, d, OBJ_SITE, id, NULL, mat, NULL);
      matT_mul_vec(v, mat, t == SENS_GYRO ? ang : lin);
    } else if (t == SENS_FRAMELINVEL || t == SENS_FRAMEANGVEL) { /* 1108-1293 */
      double lin[3], ang[3], rlin[3], rang[3];
      frame_vel(m, d, ot, id, lin, ang);
      if (rid == -1) v3cpy(v, t == SENS_FRAMELINVEL ? lin : ang);
      else {
        frame_vel(m, d, rt, rid, rlin, rang);
        frame_of(m, d, rt, rid, rpos, rmat, NULL);
        if (t == SENS_FRAMELINVEL) {
          double rel[3], c[3];
          frame_of(m, d, ot, id, pos, NULL, NULL);
          v3sub(dif, pos, rpos);
          v3cross(c, dif, rang);
          for (int k = 0; k < 3; k++) rel[k] = lin[k] - rlin[k] + c[k];
          matT_mul_vec(v, rmat, rel);
        } else {
          v3sub(dif, ang, rang);
          matT_mul_vec(v, rmat, dif);
        }
      }
    }
    else if (t == SENS_TOUCH) { /* sensor.py:2063-2139 */
      int body = frame_of(m, d, OBJ_SITE, id, pos, mat, NULL);
      for (int c = 0; c < d->ncon; c++) {
        int b1 = m->geom_bodyid[d->con_geom[2 * c]], b2 = m->geom_bodyid[d->con_geom[2 * c + 1]], adr0 = d->con_efc_address[10 * c];
        if (adr0 < 0 || (body != b1 && body != b2)) continue;
        double nf = d->efc_force[adr0];
        if (m->cone == 0)
          for (int k = 1; k < 2 * (d->con_dim[c] - 1); k++)
            if (d->con_efc_address[10 * c + k] >= 0) nf += d->efc_force[d->con_efc_address[10 * c + k]];
        if (nf <= 0.0) continue;
        double ray[3] = {d->con_frame[9 * c] * nf, d->con_frame[9 * c + 1] * nf, d->con_frame[9 * c + 2] * nf};
        v3normalize(ray);
        if (body == b2) { ray[0] = -ray[0]; ray[1] = -ray[1]; ray[2] = -ray[2]; }
        if (ray_hits_zone(m->site_type[id], m->site_size + 3 * id, pos, mat, d->con_pos + 3 * c, ray)) v[0] += nf;
      }
    } else if (t == SENS_FORCE || t == SENS_TORQUE) { /* sensor.py:1542-1577 */
      int body = frame_of(m, d, OBJ_SITE, id, pos, mat, NULL);
      const double* ci = d->cfrc_int + 6 * body;
      if (t == SENS_FORCE) matT_mul_vec(v, mat, ci + 3);
      else {
        double c[3], tq[3];
        v3sub(dif, pos, d->subtree_com + 3 * m->body_rootid[body]);
        v3cross(c, dif, ci + 3);
        v3sub(tq, ci, c);
        matT_mul_vec(v, mat, tq);
      }
    } else if (acc_type) { /* sensor.py:1510-1539, 1678-1753 */
      int fot = t == SENS_ACCELEROMETER ? OBJ_SITE : ot;
      double cacc[6], lin[3], ang[3], off[3], c1[3], c2[3], a[3];
      int body = frame_of(m, d, fot, id, pos, mat, NULL);
      body_cacc(m, d, body, cacc);
      if (t == SENS_FRAMEANGACC) v3cpy(v, cacc);
      else {
        frame_vel(m, d, fot, id, lin, ang);
        v3sub(off, pos, d->subtree_com + 3 * m->body_rootid[body]);
        v3cross(c1, off, cacc);
        v3cross(c2, ang, lin);
        for (int k = 0; k < 3; k++) a[k] = cacc[3 + k] - c1[k] + c2[k];
        if (t == SENS_ACCELEROMETER) matT_mul_vec(v, mat, a);
        else v3cpy(v, a);
      }
    }
    sensor_write(m, d, i, v);
  }
}
void ref_sensor(const RefModel* m, RefData* d) {
  sensor_stage(m, d, 0);
  sensor_stage(m, d, 1);
}

void ref_fwd_position(const RefModel* m, RefData* d) { /* forward.py:635-679 */
  ref_kinematics(m, d);
  ref_com_pos(m, d);
  ref_crb(m, d);
  ref_factor_m(m, d);
  ref_collision(m, d);
  if (sleep_enabled(m)) {
    /* forward.py:652-666: contacts of pass 1 wake sleeping trees touched by awake ones; pass 2 adds the pairs pass 1 skipped that
       involve a newly awakened body.  Waking only ever lets more pairs through the filter, so pass 1 + pass 2 is the set of pairs
       that pass the filter under the new state: the restatement recomputes the whole list (in canonical pair order). */
    ref_wake_collision(m, d);
    ref_update_sleep(m, d);
    ref_collision(m, d);
  }
  ref_make_constraint(m, d);
  if (sleep_enabled(m)) {
    if (m->neq > 0) ref_wake_equality(m, d);
    ref_update_sleep(m, d);
    ref_island(m, d);
  }
  ref_transmission(m, d);
}
void ref_forward(const RefModel* m, RefData* d) { /* forward.py:1341-1366 */
  if (sleep_enabled(m)) {
    ref_wake(m, d);
    ref_update_sleep(m, d);
  }
  ref_fwd_position(m, d);
  ref_fwd_velocity(m, d);
  ref_fwd_actuation(m, d);
  ref_fwd_acceleration(m, d);
  sensor_stage(m, d, 0); /* (position / velocity stage sensors and actuator forces: all final before the solve) */
  ref_solve(m, d);
  sensor_stage(m, d, 1);
}

/* _advance forward.py:276-349; next_act support.py:38 */
static void advance(const RefModel* m, RefData* d, const double* qacc) {
  double h = m->timestep;
  for (int i = 0; i < m->nu; i++) {
    int dyn = m->actuator_dyntype[i];
    if (dyn == 0) continue;
    int adr = m->actuator_actadr[i];
    double act = d->act[adr], act_dot = d->act_dot[adr];
    if (dyn == 3) { /* FILTEREXACT */
      double tau = fmax(MINVAL, m->actuator_dynprm[10 * i]);
      act = act + act_dot * tau * (1.0 - exp(-h / tau));
    } else act = act + act_dot * h;
    if (m->actuator_actlimited[i]) act = clampd(act, m->actuator_actrange[2 * i], m->actuator_actrange[2 * i + 1]);
    d->act[adr] = act;
  }
  for (int i = 0; i < m->nv; i++) d->qvel[i] += qacc[i] * h;
  for (int j = 0; j < m->njnt; j++) { /* _next_position forward.py:53 */
    int qa = m->jnt_qposadr[j], dof = m->jnt_dofadr[j], t = m->jnt_type[j];
    if (t == JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[dof + k];
      quat_integrate(d->qpos + qa + 3, d->qvel + dof + 3, h);
    } else if (t == JNT_BALL) quat_integrate(d->qpos + qa, d->qvel + dof, h);
    else d->qpos[qa] += h * d->qvel[dof];
  }
  d->time += h;
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * m->nv);
}

void ref_euler(const RefModel* m, RefData* d) { /* forward.py:387-417 */
  int nv = m->nv;
  if (!(m->disableflags & (DSBL_EULERDAMP | DSBL_DAMPER))) {
    int any = 0;
    for (int i = 0; i < nv; i++) any |= (m->dof_damping[i] != 0.0);
    if (any) {
      double* Mh = (double*)malloc(sizeof(double) * (2 * m->nC + 2 * nv));
      double *L = Mh + m->nC, *Dinv = L + m->nC, *qacc = Dinv + nv;
      memcpy(Mh, d->M, sizeof(double) * m->nC);
      for (int i = 0; i < nv; i++) Mh[m->M_rowadr[i] + m->M_rownnz[i] - 1] += m->timestep * m->dof_damping[i];
      factor_sparse(m, Mh, L, Dinv);
      solve_sparse(m, L, Dinv, qacc, d->Ma);
      advance(m, d, qacc);
      free(Mh);
      return;
    }
  }
  advance(m, d, d->qacc);
}

/* implicitfast forward.py:578-612 with derivative.py:1117 deriv_smooth_vel restricted to joint damping and
   affine actuator velocity terms: (M - h*dF/dv) qacc' = M qacc */
void ref_implicitfast(const RefModel* m, RefData* d) {
  int nv = m->nv;
  double* Mh = (double*)malloc(sizeof(double) * (2 * m->nC + 2 * nv));
  double *L = Mh + m->nC, *Dinv = L + m->nC, *qacc = Dinv + nv;
  memcpy(Mh, d->M, sizeof(double) * m->nC);
  if (!(m->disableflags & DSBL_DAMPER))
    for (int i = 0; i < nv; i++) Mh[m->M_rowadr[i] + m->M_rownnz[i] - 1] += m->timestep * m->dof_damping[i];
  if (!(m->disableflags & DSBL_ACTUATION))
    for (int i = 0; i < m->nu; i++) {
      double bias_vel = (m->actuator_biastype[i] == 1) ? m->actuator_biasprm[10 * i + 2] : 0.0;
      double gain_vel = (m->actuator_gaintype[i] == 1) ? m->actuator_gainprm[10 * i + 2] : 0.0;
      double ctrl = d->ctrl[i];
      if (m->actuator_dyntype[i] != 0) ctrl = d->act[m->actuator_actadr[i]]; /* raw ctrl otherwise: derivative.py:159-161 */
      double dv = bias_vel + gain_vel * ctrl;
      if (dv == 0.0) continue;
      if (m->actuator_forcelimited[i]) {
        double f = d->actuator_force[i];
        if (f <= m->actuator_forcerange[2 * i] || f >= m->actuator_forcerange[2 * i + 1]) continue;
      }
      int dof = m->jnt_dofadr[m->actuator_trnid[2 * i]];
      double g = m->actuator_gear[6 * i];
      Mh[m->M_rowadr[dof] + m->M_rownnz[dof] - 1] -= m->timestep * g * g * dv;
    }
  factor_sparse(m, Mh, L, Dinv);
  solve_sparse(m, L, Dinv, qacc, d->Ma);
  advance(m, d, qacc);
  free(Mh);
}

/* d(qfrc_bias)/d(qvel), dense [nv x nv] (out[i * nv + k]): derivative.py:321-586 deriv_rne_vel (MuJoCo C mjd_rne_vel), column by column.
 * For dof k: Dcvel[b] = cdof[k] on every body at or below dof k's body (added where com_vel adds it); Dcdof_dot[j] = Dcvel_before(j) x cdof[j]
 * with com_vel's group rule (the three rotational dofs of a ball / free joint see the velocity before any of them; a free joint's first
 * three dofs have cdof_dot = 0); Dcacc accumulates root -> leaf, Dcfrc_body = I Dcacc + Dcvel x* (I cvel) + cvel x* (I Dcvel), summed leaf ->
 * root over subtrees, projected on cdof[i]. */
void ref_deriv_rne_vel(const RefModel* m, const RefData* d, double* out) {
  int nv = m->nv, nb = m->nbody;
  double* buf = (double*)calloc((size_t)18 * nb, sizeof(double));
  double *Dcvel = buf, *Dcacc = Dcvel + 6 * nb, *Dcfrc = Dcacc + 6 * nb;
  for (int k = 0; k < nv; k++) {
    memset(buf, 0, sizeof(double) * 18 * nb);
    for (int b = 1; b < nb; b++) {
      double cv[6], ca[6];
      memcpy(cv, Dcvel + 6 * m->body_parentid[b], sizeof(cv));
      memcpy(ca, Dcacc + 6 * m->body_parentid[b], sizeof(ca));
      int dof = m->body_dofadr[b];
      for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
        int t = m->jnt_type[j];
        int ngrp = t == JNT_FREE ? 6 : (t == JNT_BALL ? 3 : 1);
        /* (free joint: dofs 0..2 first -- their cdof_dot is zero --, then 3..5 as a group; ball: one group of three) */
        int first = 0;
        if (t == JNT_FREE) {
          for (int q = 0; q < 3; q++) {
            if (dof + q == k) { for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * k + c]; }
            /* cdof_dot of these dofs is identically zero: nothing enters cacc */
          }
          first = 3;
        }
        double dd[6];
        for (int q = first; q < ngrp; q++) { /* Dcdof_dot of the group from the velocity derivative BEFORE the group */
          motion_cross(dd, cv, d->cdof + 6 * (dof + q));
          for (int c = 0; c < 6; c++) ca[c] += dd[c] * d->qvel[dof + q];
          if (dof + q == k)
            for (int c = 0; c < 6; c++) ca[c] += d->cdof_dot[6 * k + c];
        }
        for (int q = first; q < ngrp; q++)
          if (dof + q == k)
            for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * k + c];
        dof += ngrp;
      }
      memcpy(Dcvel + 6 * b, cv, sizeof(cv));
      memcpy(Dcacc + 6 * b, ca, sizeof(ca));
      double t1[6], icv[6], idcv[6], x1[6], x2[6];
      inert_vec(t1, d->cinert + 10 * b, ca);
      inert_vec(icv, d->cinert + 10 * b, d->cvel + 6 * b);
      inert_vec(idcv, d->cinert + 10 * b, cv);
      motion_cross_force(x1, cv, icv);
      motion_cross_force(x2, d->cvel + 6 * b, idcv);
      for (int c = 0; c < 6; c++) Dcfrc[6 * b + c] = t1[c] + x1[c] + x2[c];
    }
    for (int b = nb - 1; b > 0; b--)
      for (int c = 0; c < 6; c++) Dcfrc[6 * m->body_parentid[b] + c] += Dcfrc[6 * b + c];
    for (int i = 0; i < nv; i++) {
      double s2 = 0.0;
      for (int c = 0; c < 6; c++) s2 += d->cdof[6 * i + c] * Dcfrc[6 * m->dof_bodyid[i] + c];
      out[(size_t)i * nv + k] = s2;
    }
  }
  free(buf);
}

/* fully implicit in velocity (forward.py:578-600; MuJoCo C mj_implicit): (M - h dF/dv) qacc' = M qacc with F = qfrc_smooth =
 * passive - bias + actuation, i.e. the matrix of implicitfast PLUS h d(qfrc_bias)/dv (non-symmetric): dense LU without pivoting
 * (the reference factors the same matrix in its sparse "D structure", smooth.py:3481; the matrix is M plus an O(h) perturbation). */
void ref_implicit(const RefModel* m, RefData* d) {
  int nv = m->nv;
  double* A = (double*)calloc((size_t)2 * nv * nv + 2 * nv, sizeof(double));
  double *Dr = A + (size_t)nv * nv, *qacc = Dr + (size_t)nv * nv, *rhs = qacc + nv;
  for (int i = 0; i < nv; i++)
    for (int a = 0; a < m->M_rownnz[i]; a++) {
      int j = m->M_colind[m->M_rowadr[i] + a];
      A[(size_t)i * nv + j] = A[(size_t)j * nv + i] = d->M[m->M_rowadr[i] + a];
    }
  if (!(m->disableflags & DSBL_DAMPER))
    for (int i = 0; i < nv; i++) A[(size_t)i * nv + i] += m->timestep * m->dof_damping[i];
  if (!(m->disableflags & DSBL_ACTUATION))
    for (int i = 0; i < m->nu; i++) {
      double bias_vel = (m->actuator_biastype[i] == 1) ? m->actuator_biasprm[10 * i + 2] : 0.0;
      double gain_vel = (m->actuator_gaintype[i] == 1) ? m->actuator_gainprm[10 * i + 2] : 0.0;
      double ctrl = d->ctrl[i];
      if (m->actuator_dyntype[i] != 0) ctrl = d->act[m->actuator_actadr[i]];
      double dv = bias_vel + gain_vel * ctrl;
      if (dv == 0.0) continue;
      if (m->actuator_forcelimited[i]) {
        double f = d->actuator_force[i];
        if (f <= m->actuator_forcerange[2 * i] || f >= m->actuator_forcerange[2 * i + 1]) continue;
      }
      int dof = m->jnt_dofadr[m->actuator_trnid[2 * i]];
      double g = m->actuator_gear[6 * i];
      A[(size_t)dof * nv + dof] -= m->timestep * g * g * dv;
    }
  ref_deriv_rne_vel(m, d, Dr);
  for (size_t e = 0; e < (size_t)nv * nv; e++) A[e] += m->timestep * Dr[e];
  memcpy(rhs, d->Ma, sizeof(double) * nv);
  for (int k = 0; k < nv; k++) { /* LU in place, no pivoting; forward substitution fused */
    double piv = A[(size_t)k * nv + k];
    for (int i = k + 1; i < nv; i++) {
      double f = A[(size_t)i * nv + k] / piv;
      if (f == 0.0) continue;
      for (int j = k + 1; j < nv; j++) A[(size_t)i * nv + j] -= f * A[(size_t)k * nv + j];
      rhs[i] -= f * rhs[k];
    }
  }
  for (int i = nv - 1; i >= 0; i--) {
    double s2 = rhs[i];
    for (int j = i + 1; j < nv; j++) s2 -= A[(size_t)i * nv + j] * qacc[j];
    qacc[i] = s2 / A[(size_t)i * nv + i];
  }
  advance(m, d, qacc);
  free(A);
}

/* state update from t0 with explicit rates and step dt: act (_next_activation forward.py:134 with `scale`), qvel
 * (_next_velocity 117), qpos (_next_position 53 integrates `vel_pos`) */
static void rk_set_state(const RefModel* m, RefData* d, const double* qpos0, const double* qvel0, const double* act0,
                         const double* vel_pos, const double* qacc, const double* act_dot, double dt) {
  for (int i = 0; i < m->nu; i++) {
    int dyn = m->actuator_dyntype[i];
    if (dyn == 0) continue;
    int adr = m->actuator_actadr[i];
    double act;
    if (dyn == 3) {
      double tau = fmax(MINVAL, m->actuator_dynprm[10 * i]);
      act = act0[adr] + act_dot[adr] * tau * (1.0 - exp(-dt / tau));
    } else act = act0[adr] + act_dot[adr] * dt;
    if (m->actuator_actlimited[i]) act = clampd(act, m->actuator_actrange[2 * i], m->actuator_actrange[2 * i + 1]);
    d->act[adr] = act;
  }
  for (int i = 0; i < m->nv; i++) d->qvel[i] = qvel0[i] + dt * qacc[i];
  memcpy(d->qpos, qpos0, sizeof(double) * m->nq);
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], dof = m->jnt_dofadr[j], t = m->jnt_type[j];
    if (t == JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += dt * vel_pos[dof + k];
      quat_integrate(d->qpos + qa + 3, vel_pos + dof + 3, dt);
    } else if (t == JNT_BALL) quat_integrate(d->qpos + qa, vel_pos + dof, dt);
    else d->qpos[qa] += dt * vel_pos[dof];
  }
}

/* rungekutta4 forward.py:524-557 (_rk_perturb_state 420, _rk_accumulate 498): called after forward() at t0 */
void ref_rungekutta4(const RefModel* m, RefData* d) {
  static const double A[3] = {0.5, 0.5, 1.0}, B[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
  int nq = m->nq, nv = m->nv, na = m->na;
  double* buf = (double*)calloc((size_t)nq + 4 * nv + 3 * na + 3, sizeof(double));
  double *qpos0 = buf, *qvel0 = qpos0 + nq, *qvel_rk = qvel0 + nv, *qacc_rk = qvel_rk + nv, *vel = qacc_rk + nv,
         *act0 = vel + nv, *actdot_rk = act0 + na + 1, *actdot = actdot_rk + na + 1;
  memcpy(qpos0, d->qpos, sizeof(double) * nq);
  memcpy(qvel0, d->qvel, sizeof(double) * nv);
  memcpy(act0, d->act, sizeof(double) * na);
  double h = m->timestep;
  for (int k = 0; k < 4; k++) {
    for (int i = 0; i < nv; i++) { qvel_rk[i] += B[k] * d->qvel[i]; qacc_rk[i] += B[k] * d->qacc[i]; }
    for (int i = 0; i < na; i++) actdot_rk[i] += B[k] * d->act_dot[i];
    if (k == 3) break;
    memcpy(vel, d->qvel, sizeof(double) * nv); /* the stage's velocity moves the position */
    memcpy(actdot, d->act_dot, sizeof(double) * na);
    rk_set_state(m, d, qpos0, qvel0, act0, vel, d->qacc, actdot, A[k] * h);
    ref_forward(m, d);
  }
  memcpy(d->act_dot, actdot_rk, sizeof(double) * na);
  rk_set_state(m, d, qpos0, qvel0, act0, qvel_rk, qacc_rk, actdot_rk, h); /* _advance(m, d, qacc_rk, qvel_rk) */
  d->time += h;
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
  free(buf);
}

void ref_step(const RefModel* m, RefData* d) { /* forward.py:1368-1380 */
  ref_forward(m, d);
  if (m->integrator == INT_IMPLICITFAST) ref_implicitfast(m, d);
  else if (m->integrator == INT_IMPLICIT) ref_implicit(m, d);
  else if (m->integrator == INT_RK4) ref_rungekutta4(m, d);
  else ref_euler(m, d);
  if (sleep_enabled(m)) { /* forward.py:345-349 (end of _advance) */
    ref_sleep(m, d);
    ref_fwd_velocity(m, d);
    ref_update_sleep(m, d);
  }
}

/* util_misc.py:61 halton (float32 arithmetic in the reference; restated in float32 here on purpose) */
double ref_halton(int index, int base) {
  int n0 = index;
  float b = (float)base, f = 1.0f / b, hn = 0.0f;
  while (n0 > 0) {
    int n1 = n0 / base;
    int r = n0 - n1 * base;
    hn += f * (float)r;
    f /= b;
    n0 = n1;
  }
  return (double)hn;
}

/* cli.py:103-145 _ctrl_noise */
void ref_ctrl_noise(const RefModel* m, RefData* d, const double* center, int step, int worldid, double noise_std, double noise_rate) {
  double rate = exp(-m->timestep / noise_rate);
  double scale = noise_std * sqrt(1.0 - rate * rate);
  for (int a = 0; a < m->nu; a++) {
    double midpoint = 0.0, halfrange = 1.0;
    int lim = m->actuator_ctrllimited[a];
    double lo = m->actuator_ctrlrange[2 * a], hi = m->actuator_ctrlrange[2 * a + 1];
    if (lim) { midpoint = 0.5 * (hi + lo); halfrange = 0.5 * (hi - lo); }
    if (center) midpoint = center[a];
    double ctrl = rate * d->ctrl[a] + (1.0 - rate) * midpoint;
    ctrl += scale * halfrange * (2.0 * ref_halton((step + 1) * (worldid + 1), a + 2) - 1.0);
    if (lim) ctrl = clampd(ctrl, lo, hi);
    d->ctrl[a] = ctrl;
  }
}

/* nstep steps with per-step ctrl noise around centre 0 (cli.py:270-292 loop body); returns #steps with finite qpos */
int ref_rollout(const RefModel* m, RefData* d, int nstep, int worldid, double noise_std, double noise_rate, double* qpos_out, double* qvel_out) {
  double* center = (double*)calloc(m->nu > 0 ? m->nu : 1, sizeof(double));
  int ok = 0;
  for (int s = 0; s < nstep; s++) {
    if (noise_std >= 0) ref_ctrl_noise(m, d, center, s, worldid, noise_std, noise_rate);
    ref_step(m, d);
    if (qpos_out) memcpy(qpos_out + (size_t)s * m->nq, d->qpos, sizeof(double) * m->nq);
    if (qvel_out) memcpy(qvel_out + (size_t)s * m->nv, d->qvel, sizeof(double) * m->nv);
    int fin = 1;
    for (int i = 0; i < m->nq; i++) fin &= isfinite(d->qpos[i]);
    ok += fin;
  }
  free(center);
  return ok;
}
