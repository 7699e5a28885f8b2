/* oracle/ccd.c -- TEST INFRASTRUCTURE ONLY (textually included by mjref.c; never linked into the product).
 *
 * float64 restatement of the reference's general convex collision detection for the primitive convex shapes
 * (sphere, capsule, ellipsoid, cylinder, box) and convex meshes (exhaustive vertex search: the reference's path for meshes without a
 * hill-climbing graph or with fewer than 10 vertices, collision_gjk.py:154-169; height fields are out of scope):
 *   collision_gjk.py  support 116-223 | sub-distance (signed volumes) 281-593 | gjk 635-770 | polytope seeds 1021-1286 |
 *                     EPA 1319-1454 | witness points 947-1018 | gjk_phase / epa_phase / ccd 2303-2575
 *   collision_convex.py eval_ccd_write_contact 747-977 (margins, cutoff, frame, midpoint)
 * PINNED by the numbers the reference's own tests hold (collision_gjk_test.py: sphere / box / cylinder / capsule cases,
 * tests/test_convex.py reproduces them).
 */

#define CCD_FLOAT_MAX 1e30
#define CCD_MINVAL 1e-15
#define CCD_MINVAL2 1e-30
#define CCD_MIN_DIST2 1e-10
#define CCD_MIN_DIST3 1e-10
#define CCD_MIN_DIST4 1e-17
#define CCD_MIN_EPATOL 1e-7
#define CCD_MAX_ITER 128
#define CCD_MAX_HORIZON 24 /* types.py:31 MJ_MAX_EPAHORIZON */
#define CCD_EPAFACES 5     /* types.py:33 MJ_MAX_EPAFACES */
#define OVF_EPA_HORIZON (1 << 8) /* types.py OverflowType.EPA_HORIZON */

typedef struct CcdGeom {
  int type;
  double pos[3], rot[9], size[3], margin;
  const double* vert; /* mesh: vertices in the geom frame */
  int nvert;
  double prism[6][3]; /* height-field prism (type G_HFIELD): vertices in the height field's frame (collision_gjk.py:197-206) */
  int index;          /* mesh: cache of the last support call (warm start), -1 at the start (Geom.index): a vertex id on the exhaustive
                         path, a graph-local vertex id on the hill-climbing path */
  int cache;          /* out of ccd_support: SupportPoint.cached_index */
  const int* graph;   /* mesh: hill-climbing graph (MuJoCo's mesh_graph block of this mesh: numvert, numface, vert_edgeadr[numvert],
                         vert_globalid[numvert], edge_localid[numvert + 3 numface], ...) or NULL */
  /* mesh polygon tables for multi-contact recovery (types.py:1707-1733), already offset to this mesh; NULL when absent */
  const double* polynormal; /* [npoly, 3] */
  const int* polyvertadr;   /* [npoly] into polyvert (global array) */
  const int* polyvertnum;
  const int* polyvert;      /* global array of mesh-local vertex ids */
  const int* polymapadr;    /* [nvert] into polymap (global array) */
  const int* polymapnum;
  const int* polymap;       /* global array of mesh-local polygon ids */
} CcdGeom;
#define MC_MAXN 128 /* normals / polygon vertices the restatement holds per feature (the reference sizes its buffers from the model: ref.py refuses models beyond this) */

typedef struct GjkOut {
  int separated, dim;
  double dist, x1[3], x2[3];
  double s[4][3], s1[4][3], s2[4][3]; /* Minkowski-difference simplex and its preimages on the two geoms */
  int i1[4], i2[4];                   /* vertex ids of the preimages (boxes), -1 for smooth shapes */
} GjkOut;

static double ccd_sign(double x) { return x < 0.0 ? -1.0 : 1.0; } /* (warp's sign: +1 at zero) */

/* support point of a geom in world direction dir (unit): collision_gjk.py:116-223 */
static int ccd_support(const CcdGeom* g, const double* dir, double* out) {
  int vid = -1;
  if (g->type == G_SPHERE) {
    for (int k = 0; k < 3; k++) out[k] = g->pos[k] + (g->size[0] + 0.5 * g->margin) * dir[k];
    return vid;
  }
  if (g->type == G_HFIELD) { /* 197-206: the furthest of the prism's six vertices (the prism lives in the frame the pair is solved in) */
    double best = -CCD_FLOAT_MAX;
    vid = dir[2] < 0.0 ? -2 : -3;
    for (int i = 0; i < 6; i++) {
      double dd = v3dot(g->prism[i], dir);
      if (dd > best) { best = dd; v3cpy(out, g->prism[i]); }
    }
    if (g->margin > 0.0)
      for (int k = 0; k < 3; k++) out[k] += dir[k] * (0.5 * g->margin);
    return vid;
  }
  double l[3], r[3] = {0, 0, 0};
  matT_mul_vec(l, g->rot, dir);
  if (g->type == G_BOX) {
    vid = 0;
    for (int k = 0; k < 3; k++) {
      double sg = ccd_sign(l[k]);
      r[k] = sg * g->size[k];
      if (sg > 0.0) vid += 1 << k;
    }
  } else if (g->type == G_CAPSULE) {
    for (int k = 0; k < 3; k++) r[k] = l[k] * g->size[0];
    r[2] += ccd_sign(l[2]) * g->size[1];
  } else if (g->type == G_ELLIPSOID) {
    for (int k = 0; k < 3; k++) r[k] = l[k] * g->size[k];
    v3normalize(r);
    for (int k = 0; k < 3; k++) r[k] *= g->size[k];
  } else if (g->type == G_MESH && (!g->graph || g->nvert < 10)) { /* 154-169: exhaustive search, the cached vertex first */
    double best = -CCD_FLOAT_MAX;
    if (g->index > -1) {
      vid = g->index;
      best = v3dot(g->vert + 3 * vid, l);
    }
    for (int i = 0; i < g->nvert; i++) {
      double dd = v3dot(g->vert + 3 * i, l);
      if (dd > best) { best = dd; vid = i; }
    }
    ((CcdGeom*)g)->cache = vid;
    v3cpy(r, g->vert + 3 * vid);
  } else if (g->type == G_MESH) { /* 170-196: hill climbing on the hull's vertex graph from the cached vertex */
    int numvert = g->graph[0];
    const int *edgeadr = g->graph + 2, *globalid = g->graph + 2 + numvert, *edge = g->graph + 2 + 2 * numvert;
    int prev = -1, imax = g->index > -1 ? g->index : 0;
    double best = v3dot(l, g->vert + 3 * globalid[imax]);
    while (imax != prev) {
      prev = imax;
      for (int i = edgeadr[imax]; edge[i] >= 0; i++) {
        double dd = v3dot(l, g->vert + 3 * globalid[edge[i]]);
        if (dd > best) { best = dd; imax = edge[i]; }
      }
    }
    ((CcdGeom*)g)->cache = imax;
    vid = globalid[imax];
    v3cpy(r, g->vert + 3 * vid);
  } else if (g->type == G_CYLINDER) {
    double d = sqrt(l[0] * l[0] + l[1] * l[1]);
    if (d > CCD_MINVAL) {
      r[0] = l[0] * g->size[0] / d;
      r[1] = l[1] * g->size[0] / d;
    }
    r[2] = ccd_sign(l[2]) * g->size[1];
  }
  mat_mul_vec(out, g->rot, r);
  for (int k = 0; k < 3; k++) out[k] += g->pos[k];
  if (g->margin > 0.0)
    for (int k = 0; k < 3; k++) out[k] += dir[k] * (0.5 * g->margin);
  return vid;
}

static double ccd_det3(const double* a, const double* b, const double* c) {
  double bc[3];
  v3cross(bc, b, c);
  return v3dot(a, bc);
}
static int ccd_same_sign(double a, double b) { return (a > 0.0 && b > 0.0) ? 1 : ((a < 0.0 && b < 0.0) ? -1 : 0); }

/* projection of the origin on the line through v1, v2 / on the plane through v1, v2, v3 (collision_gjk.py:308-343) */
static void ccd_origin_on_line(const double* v1, const double* v2, double* out) {
  double df[3];
  v3sub(df, v2, v1);
  double scl = -(v3dot(v2, df) / v3dot(df, df));
  for (int k = 0; k < 3; k++) out[k] = v2[k] + scl * df[k];
}
static int ccd_origin_on_plane(const double* v1, const double* v2, const double* v3, double* out) {
  double d21[3], d31[3], d32[3], n[3];
  v3sub(d21, v2, v1);
  v3sub(d31, v3, v1);
  v3sub(d32, v3, v2);
  const double* as[3] = {d32, d21, d31};
  const double* bs[3] = {d21, d31, d32};
  const double* vs[3] = {v2, v1, v3};
  for (int t = 0; t < 3; t++) { /* three equivalent normals; the first well-conditioned one is used */
    v3cross(n, as[t], bs[t]);
    double nv = v3dot(n, vs[t]), nn = v3dot(n, n);
    if (t < 2 && nn == 0.0) { v3set(out, 0, 0, 0); return 1; }
    if (t == 2 || (nv != 0.0 && nn > CCD_MINVAL)) {
      for (int k = 0; k < 3; k++) out[k] = (nv / nn) * n[k];
      return 0;
    }
  }
  return 0;
}

/* barycentric coordinates of the point of a simplex closest to the origin: signed-volume method, 1 / 2 / 3-simplex
 * (collision_gjk.py:568-593, 423-565, 347-420) */
static void ccd_s1d(const double* s1, const double* s2, double* lam) {
  double po[3];
  ccd_origin_on_line(s1, s2, po);
  double mu_max = s1[0] - s2[0];
  int idx = 0;
  for (int k = 1; k < 3; k++) {
    double mu = s1[k] - s2[k];
    if (fabs(mu) >= fabs(mu_max)) { mu_max = mu; idx = k; }
  }
  double c1 = po[idx] - s2[idx], c2 = s1[idx] - po[idx];
  if (ccd_same_sign(mu_max, c1) && ccd_same_sign(mu_max, c2)) { lam[0] = c1 / mu_max; lam[1] = c2 / mu_max; }
  else { lam[0] = 0.0; lam[1] = 1.0; }
}
static void ccd_s2d(const double* s1, const double* s2, const double* s3, double* lam) {
  double po[3];
  if (ccd_origin_on_plane(s1, s2, s3, po)) {
    ccd_s1d(s1, s2, lam);
    lam[2] = 0.0;
    return;
  }
  /* drop the coordinate axis along which the triangle's projected area is smallest */
  double m14 = s2[1] * s3[2] - s2[2] * s3[1] - s1[1] * s3[2] + s1[2] * s3[1] + s1[1] * s2[2] - s1[2] * s2[1];
  double m24 = s2[0] * s3[2] - s2[2] * s3[0] - s1[0] * s3[2] + s1[2] * s3[0] + s1[0] * s2[2] - s1[2] * s2[0];
  double m34 = s2[0] * s3[1] - s2[1] * s3[0] - s1[0] * s3[1] + s1[1] * s3[0] + s1[0] * s2[1] - s1[1] * s2[0];
  double mu1 = fabs(m14), mu2 = fabs(m24), mu3 = fabs(m34), mmax;
  int x, y;
  if (mu1 >= mu2 && mu1 >= mu3) { mmax = m14; x = 1; y = 2; }
  else if (mu2 >= mu3) { mmax = m24; x = 0; y = 2; }
  else { mmax = m34; x = 0; y = 1; }
  double a[2] = {s1[x], s1[y]}, b[2] = {s2[x], s2[y]}, c[2] = {s3[x], s3[y]}, p[2] = {po[x], po[y]};
  double c31 = p[0] * b[1] + p[1] * c[0] + b[0] * c[1] - p[0] * c[1] - p[1] * b[0] - c[0] * b[1];
  double c32 = p[0] * c[1] + p[1] * a[0] + c[0] * a[1] - p[0] * a[1] - p[1] * c[0] - a[0] * c[1];
  double c33 = p[0] * a[1] + p[1] * b[0] + a[0] * b[1] - p[0] * b[1] - p[1] * a[0] - b[0] * a[1];
  int k1 = ccd_same_sign(mmax, c31), k2 = ccd_same_sign(mmax, c32), k3 = ccd_same_sign(mmax, c33);
  if (k1 && k2 && k3) { lam[0] = c31 / mmax; lam[1] = c32 / mmax; lam[2] = c33 / mmax; return; }
  double dmin = CCD_FLOAT_MAX, sub[2], xx[3];
  lam[0] = lam[1] = lam[2] = 0.0;
  if (!k1) {
    ccd_s1d(s2, s3, sub);
    for (int k = 0; k < 3; k++) xx[k] = sub[0] * s2[k] + sub[1] * s3[k];
    lam[0] = 0.0; lam[1] = sub[0]; lam[2] = sub[1];
    dmin = v3dot(xx, xx);
  }
  if (!k2) {
    ccd_s1d(s1, s3, sub);
    for (int k = 0; k < 3; k++) xx[k] = sub[0] * s1[k] + sub[1] * s3[k];
    double dd = v3dot(xx, xx);
    if (dd < dmin) { lam[0] = sub[0]; lam[1] = 0.0; lam[2] = sub[1]; dmin = dd; }
  }
  if (!k3) {
    ccd_s1d(s1, s2, sub);
    for (int k = 0; k < 3; k++) xx[k] = sub[0] * s1[k] + sub[1] * s2[k];
    double dd = v3dot(xx, xx);
    if (dd < dmin) { lam[0] = sub[0]; lam[1] = sub[1]; lam[2] = 0.0; }
  }
}
static void ccd_s3d(const double* s1, const double* s2, const double* s3, const double* s4, double* lam) {
  double c41 = -ccd_det3(s2, s3, s4), c42 = ccd_det3(s1, s3, s4), c43 = -ccd_det3(s1, s2, s4), c44 = ccd_det3(s1, s2, s3);
  double mdet = c41 + c42 + c43 + c44;
  int k1 = ccd_same_sign(mdet, c41), k2 = ccd_same_sign(mdet, c42), k3 = ccd_same_sign(mdet, c43), k4 = ccd_same_sign(mdet, c44);
  if (k1 && k2 && k3 && k4) { /* origin inside the tetrahedron */
    lam[0] = c41 / mdet; lam[1] = c42 / mdet; lam[2] = c43 / mdet; lam[3] = c44 / mdet;
    return;
  }
  const double* v[4] = {s1, s2, s3, s4};
  const int ks[4] = {k1, k2, k3, k4};
  double dmin = CCD_FLOAT_MAX;
  lam[0] = lam[1] = lam[2] = lam[3] = 0.0;
  for (int omit = 0; omit < 4; omit++) { /* faces opposite the vertices whose cofactor has the wrong sign */
    if (ks[omit]) continue;
    int id[3], n = 0;
    for (int i = 0; i < 4; i++)
      if (i != omit) id[n++] = i;
    double sub[3], xx[3];
    ccd_s2d(v[id[0]], v[id[1]], v[id[2]], sub);
    for (int k = 0; k < 3; k++) xx[k] = sub[0] * v[id[0]][k] + sub[1] * v[id[1]][k] + sub[2] * v[id[2]][k];
    double dd = v3dot(xx, xx);
    if (dd < dmin) {
      lam[omit] = 0.0;
      lam[id[0]] = sub[0]; lam[id[1]] = sub[1]; lam[id[2]] = sub[2];
      dmin = dd;
    }
  }
}
static void ccd_subdistance(int n, double s[4][3], double* lam) {
  lam[0] = 1.0; lam[1] = lam[2] = lam[3] = 0.0;
  if (n == 4) ccd_s3d(s[0], s[1], s[2], s[3], lam);
  else if (n == 3) ccd_s2d(s[0], s[1], s[2], lam);
  else if (n == 2) ccd_s1d(s[0], s[1], lam);
}
static void ccd_combine(int n, const double* lam, double m[4][3], double* out) {
  v3set(out, 0, 0, 0);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) out[k] += lam[i] * m[i][k];
}

/* gjk: collision_gjk.py:596-770.  x1_0 / x2_0: initial guess (the geom centres) */
static void ccd_gjk(double tolerance, int iterations, const CcdGeom* g1, const CcdGeom* g2, const double* x1_0, const double* x2_0,
                    double cutoff, int is_discrete, GjkOut* res) {
  int n = 0;
  double lam[4] = {1, 0, 0, 0}, xk[3];
  double epsilon = is_discrete ? 0.0 : 0.5 * tolerance * tolerance, min_norm = is_discrete ? CCD_MINVAL : tolerance;
  memset(res, 0, sizeof(*res));
  v3sub(xk, x1_0, x2_0);
  double xnorm = sqrt(v3dot(xk, xk)), xnorm_prev = 0.0;
  for (int it = 0; it < iterations; it++) {
    if (xnorm < min_norm || fabs(xnorm_prev - xnorm) < CCD_MINVAL) break;
    double dneg[3] = {xk[0] / xnorm, xk[1] / xnorm, xk[2] / xnorm}, dpos[3];
    if (is_discrete && xnorm < 1e-4) { /* noisy direction near contact of polytopes: use the simplex's own geometry (596-632) */
      if (n == 2) {
        double e[3];
        v3sub(e, res->s[1], res->s[0]);
        double e2 = v3dot(e, e);
        if (e2 > CCD_MINVAL2) {
          double proj = v3dot(dneg, e) / e2;
          for (int k = 0; k < 3; k++) dneg[k] -= proj * e[k];
          double dn = v3len(dneg);
          if (dn > CCD_MINVAL)
            for (int k = 0; k < 3; k++) dneg[k] /= dn;
        }
      } else if (n == 3) {
        double e1[3], e2[3], nr[3];
        v3sub(e1, res->s[1], res->s[0]);
        v3sub(e2, res->s[2], res->s[0]);
        v3cross(nr, e1, e2);
        double nn = v3len(nr);
        if (nn > CCD_MINVAL) {
          double sg = ccd_sign(v3dot(dneg, nr));
          for (int k = 0; k < 3; k++) dneg[k] = sg * nr[k] / nn;
        }
      }
    }
    for (int k = 0; k < 3; k++) dpos[k] = -dneg[k];
    res->i1[n] = ccd_support(g1, dpos, res->s1[n]);
    res->i2[n] = ccd_support(g2, dneg, res->s2[n]);
    ((CcdGeom*)g1)->index = g1->cache; /* 675-680 (only meshes read it) */
    ((CcdGeom*)g2)->index = g2->cache;
    v3sub(res->s[n], res->s1[n], res->s2[n]);
    double gap[3];
    v3sub(gap, xk, res->s[n]);
    if (v3dot(xk, gap) < epsilon) break; /* Frank-Wolfe duality gap */
    double lower = v3dot(xk, res->s[n]);
    if ((cutoff == 0.0 && lower > 0.0) || (cutoff != 0.0 && cutoff < CCD_FLOAT_MAX && lower > 0.0 && lower >= cutoff * xnorm)) {
      memset(res, 0, sizeof(*res));
      res->separated = 1;
      res->dist = CCD_FLOAT_MAX;
      return;
    }
    ccd_subdistance(n + 1, res->s, lam);
    int m = 0;
    for (int i = 0; i < 4; i++) {
      if (lam[i] == 0.0) continue;
      v3cpy(res->s[m], res->s[i]);
      v3cpy(res->s1[m], res->s1[i]);
      v3cpy(res->s2[m], res->s2[i]);
      res->i1[m] = res->i1[i];
      res->i2[m] = res->i2[i];
      lam[m] = lam[i];
      m++;
    }
    n = m;
    if (n < 1) break;
    ccd_combine(n, lam, res->s, xk);
    xnorm_prev = xnorm;
    xnorm = sqrt(v3dot(xk, xk));
    if (n == 4) break;
  }
  res->separated = 0;
  if (n == 0) { v3cpy(res->x1, x1_0); v3cpy(res->x2, x2_0); }
  else { ccd_combine(n, lam, res->s1, res->x1); ccd_combine(n, lam, res->s2, res->x2); }
  if (xnorm > 0.0) {
    double dir[3] = {xk[0] / xnorm, xk[1] / xnorm, xk[2] / xnorm}, ndir[3] = {-dir[0], -dir[1], -dir[2]}, p1[3], p2[3], df[3];
    ccd_support(g1, ndir, p1);
    ccd_support(g2, dir, p2);
    v3sub(df, p1, p2);
    res->separated = v3dot(xk, df) > 0.0;
  }
  res->dist = (n == 4 && !res->separated) ? 0.0 : xnorm;
  res->dim = n;
  /* The reference's separation test (690-710) fires once the lower bound x_k . s reaches cutoff |x_k| (any positive value for cutoff 0).
     At convergence that bound equals |x_k|^2, so a pair whose distance is at least the cutoff is always reported as separated -- except
     that the duality-gap test above comes first in the loop and can stop the very iteration in which the bound would have been reached
     (always in float64; in float32, the reference's arithmetic, the gap x_k . (x_k - s) is rounding noise at that point and the outcome is
     a coin flip).  Only the height-field collider can tell the two outcomes apart (it keeps the witness points of separated prisms in its
     contact selection); this engine and its oracle both normalise to the separation test's verdict: */
  if (res->separated && xnorm > 0.0 && ((cutoff == 0.0) || (cutoff < CCD_FLOAT_MAX && xnorm >= cutoff))) {
    memset(res, 0, sizeof(*res));
    res->separated = 1;
    res->dist = CCD_FLOAT_MAX;
  }
}

/* ---- EPA ---------------------------------------------------------------------------------------------------------------- */
typedef struct Polytope {
  int status, nvert, nface, nhorizon, vcap, fcap;
  double center[3];
  double vert[2 * (5 + CCD_MAX_ITER)][3]; /* vertex i: 2i on geom 1, 2i + 1 on geom 2 */
  int vidx[2 * (5 + CCD_MAX_ITER)];
  int fv[6 + CCD_EPAFACES * CCD_MAX_ITER][3], fdel[6 + CCD_EPAFACES * CCD_MAX_ITER], finv[6 + CCD_EPAFACES * CCD_MAX_ITER];
  double fpr[6 + CCD_EPAFACES * CCD_MAX_ITER][3], fn2[6 + CCD_EPAFACES * CCD_MAX_ITER];
  int horizon[CCD_MAX_HORIZON];
} Polytope;

static void pt_diff(const Polytope* pt, int v, double* out) { v3sub(out, pt->vert[2 * v], pt->vert[2 * v + 1]); }

/* collision_gjk.py:227-250: face (v1, v2, v3) at slot idx; returns |projection of the origin on its plane|^2, 0 on failure */
static double pt_attach_face(Polytope* pt, int idx, int v1, int v2, int v3) {
  if (pt->nface == pt->fcap) return 0.0;
  double p1[3], p2[3], p3[3], r[3], df[3];
  pt_diff(pt, v1, p1);
  pt_diff(pt, v2, p2);
  pt_diff(pt, v3, p3);
  if (ccd_origin_on_plane(p3, p2, p1, r)) return 0.0;
  v3sub(df, p1, pt->center);
  if (v3dot(r, df) < 0.0)
    for (int k = 0; k < 3; k++) r[k] = -r[k];
  pt->fv[idx][0] = v1; pt->fv[idx][1] = v2; pt->fv[idx][2] = v3;
  pt->fdel[idx] = pt->finv[idx] = 0;
  v3cpy(pt->fpr[idx], r);
  pt->fn2[idx] = v3dot(r, r);
  return pt->fn2[idx];
}
static void pt_support(Polytope* pt, int idx, const CcdGeom* g1, const CcdGeom* g2, const double* dir) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  pt->vidx[2 * idx] = ccd_support(g1, dir, pt->vert[2 * idx]);
  pt->vidx[2 * idx + 1] = ccd_support(g2, nd, pt->vert[2 * idx + 1]);
}
static void pt_replace_simplex3(const Polytope* pt, int v1, int v2, int v3, GjkOut* res) { /* 846-882 */
  const int v[3] = {v1, v2, v3};
  for (int i = 0; i < 3; i++) {
    v3cpy(res->s1[i], pt->vert[2 * v[i]]);
    v3cpy(res->s2[i], pt->vert[2 * v[i] + 1]);
    v3sub(res->s[i], res->s1[i], res->s2[i]);
    res->i1[i] = pt->vidx[2 * v[i]];
    res->i2[i] = pt->vidx[2 * v[i] + 1];
  }
  res->dim = 3;
}
static int ccd_same_side(const double* p0, const double* p1, const double* p2, const double* p3) {
  double a[3], b[3], n[3], c[3], m0[3] = {-p0[0], -p0[1], -p0[2]};
  v3sub(a, p1, p0);
  v3sub(b, p2, p0);
  v3cross(n, a, b);
  v3sub(c, p3, p0);
  double d1 = v3dot(n, c), d2 = v3dot(n, m0);
  return (d1 > 0.0 && d2 > 0.0) || (d1 < 0.0 && d2 < 0.0);
}
static int ccd_test_tetra(const double* p0, const double* p1, const double* p2, const double* p3) {
  return ccd_same_side(p0, p1, p2, p3) && ccd_same_side(p1, p2, p3, p0) && ccd_same_side(p2, p3, p0, p1) && ccd_same_side(p3, p0, p1, p2);
}
static void ccd_tri_affine(const double* v1, const double* v2, const double* v3, const double* p, double* out) { /* 786-826 */
  double m14 = v2[1] * v3[2] - v2[2] * v3[1] - v1[1] * v3[2] + v1[2] * v3[1] + v1[1] * v2[2] - v1[2] * v2[1];
  double m24 = v2[0] * v3[2] - v2[2] * v3[0] - v1[0] * v3[2] + v1[2] * v3[0] + v1[0] * v2[2] - v1[2] * v2[0];
  double m34 = v2[0] * v3[1] - v2[1] * v3[0] - v1[0] * v3[1] + v1[1] * v3[0] + v1[0] * v2[1] - v1[1] * v2[0];
  double mu1 = fabs(m14), mu2 = fabs(m24), mu3 = fabs(m34), mmax;
  int x, y;
  if (mu1 >= mu2 && mu1 >= mu3) { mmax = m14; x = 1; y = 2; }
  else if (mu2 >= mu3) { mmax = m24; x = 0; y = 2; }
  else { mmax = m34; x = 0; y = 1; }
  out[0] = (p[x] * v2[y] + p[y] * v3[x] + v2[x] * v3[y] - p[x] * v3[y] - p[y] * v2[x] - v3[x] * v2[y]) / mmax;
  out[1] = (p[x] * v3[y] + p[y] * v1[x] + v3[x] * v1[y] - p[x] * v1[y] - p[y] * v3[x] - v1[x] * v3[y]) / mmax;
  out[2] = (p[x] * v1[y] + p[y] * v2[x] + v1[x] * v2[y] - p[x] * v2[y] - p[y] * v1[x] - v2[x] * v1[y]) / mmax;
}
static int ccd_tri_point_intersect(const double* v1, const double* v2, const double* v3, const double* p) {
  double l[3], pr[3], df[3];
  ccd_tri_affine(v1, v2, v3, p, l);
  if (l[0] < 0.0 || l[1] < 0.0 || l[2] < 0.0) return 0;
  for (int k = 0; k < 3; k++) pr[k] = v1[k] * l[0] + v2[k] * l[1] + v3[k] * l[2];
  v3sub(df, pr, p);
  return v3len(df) < CCD_MINVAL;
}
static int ccd_ray_triangle(const double* v1, const double* v2, const double* v3, const double* v4, const double* v5) { /* 907 */
  double a[3], b[3], c[3], e[3];
  v3sub(a, v3, v1);
  v3sub(b, v4, v1);
  v3sub(c, v5, v1);
  v3sub(e, v2, v1);
  double vol1 = ccd_det3(a, b, e), vol2 = ccd_det3(b, c, e), vol3 = ccd_det3(c, a, e);
  if (vol1 >= 0.0 && vol2 >= 0.0 && vol3 >= 0.0) return 1;
  if (vol1 <= 0.0 && vol2 <= 0.0 && vol3 <= 0.0) return -1;
  return 0;
}

/* seed polytopes from a 1-, 2-, 3-simplex (collision_gjk.py:1021-1111, 1114-1204, 1207-1286).  status 0: ready; -1: fall back
 * to the 2-simplex written into res; > 0: origin on the boundary (no penetration to recover) */
static void pt_seed2(Polytope* pt, GjkOut* res, const CcdGeom* g1, const CcdGeom* g2) {
  double df[3], e[3] = {0, 0, 0}, d1[3], d2[3], d3[3];
  v3sub(df, res->s[1], res->s[0]);
  for (int k = 0; k < 3; k++) pt->center[k] = 0.5 * (res->s[0][k] + res->s[1][k]);
  double val = CCD_FLOAT_MAX;
  int index = 0;
  for (int k = 0; k < 3; k++)
    if (fabs(df[k]) < val) { val = fabs(df[k]); index = k; }
  e[index] = 1.0;
  v3cross(d1, e, df);
  { /* rotation by 120 degrees about the segment (885-904) */
    double n = v3len(df), u1 = df[0] / n, u2 = df[1] / n, u3 = df[2] / n, sn = 0.86602540378, cs = -0.5;
    double R[9] = {cs + u1 * u1 * (1 - cs),      u1 * u2 * (1 - cs) - u3 * sn, u1 * u3 * (1 - cs) + u2 * sn,
                   u2 * u1 * (1 - cs) + u3 * sn, cs + u2 * u2 * (1 - cs),      u2 * u3 * (1 - cs) - u1 * sn,
                   u1 * u3 * (1 - cs) - u2 * sn, u2 * u3 * (1 - cs) + u1 * sn, cs + u3 * u3 * (1 - cs)};
    mat_mul_vec(d2, R, d1);
    mat_mul_vec(d3, R, d2);
  }
  for (int i = 0; i < 2; i++) {
    v3cpy(pt->vert[2 * i], res->s1[i]);
    v3cpy(pt->vert[2 * i + 1], res->s2[i]);
    pt->vidx[2 * i] = res->i1[i];
    pt->vidx[2 * i + 1] = res->i2[i];
  }
  v3normalize(d1); v3normalize(d2); v3normalize(d3);
  pt_support(pt, 2, g1, g2, d1);
  pt_support(pt, 3, g1, g2, d2);
  pt_support(pt, 4, g1, g2, d3);
  static const int F[6][3] = {{0, 2, 3}, {0, 4, 2}, {0, 3, 4}, {1, 3, 2}, {1, 2, 4}, {1, 4, 3}};
  for (int f = 0; f < 6; f++)
    if (pt_attach_face(pt, f, F[f][0], F[f][1], F[f][2]) < CCD_MIN_DIST2) {
      pt->status = -1;
      pt_replace_simplex3(pt, F[f][0], F[f][1], F[f][2], res);
      return;
    }
  double v2[3], v3[3], v4[3];
  pt_diff(pt, 2, v2);
  pt_diff(pt, 3, v3);
  pt_diff(pt, 4, v4);
  if (!ccd_ray_triangle(res->s[0], res->s[1], v2, v3, v4)) { pt->status = 1; return; }
  pt->nvert = 5; pt->nface = 6; pt->status = 0;
}
static void pt_seed3(Polytope* pt, const GjkOut* res, const CcdGeom* g1, const CcdGeom* g2) {
  double a[3], b[3], n[3], nn[3];
  for (int k = 0; k < 3; k++) pt->center[k] = (res->s[0][k] + res->s[1][k] + res->s[2][k]) * (1.0 / 3.0);
  v3sub(a, res->s[1], res->s[0]);
  v3sub(b, res->s[2], res->s[0]);
  v3cross(n, a, b);
  double norm = v3len(n);
  if (norm < CCD_MINVAL) { pt->status = 2; return; }
  for (int k = 0; k < 3; k++) { n[k] /= norm; nn[k] = -n[k]; }
  for (int i = 0; i < 3; i++) {
    v3cpy(pt->vert[2 * i], res->s1[i]);
    v3cpy(pt->vert[2 * i + 1], res->s2[i]);
    pt->vidx[2 * i] = res->i1[i];
    pt->vidx[2 * i + 1] = res->i2[i];
  }
  pt_support(pt, 3, g1, g2, nn);
  pt_support(pt, 4, g1, g2, n);
  double v4[3], v5[3];
  pt_diff(pt, 3, v4);
  pt_diff(pt, 4, v5);
  if (ccd_tri_point_intersect(res->s[0], res->s[1], res->s[2], v4)) { pt->status = 3; return; }
  if (ccd_tri_point_intersect(res->s[0], res->s[1], res->s[2], v5)) { pt->status = 4; return; }
  if (res->dist > 1e-5 && !ccd_test_tetra(res->s[0], res->s[1], res->s[2], v4) && !ccd_test_tetra(res->s[0], res->s[1], res->s[2], v5)) {
    pt->status = 5;
    return;
  }
  static const int F[6][3] = {{4, 0, 1}, {4, 2, 0}, {4, 1, 2}, {3, 1, 0}, {3, 0, 2}, {3, 2, 1}};
  for (int f = 0; f < 6; f++)
    if (pt_attach_face(pt, f, F[f][0], F[f][1], F[f][2]) < CCD_MIN_DIST3) { pt->status = 6 + f; return; }
  pt->nvert = 5; pt->nface = 6; pt->status = 0;
}
static void pt_seed4(Polytope* pt, GjkOut* res) {
  for (int k = 0; k < 3; k++) pt->center[k] = 0.25 * (res->s[0][k] + res->s[1][k] + res->s[2][k] + res->s[3][k]);
  for (int i = 0; i < 4; i++) {
    v3cpy(pt->vert[2 * i], res->s1[i]);
    v3cpy(pt->vert[2 * i + 1], res->s2[i]);
    pt->vidx[2 * i] = res->i1[i];
    pt->vidx[2 * i + 1] = res->i2[i];
  }
  static const int F[4][3] = {{0, 1, 2}, {0, 3, 1}, {0, 2, 3}, {3, 2, 1}};
  double dist[4];
  int idx = 0;
  for (int f = 0; f < 4; f++) { /* origin on a face: continue from that face as a 2-simplex */
    dist[f] = pt_attach_face(pt, f, F[f][0], F[f][1], F[f][2]);
    if (dist[f] < CCD_MIN_DIST4) {
      pt->status = -1;
      pt_replace_simplex3(pt, F[f][0], F[f][1], F[f][2], res);
      return;
    }
    if (f > 0 && dist[f] < dist[idx]) idx = f;
  }
  if (!ccd_test_tetra(res->s[0], res->s[1], res->s[2], res->s[3])) {
    if (dist[idx] > CCD_MINVAL) { pt->status = 12; return; }
    pt->status = -1;
    pt_replace_simplex3(pt, F[idx][0], F[idx][1], F[idx][2], res);
    return;
  }
  pt->nvert = 4; pt->nface = 4; pt->status = 0;
}

static int pt_add_edge(Polytope* pt, int e1, int e2) { /* 925-944: an edge shared by two deleted faces leaves the horizon */
  int n = pt->nhorizon;
  if (n < 0) return -1;
  int edge = ((e1 < e2 ? e1 : e2) << 10) | (e1 < e2 ? e2 : e1);
  for (int i = 0; i < n; i++)
    if (pt->horizon[i] == edge) { pt->horizon[i] = pt->horizon[n - 1]; return n - 1; }
  if (n == CCD_MAX_HORIZON) return -1;
  pt->horizon[n] = edge;
  return n + 1;
}

/* collision_gjk.py:1319-1454; returns the index of the closest face (-1: no contact) and dist / witness points */
static int ccd_epa(double tolerance, int iterations, Polytope* pt, const CcdGeom* g1, const CcdGeom* g2, int is_discrete, int* overflow,
                   double* dist_out, double* x1, double* x2) {
  double upper = CCD_FLOAT_MAX, upper2 = CCD_FLOAT_MAX, epsilon = is_discrete ? CCD_MIN_EPATOL : tolerance;
  int idx = -1, pidx, nvalid = pt->nface;
  if (iterations > 1000) iterations = 1000;
  for (int it = 0; it < iterations; it++) {
    pidx = idx;
    idx = -1;
    double lower2 = CCD_FLOAT_MAX;
    for (int i = 0; i < pt->nface; i++)
      if (!pt->fdel[i] && !pt->finv[i] && pt->fn2[i] < lower2) { idx = i; lower2 = pt->fn2[i]; }
    if (lower2 > upper2 || idx < 0) { idx = pidx; break; }
    if (lower2 <= 0.0) break;
    double lower = sqrt(lower2), dir[3], w[3];
    int wi = pt->nvert;
    for (int k = 0; k < 3; k++) dir[k] = pt->fpr[idx][k] / lower;
    pt_support(pt, wi, g1, g2, dir);
    ((CcdGeom*)g1)->index = g1->cache; /* 1370-1373 */
    ((CcdGeom*)g2)->index = g2->cache;
    pt_diff(pt, wi, w);
    pt->nvert++;
    double upper_k = v3dot(pt->fpr[idx], w) / lower;
    if (upper_k < upper) { upper = upper_k; upper2 = upper * upper; }
    if (upper - lower < epsilon) break;
    if (is_discrete) { /* a repeated support vertex pair: the polytope cannot grow further */
      int rep = 0;
      for (int i = 0; i < pt->nvert - 1 && !rep; i++)
        rep = pt->vidx[2 * i] == pt->vidx[2 * wi] && pt->vidx[2 * i + 1] == pt->vidx[2 * wi + 1];
      if (rep) break;
    }
    nvalid--;
    pt->fdel[idx] = 1;
    pt->nhorizon = pt_add_edge(pt, pt->fv[idx][0], pt->fv[idx][1]);
    pt->nhorizon = pt_add_edge(pt, pt->fv[idx][1], pt->fv[idx][2]);
    pt->nhorizon = pt_add_edge(pt, pt->fv[idx][2], pt->fv[idx][0]);
    if (pt->nhorizon == -1) { *overflow |= OVF_EPA_HORIZON; idx = -1; break; }
    for (int i = 0; i < pt->nface; i++) { /* every face that sees w goes; its edges toggle in the horizon */
      if (pt->fdel[i]) continue;
      if (v3dot(pt->fpr[i], w) - pt->fn2[i] > 1e-10) {
        if (!pt->finv[i]) nvalid--;
        pt->fdel[i] = 1;
        pt->nhorizon = pt_add_edge(pt, pt->fv[i][0], pt->fv[i][1]);
        pt->nhorizon = pt_add_edge(pt, pt->fv[i][1], pt->fv[i][2]);
        pt->nhorizon = pt_add_edge(pt, pt->fv[i][2], pt->fv[i][0]);
        if (pt->nhorizon == -1) { *overflow |= OVF_EPA_HORIZON; idx = -1; break; }
      }
    }
    for (int i = 0; i < pt->nhorizon; i++) {
      int e0 = pt->horizon[i] & 0x3FF, e1 = (pt->horizon[i] >> 10) & 0x3FF;
      double d2 = pt_attach_face(pt, pt->nface, wi, e0, e1);
      if (d2 == 0.0) { idx = -1; break; }
      pt->nface++;
      if (d2 >= lower2 && d2 <= upper2) nvalid++;
      else pt->finv[pt->nface - 1] = 1;
    }
    if (nvalid == 0 || idx == -1) break;
    pt->nhorizon = 0;
  }
  if (idx < 0) { *dist_out = 0.0; return -1; }
  { /* witness points: affine coordinates of the origin's projection on the closest face (947-1018) */
    double v1[3], v2[3], v3[3], l[3];
    const int* f = pt->fv[idx];
    pt_diff(pt, f[0], v1);
    pt_diff(pt, f[1], v2);
    pt_diff(pt, f[2], v3);
    ccd_tri_affine(v1, v2, v3, pt->fpr[idx], l);
    for (int k = 0; k < 3; k++) {
      x1[k] = pt->vert[2 * f[0]][k] * l[0] + pt->vert[2 * f[1]][k] * l[1] + pt->vert[2 * f[2]][k] * l[2];
      x2[k] = pt->vert[2 * f[0] + 1][k] * l[0] + pt->vert[2 * f[1] + 1][k] * l[1] + pt->vert[2 * f[2] + 1][k] * l[2];
    }
    *dist_out = -sqrt(pt->fn2[idx]);
  }
  return idx;
}

/* Branch trace of ccd_run (round 5: which gate of gjk_phase, collision_gjk.py:2376-2414, a pair leaves through; read by tools / tests through
 * ref_ccd_trace in mjref.c): 1 shrunk cores separated (inflate, 2396-2400) | 2 GJK distance above the tolerance (2412) | 3 simplex of fewer than
 * two points (2412) | 4 GJK's own `separated` flag (2412) | 5 polytope seed degenerate: GJK's answer stands | 6 EPA failed | 7 EPA depth. */
static int g_ccd_branch, g_ccd_gjk_dim, g_ccd_gjk_sep;
static double g_ccd_gjk_dist;

static int ccd_discrete(int t1, int t2) { return (t1 == G_BOX || t1 == G_MESH || t1 == G_HFIELD) && (t2 == G_BOX || t2 == G_MESH || t2 == G_HFIELD); } /* 109 */

/* ccd = gjk_phase + epa_phase (collision_gjk.py:2350-2575).  Returns the number of contacts (0 / 1); *face_out = closest EPA
 * face when the pair qualifies for multi-contact recovery (boxes, zero margin), else -1; pt_out receives the final polytope */
static int ccd_run(double tolerance, double cutoff, int gjk_iterations, int epa_iterations, CcdGeom g1, CcdGeom g2, double* dist_out,
                   double* x1, double* x2, int* overflow, int* face_out, Polytope* pt) {
  const CcdGeom o1 = g1, o2 = g2;
  double full1 = 0.0, full2 = 0.0, size1 = 0.0, size2 = 0.0;
  int is_discrete = ccd_discrete(g1.type, g2.type) && g1.margin == 0.0 && g2.margin == 0.0;
  GjkOut res;
  *face_out = -1;
  /* spheres and capsules shrink to a point / segment: shallow penetrations are then separations of the cores (2376-2400) */
  if (g1.type == G_SPHERE || g1.type == G_CAPSULE) { size1 = g1.size[0]; full1 = size1 + 0.5 * g1.margin; g1.margin = 0.0; g1.size[0] = 0.0; }
  if (g2.type == G_SPHERE || g2.type == G_CAPSULE) { size2 = g2.size[0]; full2 = size2 + 0.5 * g2.margin; g2.margin = 0.0; g2.size[0] = 0.0; }
  if (size1 + size2 > 0.0) {
    cutoff += full1 + full2;
    ccd_gjk(tolerance, gjk_iterations, &g1, &g2, g1.pos, g2.pos, cutoff, is_discrete, &res);
    if (res.dist > tolerance) {
      g_ccd_branch = 1; g_ccd_gjk_dist = res.dist; g_ccd_gjk_dim = res.dim; g_ccd_gjk_sep = res.separated;
      *dist_out = res.dist;
      v3cpy(x1, res.x1);
      v3cpy(x2, res.x2);
      if (res.dist == CCD_FLOAT_MAX) return 1;
      double n[3];
      v3sub(n, res.x2, res.x1);
      v3normalize(n);
      if (full1 > 0.0) for (int k = 0; k < 3; k++) x1[k] += full1 * n[k];
      if (full2 > 0.0) for (int k = 0; k < 3; k++) x2[k] -= full2 * n[k];
      *dist_out = res.dist - (full1 + full2);
      return 1;
    }
    g1.margin = o1.margin; /* (the cached mesh vertex of the first run stays: collision_gjk.py:2403-2406 restores margin and size only) */
    g2.margin = o2.margin;
    for (int k = 0; k < 3; k++) { g1.size[k] = o1.size[k]; g2.size[k] = o2.size[k]; }
    cutoff -= full1 + full2;
  }
  ccd_gjk(tolerance, gjk_iterations, &g1, &g2, g1.pos, g2.pos, cutoff, is_discrete, &res);
  *dist_out = res.dist;
  v3cpy(x1, res.x1);
  v3cpy(x2, res.x2);
  g_ccd_gjk_dist = res.dist; g_ccd_gjk_dim = res.dim; g_ccd_gjk_sep = res.separated;
  g_ccd_branch = res.dist > tolerance ? 2 : (res.dim < 2 ? 3 : (res.separated ? 4 : 7));
  if (res.dist > tolerance || res.dim < 2 || res.separated) return 1;
  /* ---- EPA ---- */
  if (epa_iterations > CCD_MAX_ITER) epa_iterations = CCD_MAX_ITER;
  memset(pt, 0, sizeof(*pt));
  pt->vcap = 5 + epa_iterations;
  pt->fcap = 6 + CCD_EPAFACES * epa_iterations;
  if (res.dim == 2) pt_seed2(pt, &res, &g1, &g2);
  else if (res.dim == 4) pt_seed4(pt, &res);
  if (res.dim == 3) { /* also the fall-back of the other two seeds */
    pt->status = 0;
    pt_seed3(pt, &res, &g1, &g2);
  }
  if (pt->status) { g_ccd_branch = 5; return 1; } /* origin on the boundary: GJK's answer stands */
  double dist;
  int idx = ccd_epa(tolerance, epa_iterations, pt, &g1, &g2, is_discrete, overflow, &dist, x1, x2);
  if (idx == -1) { g_ccd_branch = 6; *dist_out = CCD_FLOAT_MAX; return 0; }
  *dist_out = dist;
  if (g1.margin == 0.0 && g2.margin == 0.0 && (g1.type == G_BOX || g1.type == G_MESH) && (g2.type == G_BOX || g2.type == G_MESH)) *face_out = idx; /* 2517-2523 */
  return 1;
}

/* ---- multi-contact recovery for box pairs (collision_gjk.py:2076-2300 multicontact, box branches only) -------------------------------
 * From the EPA face closest to the origin: the features (vertex / edge / face) of the two boxes it was built from, the box
 * faces whose normals oppose each other within FACE_TOL (or an edge perpendicular to a face within EDGE_TOL), then the clipping of
 * one face (or edge) against the side planes of the other (Sutherland-Hodgman), pruned to the quadrilateral of largest area. */
#define CCD_FACE_TOL 0.99999872000027307    /* cos(0.0016) */
#define CCD_EDGE_TOL 0.0015999993173334207  /* sin(0.0016) */
#define CCD_INTERSECT_TOL 0.0000003

static int mc_feature_dim(const Polytope* pt, const int* face, int offset, int* fidx, double fvert[3][3]) { /* 1503-1526 */
  for (int k = 0; k < 3; k++) {
    fidx[k] = pt->vidx[2 * face[k] + offset];
    v3cpy(fvert[k], pt->vert[2 * face[k] + offset]);
  }
  if (fidx[0] != fidx[1]) return (fidx[2] == fidx[0] || fidx[2] == fidx[1]) ? 2 : 3;
  fidx[1] = fidx[2];
  v3cpy(fvert[1], fvert[2]);
  return fidx[0] != fidx[2] ? 2 : 1;
}
static const double MC_FACE_NORMALS[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
static int mc_box_normals2(const double* mat, const double* n, double nout[3][3], int* iout) { /* 1703-1733 */
  double ln[3];
  matT_mul_vec(ln, mat, n);
  v3normalize(ln);
  for (int i = 0; i < 6; i++)
    if (v3dot(ln, MC_FACE_NORMALS[i]) > CCD_FACE_TOL) {
      mat_mul_vec(nout[0], mat, MC_FACE_NORMALS[i]);
      iout[0] = i;
      return 1;
    }
  return 0;
}
static int mc_box_normals(int dim, const int* fi, const double* mat, const double* dir, double nout[3][3], int* iout) { /* 1736-1807 */
  int v1 = fi[0], v2 = fi[1], v3 = fi[2];
  if (dim == 3) {
    int c = 0;
    double ax[3];
    for (int k = 0; k < 3; k++) {
      int b = 1 << k;
      ax[k] = (double)((v1 & b) && (v2 & b) && (v3 & b)) - (double)(!(v1 & b) && !(v2 & b) && !(v3 & b));
    }
    mat_mul_vec(nout[0], mat, ax);
    double sgn = ax[0] + ax[1] + ax[2];
    for (int k = 0; k < 3; k++)
      if (ax[k] != 0.0) iout[c++] = 2 * k;
    if (sgn == -1.0) iout[0] = iout[0] + 1;
    if (c == 1) return 1;
    return mc_box_normals2(mat, dir, nout, iout);
  }
  if (dim == 2) {
    int c = 0;
    for (int k = 0; k < 3; k++) {
      int b = 1 << k;
      double a = (double)((v1 & b) && (v2 & b)) - (double)(!(v1 & b) && !(v2 & b));
      if (a != 0.0) {
        double e[3] = {0, 0, 0};
        e[k] = a;
        mat_mul_vec(nout[c], mat, e);
        iout[c] = a > 0.0 ? 2 * k : 2 * k + 1;
        c++;
      }
    }
    if (c == 1 || c == 2) return c; /* 1: diagonal of a face, 2: a box edge */
    return mc_box_normals2(mat, dir, nout, iout);
  }
  if (dim == 1) {
    for (int k = 0; k < 3; k++) {
      double e[3] = {0, 0, 0};
      e[k] = (v1 & (1 << k)) ? 1.0 : -1.0;
      mat_mul_vec(nout[k], mat, e);
      iout[k] = e[k] > 0.0 ? 2 * k : 2 * k + 1;
    }
    return 3;
  }
  return 0;
}
static int mc_box_edge_normals(int dim, const CcdGeom* g, const double* v1, const double* v2, int v1i, double nout[3][3], double endv[3][3]) {
  if (dim == 2) { /* 1810-1845 */
    v3cpy(endv[0], v2);
    v3sub(nout[0], v2, v1);
    v3normalize(nout[0]);
    return 1;
  }
  if (dim == 1) {
    double c[3] = {(v1i & 1) ? g->size[0] : -g->size[0], (v1i & 2) ? g->size[1] : -g->size[1], (v1i & 4) ? g->size[2] : -g->size[2]};
    for (int k = 0; k < 3; k++) {
      double a[3] = {c[0], c[1], c[2]};
      a[k] = -a[k];
      mat_mul_vec(endv[k], g->rot, a);
      v3add(endv[k], endv[k], g->pos);
      v3sub(nout[k], endv[k], v1);
      v3normalize(nout[k]);
    }
    return 3;
  }
  return 0;
}
static int mc_box_face(const CcdGeom* g, int idx, double face[4][3]) { /* 1848-1888 */
  static const double C[6][4][3] = {
    {{1, 1, 1}, {1, 1, -1}, {1, -1, -1}, {1, -1, 1}},     {{-1, 1, -1}, {-1, 1, 1}, {-1, -1, 1}, {-1, -1, -1}},
    {{-1, 1, -1}, {1, 1, -1}, {1, 1, 1}, {-1, 1, 1}},     {{-1, -1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1}},
    {{-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}},     {{1, 1, -1}, {-1, 1, -1}, {-1, -1, -1}, {1, -1, -1}}};
  if (idx < 0 || idx > 5) return 0;
  for (int i = 0; i < 4; i++) {
    double c[3] = {C[idx][i][0] * g->size[0], C[idx][i][1] * g->size[1], C[idx][i][2] * g->size[2]};
    mat_mul_vec(face[i], g->rot, c);
    v3add(face[i], face[i], g->pos);
  }
  return 4;
}
static double mc_area4(const double* a, const double* b, const double* c, const double* d) { /* 1457 */
  double ad[3], db[3], bc[3], ca[3], x1[3], x2[3];
  v3sub(ad, a, d); v3sub(db, d, b); v3sub(bc, b, c); v3sub(ca, c, a);
  v3cross(x1, ad, db);
  v3cross(x2, bc, ca);
  v3add(x1, x1, x2);
  return 0.5 * v3len(x1);
}
static void mc_polygon_quad(double poly[][3], int n, int* res) { /* 1463-1499: rotating search for the largest quadrilateral */
  int b = 1, c = 2, d = 3;
  res[0] = 0; res[1] = b; res[2] = c; res[3] = d;
  double m = mc_area4(poly[0], poly[b], poly[c], poly[d]);
  for (int a = 0; a < n; a++) {
    for (;;) {
      double mn = mc_area4(poly[a], poly[b], poly[c], poly[(d + 1) % n]);
      if (mn <= m) break;
      m = mn;
      d = (d + 1) % n;
      res[0] = a; res[1] = b; res[2] = c; res[3] = d;
      for (;;) {
        mn = mc_area4(poly[a], poly[b], poly[(c + 1) % n], poly[d]);
        if (mn <= m) break;
        m = mn;
        c = (c + 1) % n;
        res[0] = a; res[1] = b; res[2] = c; res[3] = d;
      }
      for (;;) {
        mn = mc_area4(poly[a], poly[(b + 1) % n], poly[c], poly[d]);
        if (mn <= m) break;
        m = mn;
        b = (b + 1) % n;
        res[0] = a; res[1] = b; res[2] = c; res[3] = d;
      }
    }
    if (b == a) {
      b = (b + 1) % n;
      if (c == b) {
        c = (c + 1) % n;
        if (d == c) d = (d + 1) % n;
      }
    }
  }
}
/* clip polygon face2 against the side planes of face1 (normal n); returns the number of contacts, w2 = clipped points, w1 = w2 - dir */
static int g_mc_cap = 8; /* slots of the clip buffers = 2 * npolygonmax (collision_convex.py:1226-1234): 8 for box-only models */
static int mc_polygon_clip(double face1[][3], int nface1, double face2[][3], int nface2, const double* n, const double* dir, double w1[4][3],
                           double w2[4][3]) { /* 1941-2056 */
  if (nface1 < 3) return 0;
  double pn[MC_MAXN][3], pd[MC_MAXN], bufa[2 * MC_MAXN][3], bufb[2 * MC_MAXN][3];
  const int cap = g_mc_cap < 2 * MC_MAXN ? g_mc_cap : 2 * MC_MAXN;
  double(*poly)[3] = bufa;
  double(*clip)[3] = bufb;
  for (int i = 0; i < nface1; i++) {
    const double *a = face1[i], *b = face1[(i + 1) % nface1];
    double v3[3], e1[3], e2[3];
    v3add(v3, a, n);
    v3sub(e1, b, a);
    v3sub(e2, v3, a);
    v3cross(pn[i], e1, e2);
    pd[i] = v3dot(pn[i], a);
  }
  int np = nface2, nc = 0;
  for (int i = 0; i < nface2; i++) v3cpy(poly[i], face2[i]);
  for (int e = 0; e < nface1; e++) {
    for (int i = 0; i < np; i++) {
      const double *P = poly[i], *Q = poly[(i + 1) % np];
      double dp[3], dq[3];
      v3sub(dp, P, face1[e]);
      v3sub(dq, Q, face1[e]);
      int in1 = v3dot(dp, pn[e]) > -1e-10, in2 = v3dot(dq, pn[e]) > -1e-10;
      if (!in1 && !in2) continue;
      if (in1 && in2) { if (nc < cap) v3cpy(clip[nc], Q); nc++; continue; }
      double pq[3];
      v3sub(pq, Q, P);
      double dt = v3dot(pn[e], pq), t = fabs(dt) < 1e-10 ? CCD_FLOAT_MAX : (pd[e] - v3dot(pn[e], P)) / dt;
      if (t > -CCD_INTERSECT_TOL && t < 1.0 + CCD_INTERSECT_TOL) {
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        if (nc < cap) v3addscl(clip[nc], P, pq, t);
        nc++;
      }
      if (in2) { if (nc < cap) v3cpy(clip[nc], Q); nc++; }
    }
    if (nc > cap) nc = cap; /* (the reference's buffers hold 2 * npolygonmax points) */
    double(*tmp)[3] = poly; poly = clip; clip = tmp;
    np = nc;
    nc = 0;
  }
  if (np < 1) return 0;
  if (nface2 == 2 && np > 2) { /* an edge: keep the two points farthest apart */
    int b1 = 0, b2 = 1;
    double maxd = 0.0;
    for (int i = 0; i < np; i++)
      for (int j = i + 1; j < np; j++) {
        double df[3];
        v3sub(df, poly[j], poly[i]);
        double d2 = v3dot(df, df);
        if (d2 > maxd) { maxd = d2; b1 = i; b2 = j; }
      }
    v3cpy(w2[0], poly[b1]); v3sub(w1[0], w2[0], dir);
    v3cpy(w2[1], poly[b2]); v3sub(w1[1], w2[1], dir);
    return 2;
  }
  if (np > 4) {
    int q[4];
    mc_polygon_quad(poly, np, q);
    for (int i = 0; i < 4; i++) { v3cpy(w2[i], poly[q[i]]); v3sub(w1[i], w2[i], dir); }
    return 4;
  }
  for (int i = 0; i < np; i++) { v3cpy(w2[i], poly[i]); v3sub(w1[i], w2[i], dir); }
  return np;
}
/* ---- mesh features (collision_gjk.py:1556-1700, 1891-1913) ---- */
static int mc_intersect(const int* a1, int n1, const int* a2, int n2, int* res) { /* up to two common entries: _intersect1 / _intersect2 */
  int count = 0;
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++)
      if (a1[i] == a2[j]) {
        res[count++] = a1[i];
        if (count == 2) return 2;
      }
  return count;
}
static int mc_mesh_normals(int dim, const int* fi, const CcdGeom* g, double nout[][3], int* iout) { /* 1585-1652 */
  const int *m1 = g->polymap + g->polymapadr[fi[0]], n1 = g->polymapnum[fi[0]];
  if (dim == 3) {
    int edgeset[2], faceset[2];
    int n = mc_intersect(m1, n1, g->polymap + g->polymapadr[fi[1]], g->polymapnum[fi[1]], edgeset);
    if (n == 0) return 0;
    n = mc_intersect(edgeset, n, g->polymap + g->polymapadr[fi[2]], g->polymapnum[fi[2]], faceset);
    if (n == 0) return 0;
    mat_mul_vec(nout[0], g->rot, g->polynormal + 3 * faceset[0]); /* three vertices of a mesh define one face */
    iout[0] = faceset[0];
    return 1;
  }
  if (dim == 2) {
    int edgeset[2];
    int n = mc_intersect(m1, n1, g->polymap + g->polymapadr[fi[1]], g->polymapnum[fi[1]], edgeset);
    for (int i = 0; i < n; i++) {
      mat_mul_vec(nout[i], g->rot, g->polynormal + 3 * edgeset[i]);
      iout[i] = edgeset[i];
    }
    return n;
  }
  if (dim == 1) {
    int n = n1 < MC_MAXN ? n1 : MC_MAXN;
    for (int i = 0; i < n; i++) {
      mat_mul_vec(nout[i], g->rot, g->polynormal + 3 * m1[i]);
      iout[i] = m1[i];
    }
    return n;
  }
  return 0;
}
static int mc_mesh_edge_normals(int dim, const CcdGeom* g, const double* v1, const double* v2, int v1i, double nout[][3], double endv[][3]) { /* 1655-1700 */
  if (dim == 2) {
    v3cpy(endv[0], v2);
    v3sub(nout[0], v2, v1);
    v3normalize(nout[0]);
    return 1;
  }
  if (dim == 1) {
    const int* pm = g->polymap + g->polymapadr[v1i];
    int n = g->polymapnum[v1i] < MC_MAXN ? g->polymapnum[v1i] : MC_MAXN;
    for (int i = 0; i < n; i++) { /* in every polygon around the vertex: the edge to the previous vertex of the polygon */
      int adr = g->polyvertadr[pm[i]], nv = g->polyvertnum[pm[i]];
      for (int j = 0; j < nv; j++)
        if (g->polyvert[adr + j] == v1i) {
          int k = j == 0 ? nv - 1 : j - 1;
          mat_mul_vec(endv[i], g->rot, g->vert + 3 * g->polyvert[adr + k]);
          v3add(endv[i], endv[i], g->pos);
          v3sub(nout[i], endv[i], v1);
          v3normalize(nout[i]);
        }
    }
    return n;
  }
  return 0;
}
static int mc_mesh_face(const CcdGeom* g, int idx, double face[][3]) { /* 1891-1913: the polygon in reverse order */
  int adr = g->polyvertadr[idx], nv = g->polyvertnum[idx], j = 0;
  if (nv > MC_MAXN) nv = MC_MAXN;
  for (int i = nv - 1; i >= 0; i--) {
    mat_mul_vec(face[j], g->rot, g->vert + 3 * g->polyvert[adr + i]);
    v3add(face[j], face[j], g->pos);
    j++;
  }
  return nv;
}
/* multicontact (collision_gjk.py:2076-2300), boxes and meshes.  Returns the number of contacts (>= 1) and their witness points; x1 / x2 =
   EPA's witness points (contact 0 when nothing is recovered) */
static int ccd_multicontact_box(const Polytope* pt, int epa_face, const double* x1, const double* x2, const CcdGeom* g1, const CcdGeom* g2,
                                double w1[4][3], double w2[4][3]) {
  v3cpy(w1[0], x1);
  v3cpy(w2[0], x2);
  const int* face = pt->fv[epa_face];
  const int mesh1 = g1->type == G_MESH, mesh2 = g2->type == G_MESH;
  int fi1[3], fi2[3], idx1[MC_MAXN], idx2[MC_MAXN];
  double fv1[3][3], fv2[3][3], n1[MC_MAXN][3], n2[MC_MAXN][3], endv[MC_MAXN][3], dir[3], dneg[3];
  int nf1 = mc_feature_dim(pt, face, 0, fi1, fv1), nf2 = mc_feature_dim(pt, face, 1, fi2, fv2);
  v3sub(dir, x2, x1);
  for (int k = 0; k < 3; k++) dneg[k] = -dir[k];
  int nn1 = mesh1 ? mc_mesh_normals(nf1, fi1, g1, n1, idx1) : mc_box_normals(nf1, fi1, g1->rot, dneg, n1, idx1);
  int nn2 = mesh2 ? mc_mesh_normals(nf2, fi2, g2, n2, idx2) : mc_box_normals(nf2, fi2, g2->rot, dir, n2, idx2);
  int edge1 = 0, edge2 = 0, found = 0, ri = 0, rj = 0;
  for (int i = 0; i < nn1 && !found; i++) /* _aligned_faces 1529 */
    for (int j = 0; j < nn2 && !found; j++)
      if (v3dot(n1[i], n2[j]) < -CCD_FACE_TOL) { ri = i; rj = j; found = 1; }
  if (!found) {
    if (nf1 < 3 && nf1 <= nf2) { /* an edge (or vertex) of geom 1 against a face of geom 2 */
      nn1 = mesh1 ? mc_mesh_edge_normals(nf1, g1, fv1[0], fv1[1], fi1[0], n1, endv) : mc_box_edge_normals(nf1, g1, fv1[0], fv1[1], fi1[0], n1, endv);
      for (int i = 0; i < nn2 && !found; i++) /* _aligned_face_edge(edge = n1, face = n2) 1543 */
        for (int j = 0; j < nn1 && !found; j++)
          if (fabs(v3dot(n1[j], n2[i])) < CCD_EDGE_TOL) { ri = j; rj = i; found = 1; }
      if (!found) return 1;
      edge1 = 1;
    } else if (nf2 < 3) {
      nn2 = mesh2 ? mc_mesh_edge_normals(nf2, g2, fv2[0], fv2[1], fi2[0], n2, endv) : mc_box_edge_normals(nf2, g2, fv2[0], fv2[1], fi2[0], n2, endv);
      for (int i = 0; i < nn1 && !found; i++)
        for (int j = 0; j < nn2 && !found; j++)
          if (fabs(v3dot(n2[j], n1[i])) < CCD_EDGE_TOL) { ri = j; rj = i; found = 1; }
      if (!found) return 1;
      edge2 = 1;
    } else {
      return 1;
    }
  }
  double face1[MC_MAXN][3], face2[MC_MAXN][3], approx[3];
  int nface1, nface2;
  if (edge1) { v3cpy(face1[0], pt->vert[2 * face[0]]); v3cpy(face1[1], endv[ri]); nface1 = 2; }
  else {
    int ind = edge2 ? idx1[rj] : idx1[ri];
    nface1 = mesh1 ? mc_mesh_face(g1, ind, face1) : mc_box_face(g1, ind, face1);
  }
  if (edge2) { v3cpy(face2[0], pt->vert[2 * face[0] + 1]); v3cpy(face2[1], endv[ri]); nface2 = 2; }
  else nface2 = mesh2 ? mc_mesh_face(g2, idx2[rj], face2) : mc_box_face(g2, idx2[rj], face2);
  double dn = v3len(dir);
  if (edge1) { /* clip the edge of geom 1 against the face of geom 2; the roles of the witness arrays swap back afterwards */
    for (int k = 0; k < 3; k++) approx[k] = -dn * n2[rj][k];
    return mc_polygon_clip(face2, nface2, face1, nface1, n2[rj], approx, w2, w1);
  }
  if (edge2) {
    for (int k = 0; k < 3; k++) approx[k] = -dn * n1[rj][k];
    return mc_polygon_clip(face1, nface1, face2, nface2, n1[rj], approx, w1, w2);
  }
  for (int k = 0; k < 3; k++) approx[k] = dn * n2[rj][k];
  return mc_polygon_clip(face1, nface1, face2, nface2, n1[ri], approx, w1, w2);
}
