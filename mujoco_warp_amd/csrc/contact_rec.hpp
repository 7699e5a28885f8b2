// contact_rec.hpp -- the contact record handed from the narrowphase to make_constraint / the solvers, and its publication to the public
// contact.* arrays.  Split from collide.hpp (round 4) so that the solver translation units, whose launches carry the publication riders,
// do not recompile when the colliders change.
#pragma once
#include "dev_common.hpp"

// Contact record (CON_REC words; CON_STRIDE in the per-world hand-off buffer d.ws_contact, so that a record is one
// aligned 128-byte line):
//   0 dist | 1-3 pos | 4-12 frame | 13 includemargin | 14-16 friction (slide, spin, roll) | 17-18 solref |
//   19-23 solimp | 24 condim | 25-26 geoms | 27 collider contact id | (explicit pair id + 1) << 8 | 28 first efc row or -1 | 29 number of rows |
//   30-31 friction of tangent 2 / roll 2 (explicit <contact><pair> entries may be anisotropic; geom pairs repeat words 14 / 16)
// (28-29 are filled by k_make_constraint).  k_collision hands the contacts of a world to k_make_constraint through
// d.ws_contact[w]; the public, compact contact_* arrays are produced from the same records by publish_body
// (appended to the integrator launch or run as k_publish_contacts).  The reference reserves public slots with one global atomic per
// contact (collision_core.py write_contact); on MI355X one same-address device atomic per WORLD already cost 25-50 us
// per launch (they resolve at the memory side, ~6 ns each, and every later load of the wave waits behind them).
#define CON_WINDOW 16
#define CON_REC 32
#define CON_LDS 33  /* odd LDS stride: lane-per-contact reads are bank-conflict free */

// ---- publication of the compact public contact arrays (off the critical path) ---------------------------------
// publish_body: one group per world copies its records to the public SoA arrays (consecutive addresses per array),
// fills contact.efc_address and the contact rows of efc.id.  Worlds are published in world order, so the public
// arrays are deterministic (the reference's order depends on atomic arrival).
// self_prefix: the workgroup first sums ws_ncon over all earlier worlds itself (the counts are L2-resident: 32 KB at
// 8192 worlds), so that no scan kernel has to run before it; the workgroup of the last world also writes the totals.
// `sh` needs 64 ints of LDS.
template <int G>
DEV void publish_body(const MjhData& d, int self_prefix, int* sh, const Blk& b, const float* pair_solreffriction = nullptr) {
  if ((int)threadIdx.x >= b.nthreads) return;
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  int adr_self = 0;
  if (self_prefix) {
    const int t = threadIdx.x, nwave = (b.nthreads + 63) / 64;
    const bool last = b.w0 + b.nw >= d.nworld;
    int s = 0, s2 = 0;
    {  // 16-byte loads, all in flight before the first add (a dependent scalar loop costs ~0.5 us per trip)
      const int4* p4 = reinterpret_cast<const int4*>(d.ws_ncon);
      const int n4 = b.w0 >> 2;
#pragma unroll 8
      for (int i = t; i < n4; i += b.nthreads) {
        const int4 v = p4[i];
        s += v.x + v.y + v.z + v.w;
      }
      for (int i = (n4 << 2) + t; i < b.w0; i += b.nthreads) s += d.ws_ncon[i];
    }
    if (last) {
      const int4* p4 = reinterpret_cast<const int4*>(d.ws_ncollision);
      const int n4 = d.nworld >> 2;
#pragma unroll 8
      for (int i = t; i < n4; i += b.nthreads) {
        const int4 v = p4[i];
        s2 += v.x + v.y + v.z + v.w;
      }
      for (int i = (n4 << 2) + t; i < d.nworld; i += b.nthreads) s2 += d.ws_ncollision[i];
    }
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    if ((t & 63) == 0) {
      sh[t >> 6] = s;
      sh[16 + (t >> 6)] = s2;
    }
    __syncthreads();
    int base = 0, tot2 = 0;
    for (int k = 0; k < nwave; ++k) {
      base += sh[k];
      tot2 += sh[16 + k];
    }
    if (w < d.nworld) {
      // exclusive prefix inside the workgroup: worlds of a workgroup are consecutive
      int adr = base;
      for (int i = b.w0; i < w; ++i) adr += d.ws_ncon[i];
      adr_self = adr;
      if (lig == 0) d.ws_conadr[w] = adr;
      if (last && w == d.nworld - 1 && lig == 0) {
        d.nacon[0] = adr + d.ws_ncon[w];
        d.ncollision[0] = tot2;
      }
    }
  }
  if (w >= d.nworld) return;
  const int ncon = d.ws_ncon[w], adr = self_prefix ? adr_self : d.ws_conadr[w], njmax = d.njmax, npyr = d.nmaxpyramid;
  int n = ncon;
  if (adr + n > d.naconmax) n = max(0, d.naconmax - adr);
  if (n < ncon && lig == 0) atomicOr(d.overflow + w, OVF_NARROWPHASE);
  const float* rec = d.ws_contact + (size_t)w * d.concap * CON_STRIDE;
  const int* reci = reinterpret_cast<const int*>(rec);
  const size_t o0 = (size_t)adr;
  for (int c = lig; c < n; c += G) {  // one contact per lane: the scalar-per-contact arrays
    const float* r = rec + c * CON_STRIDE;
    const int* ri = reci + c * CON_STRIDE;
    const size_t o = o0 + c;
    d.contact_dist[o] = r[0];
    d.contact_includemargin[o] = r[13];
    *reinterpret_cast<float2*>(d.contact_solref + 2 * o) = float2{r[17], r[18]};
    {  // only explicit <contact><pair> entries carry a solreffriction (collision_core.py contact_params)
      const int pid = (ri[27] >> 8) - 1;
      *reinterpret_cast<float2*>(d.contact_solreffriction + 2 * o) = (pid >= 0 && pair_solreffriction) ? float2{pair_solreffriction[2 * pid], pair_solreffriction[2 * pid + 1]} : float2{0.0f, 0.0f};
    }
    d.contact_dim[o] = ri[24];
    *reinterpret_cast<int2*>(d.contact_geom + 2 * o) = int2{ri[25], ri[26]};
    d.contact_worldid[o] = w;
    d.contact_type[o] = CONTACT_TYPE_CONSTRAINT;
    d.contact_geomcollisionid[o] = ri[27] & 255;
    const int rbase = ri[28], ndim = ri[29];
    for (int k = 0; k < ndim; ++k)
      if (rbase >= 0 && rbase + k < njmax) d.efc_id[(size_t)w * njmax + rbase + k] = adr + c;
  }
  for (int idx = lig; idx < 3 * n; idx += G) d.contact_pos[3 * o0 + idx] = rec[(idx / 3) * CON_STRIDE + 1 + idx % 3];
  for (int idx = lig; idx < 9 * n; idx += G) d.contact_frame[9 * o0 + idx] = rec[(idx / 9) * CON_STRIDE + 4 + idx % 9];
  for (int idx = lig; idx < 5 * n; idx += G) {
    const int q = idx % 5;
    d.contact_friction[5 * o0 + idx] = rec[(idx / 5) * CON_STRIDE + CON_FRICTION_WORD(q)];
    d.contact_solimp[5 * o0 + idx] = rec[(idx / 5) * CON_STRIDE + 19 + q];
  }
  for (int idx = lig; idx < npyr * n; idx += G) {
    const int c = idx / npyr, k = idx % npyr;
    const int rbase = reci[c * CON_STRIDE + 28], ndim = reci[c * CON_STRIDE + 29];
    d.contact_efc_address[npyr * o0 + idx] = (rbase >= 0 && k < ndim && rbase + k < njmax) ? rbase + k : -1;
  }
}

