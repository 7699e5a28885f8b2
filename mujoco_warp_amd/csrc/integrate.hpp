// integrate.hpp -- integrators and small utility kernels.
//
// Reference: forward.py:53-131 (_next_position/_next_velocity), 134-219 (_next_activation), 221-273 (_next_time
// overflow flags), 276-349 (_advance), 353-417 (euler with implicit joint damping), 578-612 (implicitfast) with
// derivative.py:1117 (deriv_smooth_vel; joint damping + affine actuator velocity terms, joint transmissions);
// cli.py:103-145 (_ctrl_noise) and util_misc.py:61 (halton).
#pragma once
#include "dev_common.hpp"
#include "smooth.hpp"

struct IntLayout {
  int L, dinv, x, qvel, total;
};
__host__ __device__ inline IntLayout int_layout(int nv, int nC) {
  IntLayout p;
  int o = 0;
  p.L = o; o += nC;
  p.dinv = o; o += nv;
  p.x = o; o += nv;
  p.qvel = o; o += nv;
  p.total = ((o + 3) / 4) * 4 + 1;
  return p;
}

// mode: 0 = Euler (forward.py:387), 1 = implicitfast (forward.py:578), 2 = fully implicit (the solve is csrc/implicit.hpp; here: _advance)
template <int G>
DEV void integrate_body(const MjhModel& m, const MjhData& d, int mode, float* smem, const Blk& b) {
  if ((int)threadIdx.x >= b.nthreads) return;
  const int nq = m.nq, nv = m.nv, nC = m.nC, nu = m.nu, njnt = m.njnt;
  const IntLayout lay = int_layout(nv, nC);
  int* shi = reinterpret_cast<int*>(smem);
  const MStruct ms = load_mstruct<G>(m, shi, b.nthreads);
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  if (w >= d.nworld) return;
  float* S = smem + mstruct_ints(nv, nC) + (size_t)gib * lay.total;
  float *L = S + lay.L, *dinv = S + lay.dinv, *x = S + lay.x, *qvel = S + lay.qvel;
  const int dsbl = m.disableflags;
  const float h = bf(m.opt_timestep, m.opt_timestep_nb, w, 1)[0];
  const size_t vo = (size_t)w * nv;
  const float* damp = bf(m.dof_damping, m.dof_damping_nb, w, nv);
  PhaseClock pc(6, lig);

  // does the velocity update need an implicit solve?  (mode 2: the fully implicit integrator solved its system in k_implicit_solve)
  bool implicit = false;
  if (mode == 1) implicit = true;
  else if (mode == 2) implicit = false;
  else if (mode == 0 && !(dsbl & (DSBL_EULERDAMP | DSBL_DAMPER))) {
    float mx = 0.0f;
    for (int i = lig; i < nv; i += G) mx = fmaxf(mx, fabsf(damp[i]));
    implicit = gmax<G>(mx) > 0.0f;
  }
  pc.mark(0);
  if (implicit) {
    gcopy<G>(L, d.M + (size_t)w * nC, nC, lig);
    gcopy<G>(x, d.efc_Ma + vo, nv, lig);
    gsync();
    if (!(dsbl & DSBL_DAMPER))
      for (int i = lig; i < nv; i += G) L[ms.rowadr[i] + ms.rownnz[i] - 1] += h * damp[i];
    if (mode == 1 && !(dsbl & DSBL_ACTUATION)) {
      // d(qfrc_actuator)/d(qvel) for joint transmissions is diagonal (smooth.hpp actuator_vel_diag); dinv is free until factor_ld
      for (int i = lig; i < nv; i += G) dinv[i] = 0.0f;
      gsync();
      actuator_vel_diag<G>(m, d, w, lig, h, dinv);
      gsync();
      for (int i = lig; i < nv; i += G) L[ms.rowadr[i] + ms.rownnz[i] - 1] += dinv[i];
    }
    gsync();
    pc.mark(1);
    factor_ld<G>(ms, L, dinv, nv, lig, &m);
    pc.mark(2);
    solve_ld<G>(m, ms, L, dinv, x, nv, lig);
    pc.mark(3);
  } else {
    gcopy<G>(x, (mode == 2 ? d.ws_iacc : d.qacc) + vo, nv, lig);
    gsync();
  }

  // ---- _advance (forward.py:276-349) -------------------------------------------------------------------
  for (int u = lig; u < nu; u += G) {  // activations (support.py:38 next_act)
    const int dyn = m.actuator_dyntype[u];
    if (dyn == 0) continue;
    const size_t a = (size_t)w * m.na + m.actuator_actadr[u];
    float act = d.act[a];
    const float act_dot = d.act_dot[a];
    if (dyn == 3) {
      const float tau = fmaxf(MJ_MINVAL, bf(m.actuator_dynprm, m.actuator_dynprm_nb, w, 10 * nu)[10 * u]);
      act = act + act_dot * tau * (1.0f - expf(-h / tau));
    } else {
      act = act + act_dot * h;
    }
    if (m.actuator_actlimited[u]) {
      const float* ar = bf(m.actuator_actrange, m.actuator_actrange_nb, w, 2 * nu) + 2 * u;
      act = clampf(act, ar[0], ar[1]);
    }
    d.act[a] = act;
  }
  for (int i = lig; i < nv; i += G) {
    const float v = d.qvel[vo + i] + x[i] * h;
    qvel[i] = v;
    d.qvel[vo + i] = v;
    d.qacc_warmstart[vo + i] = d.qacc[vo + i];
  }
  gsync();
  float* qpos = d.qpos + (size_t)w * nq;
  for (int j = lig; j < njnt; j += G) {  // _next_position forward.py:53
    const int qa = m.jnt_qposadr[j], dof = m.jnt_dofadr[j], t = m.jnt_type[j];
    // joints of a sleeping tree are not integrated at all (the reference advances the awake dofs only, forward.py:276-349 over
    // dof_awake_ind): their velocity is zero, but re-normalising a quaternion would still move it by an ulp per step
    if (m.sleep_enabled && d.tree_awake && !d.tree_awake[(size_t)w * m.ntree + m.dof_treeid[dof]]) continue;
    if (t == JNT_FREE) {
      for (int k = 0; k < 3; ++k) qpos[qa + k] += h * qvel[dof + k];
      st4(qpos + qa + 3, quat_integrate(ld4(qpos + qa + 3), ld3(qvel + dof + 3), h));
    } else if (t == JNT_BALL) {
      st4(qpos + qa, quat_integrate(ld4(qpos + qa), ld3(qvel + dof), h));
    } else {
      qpos[qa] += h * qvel[dof];
    }
  }
  pc.mark(4);
  if (lig == 0) {
    d.time[w] += h;
    // end-of-step overflow flags that depend on counters (forward.py:221-273)
    int o = 0;
    if (d.nefc[w] > d.njmax) o |= OVF_NEFC;
    if (o) atomicOr(d.overflow + w, o);  // k_publish_contacts may be flagging the same world concurrently
  }
}

// Runge-Kutta 4 (forward.py:420-557 rungekutta4, _rk_perturb_state, _rk_accumulate): step = forward, then three more
// forwards at perturbed states.  One launch of this kernel sits after each forward:
//   stage 0    : save (qpos, qvel, act) at t0; sums = B0 * (qvel, qacc, act_dot); perturb the state with A0
//   stage 1, 2 : sums += B_s * (qvel, qacc, act_dot) of the stage just evaluated; perturb with A_s
//   stage 3    : sums += B3 * (...); restore t0 and advance with the sums (_advance with qacc_rk, qvel_rk)
// perturb(a): qpos = qpos_t0 (+) a h qvel_stage, qvel = qvel_t0 + a h qacc_stage, act = next_act(act_t0, act_dot_stage, a h).
// RK4 tableau: A = (1/2, 1/2, 1), B = (1/6, 1/3, 1/3, 1/6).  smem: nv floats per world.
template <int G>
__global__ void __launch_bounds__(256) k_rk4(MjhModel m, MjhData d, int stage, float a, float bw) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nq = m.nq, nv = m.nv, na = m.na, nu = m.nu, njnt = m.njnt;
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = blockIdx.x * (blockDim.x / G) + gib;
  if (w >= d.nworld) return;
  float* vel = smem + (size_t)gib * (nv + 1);  // the velocity the position update integrates
  float* rk = d.ws_rk + (size_t)w * (nq + 3 * nv + 2 * na);
  float *qpos0 = rk, *qvel0 = rk + nq, *qvel_rk = qvel0 + nv, *qacc_rk = qvel_rk + nv, *act0 = qacc_rk + nv, *actdot_rk = act0 + na;
  const float h = bf(m.opt_timestep, m.opt_timestep_nb, w, 1)[0];
  const bool last = stage == 3;
  const float dt = last ? h : a * h;
  const size_t vo = (size_t)w * nv;
  for (int i = lig; i < nv; i += G) {
    const float cv = d.qvel[vo + i], ca = d.qacc[vo + i];
    const float v0 = stage == 0 ? cv : qvel0[i];
    const float sv = (stage == 0 ? 0.0f : qvel_rk[i]) + bw * cv, sa = (stage == 0 ? 0.0f : qacc_rk[i]) + bw * ca;
    if (stage == 0) qvel0[i] = cv;
    qvel_rk[i] = sv;
    qacc_rk[i] = sa;
    vel[i] = last ? sv : cv;
    d.qvel[vo + i] = v0 + dt * (last ? sa : ca);
    if (last) d.qacc_warmstart[vo + i] = ca;  // qacc of the last evaluation (forward.py:343)
  }
  for (int u = lig; u < nu; u += G) {  // activations (support.py:38 next_act with the stage's act_dot)
    const int dyn = m.actuator_dyntype[u];
    if (dyn == 0) continue;
    const int ai = m.actuator_actadr[u];
    const size_t ga = (size_t)w * na + ai;
    const float cd = d.act_dot[ga];
    const float a0 = stage == 0 ? d.act[ga] : act0[ai];
    const float sd = (stage == 0 ? 0.0f : actdot_rk[ai]) + bw * cd;
    if (stage == 0) act0[ai] = a0;
    actdot_rk[ai] = sd;
    const float rate = last ? sd : cd;
    float act;
    if (dyn == 3) {
      const float tau = fmaxf(MJ_MINVAL, bf(m.actuator_dynprm, m.actuator_dynprm_nb, w, 10 * nu)[10 * u]);
      act = a0 + rate * tau * (1.0f - expf(-dt / tau));
    } else {
      act = a0 + rate * dt;
    }
    if (m.actuator_actlimited[u]) {
      const float* ar = bf(m.actuator_actrange, m.actuator_actrange_nb, w, 2 * nu) + 2 * u;
      act = clampf(act, ar[0], ar[1]);
    }
    d.act[ga] = act;
    if (last) d.act_dot[ga] = sd;
  }
  gsync();
  float* qpos = d.qpos + (size_t)w * nq;
  for (int j = lig; j < njnt; j += G) {  // _next_position forward.py:53 from the t0 position
    const int qa = m.jnt_qposadr[j], dof = m.jnt_dofadr[j], t = m.jnt_type[j];
    const int nqj = t == JNT_FREE ? 7 : (t == JNT_BALL ? 4 : 1);
    float q0[7];
    for (int k = 0; k < 7; ++k)
      if (k < nqj) {
        q0[k] = stage == 0 ? qpos[qa + k] : qpos0[qa + k];
        if (stage == 0) qpos0[qa + k] = q0[k];
      }
    if (t == JNT_FREE) {
      for (int k = 0; k < 3; ++k) qpos[qa + k] = q0[k] + dt * vel[dof + k];
      st4(qpos + qa + 3, quat_integrate(Q4{q0[3], q0[4], q0[5], q0[6]}, ld3(vel + dof + 3), dt));
    } else if (t == JNT_BALL) {
      st4(qpos + qa, quat_integrate(Q4{q0[0], q0[1], q0[2], q0[3]}, ld3(vel + dof), dt));
    } else {
      qpos[qa] = q0[0] + dt * vel[dof];
    }
  }
  if (last && lig == 0) {
    d.time[w] += h;
    if (d.nefc[w] > d.njmax) atomicOr(d.overflow + w, OVF_NEFC);
  }
}

// cli.py:103-145; halton in float32 exactly like util_misc.py:61
DEV float halton(int index, int base) {
  int n0 = index;
  const float b = (float)base;
  float f = 1.0f / b, hn = 0.0f;
  while (n0 > 0) {
    const int n1 = n0 / base;
    const int r = n0 - n1 * base;
    hn += f * (float)r;
    f /= b;
    n0 = n1;
  }
  return hn;
}
struct NoiseArgs {  // one application of the benchmark's control noise (cli.py:103-145); n = 0: none
  int n, step;
  float noise_std, noise_rate;
  const float* center;
  int sched;  // k_fwd_pos_plus only: its first workgroup sorts the solver schedule (0: the k_mid launch carries that workgroup)
};
DEV void ctrl_noise_elem(const MjhModel& m, const MjhData& d, const float* center, int step, float noise_std, float noise_rate, int idx) {
  const int nu = m.nu;
  if (idx >= d.nworld * nu) return;
  const int w = idx / nu, a = idx - w * nu;
  const float h = bf(m.opt_timestep, m.opt_timestep_nb, w, 1)[0];
  const float rate = expf(-h / noise_rate);
  const float scale = noise_std * sqrtf(1.0f - rate * rate);
  float midpoint = 0.0f, halfrange = 1.0f;
  const float* cr = bf(m.actuator_ctrlrange, m.actuator_ctrlrange_nb, w, 2 * nu) + 2 * a;
  const bool lim = m.actuator_ctrllimited[a];
  if (lim) {
    midpoint = 0.5f * (cr[1] + cr[0]);
    halfrange = 0.5f * (cr[1] - cr[0]);
  }
  if (center) midpoint = center[a];
  float ctrl = rate * d.ctrl[idx] + (1.0f - rate) * midpoint;
  const int gw = w + d.world_offset;  // global world id keeps trajectories shard-invariant across GPUs
  ctrl += scale * halfrange * (2.0f * halton((step + 1) * (gw + 1), a + 2) - 1.0f);
  if (lim) ctrl = clampf(ctrl, cr[0], cr[1]);
  d.ctrl[idx] = ctrl;
}
__global__ void k_ctrl_noise(MjhModel m, MjhData d, const float* center, int step, float noise_std, float noise_rate) {
  ctrl_noise_elem(m, d, center, step, noise_std, noise_rate, blockIdx.x * blockDim.x + threadIdx.x);
}

// x = M^-1 y (smooth.solve_m) and res = M vec (support.mul_m) on user arrays
template <int G>
__global__ void __launch_bounds__(256) k_solve_m(MjhModel m, MjhData d, float* xout, const float* yin, int mul) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nv = m.nv, nC = m.nC;
  const IntLayout lay = int_layout(nv, nC);
  int* shi = reinterpret_cast<int*>(smem);
  const MStruct ms = load_mstruct<G>(m, shi, blockDim.x);
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = blockIdx.x * (blockDim.x / G) + gib;
  if (w >= d.nworld) return;
  float* S = smem + mstruct_ints(nv, nC) + (size_t)gib * lay.total;
  float *L = S + lay.L, *dinv = S + lay.dinv, *x = S + lay.x, *y = S + lay.qvel;
  if (mul) {
    gcopy<G>(L, d.M + (size_t)w * nC, nC, lig);
    gcopy<G>(y, yin + (size_t)w * nv, nv, lig);
    gsync();
    mul_m_ld<G>(ms, L, y, x, nv, lig);
  } else {
    gcopy<G>(L, d.qLD + (size_t)w * nC, nC, lig);
    gcopy<G>(dinv, d.qLDiagInv + (size_t)w * nv, nv, lig);
    gcopy<G>(x, yin + (size_t)w * nv, nv, lig);
    gsync();
    solve_ld<G>(m, ms, L, dinv, x, nv, lig);
  }
  gsync();
  gcopy<G>(xout + (size_t)w * nv, x, nv, lig);
}

// Solver schedule for the NEXT step: counting sort of worlds by this step's solver_niter, longest first.
// Iteration counts are strongly correlated from step to step, so (i) the longest solves start first (LPT: no
// long straggler wave at the tail of k_solve) and (ii) the two worlds sharing a wavefront need a similar number
// of iterations (a wave runs for max(niter) of its two worlds).  Single workgroup; deterministic (stable sort).
// `sh` needs 512 ints of LDS; any workgroup size that is a multiple of 64.  Global loads are batched eight deep ahead
// of the LDS atomics (a load -> atomic -> store chain per world costs a full memory latency per iteration: 50-70 us
// for 8192 worlds on one small workgroup), the 128-bin prefix is one wavefront scan.
// cls: the row count that separates the two classes, 0: no classes (sched_cls below)
DEV void schedule_body(const MjhData& d, int* sh, int nthreads, int cls = 0) {
  // key: worlds that had more than `cls` constraint rows first, then the others (the solver's one-row-per-lane instantiation takes the worlds
  // of at most 32 rows in a launch of its own: with the two classes apart both launches run dense workgroups); inside a class by iteration
  // count, longest first.  256 bins: `sh` needs 512 ints.  Without classes (cls = 0: every solver but Newton with elliptic cones) the row
  // counts are not even loaded.
  // This single workgroup rides in the k_fwd_pos launch and was its tail (fused launch 59 us against 41 us for the plain kernel, round 4):
  // round 5 keeps a thread's keys in registers -- ONE batch of up to 32 loads per thread, all in flight together, serves both the histogram
  // and the scatter (it was two passes of 8-deep batches: eight dependent memory round trips for 8192 worlds on 256 threads).
  int* hist = sh;
  int* base = sh + 256;
  const int t = threadIdx.x, n = d.nworld;
  auto bin = [cls](int niter, int nefc) { return (cls == 0 || nefc > cls ? 0 : 128) + 127 - min(max(niter, 0), 127); };
  for (int i = t; i < 256; i += nthreads) hist[i] = 0;
  __syncthreads();
  constexpr int KD = 32;
  const bool one_batch = n <= KD * nthreads && (n & 3) == 0;
  int key[KD];
  if (one_batch) {
    // KD / 4 = eight 16-byte loads per thread and array (solver_niter; with row classes also nefc: sixteen in all) -- thread t: worlds
    // 4 (t + k nthreads) .. + 3 --, from clamped addresses and tied together below: a load
    // behind a per-key condition is a branch, and the compiler then waited for every load before it issued the next (32 dependent round
    // trips: the 13 us this workgroup took, measured as its launch's tail).
    int4 v4[KD / 4], e4[KD / 4];
#pragma unroll
    for (int k = 0; k < KD / 4; ++k) v4[k] = reinterpret_cast<const int4*>(d.solver_niter)[min(t + k * nthreads, n / 4 - 1)];
    if (cls) {
#pragma unroll
      for (int k = 0; k < KD / 4; ++k) e4[k] = reinterpret_cast<const int4*>(d.nefc)[min(t + k * nthreads, n / 4 - 1)];
    } else {
#pragma unroll
      for (int k = 0; k < KD / 4; ++k) e4[k] = make_int4(0, 0, 0, 0);
    }
    // (all in flight before the first use)
    asm volatile("" ::"v"(v4[0].x), "v"(v4[0].y), "v"(v4[0].z), "v"(v4[0].w), "v"(v4[1].x), "v"(v4[1].y), "v"(v4[1].z), "v"(v4[1].w), "v"(v4[2].x), "v"(v4[2].y),
                 "v"(v4[2].z), "v"(v4[2].w), "v"(v4[3].x), "v"(v4[3].y), "v"(v4[3].z), "v"(v4[3].w), "v"(v4[4].x), "v"(v4[5].x), "v"(v4[6].x), "v"(v4[7].x));
#pragma unroll
    for (int k = 0; k < KD / 4; ++k) {
      const bool ok = t + k * nthreads < n / 4;
      key[4 * k] = ok ? bin(v4[k].x, e4[k].x) : -1;
      key[4 * k + 1] = ok ? bin(v4[k].y, e4[k].y) : -1;
      key[4 * k + 2] = ok ? bin(v4[k].z, e4[k].z) : -1;
      key[4 * k + 3] = ok ? bin(v4[k].w, e4[k].w) : -1;
    }
#pragma unroll
    for (int k = 0; k < KD; ++k)
      if (key[k] >= 0) atomicAdd(&hist[key[k]], 1);
  } else {
    for (int w0 = t; w0 < n; w0 += 8 * nthreads) {
      int v[8], e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = w0 + k * nthreads < n ? d.solver_niter[w0 + k * nthreads] : -1;
        e[k] = cls && w0 + k * nthreads < n ? d.nefc[w0 + k * nthreads] : 0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (w0 + k * nthreads < n) atomicAdd(&hist[bin(v[k], e[k])], 1);
    }
  }
  __syncthreads();
  if (t < 64) {  // exclusive prefix over the 256 bins: four bins per lane of the first wavefront
    const int a0 = hist[4 * t], a1 = hist[4 * t + 1], a2 = hist[4 * t + 2], a3 = hist[4 * t + 3];
    const int tot = a0 + a1 + a2 + a3;
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (t >= off) incl += u;
    }
    const int ex = incl - tot;
    base[4 * t] = ex;
    base[4 * t + 1] = ex + a0;
    base[4 * t + 2] = ex + a0 + a1;
    base[4 * t + 3] = ex + a0 + a1 + a2;
  }
  __syncthreads();
  // scatter; the order inside a bucket is arbitrary (it only decides which wavefront hosts a world, never a result)
  if (one_batch) {
#pragma unroll
    for (int k = 0; k < KD; ++k)
      if (key[k] >= 0) d.ws_order[atomicAdd(&base[key[k]], 1)] = 4 * (t + (k >> 2) * nthreads) + (k & 3);
    return;
  }
  for (int w0 = t; w0 < n; w0 += 8 * nthreads) {
    int v[8], e[8], pos[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = w0 + k * nthreads < n ? d.solver_niter[w0 + k * nthreads] : -1;
      e[k] = cls && w0 + k * nthreads < n ? d.nefc[w0 + k * nthreads] : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) pos[k] = w0 + k * nthreads < n ? atomicAdd(&base[bin(v[k], e[k])], 1) : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (w0 + k * nthreads < n) d.ws_order[pos[k]] = w0 + k * nthreads;
  }
}
// The row-count classes matter (i) where the solver launches a one-row-per-lane instantiation for the worlds of at most 32 rows (mjhip.hip
// launch_solve_any: Newton with elliptic cones at nv <= 32 and njmax > 32) and (ii) for the 64-lane kernels of models with more than 32 dofs,
// one world per wavefront: there a solve's length follows the row count (the Hessian build) more than the iteration count, and "worlds of more
// than 64 rows first" is the longest-first order (G1, 4096 worlds: 8.2 M env-steps/s with the classes, 6.9 M without -- measured in round 5
// when the classes were dropped for every model).
DEV int sched_cls(const MjhModel& m, const MjhData& d) {
  if (m.nv > 32) return 64;
  return (m.solver == SOL_NEWTON && m.cone == CONE_ELLIPTIC && d.njmax > 32) ? 32 : 0;
}
__global__ void __launch_bounds__(1024) k_schedule_worlds(MjhData d, int cls) {
  __shared__ int sh[512];
  schedule_body(d, sh, blockDim.x, cls);
}

template <int G>
__global__ void __launch_bounds__(256) k_integrate(MjhModel m, MjhData d, int mode) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  integrate_body<G>(m, d, mode, smem, blk_of_launch<G>());
}
