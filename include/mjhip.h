/* mjhip.h -- C ABI of libmjhip.so: the MI355X-native batched mj_step engine.
 *
 * This is the drop-in boundary for the reference's L3/L1 seam: the reference's Python
 * functions `step/forward/kinematics/com_pos/crb/factor_m/collision/make_constraint/fwd_velocity/
 * fwd_actuation/fwd_acceleration/solve/euler/implicit` (/root/reference/mujoco_warp/__init__.py:26-101,
 * _src/forward.py:1341-1380) each become one extern "C" entry point taking flat pointer tables.
 *
 *   MjhModel  <->  reference `Model` dataclass  (_src/types.py:982-1960; only hot-path fields)
 *   MjhData   <->  reference `Data` / `Contact` / `Constraint` (_src/types.py:1975-2374)
 *
 * Conventions: every array is float32/int32, row-major, world-major ([nworld, n, ...]); a Model
 * array with a companion `<name>_nb` field may carry a leading batch dimension of size nb and is
 * indexed `worldid % nb` (reference "*" fields, types.py:1535-1808).  All pointers are DEVICE
 * pointers owned by the caller; entry points never allocate and never synchronise.  Return value:
 * 0 on success, negative MJH_E_* otherwise; mjh_last_error() returns a static message.
 * Entry points are re-entrant per (MjhModel, MjhData, stream).
 *
 * One declaration per statement, no macros inside the structs: mujoco_warp_amd/_abi.py parses this
 * file to build the ctypes mirror, so the header is the single source of truth.
 */
#ifndef MJHIP_H
#define MJHIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MJH_OK 0
#define MJH_E_ARG -1
#define MJH_E_UNSUPPORTED -2
#define MJH_E_LAUNCH -3

/* stage ids for mjh_stage() -- names follow the reference's public stage functions */
#define MJH_STAGE_KINEMATICS 0       /* smooth.kinematics            smooth.py:447  */
#define MJH_STAGE_COM_POS 1          /* smooth.com_pos               smooth.py:824  */
#define MJH_STAGE_CRB 2              /* smooth.crb                   smooth.py:1079 */
#define MJH_STAGE_FACTOR_M 3         /* smooth.factor_m              smooth.py:1340 */
#define MJH_STAGE_COLLISION 4        /* collision_driver.collision   collision_driver.py:885 */
#define MJH_STAGE_MAKE_CONSTRAINT 5  /* constraint.make_constraint   constraint.py:4898 */
#define MJH_STAGE_TRANSMISSION 6     /* smooth.transmission          smooth.py:2890 */
#define MJH_STAGE_COM_VEL 7          /* smooth.com_vel               smooth.py:2261 */
#define MJH_STAGE_PASSIVE 8          /* passive.passive              passive.py:1257 */
#define MJH_STAGE_RNE 9              /* smooth.rne                   smooth.py:1499 */
#define MJH_STAGE_FWD_VELOCITY 10    /* forward.fwd_velocity         forward.py:732 */
#define MJH_STAGE_FWD_ACTUATION 11   /* forward.fwd_actuation        forward.py:1152 */
#define MJH_STAGE_FWD_ACCELERATION 12 /* forward.fwd_acceleration    forward.py:1290 */
#define MJH_STAGE_SOLVE 13           /* solver.solve                 solver.py:3671 */
#define MJH_STAGE_EULER 14           /* forward.euler                forward.py:387 */
#define MJH_STAGE_IMPLICIT 15        /* forward.implicit (implicitfast) forward.py:578 */
#define MJH_STAGE_FWD_POSITION 16    /* forward.fwd_position         forward.py:635 */
#define MJH_STAGE_FORWARD 17         /* forward.forward              forward.py:1341 */
#define MJH_STAGE_STEP 18            /* forward.step                 forward.py:1368 */
#define MJH_STAGE_UPDATE_SLEEP 20    /* sleep.update_sleep           sleep.py:171 */
#define MJH_STAGE_WAKE 21            /* sleep.wake (+ update_sleep)  sleep.py:721 */
#define MJH_STAGE_WAKE_COLLISION 22  /* sleep.wake_collision (+ update_sleep) sleep.py:744 */
#define MJH_STAGE_WAKE_EQUALITY 23   /* sleep.wake_equality (+ update_sleep)  sleep.py:793 */
#define MJH_STAGE_ISLAND 24          /* island.island                island.py:294 */
#define MJH_STAGE_SLEEP 25           /* sleep.sleep (+ update_sleep) sleep.py:947 */
#define MJH_STAGE_SENSOR 26          /* sensor.sensor_pos + sensor_vel, then sensor_acc  sensor.py:810, 1432, 2512 */
#define MJH_STAGE_SENSOR_POSVEL 30   /* sensor.sensor_pos + sensor_vel only (before the solver: nothing reads qacc / efc_force)  sensor.py:810, 1432 */
#define MJH_STAGE_SENSOR_ACC 31      /* sensor.sensor_acc only (after the solver; runs rne_postconstraint when a sensor needs it)  sensor.py:2512 */
#define MJH_STAGE_ENERGY 27          /* sensor.energy_pos + energy_vel  sensor.py:2934, 3003 */
#define MJH_STAGE_SUBTREE_VEL 28     /* smooth.subtree_vel  smooth.py:3614 */
#define MJH_STAGE_RNE_POSTCONSTRAINT 29 /* smooth.rne_postconstraint  smooth.py:1744 */
#define MJH_STAGE_RUNGEKUTTA4 19     /* forward.rungekutta4 (after a forward)  forward.py:524 */

typedef struct MjhModel {
  /* sizes */
  int nq; int nv; int nu; int na; int nbody; int njnt; int ngeom; int nsite; int nC; int npair;
  int nbodylevel; int ndoflevel; int nv_pad; int neq;
  int nexplicit;       /* explicit <contact><pair> entries (pair_* tables below); 0: every pair mixes its geoms' parameters */
  int nmocap;          /* mocap bodies (static bodies posed by Data.mocap_pos / mocap_quat, smooth.py:104-108) */
  int heavy_colliders; /* 1: capsule-box / box-box pairs, explicit contact pairs, AABB / OBB broadphase filters or the SAP broadphase
                          (selects the kernel instantiation that carries them) */
  int broadphase;        /* BroadphaseType (types.py:60-71): 0 NXN, 1 SAP_TILE, 2 SAP_SEGMENTED (both: in-LDS bitonic sort per world) */
  int broadphase_filter; /* BroadphaseFilter bits (types.py:73-87): 1 plane, 2 sphere, 4 AABB, 8 OBB */
  /* options (types.py:836-905); solver: 0 = PGS (extension, the reference has none: types.py:502), 1 = CG, 2 = Newton */
  int integrator; int cone; int solver; int iterations; int ls_iterations; int disableflags; int enableflags;
  int sleep_enabled;   /* EnableBit.SLEEP set and DisableBit.ISLAND clear (forward.py:345): sleeping / waking of kinematic trees
                          (csrc/sleep.hpp; staged launch sequence, Newton only like the reference io.py:359) */
  float opt_sleep_tolerance; /* Option.sleep_tolerance (types.py:845) */
  int ccd_iterations;  /* GJK iteration cap of the convex narrowphase (capped at 64 by this engine) */
  int epa_iterations;  /* EPA iteration cap: 16 when every convex pair of the model is box-box, else ccd_iterations (collision_convex.py:1223) */
  const float* opt_timestep; int opt_timestep_nb;
  const float* opt_tolerance; int opt_tolerance_nb;
  const float* opt_ls_tolerance; int opt_ls_tolerance_nb;
  const float* opt_gravity; int opt_gravity_nb;
  const float* opt_impratio_invsqrt; int opt_impratio_invsqrt_nb;
  const float* opt_ccd_tolerance; int opt_ccd_tolerance_nb;
  const float* opt_magnetic; int opt_magnetic_nb;  /* Option.magnetic (types.py: magnetometer sensor) */
  const float* stat_meaninertia; int stat_meaninertia_nb;
  const float* qpos0; int qpos0_nb;
  const float* qpos_spring; int qpos_spring_nb;
  /* bodies */
  const int* body_parentid; const int* body_rootid; const int* body_weldid;
  const int* body_jntnum; const int* body_jntadr; const int* body_dofnum; const int* body_dofadr;
  const int* body_lastdof;      /* last dof affecting the body, -1 if static                    */
  const int* body_mocapid;      /* [nbody] index into Data.mocap_*, -1 for ordinary bodies       */
  const int* body_subtreenum;   /* bodies in the (contiguous, depth-first) subtree, incl. self  */
  const int* body_tree;         /* body ids sorted by tree depth (io.py:495-500)                */
  const int* body_leveladr;     /* [nbodylevel+1] offsets into body_tree                        */
  const unsigned int* body_dofmask; /* [nbody, ceil(nv/32)] bit i set <=> dof i moves the body (io.py:536-549) */
  const float* body_pos; int body_pos_nb;
  const float* body_quat; int body_quat_nb;
  const float* body_ipos; int body_ipos_nb;
  const float* body_iquat; int body_iquat_nb;
  const float* body_mass; int body_mass_nb;
  const float* body_subtreemass; int body_subtreemass_nb;
  const float* body_inertia; int body_inertia_nb;
  const float* body_invweight0; int body_invweight0_nb;
  const float* body_gravcomp; int body_gravcomp_nb;
  /* joints */
  const int* jnt_type; const int* jnt_qposadr; const int* jnt_dofadr; const int* jnt_bodyid; const int* jnt_limited;
  const float* jnt_solref; int jnt_solref_nb;
  const float* jnt_solimp; int jnt_solimp_nb;
  const float* jnt_pos; int jnt_pos_nb;
  const float* jnt_axis; int jnt_axis_nb;
  const float* jnt_stiffness; int jnt_stiffness_nb;
  const float* jnt_range; int jnt_range_nb;
  const float* jnt_margin; int jnt_margin_nb;
  const int* jnt_actfrclimited; /* [njnt] clamp the summed actuator force on the joint's dof (forward.py:1121-1150)                */
  const float* jnt_actfrcrange; int jnt_actfrcrange_nb; /* [*, njnt, 2]                                                        */
  const int* jnt_actgravcomp;   /* [njnt] gravity compensation is applied through the actuators (passive.py:652, forward.py:1141)  */
  /* dofs */
  const int* dof_bodyid; const int* dof_jntid; const int* dof_parentid;
  const int* dof_grpadr;        /* first dof of the ball/free rotational triple containing the dof, else the dof itself */
  const int* dof_tree;          /* dof ids sorted by depth in the dof tree   */
  const int* dof_leveladr;      /* [ndoflevel+1] offsets into dof_tree       */
  /* kinematic trees (contiguous dof ranges; M is block diagonal over them): the per-tree solver dispatch for nv > 64 */
  const int* tree_sleep_policy; /* [ntree] SleepPolicy (types.py:296): 1 AUTO_NEVER, 2 AUTO_ALLOWED */
  const float* dof_length;      /* [nv] velocity weights of the sleep test (types.py:1098) */
  int act_dof_max;              /* largest number of actuators acting on one dof (implicit integrators: see csrc/integrate.hpp) */
  int act_velfeedback;          /* 1: some actuator feeds velocity back with a positive sign (affine gain on velocity, or bias velocity coefficient > 0):
                                   M + h D - h dA/dv may then be indefinite, and implicitfast keeps the integrator launch's L'DL solve instead of the
                                   solver epilogue's Cholesky (csrc/solver.hpp impfast_acc) */
  int cg_basis;                 /* 1: every contact the model can make has condim 1 or 3 (pyramidal condim 3: four rows spanned by the three basis rows
                                   J_n, mu J_t1, mu J_t2): CG at nv <= 32, njmax <= 64 runs the pooled contact-basis kernel (csrc/solver_cgp.hpp) */
  int ntree;                    /* trees with at least one dof */
  int tree_nvmax;               /* dofs of the largest tree */
  int isl_nv4;                  /* ceil(dofs / 4) of the widest island of at most 32 dofs the model can form (kernel size class) */
  int isl_wide;                 /* 1: islands of 33..64 dofs can form */
  int tree_solve;               /* 1: nv > 64, several trees of <= 64 dofs, CG / Newton with pyramidal cones: worlds whose constraint islands
                                   (trees joined by coupling rows) have <= 64 dofs are solved per island by the register-resident kernels */
  const int* tree_dofadr;       /* [ntree] first dof */
  const int* tree_dofnum;       /* [ntree] dofs */
  const int* dof_treeid;        /* [nv] */
  const int* body_treeid;       /* [nbody] tree of the body's dofs (its own or its nearest ancestor's), -1: static */
  const float* dof_solref; int dof_solref_nb;
  const float* dof_solimp; int dof_solimp_nb;
  const float* dof_frictionloss; int dof_frictionloss_nb;
  const float* dof_armature; int dof_armature_nb;
  const float* dof_damping; int dof_damping_nb;
  const float* dof_invweight0; int dof_invweight0_nb;
  const int* M_rownnz; const int* M_rowadr; const int* M_colind;
  const int* M_dense;           /* [nv, 4 ceil(nv / 4)] index into Data.M of the dense entry (i, c), -1 where M is structurally zero: the solvers gather
                                   their dense row of M with independent loads (engine-private; io.py put_model)                */
  /* geoms */
  const int* geom_type; const int* geom_condim; const int* geom_bodyid; const int* geom_priority;
  /* ray casting (ray.py:52 _ray_eliminate): group and visibility of the geoms; nmat materials */
  int nmat; const int* geom_group; const int* geom_matid; const float* geom_rgba; /* [ngeom, 4] */ const float* mat_rgba; /* [nmat, 4] */
  const int* geom_dataid;       /* [ngeom] mesh id of mesh geoms, -1 otherwise (types.py:1266)                  */
  const int* mesh_vertadr; const int* mesh_vertnum; /* [nmesh] first vertex / number of vertices (types.py:1707-1709) */
  const float* mesh_vert;       /* [nmeshvert, 3] vertices in the mesh (= geom) frame; searched exhaustively by the convex narrowphase */
  /* sensors (types.py: sensor_*; csrc/sensor.hpp computes joint / actuator / ball / frame / velocimeter / gyro / subtreecom / clock) */
  int nsensor; int nsensordata;
  int nsensor_subtree; /* subtreelinvel / subtreeangmom sensors: smooth.subtree_vel runs before the sensor launch */
  int nsensor_energy; /* e_potential / e_kinetic sensors: the energy kernel runs even without EnableBit.ENERGY */
  int nsensor_frc;  /* force / torque sensors: smooth.rne_postconstraint runs before the acceleration-stage sensor launch */
  int nsensor_acc;  /* sensors of the acceleration stage (accelerometer, framelinacc, frameangacc): one more launch between solver and integrator */
  const int* sensor_type; const int* sensor_datatype; const int* sensor_objtype; const int* sensor_objid; const int* sensor_reftype; const int* sensor_refid;
  const int* sensor_dim; const int* sensor_adr;
  const float* sensor_cutoff;
  /* height fields (types.py: hfield_*; geom_dataid of an hfield geom is its height field) */
  int nhfield;
  const float* hfield_size;     /* [nhfield, 4] x, y half sizes, top scale of the elevation data, base thickness */
  const int* hfield_nrow; const int* hfield_ncol; const int* hfield_adr;
  const float* hfield_data;     /* elevations normalised to [0, 1], row 0 at -y */
  const int* mesh_graphadr;     /* [nmesh] first word of the mesh's hill-climbing graph in mesh_graph, -1: none (types.py: mesh_graphadr) */
  const int* mesh_graph;        /* MuJoCo's layout: numvert, numface, vert_edgeadr[numvert], vert_globalid[numvert], edge_localid[...], face_globalid[...] */
  /* mesh polygon tables for the multi-contact recovery on mesh faces (types.py:1710-1733; csrc/convex.hpp ccd_multicontact_mesh) */
  int nmeshpoly;                /* polygons of all meshes; 0: no tables (mesh pairs then keep EPA's single contact)           */
  int npolygonmax;              /* the clip buffers hold 2 * npolygonmax points (collision_convex.py:1226-1234)               */
  int nmeshdegmax;              /* polygons around one mesh vertex, at most (collision_convex.py:1233); 0: no multi-contact recovery on mesh faces */
  const int* mesh_polyadr;      /* [nmesh] first polygon                                                                     */
  const float* mesh_polynormal; /* [nmeshpoly, 3] outward normals, mesh frame                                                */
  const int* mesh_polyvertadr; const int* mesh_polyvertnum; /* [nmeshpoly] into mesh_polyvert                                */
  const int* mesh_polyvert;     /* mesh-local vertex ids, counter-clockwise seen from outside                                */
  const int* mesh_polymapadr; const int* mesh_polymapnum;   /* [nmeshvert] into mesh_polymap                                 */
  const int* mesh_polymap;      /* mesh-local ids of the polygons around each vertex                                         */
  const float* geom_solmix; int geom_solmix_nb;
  const float* geom_solref; int geom_solref_nb;
  const float* geom_solimp; int geom_solimp_nb;
  const float* geom_size; int geom_size_nb;
  const float* geom_rbound; int geom_rbound_nb;
  const float* geom_aabb; int geom_aabb_nb;   /* [ngeom, 6]: centre, half sizes in the geom frame (types.py Model.geom_aabb) */
  const float* geom_pos; int geom_pos_nb;
  const float* geom_quat; int geom_quat_nb;
  const float* geom_friction; int geom_friction_nb;
  const float* geom_margin; int geom_margin_nb;
  const float* geom_gap; int geom_gap_nb;
  const int* nxn_geom_pair;     /* [npair, 2] pre-filtered geom pairs, upper-triangular order (io.py:551-640) */
  const int* nxn_pairid;        /* [npair] explicit pair index or -1 (io.py:575-590)             */
  const int* nxn_pairindex;     /* [ngeom (ngeom - 1) / 2] index into nxn_geom_pair of the unordered geom pair (math.upper_tri_index
                                   order) or -1 when the pair is filtered out: the SAP sweep's lookup (collision_driver.py:485) */
  /* k_broad_mask's group pre-test (host: io.py cull_tables): the pair list regrouped by pairs of geom groups (the colliding geoms among
     the consecutive geoms of one moving body; every static geom its own group); a group's sphere is centred on its centre geom */
  const int* cull_geom;         /* [ncullgeom, 2] geom + (group << 16), the group's centre geom: the colliding geoms, group by group   */
  const int* cull_group;        /* [ncullgroup, 2] centre geom, number of colliding geoms (description; not read by the kernels)       */
  const int* cull_pair;         /* [ncullpair, 4] group, group (-1: explicit pairs, always tested), centre geom 1 + (centre geom 2 << 16),
                                   first entry of cull_list + (count << 24), count <= 16                                              */
  const int* cull_list;         /* [npair, 2] index into nxn_geom_pair, g1 + (g2 << 16)                                               */
  int ncullgeom; int ncullgroup; int ncullpair;
  /* explicit contact pairs (types.py Model.pair_*): parameters that replace the geom mixing */
  const int* pair_dim; const float* pair_friction; const float* pair_solref; const float* pair_solreffriction;
  const float* pair_solimp; const float* pair_margin; const float* pair_gap;
  /* sites */
  const int* site_bodyid;
  const float* site_pos; int site_pos_nb;
  const float* site_quat; int site_quat_nb;
  const int* site_type;         /* [nsite] GeomType of the site's shape (touch sensor zones) */
  const float* site_size;       /* [nsite, 3] */
  /* actuators (joint transmission) */
  const int* actuator_dyntype; const int* actuator_gaintype; const int* actuator_biastype; const int* actuator_trnid;
  const int* actuator_actadr; const int* actuator_ctrllimited; const int* actuator_forcelimited; const int* actuator_actlimited;
  const float* actuator_dynprm; int actuator_dynprm_nb;
  const float* actuator_gainprm; int actuator_gainprm_nb;
  const float* actuator_biasprm; int actuator_biasprm_nb;
  const float* actuator_ctrlrange; int actuator_ctrlrange_nb;
  const float* actuator_forcerange; int actuator_forcerange_nb;
  const float* actuator_actrange; int actuator_actrange_nb;
  const float* actuator_gear; int actuator_gear_nb;
  /* equality constraints: joint couplings only (constraint.py:500-640); eq_data = polycoef[0..4] */
  const int* eq_obj1id; const int* eq_obj2id;
  const float* eq_solref; int eq_solref_nb;
  const float* eq_solimp; int eq_solimp_nb;
  const float* eq_data; int eq_data_nb;
} MjhModel;

typedef struct MjhData {
  int nworld; int nconmax; int naconmax; int njmax; int njmax_pad; int nv_pad; int nmaxpyramid; int world_offset;
  int concap;          /* per-world contact capacity of ws_contact: clamp(2 nconmax, 16, 256)          */
  int nccdhand;        /* EPA entries the convex narrowphase can hand from k_ccd_gjk to k_ccd_epa per step (csrc/collide.hpp ccd_handcap) */
  /* state (types.py:2240-2262) */
  float* time; float* qpos; float* qvel; float* act; float* ctrl; float* qacc_warmstart;
  float* qfrc_applied; float* xfrc_applied;
  float* mocap_pos; float* mocap_quat;  /* [nworld, nmocap, 3 / 4] (types.py Data.mocap_pos / mocap_quat) */
  /* position-dependent */
  float* xpos; float* xquat; float* xmat; float* xipos; float* ximat; float* xanchor; float* xaxis;
  float* geom_xpos; float* geom_xmat; float* site_xpos; float* site_xmat;
  float* subtree_com; float* cinert; float* cdof; float* crb; float* M; float* qLD; float* qLDiagInv;
  float* actuator_length; float* actuator_moment;
  /* velocity-dependent */
  float* actuator_velocity; float* cvel; float* cdof_dot;
  float* qfrc_spring; float* qfrc_damper; float* qfrc_gravcomp; float* qfrc_passive; float* qfrc_bias;
  float* cacc; float* cfrc_int; float* cfrc_ext; /* cfrc_ext [nworld, nbody, 6]: smooth.rne_postconstraint (which also rewrites cacc / cfrc_int) */
  /* actuation / acceleration */
  float* act_dot; float* actuator_force; float* qfrc_actuator; float* qfrc_smooth; float* qacc_smooth;
  /* constraint solver outputs */
  float* qacc; float* qfrc_constraint; float* efc_Ma;
  int* solver_niter; int* ne; int* nf; int* nl; int* nefc; int* overflow;
  /* contacts: public flat arrays [naconmax] (types.py:1975-2018) + counters */
  int* nacon; int* ncollision;
  float* contact_dist; float* contact_pos; float* contact_frame; float* contact_includemargin;
  float* contact_friction; float* contact_solref; float* contact_solreffriction; float* contact_solimp;
  int* contact_dim; int* contact_geom; int* contact_efc_address; int* contact_worldid; int* contact_type;
  int* contact_geomcollisionid;
  /* constraints (types.py:2021-2072): J is [nworld, njmax_pad, nv_pad] */
  int* efc_type; int* efc_id; int* efc_state;
  float* efc_J; float* efc_pos; float* efc_margin; float* efc_D; float* efc_vel; float* efc_aref;
  float* efc_frictionloss; float* efc_force;
  /* engine workspace (pre-allocated by make_data; replaces the reference's per-step temporaries) */
  int* ws_ncon;        /* [nworld]   contacts found per world                         */
  int* ws_conadr;      /* [nworld]   exclusive scan of ws_ncon = first public slot (k_contact_scan) */
  int* ws_ncollision;  /* [nworld]   broadphase candidates per world                  */
  float* ws_ccd;       /* workspace of the convex narrowphase (layout: csrc/convex.hpp ccd_layout -- per world the height-field prisms' polytopes, the
                          per-candidate result cache, the candidate list and the broadphase mask; then the counters, the convex-pair mask and
                          nccdhand EPA hand-over records of 64 floats); empty unless the model has convex (GJK) pairs */
  /* constraint islands at tree granularity (MjhModel.tree_solve, csrc/constraint.hpp k_tree_rows); island k of a world: */
  int* ws_tree_rowadr; /* [nworld, ntree + 1] its rows are ws_tree_rowmap[rowadr[k] .. rowadr[k + 1])                       */
  int* ws_tree_rowmap; /* [nworld, njmax] constraint rows grouped by island                                                */
  int* ws_isl_dofadr;  /* [nworld, ntree + 1] its dofs are ws_isl_dofmap[dofadr[k] .. dofadr[k + 1])                       */
  int* ws_isl_dofmap;  /* [nworld, nv] island-local dof -> dof, grouped by island                                         */
  int* ws_isl_dofinv;  /* [nworld, nv] dof -> index within its island                                                     */
  int* ws_nisland;     /* [nworld] islands                                                                                */
  int* ws_isl_flags;   /* [nworld] bit 0: an island of 33..64 dofs, bit 1: an island of <= 32 dofs with more than 64 rows          */
  int* ws_isl_list;    /* [3, nworld] worlds holding an island with > 64 rows | of 33..64 dofs | of > 64 dofs (generic solver)      */
  int* ws_isl_count;   /* [4] entries of the three lists                                                                        */
  int* ws_separable;   /* [nworld] 1: every island has at most 64 dofs (solved per island), 0: generic solver             */
  /* sleeping (types.py:2330-2345; all empty unless MjhModel.sleep_enabled) */
  float* sensordata;   /* [nworld, nsensordata] Data.sensordata (types.py) */
  float* subtree_linvel; float* subtree_angmom; /* [nworld, nbody, 3] smooth.subtree_vel (computed on request or for sensors that read them) */
  float* energy;       /* [nworld, 2] potential, kinetic energy (EnableBit.ENERGY; zero otherwise) */
  int* tree_asleep;    /* [nworld, ntree] < 0: awake (counts up to -1 while the tree could sleep), >= 0: next tree of its sleep cycle */
  int* tree_awake;     /* [nworld, ntree] */
  int* body_awake;     /* [nworld, nbody] SleepState: -1 static, 0 asleep, 1 awake */
  int* body_awake_ind; /* [nworld, nbody] bodies that are not asleep (ascending; the reference's order depends on atomics) */
  int* dof_awake_ind;  /* [nworld, nv] dofs of awake trees (ascending) */
  int* ntree_awake; int* nbody_awake; int* nv_awake; /* [nworld] */
  int* tree_island;    /* [nworld, ntree] constraint island of each tree (numbered by smallest tree), -1: no constraint row (island.py:206) */
  int* nisland;        /* [nworld] */
  float* ws_iacc;      /* [nworld, nv] qacc' of the fully implicit integrator (csrc/implicit.hpp), consumed by the integrator launch; empty unless opt.integrator == IMPLICIT */
  float* ws_pgsB;      /* [nworld, njmax_pad, nv_pad] rows of J M^-1 for the generic PGS kernel (csrc/pgs_big.hpp); empty unless the model is solved by it
                          (solver PGS with more than 64 dofs or elliptic cones) */
  float* ws_sleep_J;   /* [nworld, njmax_pad, nv_pad] efc.J with the columns of sleeping dofs zeroed: what the solver reads (see csrc/sleep.hpp) */
  float* ws_sleep_warm; /* [nworld, nv] qacc_warmstart with sleeping dofs zeroed */
  int* ws_sleep_flag;  /* [nworld] a tree of the world was woken by a contact of collision pass 1 (forward.py:652-666) */
  int sleep_pass;      /* launch-local: 0 plain collision, 2 second pass (only worlds with ws_sleep_flag set recompute) */
  int nvmax;           /* capacity for awake dofs per world (make_data / put_data nvmax, reference io.py:1704; default nv): a world with
                          more awake dofs gets OverflowType.NVMAX (island.py:1010-1019) -- this engine still solves it in full */
  int* ws_efc_con;     /* [nworld, njmax] contact rows: 16 * (world-local contact) + row within the contact (make_constraint -> solver,
                          elliptic cones only) */
  int* ws_order;       /* [nworld]   solver schedule: worlds sorted by last step's solver_niter (longest first) */
  int* eq_active;      /* [nworld, neq] Data.eq_active (types.py:2262), initialised from eq_active0 */
  float* ws_rk;        /* [nworld, nq + 3 nv + 2 na] RK4 scratch: qpos, qvel, act at t0 and the weighted sums of qvel, qacc,
                          act_dot (the temporaries forward.rungekutta4 allocates per step, forward.py:530-540) */
  float* ws_contact;   /* [nworld, concap, 32] per-world contact records (collision -> make_constraint hand-off;
                          the public contact_* arrays are compacted from these off the critical path) */
} MjhData;

/* One launch sequence for a reference stage function (MJH_STAGE_*); `stream` is a hipStream_t. */
int mjh_stage(const MjhModel* m, const MjhData* d, int stage, void* stream);

/* step == mjh_stage(MJH_STAGE_STEP): forward.step forward.py:1368 */
int mjh_step(const MjhModel* m, const MjhData* d, void* stream);
int mjh_forward(const MjhModel* m, const MjhData* d, void* stream);

/* x = M^-1 y with the stored factor (smooth.solve_m smooth.py:3214); x,y: [nworld, nv] device */
int mjh_solve_m(const MjhModel* m, const MjhData* d, float* x, const float* y, void* stream);
/* res = M vec (support.mul_m support.py:218) */
int mjh_mul_m(const MjhModel* m, const MjhData* d, float* res, const float* vec, void* stream);
/* CSR copy of the dense efc.J in the reference's sparse layout (types.py:2021-2070: J_rownnz / J_rowadr [nworld, njmax], J_colind / J
 * [nworld, njmax_nnz]; the reference stores efc.J this way for nv > 32, io.py:1804-1808).  Entries = the numerically non-zero values of
 * each row (a subset of the reference's structural pattern); rows whose entries do not fit njmax_nnz are truncated and
 * OverflowType.NJMAX_NNZ is set.  Opt-in: nothing inside step reads it. */
/* Dense Cholesky factors of M's diagonal (kinematic-tree) blocks in the reference's packed layout (io.py:173-211 m_block_layout,
 * smooth.py:3256: upper factor U with M = U^T U, n x n row-major per tree at qld_dense + nworld stride * w + block offset, blocks in tree
 * order; entries below the diagonal are zero).  Opt-in: Data.qLD of this engine stays MuJoCo's sparse L^T D L factor (DESIGN.md, section
 * 2).  Trees of more than 64 dofs are skipped (the reference keeps the sparse factor for them too). */
int mjh_qld_dense(const MjhModel* m, const MjhData* d, float* qld_dense, int stride, void* stream);
/* support.contact_force (support.py:445): 6D force (normal, tangent 1, tangent 2, spin, roll 1, roll 2) of the public contacts contact_ids[0..n),
 * in the contact frame or (to_world_frame != 0) rotated to world axes; force is [n, 6] device memory; slots of ids >= nacon are left untouched */
int mjh_contact_force(const MjhModel* m, const MjhData* d, const int* contact_ids, int n, int to_world_frame, float* force, void* stream);
/* support.jac (support.py:581): Jacobians [nworld, 3, nv] of point[w] (world coordinates) moving with body[w]; jacp or jacr may be NULL */
int mjh_jac(const MjhModel* m, const MjhData* d, float* jacp, float* jacr, const float* point, const int* body, void* stream);
/* ray.rays (ray.py:1219): nearest intersection of nray rays per world with the primitive geoms (mesh / height-field geoms are not
   intersected).  pnt, vec [pnt_nworld (1 or nworld), nray, 3] device; geomgroup: 6 host floats (NULL or six -1: every group; otherwise
   groups whose entry is 0 are skipped); bodyexclude [nray] device ints or NULL; dist [nworld, nray] (-1: no hit); geomid, normal may be NULL */
int mjh_rays(const MjhModel* m, const MjhData* d, const float* pnt, const float* vec, int pnt_nworld, int nray, const float* geomgroup,
             int flg_static, const int* bodyexclude, float* dist, int* geomid, float* normal, void* stream);
int mjh_efc_j_sparse(const MjhModel* m, const MjhData* d, int njmax_nnz, int* rownnz, int* rowadr, int* colind, float* values, void* stream);

/* cli._ctrl_noise cli.py:103-145; ctrl_center may be NULL (-> actuator midpoint); worldid is global */
int mjh_ctrl_noise(const MjhModel* m, const MjhData* d, const float* ctrl_center, int step, float noise_std,
                   float noise_rate, void* stream);

/* hipGraph capture/replay of one step (the reference captures step in a CUDA graph, cli.py:262-265) */
int mjh_graph_create(const MjhModel* m, const MjhData* d, void* stream, void** graph_exec_out);
int mjh_graph_launch(void* graph_exec, void* stream);
int mjh_graph_destroy(void* graph_exec);

/* timed loop helper: runs `nstep` x (ctrl_noise + step) on `stream`, bracketed by hipEvents recorded on
 * that stream; returns elapsed milliseconds through *ms_out (events, not host clocks).  If per_kernel_ms
 * is non-NULL it must hold MJH_NKERNEL floats and receives the summed duration of each kernel class:
 *   0 ctrl_noise | 1 fwd_pos | 2 collision | 3 make_constraint | 4 fwd_vel | 5 solve | 6 integrate | 7 other | 8 mid
 * plain_kernels = 0 times the launches of the fused step (classes 0, 1, 8 = collision+make_constraint+fwd_vel in one
 * launch, 5, 6); plain_kernels = 1 runs one plain kernel per stage instead (classes 0-7), for the per-stage trace. */
#define MJH_NKERNEL 9
int mjh_timed_steps(const MjhModel* m, const MjhData* d, int nstep, int step0, float noise_std, float noise_rate,
                    void* stream, float* ms_out, float* per_kernel_ms, int plain_kernels);

/* Frees what the calling host thread holds inside the library (the low-priority side stream and its two events that a Newton
 * step forks onto, one set per device).  Optional: call when a stepping thread ends; safe to call more than once. */
int mjh_release_thread_resources(void);

const char* mjh_last_error(void);
/* Developer knobs -- the ONE test hook.  The library snapshots the MJH_* environment variables once, when it is loaded, and never calls getenv on a
 * launch path; this call sets (value != NULL) or clears (NULL) an entry of that snapshot afterwards, e.g. mjh_dev_knob("MJH_CG_KERNEL", "pair") to run
 * the two-worlds-per-wavefront CG kernel where the dispatch would pick the pooled one.  Knobs select between kernels / launch shapes that all pass the
 * parity suite; none skips work.  Knobs cached at first use (most of them: see csrc) keep their first value.  Not to be called while another
 * thread is inside a mjh_* call.  Host only. */
int mjh_dev_knob(const char* name, const char* value);
/* Name of the solver mapping the dispatch picks for (m, d) in a fused step -- "cgp" (pooled contact-basis CG, csrc/solver_cgp.hpp), "cgw" (one world per
 * wavefront), "pair" (k_solve<cg>), "newton_mfma", "newton32", "cg64", "newton64", "*_ell", "tree+big", "big", "pgs", "pgs_big", "unsupported" -- so that a
 * test or a bench line can say which kernel it measured without a knob.  Static string; host only; launches nothing. */
const char* mjh_solver_kernel(const MjhModel* m, const MjhData* d);
#define MJH_ABI_VERSION 43
/* floats of Data.ws_ccd for a model with GJK pairs (csrc/convex.hpp ccd_layout: per world the candidate list, the per-candidate result cache and the
   broadphase mask; then the counters, the convex-pair mask and the EPA hand-over records) -- what a binding that allocates Data itself must
   provide; iterations = max(ccd_iterations, epa_iterations), concap = Data.concap.  Also returns the default Data.nccdhand through *nccdhand_out
   (may be NULL); a binding that sets its own nccdhand (the reference's nccdmax x nworld / naccdmax) adds 64 floats per entry beyond it.
   npolygonmax / nmeshdegmax no longer enter (the multi-contact recovery works in LDS since round 5).  Host only. */
int mjh_ws_ccd_floats(int nworld, int iterations, int nhfield, int npolygonmax, int nmeshdegmax, int npair, int concap, double* floats_out, int* nccdhand_out);
int mjh_abi_version(void); /* returns MJH_ABI_VERSION of the library that was loaded */
/* hash of the sources and compiler flags the library was built from (csrc/build_id.hip); the Python loader compares it with the hash of the
   sources on disk and rebuilds -- or refuses to load -- a library that does not match */
const char* mjh_build_id(void);

#ifdef __cplusplus
}
#endif
#endif
