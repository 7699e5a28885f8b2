/* oracle/mjref.h -- TEST INFRASTRUCTURE ONLY.
 *
 * float64, single-world, plain-C restatement of the reference's mj_step hot path
 * (/root/reference/mujoco_warp/_src/{smooth,passive,forward,collision_*,constraint,solver}.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (mujoco_warp_amd) never does.
 *
 * PIN STATUS (round 4, unchanged in round 5).  The arithmetic of record is MuJoCo C (mujoco 3.11.1.dev954833728, uv.lock:1053), absent from /root/reference
 * and from this environment.  What pins this restatement instead:
 *   - EXTERNALLY: the states MuJoCo C itself went through, recorded in the reference's benchmark input
 *     benchmarks/unitree_g1/shuffle_dance.npz (qpos [251, 36], qvel [251, 35] at 50 Hz next to the controls): from every recorded frame,
 *     four oracle steps land on the next recorded frame to the recording's float32 storage precision (qpos median 1.7e-7, qvel median
 *     1.0e-5 over all 250 intervals; a free run tracks the recording for 100 frames).  That pins, for the G1 chain, MJCF compilation ->
 *     FK -> CRBA -> RNE -> position actuators -> collision -> constraint rows -> Newton -> implicitfast (tests/test_reference_trajectory.py);
 *   - EXTERNALLY, second anchor (round 4): the reference's own golden for the aloha task (unroll_test.py:40-58: replay lift_pot.npz on
 *     test_data/aloha_pot/scene.xml, the pot body ends above z = 0.069 and the lid body above z = 0.16).  Both cones reproduce it in
 *     float64 (pot z 0.108, lid 0.168: tests/test_aloha_pot.py) -- mesh compilation (134 files), joint-level actuator force range + gravity
 *     compensation routing, the zero-quaternion rule, GJK / EPA / mesh multicontact on real assets, elliptic and pyramidal rows;
 *   - every number the reference's own tests hold for this path (math vectors, key-0 contact / row counts, broadphase counts, the
 *     GJK / EPA / multi-contact values of collision_gjk_test.py: tests/test_reference_vectors.py, test_convex.py, test_reference_gjk_gpu.py);
 *   - MuJoCo-independent identities for the rest (energy, M (M^-1 y) = y, KKT residuals, Newton = CG = PGS fixed points, cone membership).
 * UNPINNED still: CG and PGS as algorithms (only their fixed points are pinned, through Newton), elliptic cones, sleeping policy
 * stand-ins, meshes / height fields beyond the reference-held numbers, sensors, rays.
 *
 * FLOAT32 TWIN: the same sources compiled with -DREF_REAL_FLOAT (libmjref32.so, ref.RefSim(real="f32")) restate the reference's algorithm
 * in the reference's own precision.  It is the measured float32 floor the engine is compared against (tools/parity_report.py), not a
 * second oracle: the float64 build stays the arithmetic of record.
 *
 * The struct layouts below are parsed by oracle/ref.py (one declaration per line, no macros).
 */
#ifndef MJREF_H
#define MJREF_H

typedef struct RefModel {
  int nq;
  int nv;
  int nu;
  int na;
  int nbody;
  int njnt;
  int ngeom;
  int nC;
  int npair;
  int neq;
  int njmax;
  int nconmax;
  int nmocap;
  int nexplicit;
  int ntree;
  int nsite;
  int nsensor;
  int nsensordata;
  int nmeshpoly;   /* polygons of all meshes (0: no polygon tables, no multi-contact recovery on meshes) */
  int npolygonmax; /* clip buffers hold 2 * npolygonmax points (collision_convex.py:1226-1234) */
  int enableflags;       /* EnableBit: SLEEP = 1 << 5 (with DisableBit.ISLAND clear: forward.py:345) */
  int integrator;
  int cone;
  int solver;
  int iterations;
  int ls_iterations;
  int disableflags;
  int broadphase;        /* BroadphaseType: 0 NXN, 1 SAP_TILE, 2 SAP_SEGMENTED (io.py:631-636) */
  int broadphase_filter; /* BroadphaseFilter bits: 1 plane, 2 sphere, 4 AABB, 8 OBB (io.py:405) */
  int ccd_iterations;    /* GJK iteration cap (opt.ccd_iterations) */
  int epa_iterations;    /* EPA iteration cap: 16 when every convex pair of the model is box-box, else ccd_iterations (collision_convex.py:1223) */
  double timestep;
  double tolerance;
  double ls_tolerance;
  double impratio;
  double ccd_tolerance;
  double meaninertia;
  double sleep_tolerance;
  double* gravity;
  double* magnetic;
  double* qpos0;
  double* qpos_spring;
  int* body_parentid;
  int* body_mocapid;
  int* body_rootid;
  int* body_weldid;
  int* body_jntnum;
  int* body_jntadr;
  int* body_dofnum;
  int* body_dofadr;
  double* body_pos;
  double* body_quat;
  double* body_ipos;
  double* body_iquat;
  double* body_mass;
  double* body_subtreemass;
  double* body_inertia;
  double* body_invweight0;
  double* body_gravcomp;
  int* jnt_type;
  int* jnt_qposadr;
  int* jnt_dofadr;
  int* jnt_bodyid;
  int* jnt_limited;
  double* jnt_solref;
  double* jnt_solimp;
  double* jnt_pos;
  double* jnt_axis;
  double* jnt_stiffness;
  double* jnt_range;
  double* jnt_margin;
  int* jnt_actfrclimited;   /* types.py:1075: clamp of the summed actuator force on the joint (forward.py:1145-1147) */
  double* jnt_actfrcrange;  /* [njnt, 2] */
  int* jnt_actgravcomp;     /* gravity compensation routed through the actuators (passive.py:652, forward.py:1141) */
  int* dof_bodyid;
  int* dof_jntid;
  int* dof_parentid;
  double* dof_solref;
  double* dof_solimp;
  double* dof_frictionloss;
  double* dof_armature;
  double* dof_damping;
  double* dof_invweight0;
  int* M_rownnz;
  int* M_rowadr;
  int* M_colind;
  int* geom_type;
  int* geom_condim;
  int* geom_bodyid;
  int* geom_priority;
  int* geom_group;  /* ray casting (ray.py:52): group, material and colour (alpha 0 = invisible to rays) */
  int* geom_matid;
  double* geom_rgba; /* [ngeom, 4] */
  double* mat_rgba;  /* [nmat, 4] */
  double* geom_solmix;
  double* geom_solref;
  double* geom_solimp;
  double* geom_size;
  double* geom_rbound;
  double* geom_aabb; /* [ngeom, 6]: centre, half sizes in the geom frame */
  double* geom_pos;
  double* geom_quat;
  double* geom_friction;
  double* geom_margin;
  double* geom_gap;
  int* pair_geom;
  int* nxn_pairid;      /* explicit <contact><pair> index or -1, per filtered pair */
  int* xpair_dim;
  double* xpair_friction;
  double* xpair_solref;
  double* xpair_solreffriction;
  double* xpair_solimp;
  double* xpair_margin;
  double* xpair_gap;
  int* actuator_dyntype;
  int* actuator_gaintype;
  int* actuator_biastype;
  int* actuator_trnid;
  int* actuator_actadr;
  int* actuator_ctrllimited;
  int* actuator_forcelimited;
  int* actuator_actlimited;
  double* actuator_dynprm;
  double* actuator_gainprm;
  double* actuator_biasprm;
  double* actuator_ctrlrange;
  double* actuator_forcerange;
  double* actuator_actrange;
  double* actuator_gear;
  int* eq_obj1id;
  int* eq_obj2id;
  int* eq_active0;
  double* eq_solref;
  double* eq_solimp;
  double* eq_data;
  int* geom_dataid;   /* [ngeom] mesh id of mesh geoms, -1 otherwise */
  int* mesh_vertadr;
  int* mesh_vertnum;
  double* mesh_vert;  /* [nmeshvert, 3] vertices in the mesh (= geom) frame */
  int* site_bodyid;
  double* site_pos;
  double* site_quat;
  int* site_type;
  double* site_size;
  int* sensor_type;     /* mjtSensor */
  int* sensor_datatype; /* mjtDataType: 0 real, 1 positive, 2 axis, 3 quaternion */
  int* sensor_objtype;  /* mjtObj: 1 body (inertial frame), 2 xbody, 5 geom, 6 site */
  int* sensor_objid;
  int* sensor_reftype;
  int* sensor_refid;
  int* sensor_dim;
  int* sensor_adr;
  double* sensor_cutoff;
  double* hfield_size; /* [nhfield, 4]: x, y half sizes, top scale of the elevation data, base thickness */
  int* hfield_nrow;
  int* hfield_ncol;
  int* hfield_adr;
  double* hfield_data; /* elevations normalised to [0, 1] */
  int* mesh_graphadr; /* [nmesh] first word of the mesh's hill-climbing graph in mesh_graph, -1: none */
  int* mesh_graph;
  int* mesh_polyadr;
  double* mesh_polynormal;
  int* mesh_polyvertadr;
  int* mesh_polyvertnum;
  int* mesh_polyvert;
  int* mesh_polymapadr;
  int* mesh_polymapnum;
  int* mesh_polymap;
  int* body_treeid;
  int* dof_treeid;
  int* tree_dofadr;
  int* tree_dofnum;
  int* tree_sleep_policy; /* SleepPolicy: 0 AUTO, 1 AUTO_NEVER, 2 AUTO_ALLOWED */
  double* dof_length;
} RefModel;

typedef struct RefData {
  double time;
  int ncon;
  int ne;
  int nf;
  int nl;
  int nefc;
  int solver_niter;
  int ncollision;
  int overflow;
  int nisland;
  int ntree_awake;
  int nbody_awake;
  int nv_awake;
  double* qpos;
  double* qvel;
  double* act;
  double* ctrl;
  double* qacc_warmstart;
  double* qfrc_applied;
  double* xfrc_applied;
  double* mocap_pos;
  double* mocap_quat;
  double* xpos;
  double* xquat;
  double* xmat;
  double* xipos;
  double* ximat;
  double* xanchor;
  double* xaxis;
  double* geom_xpos;
  double* geom_xmat;
  double* subtree_com;
  double* cinert;
  double* cdof;
  double* crb;
  double* M;
  double* qLD;
  double* qLDiagInv;
  double* cvel;
  double* cdof_dot;
  double* qfrc_spring;
  double* qfrc_damper;
  double* qfrc_gravcomp;
  double* qfrc_passive;
  double* qfrc_bias;
  double* cacc;
  double* cfrc_int;
  double* cfrc_ext;
  double* actuator_length;
  double* actuator_velocity;
  double* actuator_force;
  double* act_dot;
  double* qfrc_actuator;
  double* qfrc_smooth;
  double* qacc_smooth;
  double* qacc;
  double* qfrc_constraint;
  double* Ma;
  double* con_dist;
  double* con_pos;
  double* con_frame;
  double* con_includemargin;
  double* con_friction;
  double* con_solref;
  double* con_solreffriction;
  double* con_solimp;
  int* con_dim;
  int* con_geom;
  int* con_efc_address;
  int* efc_type;
  int* efc_id;
  int* efc_state;
  double* efc_J;
  double* efc_pos;
  double* efc_margin;
  double* efc_D;
  double* efc_vel;
  double* efc_aref;
  double* efc_frictionloss;
  double* efc_force;
  double* sensordata;
  double* subtree_linvel;
  double* subtree_angmom;
  int* tree_asleep;   /* sleep.py: < 0 awake (countdown to -1), >= 0 next tree of the sleep cycle */
  int* tree_awake;
  int* body_awake;    /* SleepState: -1 static, 0 asleep, 1 awake */
  int* tree_island;
  int* body_awake_ind;
  int* dof_awake_ind;
} RefData;

void ref_kinematics(const RefModel* m, RefData* d);
void ref_com_pos(const RefModel* m, RefData* d);
void ref_crb(const RefModel* m, RefData* d);
void ref_factor_m(const RefModel* m, RefData* d);
void ref_solve_m(const RefModel* m, const RefData* d, double* x, const double* y);
void ref_mul_m(const RefModel* m, const RefData* d, double* res, const double* vec);
void ref_collision(const RefModel* m, RefData* d);
void ref_make_constraint(const RefModel* m, RefData* d);
void ref_transmission(const RefModel* m, RefData* d);
void ref_com_vel(const RefModel* m, RefData* d);
void ref_passive(const RefModel* m, RefData* d);
void ref_rne(const RefModel* m, RefData* d);
void ref_fwd_position(const RefModel* m, RefData* d);
void ref_fwd_velocity(const RefModel* m, RefData* d);
void ref_fwd_actuation(const RefModel* m, RefData* d);
void ref_fwd_acceleration(const RefModel* m, RefData* d);
void ref_solve(const RefModel* m, RefData* d);
void ref_forward(const RefModel* m, RefData* d);
void ref_euler(const RefModel* m, RefData* d);
void ref_implicitfast(const RefModel* m, RefData* d);
void ref_rungekutta4(const RefModel* m, RefData* d); /* forward.py:524; call after ref_forward */
void ref_step(const RefModel* m, RefData* d);
void ref_subtree_vel(const RefModel* m, RefData* d); /* smooth.py:3614 */
void ref_rne_postconstraint(const RefModel* m, RefData* d); /* smooth.py:1744; call after ref_solve */
/* ray.py:907-1011 _ray for one ray: distance to the nearest primitive geom (-1: none), its id and the normal there; geomgroup: 6 doubles or NULL */
double ref_ray(const RefModel* m, const RefData* d, const double* pnt, const double* vec, const double* geomgroup, int flg_static, int bodyexclude,
               int* geomid, double* normal);
void ref_sensor(const RefModel* m, RefData* d); /* sensor.py sensor_pos / sensor_vel / sensor_acc, the subset in oracle/mjref.c; called by ref_forward */
/* sleep.py / island.py:28-310 (tree-level constraint islands, sleeping, waking) */
void ref_update_sleep(const RefModel* m, RefData* d);
void ref_wake(const RefModel* m, RefData* d);
void ref_wake_collision(const RefModel* m, RefData* d);
void ref_wake_equality(const RefModel* m, RefData* d);
void ref_island(const RefModel* m, RefData* d);
void ref_sleep(const RefModel* m, RefData* d);
void ref_ctrl_noise(const RefModel* m, RefData* d, const double* center, int step, int worldid, double noise_std, double noise_rate);
double ref_halton(int index, int base);
void ref_closest_segment_to_segment_points(const double* a0, const double* a1, const double* b0, const double* b1, double* best_a_out,
                                           double* best_b_out); /* math.py:283 */
int ref_upper_tri_index(int n, int i, int j);  /* math.py:323 */
int ref_upper_trid_index(int n, int i, int j); /* math.py:329 */
/* collision_gjk.py:2529 ccd on two posed primitive convex geoms (types: GeomType values; mat row-major 3x3); out = dist, x1[3], x2[3];
 * returns the number of contacts (box pairs: after multi-contact recovery when multiccd != 0, witness pairs in wit[8][3]) */
int ref_ccd(int type1, const double* pos1, const double* mat1, const double* size1, int type2, const double* pos2, const double* mat2,
            const double* size2, double margin, double tolerance, double cutoff, int iterations, int multiccd, double* out, double* wit);
int ref_ccd_mesh(int type1, const double* pos1, const double* mat1, const double* size1, const double* vert1, int nvert1, int type2,
                 const double* pos2, const double* mat2, const double* size2, const double* vert2, int nvert2, double margin, double tolerance,
                 double cutoff, int iterations, int multiccd, double* out, double* wit); /* ref_ccd with mesh vertices (geom frame) */
int ref_ccd_geoms(const RefModel* m, int g1, int g2, const double* pos1, const double* mat1, const double* pos2, const double* mat2, double margin,
                  double tolerance, double cutoff, int iterations, int multiccd, double* out, double* wit);
int ref_rollout(const RefModel* m, RefData* d, int nstep, int worldid, double noise_std, double noise_rate, double* qpos_out, double* qvel_out);

#endif
