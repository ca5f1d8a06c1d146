// mjb_dev.h — structures shared by the host API (mjb_api.hip) and the gfx950 kernels (mjb_step.hip).
//
// DevModel mirrors mjb_model_desc (include/mjb_model_fields.def) with every array living in one
// device blob.  The pointers are declared in the AMDGPU *constant* address space (4): model data is
// immutable and wave-uniformly indexed, so the compiler fetches it with scalar loads (s_load_*)
// into SGPRs instead of spending vector-memory issue slots and VGPRs on it.
#pragma once

#include <stdint.h>

#include "../../include/mjb.h"

#define MJB_AS4 __attribute__((address_space(4)))
typedef const int MJB_AS4 *mjb_ciptr;
typedef const double MJB_AS4 *mjb_cdptr;

struct DevModel {
#define MJB_SIZE(name) int name;
#define MJB_OPT_I(name) int name;
#define MJB_OPT_D(name, n) double name[n];
#define MJB_ARR_I(name, rows, cols) mjb_ciptr name;
#define MJB_ARR_D(name, rows, cols) mjb_cdptr name;
#include "../../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	// engine-derived constant tables (built in mjb_compile)
	mjb_ciptr M_rowdof;    // [nM] dof i of qM entry e
	mjb_ciptr M_coldof;    // [nM] ancestor dof j of qM entry e
	mjb_ciptr dof_depth;   // [nv] number of entries in row i of qM (self + ancestors)
	mjb_ciptr dof_jstart;  // [nv] first dof of the "velocity group" the dof belongs to (see com_vel)
	mjb_ciptr body_rec;    // [nbody][4] packed {parentid, dofadr, dofnum, rootid}   (one s_load_dwordx4 per body)
	mjb_ciptr body_rec2;   // [nbody][4] packed {jntadr, jntnum, sameframe, weldid}
	mjb_ciptr dof_rec;     // [nv][4]    packed {Madr, nancestor, bodyid, parentid}
	mjb_ciptr dof_rec2;    // [nv][4]    packed {translational dof of a free joint, first joint of its body, parent of its body, bodyid}
	mjb_ciptr jnt_rec;     // [njnt][4]  packed {bodyid, type, dofadr, rootid of its body}
	mjb_ciptr fac_ops;     // [nfac][4]  factorisation micro-ops {dst, srcA, srcB, 0}: LD[dst] -= LD[srcA]/LD[kk]*LD[srcB]
	mjb_ciptr fac_beg;     // [nv+1]     first micro-op of pivot k
	mjb_ciptr flv_hdr;     // [flv_n][4] the same updates by levels of the elimination tree (mjb_api.hip): { slots, longest list of slot 0 / 1 / 2 }
	mjb_ciptr flv_rec;     // [flv_n + 3][3][64][4] the levels' contributions, one 16-byte word per lane and slot
	mjb_ciptr flv_ent;     // [nM, padded to 4 x 64] row | diagonal << 16 of every qM entry
	int flv_n;
	mjb_ciptr sens_copy;     // [2][3][sens_ncopy_max][2] (layout full/compact, stage-1): {dst offset in sensordata, src frame offset}
	mjb_ciptr sens_slow;     // [3][nsensor] ids of the sensors of each stage that need real work
	mjb_ciptr dof_act_adr;   // [nv+1] CSR: the actuators driving each dof (a tendon transmission appears under every joint of its tendon)
	mjb_ciptr dof_act_id;    // [dof_act_adr[nv]]
	mjb_ciptr pair_i;        // [ncollpair][8]  candidate-pair records: g1, g2, type1, type2, condim, friction rule (0 max, 1 geom1, 2 geom2, 4 the pair's own: <contact><pair friction>), collision-function override (MJB_COLFUNC_*), 0
	mjb_cdptr pair_d;        // [ncollpair][24] size1[3] size2[3] margin gap rbound1 rbound2 solref[2] solimp[5] includemargin friction[3] tran pad[2] (rule 4: friction = tangent 1, spin, roll 1; pad = tangent 2, roll 2)
	                         //   ([21] tran = body_invweight0[2 b1] + body_invweight0[2 b2] of the two geoms' bodies: the contact rows' diagApprox)
	// limit items (joints, then tendons) of make_constraint as records in pair_d's slots -- a lane fetches ITS item's record, contact or
	// limit, with one batch of unconditional loads before the kinds diverge: [0..1] range, [6] margin, [10..11] solref, [12..16] solimp,
	// [21] dof_invweight0 / tendon_invweight0;  lim_i [4]: limited (and slide / hinge), qpos address (tendon: its id), dof address, pad
	mjb_cdptr lim_d;         // [njnt + ntendon][24]
	mjb_cdptr dof_act_mom;   // [dof_act_adr[nv]] moment arm of each (dof, actuator) entry: gear (joint transmission) / gear * coefficient (tendon transmission)
	int act_tendon;          // some actuator drives a fixed tendon (the per-actuator lane caches of the dense kernels stand down)
	mjb_cdptr dof_damping_int;  // [nv] what the integrator's implicit matrix adds to M's diagonal, per unit of h: dof_damping (Euler), -diag(D) (implicitfast)
	mjb_ciptr lim_i;         // [njnt + ntendon][4]
	int sens_ncopy[3];       // plain-copy elements per stage
	int sens_nslow[3];       // complex sensors per stage
	int sens_ncopy_max;
	mjb_ciptr body_dofmask;  // [nbody][2] bit i set: dof i moves the body (ancestor-or-self dofs), nv <= 64
	mjb_ciptr M_dense;       // [16][16] qM address of entry (i, j) (dof j ancestor-or-self of dof i) or -1; nv <= 16 only
	mjb_ciptr M_sym;         // [32][32] qM address of the symmetric entry (i, j) or -1; nv <= 32 only (primal solvers: M rows in registers)
	mjb_ciptr body_anc;      // [kin_rounds + 1][nbody] ancestor at distance 2^r (0 = world / beyond the root)
	mjb_ciptr dof_bodymask;  // [nv][2] bit b set: dof i moves body b (transpose of body_dofmask), nbody <= 64
	mjb_ciptr body_submask;  // [nbody][2] bit i set: body i belongs to the body's subtree (incl. itself), nbody <= 64
	int eulerdamp;           // any dof_damping > 0 and EULERDAMP not disabled
	int nfriction;           // number of dofs + tendons with frictionloss > 0 (0 when FRICTIONLOSS is disabled): friction rows exist
	int need_rnepost;        // an acceleration-stage sensor needs mj_rnePostConstraint (cacc / cfrc_int / cfrc_ext)
	int kin_rounds;          // ceil(log2(max body depth)): rounds of the pointer-jumping kinematics
	int maxdepth;          // max dof_depth
	// leaf-to-root sums of the smooth stages (subtree com, composite inertia, RNE's backward pass) as a product with the 0/1 subtree
	// matrix: [sub_nt][4 sub_nt][64] doubles, the A operands of v_mfma_f64_16x16x4_f64 (mjb_step.hip, subtree_sum)
	mjb_cdptr sub_S;
	int sub_nt;              // row tiles (1: nbody <= 16, 2: nbody <= 32; 0: larger models keep the flat sums)
	// root-to-leaf sums (cvel, cacc): [nbody][4] = the body's ancestor-or-self dofs, ASCENDING, one byte each (0xFF = none), at most 16
	mjb_ciptr body_dofanc;
	int dofanc_max;          // longest list (0: some body has more than 16 -- or nv > 255 --: the mask loops run instead)
	// lane = env kernel (mjb_lane_env.hip): the model's numeric constants as ONE tape in the order the kernel consumes them --
	// LeTapeHdr, LeTapeBody[nbody], LeTapeAct[nu] -- read by wide scalar loads off a single base address (NULL: no compiled-in topology)
	mjb_cdptr le_tape;
};

// (64-byte records: one s_load_dwordx16 per half)
struct LeTapeHdr { double dt, gravity[3], pad[4]; };
struct LeTapeBody {
	// pose half: what mj_kinematics needs of the body and its (single) joint
	double pos[3], quat[4], jaxis[3], jpos[3], qpos0, stiffness, spring;
	// inertial half
	double ipos[3], ibody[6] /* R(iquat) diag(inertia) R(iquat)': xx yy zz xy xz yz */, mass, damping, armature, hdamping /* timestep * damping */, pad[3];
};
struct LeTapeAct { double gear, ctrllo, ctrlhi, gain[3], bias[3], forcelo, forcehi, pad[5]; };
static_assert(sizeof(LeTapeHdr) == 64 && sizeof(LeTapeBody) == 256 && sizeof(LeTapeAct) == 128, "lane = env tape records");

// Offsets (in doubles / ints) of every data field inside one per-env frame.
struct FrameLayout {
#define MJB_DS(name, rows, cols) int name;
#define MJB_DD(name, rows, cols) int name;
#define MJB_DD2(name, rows, cols) int name;
#define MJB_DI(name, rows, cols) int name;
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	int MhB;       // [nM]  M + h*diag(damping) (Euler implicit damping), built and factorised next to qM
	int qH;        // [nM]  its L'DL factor
	int qHdi;      // [nv]  1 / diag
	int nwt_M;     // [nv*nv] Newton: dense M          (size 0 unless solver == Newton with constraint rows)
	int nwt_H;     // [nv*nv] Newton: Hessian / its Cholesky factor
	int nwt_vec;   // [5*nv]  Newton: qacc, Ma, grad, search, Mv
	int nwt_row;   // [3*nefcmax] Newton: per-row jaref, jv, Hessian weight
	int nwt_hc;    // [hcs*nconmax] Newton: Hessian blocks of the elliptic cones (size 0 unless cone == elliptic); PGS + elliptic: blocks of AR
	int hcs, hcd;  // cone blocks are hcd x hcd, hcs doubles apart (primal solvers: hcd = the model's largest contact dim, hcs = max(hcd^2, 10); PGS: 36, 6)
	int iscratch;  // transient int scratch: max(ncollpair, njnt + nconmax)
	int gravity;   // [3] this env's gravity (model value or per-env override, loaded with the state)
	int gfriction; // [3*ngeom] this env's geom friction (models with contacts only)
	int eqparam;   // [19*neq] this env's equality parameters: active | eq_data[11] | solref[2] | solimp[5] per equality
	int cwrench;  // [nconmax][6] world contact wrenches (rne_post scratch)
	int kinloc;    // [7*nbody] kinematics: pose of each body in its parent frame (transient)
	int crbbuf;    // [6*nv]    crb: crb[body(i)] * cdof_i (transient)
	int eulerx;    // [nv]      Euler: velocity-update right-hand side / solution (transient)
	int dadr;      // [64 ints] constrained kernels, nv <= 16: packed dense address map of lanes 0-15 (int frame)
	int tri;       // transient scratch: packed dense triangle of the L'DL factor (PGS, nv <= 16: 128 doubles; 16 < nv <= 32: 496, solve_tri32)
	int rk;        // [nq + 4 nv + nsensordata] RK4 bookkeeping (rk4_stage), -1 under Euler
	int jrows;     // rows of efc_J the frame holds (nefcmax, except in the fused frame of kernel variant 4: 64, the rest in DevState::efc_Jg)
	int rcap;      // rows every OTHER per-row array holds (efc_D / aref / b / force / frictionloss / type / id, nwt_row, the row metadata in
	               // iscratch): nefcmax, except in the X-layout fused frame of kernel variant 4, where it equals jrows (an env-step with
	               // more rows keeps ALL of its row data in the env's block of DevState::efc_Jg, see RowBlock)
	int hcrow;     // 1: the cone block of a contact sits at nwt_hc + hcd * (its first row) [hcd * rcap doubles]; 0: at nwt_hc + hcs * contact
	int solvescr;  // [32]      pivot-row scratch of the dense M^-1 solves in fwd_acceleration / Euler (the factorisation uses crbbuf)
	int bbscr;     // [216]     transient scratch of the box - box narrow phase (alive inside collision only)
	int ndouble;   // doubles per frame
	int nint;      // ints per frame (follow the doubles)
	int nstate;    // doubles in the persistent prefix
};

// HBM arrays of the persistent state, env-major [nenv][dim]
struct DevState {
#define MJB_DS(name, rows, cols) double *name;
#define MJB_DD(name, rows, cols)
#define MJB_DD2(name, rows, cols)
#define MJB_DI(name, rows, cols)
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	double *frame_ws;              // optional [nenv][ndouble + nint/2 padded] full-frame workspace
	unsigned long long *nwarn;     // [MJB_NWARNING] mjData.warning[].number summed over the envs (mjb_warning)
	unsigned long long *prof;      // [64] per-stage cycle sums + call counts (profiling build only), else NULL
	unsigned long long *stats;     // [MJB_NSTATS] workload statistics (mjb_set_stats), NULL when off
	unsigned long long *rowstat;         // [4] evaluations | beyond 64 rows | beyond 128 rows | - of the current long fused launch (kernel variant 4's wide-frame policy), or NULL
	int nenv;
	int frame_stride;              // doubles per env in frame_ws
	const double *env_gravity;       // [nenv][3] per-env gravity override (NULL: the model's)
	const double *env_geom_friction; // [nenv][ngeom][3] per-env geom friction override (NULL: the model's)
	const double *env_geom_size;     // [nenv][ngeom][3] per-env geom sizes (NULL: the model's)
	const int *env_geom_type;        // [nenv][ngeom] per-env geom types (NULL: the model's)
	const double *env_equality;      // [nenv][neq][19] per-env equality parameters (NULL: the model's)
	const double *env_mass;          // [nenv][7 nbody + nv + ntendon + 1] per-env inertial constants (NULL: the model's):
	                                 // body_mass | body_subtreemass | body_inertia[3] | dof_invweight0 | body_invweight0[2] | tendon_invweight0 | meaninertia
	int *sched;                      // [1 + nenv] work counter | chunks done per env, of a chunked fused launch (constrained kernels); NULL otherwise
	double *efc_Jg;                  // [nenv][mjb_rowblock_doubles] row data of the env-steps whose rows outnumber the fused frame's share (kernel variant 4, RowBlock); NULL otherwise
	unsigned long long efc_Jg_stride;  // doubles per env of efc_Jg: sized for the largest cone-block stride of the model's three frame layouts (the kernels lay a block out with the stride of the frame they run on)
	double *pgs_B;                   // [nenv][nefcmax * nv] rows of J M^-1 of the PGS steps beyond 64 rows (nv <= 16 models keep them out of LDS); NULL otherwise
	int use_xfrc;                  // xfrc_applied has ever been written
	const double *zbuf;            // [nsteps][nenv][nu] standard normals of the ctrl-noise injector for ONE fused launch, pre-generated by
	                               // mjb_noise_kernel on the same stream (NULL: generated inside the step kernel)
	const unsigned int *zinfo;     // two records { step0, nsteps, nenv, - }: what the generator wrote each HALF of zbuf for; a launch uses the half
	                               // whose record is its own (the other half is being filled for the NEXT launch, on a side stream, meanwhile)
	unsigned long long zhalf;      // doubles per half
	int keep_frame;                // fused mjb_step also dumps the last step's full frame to frame_ws
	double *handoff;               // [nenv][handoff_stride] hand-off records of the split step (HandoffLayout): written by the lane = env smooth kernel
	                               // (mjb_smooth_kernel.h), read by mjb_cstep_kernel; NULL until the batch first steps that way
	int handoff_stride;
	int sens_every_step;           // lane = env kernel: evaluate the sensors at every step of a fused launch instead of its last one (mjb_set_sensors_every_step)
	int colfunc[64];               // mjCOLLISIONFUNC-style table of MujocoEnv::registerCollisionFunction's overrides by (geom type 1, geom type 2), type1 <= type2,
	                               // index 8 * type1 + type2: read instead of the pair record's copy when per-env geom TYPES are in play (a pair's current types)
	const double *reset_step;      // [nq + nv + nv] qpos | qvel | qacc_warmstart one step after mj_resetData (mj_checkAcc's reset inside the split step), or NULL
	int prof_base;                 // profiling build: first of the two probe ids this launch records (mjb_debug_profile_window)
};

// The env's spill-over block in DevState::efc_Jg (kernel variant 4): efc_J [nefcmax][nv], then -- used only when the frame's row
// arrays are capped too (FrameLayout::rcap < nefcmax) -- D | aref | b | force | frictionloss | jaref | jv | hw [nefcmax each],
// the cone blocks [hcs * nconmax], and three int arrays type | id | meta [nefcmax each].
#if defined(__HIPCC__)
#define MJB_HD __host__ __device__
#else
#define MJB_HD
#endif
MJB_HD inline size_t mjb_rowblock_doubles(int nefcmax, int nv, int nconmax, int hcs)
{
	return (size_t)nefcmax * nv + (size_t)8 * nefcmax + (size_t)hcs * nconmax + (size_t)2 * nefcmax;
}
struct RowBlock {
	double *J, *D, *aref, *b, *force, *fl, *nwt_row, *hc;
	int *type, *id, *meta;
};
MJB_HD inline RowBlock mjb_rowblock(double *base, int nefcmax, int nv, int nconmax, int hcs)
{
	RowBlock g;
	g.J = base;
	g.D = base + (size_t)nefcmax * nv;
	g.aref = g.D + nefcmax;
	g.b = g.aref + nefcmax;
	g.force = g.b + nefcmax;
	g.fl = g.force + nefcmax;
	g.nwt_row = g.fl + nefcmax;
	g.hc = g.nwt_row + 3 * nefcmax;
	g.type = reinterpret_cast<int *>(g.hc + (size_t)hcs * nconmax);
	g.id = g.type + nefcmax;
	g.meta = g.id + nefcmax;
	return g;
}

// The hand-off record of the split step (mjb_smooth_kernel.h -> mjb_cstep_kernel): what the constraint stages read of the smooth stages'
// results, one contiguous record per env in HBM, in this order.  subtree_com holds, for a tree's ROOT body, the point the tree's spatial
// quantities (cdof, hence the contact Jacobians) are taken about -- the root body's origin; the other bodies' entries are not written.
struct HandoffLayout {
	int geom_xpos, geom_xmat, cdof, subtree_com, qLD, qLDiagInv, qH, qHdi, qfrc_smooth, qacc_smooth, ndouble, stride;
};
MJB_HD inline HandoffLayout mjb_handoff_layout(int ngeom, int nv, int nbody, int nM)
{
	HandoffLayout h;
	int o = 0;
	h.geom_xpos = o; o += 3 * ngeom;
	h.geom_xmat = o; o += 9 * ngeom;
	h.cdof = o; o += 6 * nv;
	h.subtree_com = o; o += 3 * nbody;
	h.qLD = o; o += nM;
	h.qLDiagInv = o; o += nv;
	h.qH = o; o += nM;
	h.qHdi = o; o += nv;
	h.qfrc_smooth = o; o += nv;
	h.qacc_smooth = o; o += nv;
	h.ndouble = o;
	h.stride = (o + 7) & ~7;  // records start on 64-byte boundaries
	return h;
}

struct NoiseCfg {
	double rate;   // exp(-dt / max(ctrl_noise_rate, mjMINVAL))
	double scale;  // ctrl_noise_std * sqrt(1 - rate^2)
	unsigned long long seed;
	long long env_offset;
	int enabled;
	int pad;
};

// device-side DefaultRobotHWSim::writeSim (mjb_hwsim_*): controlled joints and their per-env commands / PID state
struct HwSim {
	int n;             // controlled joints (0: stage off)
	int estop;         // e-stop active
	const int *joint, *method, *kind, *antiwindup;          // [n]
	const double *gains;                                     // [n][8]: p, i, d, i_max, i_min, effort_limit, lower, upper
	const double *cmd_pos, *cmd_vel, *cmd_eff, *cmd_hold;    // [nenv][n]; cmd_hold = position commands frozen at e-stop
	double *pid;                                             // [nenv][n][2]: integral of the error, previous error
	// controller cadence of MujocoRosControlPlugin::controlCallback (mjb_hwsim_set_period; period_ns == 0: writeSim at every step on
	// the step's own joint state with the model's timestep, the round-2 behaviour)
	long long period_ns;                                     // control_period as ros::Duration counts it
	double *cad;                                             // [nenv][2 + 2 n]: last update / last write [ns], joint_position_[n], joint_velocity_[n]
};

// Everything a launch needs, resident in device memory (uploaded when it changes); kernels get one
// constant-address-space pointer to it.
struct KernelParams {
	DevModel m;
	FrameLayout L;    // full layout: every field has its own storage (forward / split step / dumps)
	FrameLayout Lc;   // compact layout of the fused step: fields with disjoint lifetimes share storage
	int use_compact;  // fused launches use Lc
	int pad0;
	DevState s;
	NoiseCfg nz;
	HwSim hw;
};

enum { MJB_MODE_STEP = 0, MJB_MODE_FORWARD = 1, MJB_MODE_STEP1 = 2, MJB_MODE_STEP2 = 3, MJB_MODE_STEP21 = 4 /* STEP2 of one step, then STEP1 of the next */,
       // STEP2 of an RK4 step cut at the callback points of its evaluations (mjb_step2_rk_prefix; the evaluation index travels in the
       // kernel's `nsteps` argument): RKMID = second half of evaluation rk, rk4_stage(rk), first half of evaluation rk + 1;
       // RKLAST = second half of evaluation 3 + the final advance
       MJB_MODE_RKMID = 5, MJB_MODE_RKLAST = 6 };

// launches (implemented in mjb_step.hip); returns hipError_t as int
// (steps envs [env_lo, nenv): the whole batch, or a prefix / the rest for the split steps of the host runtime)
int mjb_launch_noise(const KernelParams *Pdev, double *zbuf, unsigned int *zinfo, int nenv, int nu, int nsteps, unsigned int step0, void *stream);
int mjb_launch_step(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0,
                    int lanes_per_env, int envs_per_block, int constrained, int dense, void *stream);
int mjb_launch_reset(const KernelParams *Pdev, int nenv, const unsigned char *mask_dev, void *stream);
int mjb_max_lds_bytes();
// lane = env form of the unconstrained fused step (mjb_lane_env.hip)
int mjb_lane_env_match(const mjb_model_desc *h);
size_t mjb_lane_env_tape_doubles(const mjb_model_desc *h);      // size of the constant tape
void mjb_lane_env_tape(const mjb_model_desc *h, double *tape);  // fills it
const char *mjb_lane_env_name(int topo);
enum { MJB_LE_TOPO_NONE = -1, MJB_LE_TOPO_JIT = -2, MJB_LE_UNAVAILABLE = -1000 };
int mjb_lane_env_eligible(const mjb_model_desc *h);   // the model's structure fits the kernel (compiled in or not)
const char *mjb_lane_env_jit_error(void);
void mjb_lane_env_jit_stats(int *compiled, int *disk_hits);  // hiprtc builds of this process / builds taken from the disk cache instead              // why the last hiprtc build of a topology was not available ("" if none failed)
int mjb_launch_lane_env(const KernelParams *Pdev, int topo, const mjb_model_desc *h, int nenv_batch, int env_lo, int env_hi, int nsteps, unsigned int step0,
                        void *stream);
// the split step (mjb_smooth_kernel.h + mjb_cstep_kernel): one step of envs [env_lo, env_hi) -- smooth half in lane = env form, then the constraint half
int mjb_smooth_match(const mjb_model_desc *h);  // index of the compiled-in SmTopo the model has, or -1
const char *mjb_smooth_name(int topo);
int mjb_launch_smooth(const KernelParams *Pdev, int topo, int env_lo, int env_hi, unsigned int step, int flags, void *stream);
int mjb_launch_cstep(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int env_hi, int envs_per_block, int flags, void *stream);
// sensors-plugin equivalent (mjb_sensor_pack.hip)
int mjb_launch_sensor_pack(const KernelParams *Pdev, int nenv, int nsensor, const int *set_flag, const double *mean,
                           const double *sigma, unsigned long long seed, long long env_offset, unsigned int step,
                           float *value, float *truth, void *stream);
