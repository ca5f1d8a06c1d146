// mjb_step.hip — gfx950 (MI355X / CDNA4) kernels of the batched step engine.
//
// One kernel advances every env of a batch by K full steps (the work of K x `mj_step(model,data)` per
// env; reference call sites /root/reference mujoco_ros/src/mujoco_env.cpp:498,552,593).
//
// Mapping.  G lanes of ONE wavefront (G = 8/16/32/64, template) cooperate on one env; a workgroup
// holds EPB envs.  Each env owns a frame in LDS that holds its whole mjData-like working set
// (layout: FrameLayout, mjb_dev.h) for all K steps; HBM is touched only for the persistent state
// (env-major arrays, one contiguous segment per env and field) at launch begin/end.  A group never
// spans wavefronts, so lanes of a group run in lockstep and LDS ops of a wave retire in order:
// `gsync` is a compiler-level fence + wave_barrier, never an s_barrier, and waves never wait for
// each other.  Tree recursions run component-per-lane (no cross-lane dependency along the chain),
// everything else item-per-lane (body / dof / joint / qM entry / sensor / actuator).
// The model is immutable and indexed wave-uniformly wherever possible, so it is read through
// constant-address-space pointers (scalar loads).
//
// Stage functions are named after the MuJoCo 2.3.7 stage whose result they produce.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include <utility>

#include "mjb_dev.h"
#include "mjb_math.h"

namespace {

typedef const DevModel MJB_AS4 &CModel;
typedef const FrameLayout MJB_AS4 &CLayout;
typedef const DevState MJB_AS4 &CState;
typedef const NoiseCfg MJB_AS4 &CNoise;

// Marginal-cost profile of the SHIPPED kernels (tools/stage_marginal.py): a build with -DMJB_DOUBLE_STAGE=<id> runs that stage (every one of them
// recomputes its outputs from its inputs) TWICE; launch time minus the production build's = what the stage costs in THROUGHPUT on the production register allocation,
// partner wavefront and all -- the windowed cycle probes measure one wavefront's latency and perturb the 256-register allocation.  -1: production (the loops vanish).
#ifndef MJB_DOUBLE_STAGE
#define MJB_DOUBLE_STAGE -1
#endif
#define MJB_REP(id) _Pragma("nounroll") for (int rep_ = 0; rep_ < (MJB_DOUBLE_STAGE == (id) ? 2 : 1); rep_++)

#ifdef MJB_STAGE_NOINLINE
#define STAGE static __device__ __noinline__
#else
#define STAGE static __device__ __forceinline__
#endif

// (MJB_GSYNC_LOCAL -- the slice of the kernels without constraint rows, whose lanes exchange data through the LDS frame only: the
//  fences name the LDS address space.  A plain wavefront-scope release fence makes the compiler wait for EVERY outstanding memory
//  operation, vmcnt(0) included: each stage boundary then also waited for global loads nobody needs yet -- the model-table reads of
//  the next stage, the next step's ctrl-noise value fetched a step ahead.  The constrained kernels keep the full fence: rows beyond
//  the frame's share travel between lanes through HBM there.)
#if !defined(MJB_GSYNC_LOCAL) && defined(MJB_GROUP)
#if MJB_GROUP == 0 && !defined(MJB_DEV_ONLY_CON)
#define MJB_GSYNC_LOCAL 1
#endif
#endif
#ifndef MJB_GSYNC_LOCAL
#define MJB_GSYNC_LOCAL 0
#endif
template <int G> DEVI void gsync()
{
#if MJB_GSYNC_LOCAL
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#else
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// (the `asm volatile("")` inside wave-uniform branches keeps them real scalar branches: without it the compiler
//  if-converts the unrolled pivots into selects over the whole register matrix)
#define MJB_KEEP_BRANCH() asm volatile("" ::: "memory")

// value of lane I of this lane's 16-lane DPP row, one v_mov_b32_dpp row_newbcast per dword: what carries pivots and pivot rows
// between the lanes of a 16-lane env group (G == 16: group == row; G == 64: the matrix sits in row 0).  Round 3: replaces the
// LDS publish / wave-uniform read round trip (~110 cycles + two syncs per pivot) the factorisation and both sweeps of the solve
// went through -- same values, same arithmetic, no memory.
template <int I> DEVI double row_bcast16(double v)
{
	static_assert(I >= 0 && I < 16, "row_newbcast selects one of the 16 lanes of a row");
	return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + I, 0xF, 0xF, true),
	                        __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + I, 0xF, 0xF, true));
}
// ... one env per wavefront (G == 64, the matrix in lanes 0 - 15): the source lane is wave-uniform, v_readlane puts the value in a
// scalar register pair that the fma reads directly
template <int G, int I> DEVI double group_bcast(double v)
{
	if constexpr (G == 64)
		return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), I), __builtin_amdgcn_readlane(__double2loint(v), I));
	else
		return row_bcast16<I>(v);
}
template <typename F, int... Is> DEVI void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> DEVI void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }


// Optional per-stage cycle accounting (build with -DMJB_PROFILE -> libmjb_prof.so): env 0 / lane 0 adds the
// s_memtime delta of the probes inside the launch's WINDOW (two consecutive probe ids, DevState::prof_base) to DevState::prof.
#ifdef MJB_PROFILE
// (sums are kept in LDS by the recording lane and flushed to DevState::prof once per launch: a global read-modify-write per
//  probe costs ~1 k cycles, more than the small stages it measures.  The LDS block is 32 BYTES -- two probes per launch, the tool
//  sweeps the window over the ids -- because that is what the lean frames leave of their last LDS granule: config 3's 8 x 20448 B
//  and config 5's 4 x 40864 B keep their residency, so the kernel profiled is the kernel shipped; round 3's 512-byte block cost
//  config 5 its fourth env per CU and config 3 its second wave per SIMD.)
struct ProfLds {
	unsigned long long sum[2];
	unsigned int cnt[2];
	unsigned int base, pad;
};
__shared__ ProfLds mjb_prof_lds;
// (NO lane-divergent branch: every lane of the wave that holds env 0 adds the same -- scalar -- delta to the same address and stores
//  the same sum.  With `if (env == 0 && lane == 0)` around the update, LLVM threaded the recording lane separately through the
//  two-trip loop of forward_first -- the stages' cross-lane LDS exchanges assume the wave runs them together -- and the probes at
//  that loop's edges (windows 8 and 16) ran env 0 into mj_checkAcc resets at every step)
DEVI void prof_rec(int env, int lane, int id, unsigned long long v, unsigned int n = 1)
{
	(void)lane;
	if (__builtin_amdgcn_readfirstlane(env) != 0) return;  // (G < 64: the wave that holds env 0 holds it in its first lanes)
	const unsigned int rel = (unsigned int)id - (unsigned int)__builtin_amdgcn_readfirstlane((int)mjb_prof_lds.base);
	if (rel < 2u) {
		mjb_prof_lds.sum[rel] += v;
		mjb_prof_lds.cnt[rel] += n;
	}
}
#define PROF_BEGIN() unsigned long long _t0 = __builtin_readcyclecounter()
#define PROF(id)                                                                      \
	do {                                                                              \
		unsigned long long _t1 = __builtin_readcyclecounter();                        \
		prof_rec(e.env, e.lane, id, _t1 - _t0);                        \
		_t0 = __builtin_readcyclecounter();                                           \
	} while (0)
// sub-stage accounting inside stages that do not see DevState (the solvers)
#define EPROF_BEGIN() unsigned long long _et0 = __builtin_readcyclecounter()
#define EPROF(id)                                                                     \
	do {                                                                              \
		unsigned long long _et1 = __builtin_readcyclecounter();                       \
		prof_rec(e.env, e.lane, id, _et1 - _et0);                      \
		_et0 = __builtin_readcyclecounter();                                          \
	} while (0)
#else
#define PROF_BEGIN() do { } while (0)
#define PROF(id) do { } while (0)
#define EPROF_BEGIN() do { } while (0)
#define EPROF(id) do { } while (0)
#endif
// libmjb_prof_sm.so: slots 20 - 31 = phases of the smooth stages (com_pos, crb, com_vel, rne, acceleration, euler)
#ifdef MJB_PROFILE_SM
#undef EPROF_BEGIN
#undef EPROF
#define EPROF_BEGIN() do { } while (0)
#define EPROF(id) do { } while (0)
#define SPROF_BEGIN() unsigned long long _st0 = __builtin_readcyclecounter()
#define SPROF(id)                                                                     \
	do {                                                                              \
		unsigned long long _st1 = __builtin_readcyclecounter();                       \
		prof_rec(e.env, e.lane, id, _st1 - _st0);                                     \
		_st0 = __builtin_readcyclecounter();                                          \
	} while (0)
#else
#define SPROF_BEGIN() do { } while (0)
#define SPROF(id) do { } while (0)
#endif

// Integer members of LaneConst re-launder themselves at every read (like LaneId): as plain values they are invariants of the K-step
// loop, and every predicate computed from them -- `sc_dst >= 0`, `q_row == q_col`, the bits of the ancestor masks: ~90 lane masks --
// was hoisted to the top of the kernel, spilled there (183 v_writelane in the prologue of the config-2 kernel) and fetched back with a
// v_readlane pair + s_nop at every use, where one v_cmp against an immediate rebuilds it.
struct LInt {
	int x;
	__device__ __forceinline__ LInt &operator=(int v) { x = v; return *this; }
	__device__ __forceinline__ LInt &operator|=(int v) { x |= v; return *this; }
	__device__ __forceinline__ operator int() const
	{
		int v = x;
		asm volatile("" : "+v"(v));
		return v;
	}
};
struct LUInt {
	unsigned int x;
	__device__ __forceinline__ LUInt &operator=(unsigned int v) { x = v; return *this; }
	__device__ __forceinline__ operator unsigned int() const
	{
		unsigned int v = x;
		asm volatile("" : "+v"(v));
		return v;
	}
};
// Model constants of body `lane`, fetched once per kernel by the dense kernels (nbody <= 16 = G; 512 VGPRs to spend) instead of
// once per stage and step (per-lane table reads are vector-memory loads: ~0.5 us of exposed latency per step each)
struct LaneConst {
	LUInt dmlo, dmhi, smlo, smhi;  // ancestor-dof and subtree-body masks
	LInt jntadr, jntnum, jtype, qa, simple, rootid;
	double bpos[3], bquat[4], jaxis[3], jpos[3], q0, ipos[3], iquat[4], mass, stmass, inertia[3];
	// ... and of dof `lane` (nv <= 16): its body, and where the body velocity "before" the dof's joint comes from
	LInt d_body, d_zero, d_simple, d_parent;
	LUInt d_bmlo, d_bmhi;  // bodies moved by the dof  // d_zero: translational dof of a free joint; d_simple: first joint of its body
	// ... and what the SMALL stages read per lane and step (round 3): transmission, passive, actuation, the sensors' plain copies and
	// Euler each cost 1 - 2 k cycles of which the arithmetic is a handful of fma -- the rest is a chain of two or three dependent
	// loads from the model blob (dof -> actuator list -> actuator -> parameters).  nu, njnt <= 16 (mjb_api.hip picks the kernel).
	LInt u_qa, u_da;        // lane = actuator: qpos / dof address of its joint (joint transmission)
	double u_gear;
	LInt a_n, a_id, a_flags;  // lane = dof: number of actuators driving it (a_n > 1: table walk), the single one's id, flags:
	                         // 1 ctrllimited (and clamping on), 2 affine gain, 4 affine bias, 8 forcelimited
	double a_clo, a_chi, a_g[3], a_b[3], a_flo, a_fhi, a_gear;
	LInt j_type, j_qa, j_da;  // lane = joint
	double j_stiff, j_spring, j_damp;  // (spring reference / damping of a hinge or slide joint; ball / free joints walk the tables)
	LInt sc_dst[3][2], sc_src[3][2];  // lane's plain sensor copies of the three stages in the layout of this launch (-1: none)
	LInt anc[6];            // lane = body: its ancestors at distance 1, 2, 4, 8, 16 (0 = world or beyond): kinematics' pointer jumping
	LInt j_body, j_root;    // lane = joint: its body and that body's root (comPos: cdof)
	// lane = item of kinematics' phase D (joint / geom / site; njnt + ngeom + nsite <= 16, else k_kind = -1: table walk)
	LInt k_kind, k_id, k_body, k_same;  // kind: 0 joint (k_body = the PARENT of the joint's body; k_same = joint is free), 1 geom, 2 site
	double k_pos[3], k_quat[4];
	// lane's qM entries en = lane, lane + 16, lane + 32 (nM <= 48, else q_row[0] = -2: table walk): row / column dof, and for a
	// diagonal entry its armature and h * damping
	LInt q_row[3], q_col[3];
	double q_arm[3], q_hd[3];
};

// The lane's index inside its env group, re-derived from the hardware lane id at EVERY read (two VALU instructions behind an
// `asm volatile`): a plain int member is a loop invariant of the K-step loop, and everything computed from it -- lane * stride
// addresses, the lane == k predicates of the unrolled register code -- gets hoisted to the top of the kernel, kept live across
// every stage and, in the 256-register kernels, spilled there and reloaded from scratch at each use.
struct LaneId {
	int mask;  // G - 1
	__device__ __forceinline__ operator int() const
	{
		int l;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
		return l & mask;
	}
};

// What the OUT-OF-LINE stages (rne_post, reset_frame_state, energy, hwsim_write, the noinline PGS paths) and every stage of
// mjb_constraint.h get: handing them `const Env &` made the
// whole Env -- dadr[16], the LaneConst block -- escape to memory, i.e. a private (scratch) segment in every kernel and a store of
// each member at kernel entry.
struct EnvLite {
	double *f;
	int *fi;
	LaneId lane;
	int env;
	const double *mp;
};
struct Env {
	double *f;  // LDS frame (doubles)
	int *fi;    // LDS frame (ints)
	LaneId lane;  // 0..G-1
	int env;    // batch-local env index
	LInt dadr[16];  // (self-laundering: the sixteen `dadr[i] >= 0` masks are not hoisted) dense kernels only: qM address of entry (i, lane) of the joint-space inertia, or -1
	LaneConst lc;  // one-body-per-lane kernels only
#ifdef MJB_PROFILE
	unsigned long long *prof;
#endif
	const double *mp;  // this env's inertial constants (mjb_set_env_mass_params) or nullptr: the model's.  (Last member: the
	                   // struct is spilled to scratch around the out-of-line stages, and moving the members above by 8 bytes
	                   // misaligns their 16-byte scratch accesses -- measured 2 % on config 2.)
	__device__ __forceinline__ operator EnvLite() const { return EnvLite{ f, fi, lane, env, mp }; }
};

DEVI EnvLite lite(const Env &e) { return EnvLite{ e.f, e.fi, e.lane, e.env, e.mp }; }

// per-env inertial constants (what mj_setConst derives from the masses): the env's own block in HBM when the batch carries
// overrides (mjb_set_env_mass_params; the dense kernels are not used then), else the model's tables
#define MP_BODY_MASS(m, e, i) ((e).mp ? (e).mp[(i)] : (m).body_mass[(i)])
#define MP_SUBTREEMASS(m, e, i) ((e).mp ? (e).mp[(m).nbody + (i)] : (m).body_subtreemass[(i)])
#define MP_INERTIA(m, e, i, k) ((e).mp ? (e).mp[2 * (m).nbody + 3 * (i) + (k)] : (m).body_inertia[3 * (i) + (k)])
#define MP_DOF_INVW(m, e, i) ((e).mp ? (e).mp[5 * (m).nbody + (i)] : (m).dof_invweight0[(i)])
#define MP_BODY_INVW(m, e, i) ((e).mp ? (e).mp[5 * (m).nbody + (m).nv + (i)] : (m).body_invweight0[(i)])
#define MP_TEN_INVW(m, e, i) ((e).mp ? (e).mp[7 * (m).nbody + (m).nv + (i)] : (m).tendon_invweight0[(i)])
#define MP_MEANINERTIA(m, e) ((e).mp ? (e).mp[7 * (m).nbody + (m).nv + (m).ntendon] : (m).meaninertia[0])

// ------------------------------------------------------------------------------------------------
// A1  kinematics: body frames, joint anchors/axes, inertial / geom / site frames
// ------------------------------------------------------------------------------------------------
DEVI void local2global(const double *xpos_b, const double *xquat_b, const double *xmat_b, double *opos,
                       double *omat, const double *pos, const double *quat, int sameframe, double *oquat = nullptr, int wq = -1)
{
	const bool has_quat = wq < 0 ? oquat != nullptr : wq != 0;  // (wq: the caller's per-lane flag when oquat is always a valid address)
	if (sameframe) {
		double p[3], M[9];
		ld3(p, xpos_b);
		ld9(M, xmat_b);
		st3(opos, p);
		st9(omat, M);
		if (has_quat) {
			double bq[4];
			ld4(bq, xquat_b);
			st4(oquat, bq);
		}
	} else {
		double p[3], M[9], q[4], bq[4], v[3], r[9];
		ld3(p, xpos_b);
		ld9(M, xmat_b);
		ld4(bq, xquat_b);
		matvec3(v, M, pos);
		v[0] += p[0]; v[1] += p[1]; v[2] += p[2];
		qmul(q, bq, quat);
		quat2mat(r, q);
		st3(opos, v);
		st9(omat, r);
		if (has_quat) st4(oquat, q);
	}
}

template <int G, bool SCAN, bool CACHE> STAGE void kinematics(CModel m, CLayout L, CState s, const Env &e)
{
	PROF_BEGIN();
	double *f = e.f;
	double *qpos = f + L.qpos, *xpos = f + L.xpos, *xquat = f + L.xquat, *xmat = f + L.xmat;
	double *xanchor = f + L.xanchor, *xaxis = f + L.xaxis;
	double *loc = f + L.kinloc;  // [nbody][7] local pose (pos, quat) of each body in its parent's frame
	const int lane = e.lane;
	// one body per lane of a 16-lane group (dense kernels): the body's pose travels from phase A through B to C in the lane's
	// registers -- no LDS round trip, no sync between the phases (other lanes read it by ds_bpermute in phase B)
	constexpr bool REG = SCAN && CACHE && G == 16;
	[[maybe_unused]] double kp[3] = { 0, 0, 0 }, kq[4] = { 1, 0, 0, 0 };

	// Phase A -- one body per lane: pose relative to the parent INCLUDING the joint motion, plus joint anchors
	// and axes in the parent's frame (parked in xanchor / xaxis until phase D).  Also normalises the
	// quaternions stored in qpos (ball / free joints), as mj_kinematics does.
	for (int b = lane; b < m.nbody; b += G) {
		double p[3], q[4];
		if (b == 0) {
			p[0] = p[1] = p[2] = 0;
			q[0] = 1; q[1] = q[2] = q[3] = 0;
		} else {
			const int jntadr = CACHE ? e.lc.jntadr : m.body_rec2[4 * b], jntnum = CACHE ? e.lc.jntnum : m.body_rec2[4 * b + 1];
			const int mid = m.nmocap > 0 ? m.body_mocapid[b] : -1;
			if (mid >= 0) {  // mocap body: pose straight from the (normalised) mocap fields, as mj_kinematics does
				ld3(p, f + L.mocap_pos + 3 * mid);
				ld4(q, f + L.mocap_quat + 4 * mid);
				normalize4(q);
			} else if (jntnum == 1 && (CACHE ? e.lc.jtype : m.jnt_type[jntadr]) == MJB_JNT_FREE) {
				const int qa = CACHE ? e.lc.qa : m.jnt_qposadr[jntadr];
				ld3(p, qpos + qa);
				ld4(q, qpos + qa + 3);
				normalize4(q);
				st4(qpos + qa + 3, q);
				double ax[3];
				if constexpr (CACHE) {
					ax[0] = e.lc.jaxis[0]; ax[1] = e.lc.jaxis[1]; ax[2] = e.lc.jaxis[2];
				} else {
					ldc3(ax, m.jnt_axis + 3 * jntadr);
				}
				st3(xanchor + 3 * jntadr, p);
				st3(xaxis + 3 * jntadr, ax);
			} else {
				if constexpr (CACHE) {
					for (int k = 0; k < 3; k++) p[k] = e.lc.bpos[k];
					for (int k = 0; k < 4; k++) q[k] = e.lc.bquat[k];
				} else {
					ldc3(p, m.body_pos + 3 * b);
					ldc4(q, m.body_quat + 4 * b);
				}
				for (int j = jntadr; j < jntadr + jntnum; j++) {
					const bool hit = CACHE && j == jntadr;  // the body's first joint sits in the lane's registers
					const int qa = hit ? e.lc.qa : m.jnt_qposadr[j], jt = hit ? e.lc.jtype : m.jnt_type[j];
					const double q0 = hit ? e.lc.q0 : m.qpos0[qa];
					double jaxis[3], jpos[3], ax[3], an[3];
					if (hit) {
						for (int k = 0; k < 3; k++) { jaxis[k] = e.lc.jaxis[k]; jpos[k] = e.lc.jpos[k]; }
					} else {
						ldc3(jaxis, m.jnt_axis + 3 * j);
						ldc3(jpos, m.jnt_pos + 3 * j);
					}
					rotvec_quat(ax, jaxis, q);
					rotvec_quat(an, jpos, q);
					an[0] += p[0]; an[1] += p[1]; an[2] += p[2];
					st3(xaxis + 3 * j, ax);
					st3(xanchor + 3 * j, an);
					if (jt == MJB_JNT_SLIDE) {
						const double sl = qpos[qa] - q0;
						p[0] += ax[0] * sl; p[1] += ax[1] * sl; p[2] += ax[2] * sl;
					} else {
						double ql[4], v[3];
						if (jt == MJB_JNT_BALL) {
							ld4(ql, qpos + qa);
							normalize4(ql);
							st4(qpos + qa, ql);
						} else {
							axis_angle_quat(ql, jaxis, qpos[qa] - q0);
						}
						qmul(q, q, ql);
						rotvec_quat(v, jpos, q);
						p[0] = an[0] - v[0]; p[1] = an[1] - v[1]; p[2] = an[2] - v[2];
					}
				}
			}
		}
		if constexpr (REG) {
			for (int k = 0; k < 3; k++) kp[k] = p[k];
			for (int k = 0; k < 4; k++) kq[k] = q[k];
		} else {
			st3(loc + 7 * b, p);
			st4(loc + 7 * b + 3, q);
		}
	}
	if constexpr (!REG) gsync<G>();
#if !defined(MJB_PROFILE_SUB) && !defined(MJB_PROFILE_NWT) && !defined(MJB_PROFILE_COL) && !defined(MJB_PROFILE_MK) && !defined(MJB_PROFILE_SM)  // (slots 20 - 23 carry the sub-stages of the Newton iteration's gradient step / of collision / the PGS tail in those builds)
	PROF(20);
#endif

	if constexpr (SCAN) {
		// Phase B (nbody <= G) -- pointer jumping over the tree: every body composes its pose with the pose of its
		// ancestor at distance 1, 2, 4, ... (host table body_anc), so a chain of depth d takes ceil(log2 d) rounds
		// instead of d.  Pose composition is associative; only the rounding differs from the serial walk.
		const bool act = lane < m.nbody;
		const int b = act ? lane : 0;
		double p[3], q[4];
		if constexpr (REG) {
			for (int k = 0; k < 3; k++) p[k] = kp[k];
			for (int k = 0; k < 4; k++) q[k] = kq[k];
		} else {
			ld3(p, loc + 7 * b);
			ld4(q, loc + 7 * b + 3);
		}
		if constexpr (CACHE && G == 16) {
			// one body per lane of a 16-lane env group: the ancestor's pose comes straight out of ITS lane's registers
			// (ds_bpermute: no LDS storage, no sync), the ancestor indices out of this lane's (LaneConst::anc)
			static_for<5>([&](auto rc) {
				constexpr int r = decltype(rc)::value;
				if (r < m.kin_rounds) {
					MJB_KEEP_BRANCH();
					const int A = act ? e.lc.anc[r] : 0;
					const int addr = (int)(((threadIdx.x & 48u) | (unsigned int)A) << 2);
					double pa[3], qa[4];
#pragma unroll
					for (int k = 0; k < 3; k++)
						pa[k] = __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(p[k])), __builtin_amdgcn_ds_bpermute(addr, __double2loint(p[k])));
#pragma unroll
					for (int k = 0; k < 4; k++)
						qa[k] = __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(q[k])), __builtin_amdgcn_ds_bpermute(addr, __double2loint(q[k])));
					if (A) {
						double Ma[9], v[3];
						quat2mat_nocheck(Ma, qa);
						matvec3(v, Ma, p);
						p[0] = pa[0] + v[0]; p[1] = pa[1] + v[1]; p[2] = pa[2] + v[2];
						qmul(q, qa, q);
					}
				}
			});
		} else {
		int A = b ? m.body_anc[b] : 0;
#pragma nounroll
		for (int r = 0; r < m.kin_rounds; r++) {
			if (r && act) {
				st3(loc + 7 * b, p);
				st4(loc + 7 * b + 3, q);
			}
			gsync<G>();
			const int An = A ? m.body_anc[(r + 1) * m.nbody + b] : 0;
			if (A) {
				double pa[3], qa[4], Ma[9], v[3];
				ld3(pa, loc + 7 * A);
				ld4(qa, loc + 7 * A + 3);
				quat2mat_nocheck(Ma, qa);
				matvec3(v, Ma, p);
				p[0] = pa[0] + v[0]; p[1] = pa[1] + v[1]; p[2] = pa[2] + v[2];
				qmul(q, qa, q);
			}
			A = An;
			gsync<G>();
		}
		}
		if constexpr (REG) {
			if (act) st3(xpos + 3 * b, p);  // (phase C stores the normalised xquat)
			for (int k = 0; k < 3; k++) kp[k] = p[k];
			for (int k = 0; k < 4; k++) kq[k] = q[k];
		} else {
			if (act) {
				st3(xpos + 3 * b, p);
				st4(xquat + 4 * b, q);
			}
			gsync<G>();
		}
	} else {
		// Phase B -- thin serial chain, replicated in every lane with the running parent pose in registers:
		// xquat_i = xquat_p * lq_i,  xpos_i = xpos_p + R_p lp_i.  Only xpos / xquat are stored (7 doubles per body);
		// phase C re-normalises the quaternions and derives xmat.
		{
			double cp[3] = { 0, 0, 0 }, cq[4] = { 1, 0, 0, 0 }, cM[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
			if (lane == 0) {
				st3(xpos, cp);
				st4(xquat, cq);
			}
			double lp[3], lq[4];
			ld3(lp, loc + 7);
			ld4(lq, loc + 10);
#pragma nounroll
			for (int i = 1; i < m.nbody; i++) {
				const int pid = m.body_rec[4 * i];
				if (pid != i - 1) {
					ld3(cp, xpos + 3 * pid);
					ld4(cq, xquat + 4 * pid);
					quat2mat_nocheck(cM, cq);
				}
				double v[3];
				matvec3(v, cM, lp);
				cp[0] += v[0]; cp[1] += v[1]; cp[2] += v[2];
				qmul(cq, cq, lq);
				// prefetch the next body's local pose before this body's stores enter the LDS queue
				const int nx = (i + 1 < m.nbody) ? i + 1 : i;
				ld3(lp, loc + 7 * nx);
				ld4(lq, loc + 7 * nx + 3);
				if (lane == 0) {
					st3(xpos + 3 * i, cp);
					st4(xquat + 4 * i, cq);
				}
				quat2mat_nocheck(cM, cq);
			}
		}
		gsync<G>();
	}

#if !defined(MJB_PROFILE_SUB) && !defined(MJB_PROFILE_NWT) && !defined(MJB_PROFILE_COL) && !defined(MJB_PROFILE_MK) && !defined(MJB_PROFILE_SM)  // (slots 21 / 22 carry the PGS sweep / row counts in the sub-stage build)
	PROF(21);
#endif
	// Phase C -- one body per lane: normalise xquat, final xmat, inertial frame
	for (int b = lane; b < m.nbody; b += G) {
		double q[4], M[9], p[3];
		if constexpr (REG) {
			for (int k = 0; k < 3; k++) p[k] = kp[k];
			for (int k = 0; k < 4; k++) q[k] = kq[k];
		} else {
			ld4(q, xquat + 4 * b);
			ld3(p, xpos + 3 * b);
		}
		normalize4(q);
		quat2mat(M, q);
		st4(xquat + 4 * b, q);
		st9(xmat + 9 * b, M);
		double *oip = f + L.xipos + 3 * b, *oim = f + L.ximat + 9 * b;
		if (b == 0 || (CACHE ? e.lc.simple : m.body_rec2[4 * b + 2])) {
			st3(oip, p);
			st9(oim, M);
		} else {
			double ip[3], iq[4], v[3], r[9];
			if constexpr (CACHE) {
				for (int k = 0; k < 3; k++) ip[k] = e.lc.ipos[k];
				for (int k = 0; k < 4; k++) iq[k] = e.lc.iquat[k];
			} else {
				ldc3(ip, m.body_ipos + 3 * b);
				ldc4(iq, m.body_iquat + 4 * b);
			}
			matvec3(v, M, ip);
			v[0] += p[0]; v[1] += p[1]; v[2] += p[2];
			qmul(iq, q, iq);
			quat2mat(r, iq);
			st3(oip, v);
			st9(oim, r);
		}
	}
	gsync<G>();

#if !defined(MJB_PROFILE_SUB) && !defined(MJB_PROFILE_NWT) && !defined(MJB_PROFILE_COL) && !defined(MJB_PROFILE_MK) && !defined(MJB_PROFILE_SM)
	PROF(22);
#endif
	// Phase D -- joints (anchor / axis to the world frame through the PARENT body's frame), geoms, sites
	const int nitem = m.njnt + m.ngeom + m.nsite;
	if (CACHE && e.lc.k_kind >= 0) {  // (the lane's item record sits in registers: no table walk)
		if (e.lc.k_kind == 0 && !e.lc.k_same) {
			const int j = e.lc.k_id, pid = e.lc.k_body;
			double M[9], pp[3], an[3], ax[3], v[3];
			ld9(M, xmat + 9 * pid);
			ld3(pp, xpos + 3 * pid);
			ld3(an, xanchor + 3 * j);
			ld3(ax, xaxis + 3 * j);
			matvec3(v, M, an);
			v[0] += pp[0]; v[1] += pp[1]; v[2] += pp[2];
			st3(xanchor + 3 * j, v);
			matvec3(v, M, ax);
			st3(xaxis + 3 * j, v);
		} else if (e.lc.k_kind == 1 || e.lc.k_kind == 2) {
			const bool isg = e.lc.k_kind == 1;
			const int id = e.lc.k_id, b = e.lc.k_body;
			double pos[3] = { e.lc.k_pos[0], e.lc.k_pos[1], e.lc.k_pos[2] }, quat[4] = { e.lc.k_quat[0], e.lc.k_quat[1], e.lc.k_quat[2], e.lc.k_quat[3] };
			double *opos = f + (isg ? L.geom_xpos + 3 * id : L.site_xpos + 3 * id);
			double *omat = f + (isg ? L.geom_xmat + 9 * id : L.site_xmat + 9 * id);
			double *oquat = f + L.site_xquat + (isg ? 0 : 4 * id);
			local2global(xpos + 3 * b, xquat + 4 * b, xmat + 9 * b, opos, omat, pos, quat, e.lc.k_same, oquat, isg ? 0 : 1);
		}
	} else
	for (int it = lane; it < nitem; it += G) {
		if (it < m.njnt) {
			const int j = it, b = m.jnt_bodyid[j];
			if (m.jnt_type[j] != MJB_JNT_FREE) {
				const int pid = m.body_rec[4 * b];
				double M[9], pp[3], an[3], ax[3], v[3];
				ld9(M, xmat + 9 * pid);
				ld3(pp, xpos + 3 * pid);
				ld3(an, xanchor + 3 * j);
				ld3(ax, xaxis + 3 * j);
				matvec3(v, M, an);
				v[0] += pp[0]; v[1] += pp[1]; v[2] += pp[2];
				st3(xanchor + 3 * j, v);
				matvec3(v, M, ax);
				st3(xaxis + 3 * j, v);
			}
		} else {
			// geoms and sites take ONE path (lanes of both kinds run it together: a branch per kind is a second serial pass of
			// dependent table loads, ~2.5 k cycles per step)
			const bool isg = it < m.njnt + m.ngeom;
			const int id = isg ? it - m.njnt : it - m.njnt - m.ngeom;
			const int b = isg ? m.geom_bodyid[id] : m.site_bodyid[id];
			const mjb_cdptr ppos = isg ? m.geom_pos + 3 * id : m.site_pos + 3 * id;
			const mjb_cdptr pquat = isg ? m.geom_quat + 4 * id : m.site_quat + 4 * id;
			const int same = isg ? m.geom_sameframe[id] : m.site_sameframe[id];
			double pos[3], quat[4];
			ldc3(pos, ppos);
			ldc4(quat, pquat);
			double *opos = f + (isg ? L.geom_xpos + 3 * id : L.site_xpos + 3 * id);
			double *omat = f + (isg ? L.geom_xmat + 9 * id : L.site_xmat + 9 * id);
			double *oquat = f + L.site_xquat + (isg ? 0 : 4 * id);
			local2global(xpos + 3 * b, xquat + 4 * b, xmat + 9 * b, opos, omat, pos, quat, same, oquat, isg ? 0 : 1);
		}
	}
	gsync<G>();
#if !defined(MJB_PROFILE_SUB) && !defined(MJB_PROFILE_NWT) && !defined(MJB_PROFILE_COL) && !defined(MJB_PROFILE_MK) && !defined(MJB_PROFILE_SM)
	PROF(23);
#endif
}

// ------------------------------------------------------------------------------------------------
// A1  comPos: subtree centres of mass, com-based body inertias (cinert) and motion dofs (cdof)
// ------------------------------------------------------------------------------------------------
// bit i of a 64-bit mask stored as two ints (host-built ancestor-dof / subtree-body masks; nv, nbody <= 64)
DEVI bool maskbit(unsigned int lo, unsigned int hi, int i) { return ((i < 32 ? lo >> i : hi >> (i - 32)) & 1u) != 0; }

// ---- leaf-to-root sums (subtree com, composite inertia, RNE's backward pass) as ONE product with the model's 0/1 subtree matrix on the
// matrix cores:  out[a][c] = sum_b S[a][b] x[b][c],  S[a][b] = 1 when body b belongs to body a's subtree  (host table sub_S, laid out
// as the A operands of v_mfma_f64_16x16x4_f64: [row tile][k block][lane] = S[16 t + (l & 15)][4 k + (l >> 4)]; the blocks left of a
// tile's diagonal are zero -- children carry larger ids -- and are skipped).  In place in buf [nbody][N]: all of x is in registers
// before the first result is stored.  The serial walks these stages ran before cost a round trip per BODY (crb: 25 dependent
// read-add-write steps = 7.5 k cycles on the hand model; rne: 24 masked 6-vector reads per lane = 9.1 k); the products of the 0/1
// entries are exact, the order of the additions is the matrix core's (k ascending) instead of the tree's.
typedef double mjb_sd4 __attribute__((ext_vector_type(4)));
typedef int mjb_i4 __attribute__((ext_vector_type(4)));
// (the A operands are fetched by the CALLER ahead of the phase before the product -- a trip to L2 that otherwise sits in front of the
//  matrix instructions:  [0, 4 NT) = tile 0's blocks, [8, 12) = tile 1's blocks 4 - 7)
struct SubtreeA {
	double a[12];
};
DEVI SubtreeA subtree_fetch(CModel m, int lane)
{
	SubtreeA r;
	const double MJB_AS4 *S = m.sub_S + lane;
	if (m.sub_nt == 1) {
		MJB_KEEP_BRANCH();
#pragma unroll
		for (int k = 0; k < 4; k++) r.a[k] = S[k * 64];
#pragma unroll
		for (int k = 4; k < 12; k++) r.a[k] = 0;
	} else {
		MJB_KEEP_BRANCH();
#pragma unroll
		for (int k = 0; k < 8; k++) r.a[k] = S[k * 64];
#pragma unroll
		for (int k = 4; k < 8; k++) r.a[4 + k] = S[(8 + k) * 64];
	}
	return r;
}
template <int N, int NT, bool KEEP0> DEVI void subtree_sum_tiles(CModel m, double *buf, int lane, const SubtreeA &A)
{
	static_assert(N <= 16, "one column tile");
	constexpr int KB = 4 * NT;
	const int li = lane & 15, lk = lane >> 4;
	double B[KB];
#pragma unroll
	for (int k = 0; k < KB; k++) {
		const int b = 4 * k + lk;
		const bool on = li < N && b < m.nbody;
		const double v = buf[on ? N * b + li : 0];
		B[k] = on ? v : 0.0;
	}
	mjb_sd4 acc[NT];
#pragma unroll
	for (int t = 0; t < NT; t++) acc[t] = mjb_sd4{ 0, 0, 0, 0 };
#pragma unroll
	for (int k = 0; k < KB; k++)
#pragma unroll
		for (int t = 0; t < NT; t++)
			if (k >= 4 * t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.a[t == 0 ? k : 4 + k], B[k], acc[t], 0, 0, 0);
#pragma unroll
	for (int t = 0; t < NT; t++)
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const int a = 16 * t + lk + 4 * q;
			if (li < N && a < m.nbody && !(KEEP0 && a == 0)) buf[N * a + li] = acc[t][q];
		}
}
template <int N, bool KEEP0> DEVI void subtree_sum(CModel m, double *buf, int lane, const SubtreeA &A)
{
	if (m.sub_nt == 1) {
		MJB_KEEP_BRANCH();
		subtree_sum_tiles<N, 1, KEEP0>(m, buf, lane, A);
	} else {
		MJB_KEEP_BRANCH();
		subtree_sum_tiles<N, 2, KEEP0>(m, buf, lane, A);
	}
}
// the k-th byte of a 16-byte list held in two words, as the loop shifts it down
DEVI int list_next(unsigned long long &lo, unsigned long long &hi)
{
	const int v = (int)(lo & 0xFFull);
	lo = (lo >> 8) | (hi << 56);
	hi = (hi >> 8) | (0xFFull << 56);
	return v;
}

template <int G, bool OBL> STAGE void com_pos(CModel m, CLayout L, const Env &e)
{
	double *f = e.f;
	double *sc = f + L.subtree_com, *xipos = f + L.xipos;
	const int lane = e.lane;
	SPROF_BEGIN();
	// lane = body: mass-weighted sum over the bodies of its subtree (host-built mask) -- no walk up the tree, every
	// lane reads the same xipos / mass sequence (LDS broadcast + scalar loads) and keeps what its mask selects
	if constexpr (OBL && G == 16) {
		// one body per lane of a 16-lane group: every body's inertial position and mass come out of ITS lane's registers by DPP row
		// broadcasts (the same fma in the same order as the loop below: no scalar load per body, no LDS loop)
		const int b = lane < m.nbody ? (int)lane : 0;
		double xo[3];
		ld3(xo, xipos + 3 * b);
		const double mo = lane < m.nbody ? e.lc.mass : 0.0;
		const unsigned int lo = e.lc.smlo;
		double s0 = 0, s1 = 0, s2 = 0;
		static_for<16>([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			if (i < m.nbody) {
				MJB_KEEP_BRANCH();
				const double mi = ((lo >> i) & 1u) ? row_bcast16<i>(mo) : 0.0;
				s0 += row_bcast16<i>(xo[0]) * mi;
				s1 += row_bcast16<i>(xo[1]) * mi;
				s2 += row_bcast16<i>(xo[2]) * mi;
			}
		});
		if (lane < m.nbody) {
			const double stm = e.lc.stmass;
			if (stm < MJB_MINVAL) {
				sc[3 * b] = xo[0]; sc[3 * b + 1] = xo[1]; sc[3 * b + 2] = xo[2];
			} else {
				const double inv = 1.0 / fmax(MJB_MINVAL, stm);
				sc[3 * b] = s0 * inv; sc[3 * b + 1] = s1 * inv; sc[3 * b + 2] = s2 * inv;
			}
		}
	} else if (G == 64 && !OBL && m.sub_nt > 0) {
		MJB_KEEP_BRANCH();
		const bool act = lane < m.nbody;
		const int b = act ? (int)lane : 0;
		const SubtreeA SA = subtree_fetch(m, lane);
		const double mass = MP_BODY_MASS(m, e, b), stm = MP_SUBTREEMASS(m, e, b);
		double xo[3], r[3];
		ld3(xo, xipos + 3 * b);
		if (act) {
			for (int k = 0; k < 3; k++) r[k] = xo[k] * mass;
			st3(sc + 3 * b, r);
		}
		gsync<G>();
		subtree_sum<3, false>(m, sc, lane, SA);
		gsync<G>();
		ld3(r, sc + 3 * b);
		if (act) {
			if (stm < MJB_MINVAL) {
				st3(sc + 3 * b, xo);
			} else {
				const double inv = 1.0 / fmax(MJB_MINVAL, stm);
				sc[3 * b] = r[0] * inv; sc[3 * b + 1] = r[1] * inv; sc[3 * b + 2] = r[2] * inv;
			}
		}
	} else
	for (int b = lane; b < m.nbody; b += G) {
		const unsigned int lo = OBL ? e.lc.smlo : (unsigned int)m.body_submask[2 * b], hi = OBL ? e.lc.smhi : (unsigned int)m.body_submask[2 * b + 1];
		double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll 4
		for (int i = 0; i < m.nbody; i++) {
			const double mi = maskbit(lo, hi, i) ? (OBL ? m.body_mass[i] : MP_BODY_MASS(m, e, i)) : 0.0;
			s0 += xipos[3 * i] * mi;
			s1 += xipos[3 * i + 1] * mi;
			s2 += xipos[3 * i + 2] * mi;
		}
		const double stm = OBL ? m.body_subtreemass[b] : MP_SUBTREEMASS(m, e, b);
		if (stm < MJB_MINVAL) {
			sc[3 * b] = xipos[3 * b]; sc[3 * b + 1] = xipos[3 * b + 1]; sc[3 * b + 2] = xipos[3 * b + 2];
		} else {
			const double inv = 1.0 / fmax(MJB_MINVAL, stm);
			sc[3 * b] = s0 * inv; sc[3 * b + 1] = s1 * inv; sc[3 * b + 2] = s2 * inv;
		}
	}
	gsync<G>();
	SPROF(20);
	// cinert: one body per lane
	for (int b = lane; b < m.nbody; b += G) {
		double r[10];
		if (b == 0) {
			for (int k = 0; k < 10; k++) r[k] = 0;
		} else {
			double ip[3], root[3], off[3], im[9], inert[3];
			ld3(ip, xipos + 3 * b);
			ld3(root, sc + 3 * (OBL ? e.lc.rootid : m.body_rootid[b]));
			off[0] = ip[0] - root[0]; off[1] = ip[1] - root[1]; off[2] = ip[2] - root[2];
			ld9(im, f + L.ximat + 9 * b);
			if constexpr (OBL) {
				inert[0] = e.lc.inertia[0]; inert[1] = e.lc.inertia[1]; inert[2] = e.lc.inertia[2];
			} else {
				for (int k = 0; k < 3; k++) inert[k] = MP_INERTIA(m, e, b, k);
			}
			inert_com(r, inert, im, off, OBL ? e.lc.mass : MP_BODY_MASS(m, e, b));
		}
		double *o = f + L.cinert + 10 * b;
		for (int k = 0; k < 10; k++) o[k] = r[k];
	}
	// cdof: one joint per lane
	for (int j = lane; j < m.njnt; j += G) {
		mjb_i4 jr = mjb_i4{ 0, 0, 0, 0 };
		if constexpr (!OBL) jr = reinterpret_cast<const mjb_i4 MJB_AS4 *>(m.jnt_rec)[j];
		const int bi = OBL ? e.lc.j_body : jr[0], jt = OBL ? e.lc.j_type : jr[1];
		double *cd = f + L.cdof + 6 * (OBL ? e.lc.j_da : jr[2]);
		double root[3], an[3], off[3];
		ld3(root, sc + 3 * (OBL ? e.lc.j_root : jr[3]));
		ld3(an, f + L.xanchor + 3 * j);
		off[0] = root[0] - an[0]; off[1] = root[1] - an[1]; off[2] = root[2] - an[2];
		if (jt == MJB_JNT_FREE || jt == MJB_JNT_BALL) {
			if (jt == MJB_JNT_FREE) {
				for (int k = 0; k < 18; k++) cd[k] = 0;
				cd[3] = 1; cd[10] = 1; cd[17] = 1;
				cd += 18;
			}
			double M[9];
			ld9(M, f + L.xmat + 9 * bi);
			for (int k = 0; k < 3; k++) {
				double ax[3] = { M[k], M[k + 3], M[k + 6] }, cr[3];
				cross3(cr, ax, off);
				cd[6 * k + 0] = ax[0]; cd[6 * k + 1] = ax[1]; cd[6 * k + 2] = ax[2];
				cd[6 * k + 3] = cr[0]; cd[6 * k + 4] = cr[1]; cd[6 * k + 5] = cr[2];
			}
		} else {
			double ax[3];
			ld3(ax, f + L.xaxis + 3 * j);
			if (jt == MJB_JNT_SLIDE) {
				cd[0] = cd[1] = cd[2] = 0;
				st3(cd + 3, ax);
			} else {
				double cr[3];
				cross3(cr, ax, off);
				st3(cd, ax);
				st3(cd + 3, cr);
			}
		}
	}
	// mj_tendon (fixed tendons): length = sum coef * qpos[joint]
	for (int t = lane; t < m.ntendon; t += G) {
		double len = 0;
		for (int w = m.tendon_adr[t]; w < m.tendon_adr[t] + m.tendon_num[t]; w++)
			len += m.wrap_prm[w] * f[L.qpos + m.jnt_qposadr[m.wrap_objid[w]]];
		f[L.ten_length + t] = len;
	}
	gsync<G>();
	SPROF(21);
}

// ------------------------------------------------------------------------------------------------
// A2  crb: composite inertias and the sparse joint-space inertia qM
// ------------------------------------------------------------------------------------------------
template <int G, bool OBL> STAGE void crb(CModel m, CLayout L, const Env &e)
{
	double *f = e.f;
	double *crbv = f + L.crb, *cinert = f + L.cinert, *buf = f + L.crbbuf;
	const int lane = e.lane;
	SPROF_BEGIN();
	if (G == 64 && !OBL && m.sub_nt > 0) {
		MJB_KEEP_BRANCH();
		// (nothing is added to the world body: its row stays what it was)
		const SubtreeA SA = subtree_fetch(m, lane);
		for (int k = lane; k < 10 * m.nbody; k += G) crbv[k] = cinert[k];
		gsync<G>();
		subtree_sum<10, true>(m, crbv, lane, SA);
		gsync<G>();
	} else
	for (int c = lane; c < 10; c += G) {
		for (int i = 0; i < m.nbody; i++) crbv[10 * i + c] = cinert[10 * i + c];
		for (int i = m.nbody - 1; i > 0; i--) {
			const int p = m.body_parentid[i];
			if (p > 0) crbv[10 * p + c] += crbv[10 * i + c];
		}
	}
	if (!(G == 64 && !OBL && m.sub_nt > 0)) gsync<G>();
	SPROF(22);
	// buf_i = crb[body(i)] * cdof_i, one dof per lane
	for (int i = lane; i < m.nv; i += G) {
		double I[10], v[6], r[6];
		ld10(I, crbv + 10 * (OBL ? e.lc.d_body : m.dof_bodyid[i]));
		ld6(v, f + L.cdof + 6 * i);
		mul_inert_vec(r, I, v);
		st6(buf + 6 * i, r);
	}
	gsync<G>();
	SPROF(23);
	// qM entries, one per lane: M(i,j) = cdof_j . buf_i  (+ armature on the diagonal)
	if (OBL && e.lc.q_row[0] != -2) {  // (the lane's entries -- row / column dof, armature, h * damping -- sit in registers)
#pragma unroll
		for (int q = 0; q < 3; q++) {
			const int i = e.lc.q_row[q], j = e.lc.q_col[q], en = lane + G * q;
			if (i < 0) continue;
			double a[6], b[6];
			ld6(a, f + L.cdof + 6 * j);
			ld6(b, buf + 6 * i);
			double v = e.lc.q_arm[q];
			v += dot6r(a, b);
			f[L.qM + en] = v;
			if (m.eulerdamp) f[L.MhB + en] = (i == j) ? v + e.lc.q_hd[q] : v;
		}
	} else
	for (int en = lane; en < m.nM; en += G) {
		const int i = m.M_rowdof[en], j = m.M_coldof[en];
		double a[6], b[6];
		ld6(a, f + L.cdof + 6 * j);
		ld6(b, buf + 6 * i);
		double v = (i == j) ? m.dof_armature[i] : 0.0;
		v += dot6r(a, b);
		f[L.qM + en] = v;
		if (m.eulerdamp) f[L.MhB + en] = (i == j) ? v + m.timestep[0] * m.dof_damping_int[i] : v;
	}
	gsync<G>();
	SPROF(24);
}

// ------------------------------------------------------------------------------------------------
// A3  sparse L'DL factorisation in qM layout and the matching triangular solves
// ------------------------------------------------------------------------------------------------
// Two matrices with the same sparsity are factorised in the same rounds (qM -> qLD for mj_factorM and,
// when Euler's implicit damping is on, MhB = M + h diag(B) -> qH), so the second factorisation rides on
// the first one's latency chain.  Per pivot k (descending) ONE round applies every row update
// LD[dst] -= LD[srcA] / LD[kk] * LD[srcB] listed in the host-built micro-program (one op per lane), a
// second round scales row k.
template <int G>
STAGE void factor2(CModel m, const Env &e, const double *M, double *LD, double *di, const double *M2, double *LD2,
                   double *di2, bool dual)
{
	const int lane = e.lane;
	for (int en = lane; en < m.nM; en += G) {
		LD[en] = M[en];
		if (dual) LD2[en] = M2[en];
	}
	gsync<G>();
	{
		int k = m.nv - 1;
		int na = k >= 0 ? m.dof_rec[4 * k + 1] : 0, kk = k >= 0 ? m.dof_rec[4 * k] : 0;
		int beg = k >= 0 ? m.fac_beg[k] : 0, end = k >= 0 ? m.fac_beg[k + 1] : 0;
		int o0 = 0, o1 = 0, o2 = 0;  // this lane's first micro-op of the pivot
		if (beg + lane < end) {
			o0 = m.fac_ops[4 * (beg + lane)];
			o1 = m.fac_ops[4 * (beg + lane) + 1];
			o2 = m.fac_ops[4 * (beg + lane) + 2];
		}
#pragma nounroll
		for (; k >= 0; k--) {
			const int na_c = na, kk_c = kk, beg_c = beg, end_c = end, dst = o0, sa = o1, sb = o2;
			if (k > 0) {
				na = m.dof_rec[4 * (k - 1) + 1];
				kk = m.dof_rec[4 * (k - 1)];
				beg = m.fac_beg[k - 1];
				end = beg_c;  // fac_beg[k]
				if (beg + lane < end) {
					o0 = m.fac_ops[4 * (beg + lane)];
					o1 = m.fac_ops[4 * (beg + lane) + 1];
					o2 = m.fac_ops[4 * (beg + lane) + 2];
				}
			}
			if (na_c <= 0) continue;
			const double dkk = LD[kk_c];
			const double dkk2 = dual ? LD2[kk_c] : 1.0;
			if (beg_c + lane < end_c) {
				LD[dst] -= LD[sa] / dkk * LD[sb];
				if (dual) LD2[dst] -= LD2[sa] / dkk2 * LD2[sb];
			}
			for (int t = beg_c + lane + G; t < end_c; t += G) {
				const int d2 = m.fac_ops[4 * t], a2 = m.fac_ops[4 * t + 1], b2 = m.fac_ops[4 * t + 2];
				LD[d2] -= LD[a2] / dkk * LD[b2];
				if (dual) LD2[d2] -= LD2[a2] / dkk2 * LD2[b2];
			}
			gsync<G>();
		}
	}
	// Row scaling is deferred: pivot k's updates only need the UNSCALED row k and its diagonal, and no later pivot
	// reads row k, so  L(k,a) = U(k,a) / D(k)  is applied to all entries at once here (one round instead of nv).
	for (int en = lane; en < m.nM; en += G) {
		const int row = m.M_rowdof[en];
		if (m.M_coldof[en] != row) {
			const int kk = m.dof_rec[4 * row];
			LD[en] = LD[en] / LD[kk];
			if (dual) LD2[en] = LD2[en] / LD2[kk];
		} else {
			di[row] = 1.0 / LD[en];
			if (dual) di2[row] = 1.0 / LD2[en];
		}
	}
	gsync<G>();
}

// ---- the same factorisation scheduled by LEVELS of the elimination tree (host program flv_hdr / flv_rec): one round per level of
// the dof tree instead of one per pivot -- 7 rounds for the hand model's 30 dofs.  One lane per destination entry applies the
// level's contributions to it, in descending pivot order; the rounding differs from the pivot-by-pivot order only where pivots of
// different levels meet in an entry (the entries of common ancestors).
DEVI double fast_rcp(double x)
{
	double r = __builtin_amdgcn_rcp(x);
	r = fma(fma(-x, r, 1.0), r, r);
	r = fma(fma(-x, r, 1.0), r, r);
	return r;
}

template <int G>
STAGE void factor_levels(CModel m, const Env &e, const double *M, double *LD, double *di, const double *M2, double *LD2,
                         double *di2, bool dual)
{
	static_assert(G == 64, "a level's items sit at one place per lane of the wavefront");
	const int lane = e.lane;
	// only LDS crosses lanes in this stage: its syncs do not wait for the look-ahead fetches (a plain wavefront fence waits vmcnt(0))
	auto lsync = [&]() {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
	};
	constexpr int LVSZ = 3 * 64;  // a level's words, in 16-byte units
	const mjb_i4 MJB_AS4 *recs = reinterpret_cast<const mjb_i4 MJB_AS4 *>(m.flv_rec);
	struct Lv {
		mjb_i4 w[3];
	};
	auto fetch = [&](int lev) {
		const mjb_i4 MJB_AS4 *p = recs + (size_t)lev * LVSZ + lane;
		Lv r;
#pragma unroll
		for (int q = 0; q < 3; q++) r.w[q] = p[64 * q];
		return r;
	};
	const mjb_i4 hv = reinterpret_cast<const mjb_i4 MJB_AS4 *>(m.flv_hdr)[lane < m.flv_n ? (int)lane : 0];
	Lv P0 = fetch(0), P1 = fetch(1);
	// the inverse diagonals are kept up to date as the levels go: an update  LD[dst] -= LD[a] * (1 / D_k) * LD[b]  costs two
	// multiplications (a division per contribution -- ~15 instructions of the quarter-rate kind -- was 3/4 of this stage), and the
	// lane that finishes a diagonal entry writes its reciprocal: a pivot's diagonal is final once the level below it has run
	// (the entries' row words for four trips of 64 at once -- one trip to the table for the first and the last round)
	int ent[4];
#pragma unroll
	for (int q = 0; q < 4; q++) ent[q] = m.flv_ent[lane + 64 * q];
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const int en = lane + 64 * q;
		if (64 * q < m.nM) {  // (wave-uniform)
			MJB_KEEP_BRANCH();
			if (en < m.nM) {
				const double v = M[en], v2 = dual ? M2[en] : 1.0;
				LD[en] = v;
				if (dual) LD2[en] = v2;
				if (ent[q] & 0x10000) {
					di[ent[q] & 0xFFFF] = fast_rcp(v);
					if (dual) di2[ent[q] & 0xFFFF] = fast_rcp(v2);
				}
			}
		}
	}
	for (int en = lane + 256; en < m.nM; en += G) {
		const double v = M[en], v2 = dual ? M2[en] : 1.0;
		LD[en] = v;
		if (dual) LD2[en] = v2;
		const int w = m.flv_ent[en];
		if (w & 0x10000) {
			di[w & 0xFFFF] = fast_rcp(v);
			if (dual) di2[w & 0xFFFF] = fast_rcp(v2);
		}
	}
	// one contribution per lane; an entry's contributions sit in consecutive lanes of a 16-lane row and its first lane (the owner)
	// subtracts them in order -- its own, then the neighbours' by DPP row shifts
	auto apply = [&](const mjb_i4 w, int tmax) {
		const int dst = w[0] & 0xFFFF, drow = (w[0] >> 16) - 1, fl = w[1], sa = w[2] & 0xFFFF, sb = (unsigned int)w[2] >> 16, k = w[3];
		const bool valid = (fl & 1) != 0, owner = (fl & 2) != 0;
		const int cnt = fl >> 8;
		double acc = LD[dst], acc2 = dual ? LD2[dst] : 0.0;
		const double a = LD[sa], b = LD[sb], d = di[k];
		const double a2 = dual ? LD2[sa] : 0.0, b2 = dual ? LD2[sb] : 0.0, d2 = dual ? di2[k] : 0.0;
		double u = a * d * b, u2 = a2 * d2 * b2;
		u = valid ? u : 0.0;
		u2 = valid ? u2 : 0.0;
		acc -= u;
		acc2 -= u2;
		static_for<5>([&](auto jc) {
			constexpr int j = decltype(jc)::value + 1;
			if (j < tmax) {  // (wave-uniform)
				MJB_KEEP_BRANCH();
				const double v = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(u), 0x100 + j, 0xF, 0xF, true),
				                                  __builtin_amdgcn_update_dpp(0, __double2loint(u), 0x100 + j, 0xF, 0xF, true));
				acc -= j < cnt ? v : 0.0;
				if (dual) {
					const double v2 = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(u2), 0x100 + j, 0xF, 0xF, true),
					                                   __builtin_amdgcn_update_dpp(0, __double2loint(u2), 0x100 + j, 0xF, 0xF, true));
					acc2 -= j < cnt ? v2 : 0.0;
				}
			}
		});
		if (owner) {
			LD[dst] = acc;
			if (dual) LD2[dst] = acc2;
		}
		if (__ballot(owner && drow >= 0)) {
			MJB_KEEP_BRANCH();
			const double r = fast_rcp(acc), r2 = dual ? fast_rcp(acc2) : 0.0;
			if (owner && drow >= 0) {
				di[drow] = r;
				if (dual) di2[drow] = r2;
			}
		}
	};
	auto level = [&](int lev, const Lv &C) {
		const int ns = __builtin_amdgcn_readlane(hv[0], lev);
		apply(C.w[0], __builtin_amdgcn_readlane(hv[1], lev));
		if (ns > 1) {
			MJB_KEEP_BRANCH();
			apply(C.w[1], __builtin_amdgcn_readlane(hv[2], lev));
			if (ns > 2) {
				MJB_KEEP_BRANCH();
				apply(C.w[2], __builtin_amdgcn_readlane(hv[3], lev));
			}
		}
		lsync();
	};
	lsync();
	// (two levels a trip, each refilling its OWN set of registers for the level two ahead: rotating one set into the other by
	//  copies made every trip wait for the loads it had just issued)
#pragma nounroll
	for (int lev = 0; lev < m.flv_n; lev += 2) {
		level(lev, P0);
		P0 = fetch(lev + 2);
		if (lev + 1 < m.flv_n) {
			MJB_KEEP_BRANCH();
			level(lev + 1, P1);
			P1 = fetch(lev + 3);
		}
	}
	// rows scaled at the end, as in factor2:  L(k, a) = U(k, a) / D(k)
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const int en = lane + 64 * q;
		if (64 * q < m.nM) {  // (wave-uniform)
			MJB_KEEP_BRANCH();
			if (en < m.nM && !(ent[q] & 0x10000)) {
				LD[en] = LD[en] * di[ent[q] & 0xFFFF];
				if (dual) LD2[en] = LD2[en] * di2[ent[q] & 0xFFFF];
			}
		}
	}
	for (int en = lane + 256; en < m.nM; en += G) {
		const int w = m.flv_ent[en];
		if (!(w & 0x10000)) {
			LD[en] = LD[en] * di[w & 0xFFFF];
			if (dual) LD2[en] = LD2[en] * di2[w & 0xFFFF];
		}
	}
	gsync<G>();
}

// sum over the 16 lanes of a DPP row; every lane of the row receives the total (butterfly: xor 1, xor 2,
// half-mirror, mirror)
DEVI double dpp_add(double v, const int lo2, const int hi2) { return v + __hiloint2double(hi2, lo2); }
#define MJB_DPP_STEP(v, ctrl)                                                                        \
	v = dpp_add(v, __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xF, 0xF, true),          \
	            __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xF, 0xF, true))
template <int W> DEVI double row_sum(double v)  // W = 8 or 16 participating lanes
{
	MJB_DPP_STEP(v, 0xB1);   // quad_perm [1,0,3,2]
	MJB_DPP_STEP(v, 0x4E);   // quad_perm [2,3,0,1]
	MJB_DPP_STEP(v, 0x141);  // row_half_mirror
	if (W == 16) MJB_DPP_STEP(v, 0x140);  // row_mirror
	return v;
}

// Two right-hand sides with two factors of the same sparsity are solved in the same rounds (dual == true):
// qacc_smooth = M^-1 f  and, when nothing can change the force in between (no constraint rows), Euler's
// (M + hB)^-1 f -- the second solve rides on the latency chain of the first.
template <int G>
STAGE void solve2(CModel m, const EnvLite &e, double *x, const double *LD, const double *diaginv, double *x2,
                  const double *LD2, const double *diaginv2, bool dual)
{
	const int lane = e.lane;
	constexpr int W = G < 16 ? G : 16;
	// x <- inv(L') x : column sweep, one ancestor per lane
#pragma nounroll
	for (int i = m.nv - 1; i >= 0; i--) {
		const int na = m.dof_rec[4 * i + 1], ii = m.dof_rec[4 * i];
		if (na <= 0) continue;
		const double xi = x[i];
		const double xi2 = dual ? x2[i] : 0.0;
		for (int a = lane; a < na; a += G) {
			const int j = m.M_coldof[ii + 1 + a];
			x[j] -= LD[ii + 1 + a] * xi;
			if (dual) x2[j] -= LD2[ii + 1 + a] * xi2;
		}
		gsync<G>();
	}
	for (int i = lane; i < m.nv; i += G) {
		x[i] *= diaginv[i];
		if (dual) x2[i] *= diaginv2[i];
	}
	gsync<G>();
	// x <- inv(L) x : row i needs its ancestors only; products one per lane (first min(G,16) lanes of the group),
	// summed with a DPP row butterfly
#pragma nounroll
	for (int i = 0; i < m.nv; i++) {
		const int na = m.dof_rec[4 * i + 1], ii = m.dof_rec[4 * i];
		if (na <= 0) continue;
		double part = 0, part2 = 0;
		if (lane < W)
			for (int a = lane; a < na; a += W) {
				const int j = m.M_coldof[ii + 1 + a];
				part += LD[ii + 1 + a] * x[j];
				if (dual) part2 += LD2[ii + 1 + a] * x2[j];
			}
		part = row_sum<W>(part);
		if (dual) part2 = row_sum<W>(part2);
		if (lane == 0) {
			x[i] -= part;
			if (dual) x2[i] -= part2;
		}
		gsync<G>();
	}
}

// ---- register-resident L'DL for small trees (nv <= 16, G == 16: one column of M per lane) -------------------------
// Lane j keeps column j of the matrix (entries (i, j), i >= j, zero where dof j is not an ancestor of dof i) in 16
// statically indexed registers; pivots run k = nv-1 .. 0 like mj_factorM, the pivot row and the multipliers travel
// between the lanes of the env's 16-lane group through ds_bpermute (no LDS storage, no barrier inside the
// factorisation).  `dadr[i]` = qM-layout address of entry (i, lane) or -1 (host table M_dense, fetched once per kernel).
DEVI double group_bcast16(double v, int src)  // value of lane `src` of this lane's 16-lane group
{
	const int addr = (int)(((threadIdx.x & 48u) | (unsigned int)src) << 2);  // (1-D blocks: lane in wave = threadIdx.x & 63)
	return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)),
	                        __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}


// 1 / x to ~1 ulp: hardware seed + two Newton steps (the correctly rounded division costs 12 dependent instructions)

#ifndef MJB_FACTOR64_LDS
#define MJB_FACTOR64_LDS 1  // (0: v_readlane broadcasts -- more issue slots than the LDS round trips the co-resident wave hides: 185.1 -> 186.9 ms on config 3)
#endif
template <int G, bool DUAL, int NVM, typename DA>
DEVI void factor_dense16_impl(CModel m, const Env &e, const double *M, double *LD, double *di, const double *M2, double *LD2,
                              double *di2, const DA (&dadr)[16], double *scr)
{
	const int lane = e.lane, nv = m.nv;
	double A[NVM], B[NVM];
#pragma unroll
	for (int i = 0; i < NVM; i++) {
		// unconditional loads (clamped address) + select: no divergent branch per entry
		const int a = dadr[i], ac = a >= 0 ? a : 0;
		const double va = M[ac], vb = DUAL ? M2[ac] : 0.0;
		A[i] = a >= 0 ? va : 0.0;
		B[i] = a >= 0 ? vb : 0.0;
	}
	double myinv = 0, myinv2 = 0;
	if constexpr (G == 64 && MJB_FACTOR64_LDS) {
	// (one env per wavefront, 256- / 512-register constrained kernels: the LDS round trip stays -- the broadcast forms below cost the
	//  capped PGS kernel registers, measured 42.9 -> 44.2 ms on config 3)
#pragma unroll
	for (int k = NVM - 1; k >= 0; k--) {
		if (k < nv) {
			MJB_KEEP_BRANCH();
			// the pivot row crosses lanes through LDS: every lane publishes its entry, then all read the row at uniform
			// addresses (one broadcast read per double; ds_bpermute costs two slower operations per double)
			if (G == 16 || lane < 16) {  // (G == 64: the first 16 lanes of the wavefront carry the matrix)
				scr[lane] = A[k];
				if (DUAL) scr[16 + lane] = B[k];
			}
			gsync<G>();
			const double dk = scr[k], dk2 = DUAL ? scr[16 + k] : 1.0;
			double mk[NVM], mk2[NVM];
#pragma unroll
			for (int i = 0; i < NVM; i++) {
				mk[i] = i < k ? scr[i] : 0.0;  // unscaled M(k, i), held by lane i
				mk2[i] = (DUAL && i < k) ? scr[16 + i] : 0.0;
			}
			gsync<G>();
			const double inv = fast_rcp(dk), inv2 = DUAL ? fast_rcp(dk2) : 0.0;
			const double lkj = A[k] * inv, lkj2 = B[k] * inv2;  // scaled pivot-row entry of this lane's column (lane < k)
			// (entries above the diagonal -- register i of a lane > i -- are never read or stored: no lane predicate)
#pragma unroll
			for (int i = 0; i < NVM; i++) {
				if (i >= k) continue;
				A[i] -= mk[i] * lkj;
				if (DUAL) B[i] -= mk2[i] * lkj2;
			}
			if (lane == k) {
				myinv = inv;
				myinv2 = inv2;
			} else {
				A[k] = lkj;
				B[k] = lkj2;
			}
		}
	}
	} else {
	(void)scr;
	static_for<NVM>([&](auto kc) {
		constexpr int k = NVM - 1 - decltype(kc)::value;
		if (k < nv) {
			MJB_KEEP_BRANCH();
			// the pivot row M(k, .) is spread over the lanes (lane i holds M(k, i) in register k): its entries travel by DPP row
			// broadcasts -- the diagonal first (its reciprocal heads the dependent chain), then one entry per update
			const double dk = group_bcast<G, k>(A[k]), dk2 = DUAL ? group_bcast<G, k>(B[k]) : 1.0;
			const double inv = fast_rcp(dk), inv2 = DUAL ? fast_rcp(dk2) : 0.0;
			const double lkj = A[k] * inv, lkj2 = B[k] * inv2;  // scaled pivot-row entry of this lane's column (lane < k)
			// (entries above the diagonal -- register i of a lane > i -- are never read or stored: no lane predicate)
			static_for<k>([&](auto ic) {
				constexpr int i = decltype(ic)::value;
				A[i] -= group_bcast<G, i>(A[k]) * lkj;  // unscaled M(k, i), held by lane i
				if (DUAL) B[i] -= group_bcast<G, i>(B[k]) * lkj2;
			});
			if (lane == k) {
				myinv = inv;
				myinv2 = inv2;
			} else {
				A[k] = lkj;
				B[k] = lkj2;
			}
		}
	});
	}
#pragma unroll
	for (int i = 0; i < NVM; i++) {
		const int a = dadr[i];
		if (a >= 0) {
			LD[a] = A[i];
			if (DUAL) LD2[a] = B[i];
		}
	}
	if (lane < nv) {
		di[lane] = myinv;
		if (DUAL) di2[lane] = myinv2;
	}
	gsync<G>();
}

template <int G, int NVM, typename DA>
STAGE void factor_dense16(CModel m, const Env &e, const double *M, double *LD, double *di, const double *M2, double *LD2,
                          double *di2, bool dual, const DA (&dadr)[16], double *scr)
{
	static_assert(G == 16 || G == 64, "one matrix column per lane of a 16-lane env group (G == 64: lanes 0-15 of the wavefront)");
	if (dual) {
		MJB_KEEP_BRANCH();
		factor_dense16_impl<G, true, NVM>(m, e, M, LD, di, M2, LD2, di2, dadr, scr);
	} else {
		MJB_KEEP_BRANCH();
		factor_dense16_impl<G, false, NVM>(m, e, M, LD, di, M2, LD2, di2, dadr, scr);
	}
}

// ---- the same for 16 < nv <= 32 in the 512-register constrained kernels (one env per wavefront, ONE wave per SIMD: every LDS
// round trip of the sparse factor2 -- a pivot's round per dof, ~1.3 k cycles each -- is a bare stall there).  Lane c < 32 keeps
// column c of M (and of M + h B) in 32 + 32 statically indexed registers, gathered through the host table M_sym; the pivot row
// travels by v_readlane.  The tree's zeros are skipped four columns at a time on the VALUES (a zero multiplier makes the update
// a no-op, so skipping it is exact): v_cmp -> wave mask -> scalar tests, no table.
// row `row` of the host table M_sym (symmetric: row == column) in eight 16-byte loads -- read entry by entry at stride 32 it was 32
// dependent-latency global loads per lane, twice in the factorisation and once more in the Newton solver's setup
DEVI void msym_row(CModel m, int row, int (&a)[32])
{
	const mjb_i4 MJB_AS4 *p = reinterpret_cast<const mjb_i4 MJB_AS4 *>(m.M_sym + 32 * row);
#pragma unroll
	for (int q = 0; q < 8; q++) {
		const mjb_i4 w = p[q];
		a[4 * q] = w[0]; a[4 * q + 1] = w[1]; a[4 * q + 2] = w[2]; a[4 * q + 3] = w[3];
	}
}

template <int G, bool DUAL>
DEVI void factor_dense32_impl(CModel m, const Env &e, const double *M, double *LD, double *di, const double *M2, double *LD2,
                              double *di2)
{
	static_assert(G == 64, "one env per wavefront");
	const int lane = e.lane, nv = m.nv;
	const int c = lane < 32 ? lane : 0;  // (lanes 32 .. 63 mirror lane 0 and store nothing)
	double A[32], B[32];
	int ms[32];
	msym_row(m, c, ms);
#pragma unroll
	for (int i = 0; i < 32; i++) {
		const int a = ms[i], ac = a >= 0 ? a : 0;
		const bool has = a >= 0 && i >= c;
		const double va = M[ac], vb = DUAL ? M2[ac] : 0.0;
		A[i] = has ? va : 0.0;
		B[i] = has ? vb : 0.0;
	}
	double myinv = 0, myinv2 = 0;
	static_for<32>([&](auto kc) {
		constexpr int k = 31 - decltype(kc)::value;
		if (k < nv) {
			MJB_KEEP_BRANCH();
			const double dk = group_bcast<64, k>(A[k]), dk2 = DUAL ? group_bcast<64, k>(B[k]) : 1.0;
			const double inv = fast_rcp(dk), inv2 = DUAL ? fast_rcp(dk2) : 0.0;
			const double lkj = A[k] * inv, lkj2 = B[k] * inv2;  // scaled pivot-row entry of this lane's column (lane < k)
			const unsigned int live = (unsigned int)__ballot(A[k] != 0.0);  // columns i with M(k, i) != 0
			static_for<(k + 3) / 4>([&](auto gc) {
				constexpr int i0 = 4 * decltype(gc)::value;
				if ((live >> i0) & 0xFu) {
					MJB_KEEP_BRANCH();
					// (the group's broadcasts first, then its fma: issued pairwise -- readlane, readlane, fma -- every fma sat behind
					//  the scalar-write hazard of its own operand, 526 s_nop in the stage)
					double ba[4], bb[4];
					static_for<4>([&](auto qc) {
						constexpr int q = decltype(qc)::value, i = i0 + q;
						ba[q] = i < k ? group_bcast<64, (i < k ? i : 0)>(A[k]) : 0.0;  // unscaled M(k, i), held by lane i
						bb[q] = (DUAL && i < k) ? group_bcast<64, (i < k ? i : 0)>(B[k]) : 0.0;
					});
					static_for<4>([&](auto qc) {
						constexpr int q = decltype(qc)::value, i = i0 + q;
						if constexpr (i < k) {
							A[i] -= ba[q] * lkj;
							if (DUAL) B[i] -= bb[q] * lkj2;
						}
					});
				}
			});
			if (lane == k) {
				myinv = inv;
				myinv2 = inv2;
			} else {
				A[k] = lkj;
				B[k] = lkj2;
			}
		}
	});
#pragma unroll
	for (int i = 0; i < 32; i++) {
		const int a = ms[i];
		if (a >= 0 && i >= c && lane < 32) {
			LD[a] = A[i];
			if (DUAL) LD2[a] = B[i];
		}
	}
	if (lane < nv) {
		di[lane] = myinv;
		if (DUAL) di2[lane] = myinv2;
	}
	gsync<G>();
}

template <int G>
STAGE void factor_dense32(CModel m, const Env &e, const double *M, double *LD, double *di, const double *M2, double *LD2, double *di2,
                          bool dual)
{
	if (dual) {
		MJB_KEEP_BRANCH();
		factor_dense32_impl<G, true>(m, e, M, LD, di, M2, LD2, di2);
	} else {
		MJB_KEEP_BRANCH();
		factor_dense32_impl<G, false>(m, e, M, LD, di, M2, LD2, di2);
	}
}

// x <- M^-1 x with the factor's columns re-read from LDS into registers; x lives one element per lane
#ifndef MJB_SOLVE64_LDS
#define MJB_SOLVE64_LDS 0  // (1: the LDS round trips of r02 -- 43.3 -> 42.1 ms per 200 steps of config 3 without them)
#endif
template <int G, bool DUAL, int NVM, typename DA>
DEVI void solve_dense16_impl(CModel m, const Env &e, double *x, const double *LD, const double *diaginv, double *x2,
                             const double *LD2, const double *diaginv2, const DA (&dadr)[16], double *scr)
{
	const int lane = e.lane, nv = m.nv;
	const bool act = lane < nv;
	double A[NVM], B[NVM];
#pragma unroll
	for (int i = 0; i < NVM; i++) {
		const int a = dadr[i], ac = a >= 0 ? a : 0;
		const double va = LD[ac], vb = DUAL ? LD2[ac] : 0.0;
		A[i] = (a >= 0 && i != lane) ? va : 0.0;
		B[i] = (a >= 0 && i != lane) ? vb : 0.0;
	}
	double xj = act ? x[lane] : 0.0, xj2 = (DUAL && act) ? x2[lane] : 0.0;
	const double dinv = act ? diaginv[lane] : 0.0, dinv2 = (DUAL && act) ? diaginv2[lane] : 0.0;
	if constexpr (G == 64 && MJB_SOLVE64_LDS) {
	// x <- inv(L') x : dof i pushes its value down to its ancestors j < i
#pragma unroll
	for (int i = NVM - 1; i >= 1; i--) {
		if (i < nv) {
			MJB_KEEP_BRANCH();
			// x_i crosses lanes through the LDS scratch (publish, then a wave-uniform broadcast read)
			if (G == 16 || lane < 16) {
				scr[lane] = xj;
				if (DUAL) scr[16 + lane] = xj2;
			}
			gsync<G>();
			const double xi = scr[i], xi2 = DUAL ? scr[16 + i] : 0.0;
			gsync<G>();
			xj -= A[i] * xi;  // A[i] == 0 in the lanes >= i
			if (DUAL) xj2 -= B[i] * xi2;
		}
	}
	} else {
	// x <- inv(L') x : dof i pushes its value down to its ancestors j < i
	(void)scr;
	static_for<NVM - 1>([&](auto ic) {
		constexpr int i = NVM - 1 - decltype(ic)::value;
		if (i < nv) {
			MJB_KEEP_BRANCH();
			const double xi = group_bcast<G, i>(xj), xi2 = DUAL ? group_bcast<G, i>(xj2) : 0.0;  // x_i from lane i
			xj -= A[i] * xi;  // A[i] == 0 in the lanes >= i
			if (DUAL) xj2 -= B[i] * xi2;
		}
	});
	}
	xj *= dinv;
	xj2 *= dinv2;
	// x <- inv(L) x : row i gathers L(i, j) x_j from the lanes j < i with a 16-lane butterfly sum
#pragma unroll
	for (int i = 1; i < NVM; i++) {
		if (i < nv) {
			MJB_KEEP_BRANCH();
			const double s = row_sum<16>(A[i] * xj);
			const double s2 = DUAL ? row_sum<16>(B[i] * xj2) : 0.0;
			if (lane == i) {
				xj -= s;
				xj2 -= s2;
			}
		}
	}
	if (act) {
		x[lane] = xj;
		if (DUAL) x2[lane] = xj2;
	}
	gsync<G>();
}

template <int G, int NVM, typename DA>
STAGE void solve_dense16(CModel m, const Env &e, double *x, const double *LD, const double *diaginv, double *x2,
                         const double *LD2, const double *diaginv2, bool dual, const DA (&dadr)[16], double *scr)
{
	static_assert(G == 16 || G == 64, "one matrix column per lane of a 16-lane env group (G == 64: lanes 0-15 of the wavefront)");
	if (dual) {
		MJB_KEEP_BRANCH();
		solve_dense16_impl<G, true, NVM>(m, e, x, LD, diaginv, x2, LD2, diaginv2, dadr, scr);
	} else {
		MJB_KEEP_BRANCH();
		solve_dense16_impl<G, false, NVM>(m, e, x, LD, diaginv, x2, LD2, diaginv2, dadr, scr);
	}
}

// Constrained kernels (G = 64) keep the dense address map of the register-resident factor / solve out of the registers:
// lanes 0-15 park their 16 qM addresses (< 255; 255 = none) as 4 packed ints in the frame's int area once per env and
// unpack them where a dense routine starts (4 LDS reads + 16 v_bfe) -- 16 kernel-lifetime VGPRs were the first thing the
// allocator spilled to scratch under the solvers' register pressure (reloaded at every factor / solve).
DEVI void dadr_load(const Env &e, CLayout L, int (&dl)[16])
{
	const int *src = e.fi + L.dadr + 4 * (e.lane & 15);
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const unsigned int w = (unsigned int)src[q];
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int v = (int)((w >> (8 * r)) & 255u);
			dl[4 * q + r] = (v == 255 || e.lane >= 16) ? -1 : v;
		}
	}
}

template <int G> DEVI void solve(CModel m, const EnvLite &e, double *x, const double *LD, const double *diaginv)
{
	solve2<G>(m, e, x, LD, diaginv, x, LD, diaginv, false);
}

// x <- M^-1 x for 16 < nv <= 32 in the 512-register constrained kernels (one env per wavefront).  The sparse L'DL factor is spread
// into a packed dense strictly-lower triangle T (entry (i, j), i > j, at i (i - 1) / 2 + j; zero where dof j is no ancestor of dof
// i) in a 496-double LDS scratch; lane k keeps x_k, COLUMN k of T (for the L' sweep) and ROW k (for the L sweep) in 62 statically
// indexed registers, and the pivots travel by v_readlane: 62 steps of (readlane pair, fma) instead of the generic solve's
// ~2 nv LDS round trips -- 37 k -> ~5 k cycles at nv = 30 (BASELINE config 5).  Entries / elements >= nv are zero, so the
// unrolled sweeps need no guards.  Both sweeps visit the off-diagonal entries of a row in ascending column order.
template <int G> DEVI void solve_tri32(CModel m, const Env &e, double *x, const double *LD, const double *diaginv, double *T)
{
	static_assert(G == 64, "one env per wavefront");
	const int lane = e.lane, nv = m.nv;
	for (int t = lane; t < 496; t += G) T[t] = 0;
	gsync<G>();
	for (int en = lane; en < m.nM; en += G) {
		const int i = m.M_rowdof[en], j = m.M_coldof[en];
		if (i != j) T[i * (i - 1) / 2 + j] = LD[en];
	}
	gsync<G>();
	const int k = lane < 32 ? lane : 0;  // (lanes 32 .. 63 mirror lane 0 and store nothing)
	double col[32], row[32];
#pragma unroll
	for (int i = 1; i < 32; i++) {
		const double v = T[i * (i - 1) / 2 + (k < i ? k : 0)];
		col[i] = k < i ? v : 0.0;
	}
#pragma unroll
	for (int j = 0; j < 31; j++) {
		const double v = T[k > j ? k * (k - 1) / 2 + j : 0];
		row[j] = k > j ? v : 0.0;
	}
	double xk = lane < nv ? x[lane] : 0.0;
	const double dk = lane < nv ? diaginv[lane] : 0.0;
#pragma unroll
	for (int i = 31; i >= 1; i--) {  // x <- inv(L') x
		const double xi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xk), i), __builtin_amdgcn_readlane(__double2loint(xk), i));
		xk -= col[i] * xi;
	}
	xk *= dk;
#pragma unroll
	for (int j = 0; j < 31; j++) {  // x <- inv(L) x, column by column
		const double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xk), j), __builtin_amdgcn_readlane(__double2loint(xk), j));
		xk -= row[j] * xj;
	}
	if (lane < nv) x[lane] = xk;
	gsync<G>();
}

// ------------------------------------------------------------------------------------------------
// transmission (joint) : actuator_length
// ------------------------------------------------------------------------------------------------
template <int G, bool CACHE = false> STAGE void transmission(CModel m, CLayout L, const Env &e)
{
	if constexpr (CACHE) {
		if (!m.act_tendon) {
			if (e.lane < m.nu) e.f[L.actuator_length + e.lane] = e.f[L.qpos + e.lc.u_qa] * e.lc.u_gear;
			return;
		}
	}
	for (int i = e.lane; i < m.nu; i += G) {
		const int j = m.actuator_trnid[2 * i];
		if (m.act_tendon && m.actuator_trntype[i] == MJB_TRN_TENDON) e.f[L.actuator_length + i] = e.f[L.ten_length + j] * m.actuator_gear[6 * i];  // (mj_tendon ran in com_pos)
		else e.f[L.actuator_length + i] = e.f[L.qpos + m.jnt_qposadr[j]] * m.actuator_gear[6 * i];
	}
}

// ------------------------------------------------------------------------------------------------
// A8  comVel: body spatial velocities (cvel) and dof time-derivatives (cdof_dot)
// ------------------------------------------------------------------------------------------------
// velocity of body b "before" dof d's group is applied: parent's cvel + earlier groups of the body
DEVI void cvel_before(CModel m, CLayout L, const double *f, int b, int dstop, double *v)
{
	ld6(v, f + L.cvel + 6 * m.body_parentid[b]);
	const int bda = m.body_dofadr[b];
	int d = bda;
	while (d < dstop) {
		const int gs = d;
		const int jt = m.jnt_type[m.dof_jntid[d]];
		const int glen = (jt == MJB_JNT_HINGE || jt == MJB_JNT_SLIDE) ? 1 : 3;
		double tmp[6] = { 0, 0, 0, 0, 0, 0 };
		for (int k = 0; k < glen; k++) {
			const double qv = f[L.qvel + gs + k];
			for (int c = 0; c < 6; c++) tmp[c] += f[L.cdof + 6 * (gs + k) + c] * qv;
		}
		for (int c = 0; c < 6; c++) v[c] += tmp[c];
		d += glen;
	}
}

template <int G, bool OBL> STAGE void com_vel(CModel m, CLayout L, const Env &e)
{
	double *f = e.f;
	double *cvel = f + L.cvel, *cdof = f + L.cdof, *qvel = f + L.qvel;
	const int lane = e.lane;
	SPROF_BEGIN();
	// lane = body: cvel = sum of cdof_d qvel_d over the dofs that move the body (ancestor mask, root to leaf order)
	if (!OBL && m.dofanc_max > 0 && m.nbody <= G) {
		MJB_KEEP_BRANCH();
		// (the lane's own ancestor list, ascending, four entries a trip: 7 of the hand's 30 dofs move a fingertip -- the masked loop
		//  over all dofs read 210 LDS values per lane for the 49 that count; same terms in the same order)
		const bool act = lane < m.nbody;
		const int b = act ? (int)lane : 0;
		const mjb_i4 a4 = reinterpret_cast<const mjb_i4 MJB_AS4 *>(m.body_dofanc)[b];
		unsigned long long alo = ((unsigned long long)(unsigned int)a4[1] << 32) | (unsigned int)a4[0];
		unsigned long long ahi = ((unsigned long long)(unsigned int)a4[3] << 32) | (unsigned int)a4[2];
		double v[6] = { 0, 0, 0, 0, 0, 0 };
#pragma nounroll
		for (int k0 = 0; k0 < m.dofanc_max; k0 += 4) {
			int dd[4];
			double qd[4], cd[4][6];
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const int d = list_next(alo, ahi);
				dd[q] = d != 0xFF ? d : 0;
				qd[q] = qvel[dd[q]];
				for (int c = 0; c < 6; c++) cd[q][c] = cdof[6 * dd[q] + c];
				if (d == 0xFF) qd[q] = 0.0;
			}
#pragma unroll
			for (int q = 0; q < 4; q++)
				for (int c = 0; c < 6; c++) v[c] += cd[q][c] * qd[q];
		}
		if (act) st6(cvel + 6 * b, v);
	} else
	for (int b = lane; b < m.nbody; b += G) {
		const unsigned int lo = OBL ? e.lc.dmlo : (unsigned int)m.body_dofmask[2 * b], hi = OBL ? e.lc.dmhi : (unsigned int)m.body_dofmask[2 * b + 1];
		double v[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll 3
		for (int d = 0; d < m.nv; d++) {
			const double qd = maskbit(lo, hi, d) ? qvel[d] : 0.0;
			for (int c = 0; c < 6; c++) v[c] += cdof[6 * d + c] * qd;
		}
		st6(cvel + 6 * b, v);
	}
	gsync<G>();
	SPROF(25);
	// cdof_dot: one dof per lane
	for (int d = lane; d < m.nv; d += G) {
		double r[6];
		bool zero;
		[[maybe_unused]] mjb_i4 dr = mjb_i4{ 0, 0, 0, 0 };
		if constexpr (OBL) zero = e.lc.d_zero != 0;
		else {
			dr = reinterpret_cast<const mjb_i4 MJB_AS4 *>(m.dof_rec2)[d];
			zero = dr[0] != 0;
		}
		if (zero) {
			for (int k = 0; k < 6; k++) r[k] = 0;
		} else {
			double v[6], cd[6];
			if (OBL ? e.lc.d_simple != 0 : dr[1] != 0) ld6(v, cvel + 6 * (OBL ? (int)e.lc.d_parent : dr[2]));  // first joint of its body: the parent's velocity
			else cvel_before(m, L, f, m.dof_bodyid[d], m.dof_jstart[d], v);
			ld6(cd, cdof + 6 * d);
			cross_motion(r, v, cd);
		}
		st6(f + L.cdof_dot + 6 * d, r);
	}
	if (OBL && !m.act_tendon) {
		if (lane < m.nu) f[L.actuator_velocity + lane] = e.lc.u_gear * qvel[e.lc.u_da];
	} else {
		for (int i = lane; i < m.nu; i += G) {
			const int j = m.actuator_trnid[2 * i];
			if (m.act_tendon && m.actuator_trntype[i] == MJB_TRN_TENDON) {  // moment . qvel, moment = gear * the tendon's coefficients
				double v = 0;
				for (int w = m.tendon_adr[j]; w < m.tendon_adr[j] + m.tendon_num[j]; w++)
					v += m.actuator_gear[6 * i] * m.wrap_prm[w] * qvel[m.jnt_dofadr[m.wrap_objid[w]]];
				f[L.actuator_velocity + i] = v;
			} else
				f[L.actuator_velocity + i] = m.actuator_gear[6 * i] * qvel[m.jnt_dofadr[j]];
		}
	}
	gsync<G>();
	SPROF(26);
}

// ------------------------------------------------------------------------------------------------
// A8  passive forces: joint springs and dof dampers
// ------------------------------------------------------------------------------------------------
template <int G, bool CACHE = false> STAGE void passive(CModel m, CLayout L, const Env &e)
{
	double *f = e.f;
	double *qp = f + L.qfrc_passive;
	const bool off = (m.disableflags & MJB_DSBL_PASSIVE) != 0;
	for (int j = e.lane; j < m.njnt; j += G) {
		const int jt = CACHE ? e.lc.j_type : m.jnt_type[j];
		if constexpr (CACHE) {
			if (jt == MJB_JNT_HINGE || jt == MJB_JNT_SLIDE) {  // one dof: spring and damper from the lane's registers (same expressions)
				const int pa1 = e.lc.j_qa, da1 = e.lc.j_da;
				const double k1 = e.lc.j_stiff;
				double v = 0;
				if (k1 != 0 && !off) v = -k1 * (f[L.qpos + pa1] - e.lc.j_spring);
				if (!off) v -= e.lc.j_damp * f[L.qvel + da1];
				qp[da1] = v;
				continue;
			}
		}
		int pa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
		const int nd = jt == MJB_JNT_FREE ? 6 : (jt == MJB_JNT_BALL ? 3 : 1);
		double frc[6] = { 0, 0, 0, 0, 0, 0 };
		const double k = m.jnt_stiffness[j];
		if (k != 0 && !off) {
			int o = 0;
			if (jt == MJB_JNT_FREE) {
				for (int c = 0; c < 3; c++) frc[c] = -k * (f[L.qpos + pa + c] - m.qpos_spring[pa + c]);
				pa += 3;
				o = 3;
			}
			if (jt == MJB_JNT_FREE || jt == MJB_JNT_BALL) {
				double q[4], qs[4], dif[3];
				ld4(q, f + L.qpos + pa);
				normalize4(q);
				ldc4(qs, m.qpos_spring + pa);
				quat_sub(dif, q, qs);
				for (int c = 0; c < 3; c++) frc[o + c] = -k * dif[c];
			} else {
				frc[0] = -k * (f[L.qpos + pa] - m.qpos_spring[pa]);
			}
		}
		for (int c = 0; c < nd; c++) {
			double v = frc[c];
			if (!off) v -= m.dof_damping[da + c] * f[L.qvel + da + c];
			qp[da + c] = v;
		}
	}
	gsync<G>();
	if (m.ntendon > 0) {
		// fixed tendons: velocity = J qvel (lane = tendon), then springs / dampers  qfrc_passive += J' frc  (tendons
		// may share dofs: one lane walks them in order, as mj_passive does)
		for (int t = e.lane; t < m.ntendon; t += G) {
			double v = 0;
			for (int w = m.tendon_adr[t]; w < m.tendon_adr[t] + m.tendon_num[t]; w++)
				v += m.wrap_prm[w] * f[L.qvel + m.jnt_dofadr[m.wrap_objid[w]]];
			f[L.ten_velocity + t] = v;
		}
		gsync<G>();
		if (e.lane == 0 && !off)
			for (int t = 0; t < m.ntendon; t++) {
				const double frc = -m.tendon_stiffness[t] * (f[L.ten_length + t] - m.tendon_lengthspring[t]) -
				                   m.tendon_damping[t] * f[L.ten_velocity + t];
				if (frc == 0) continue;
				for (int w = m.tendon_adr[t]; w < m.tendon_adr[t] + m.tendon_num[t]; w++)
					qp[m.jnt_dofadr[m.wrap_objid[w]]] += m.wrap_prm[w] * frc;
			}
		gsync<G>();
	}
}

// ------------------------------------------------------------------------------------------------
// A9  RNE with zero acceleration: qfrc_bias
// ------------------------------------------------------------------------------------------------
template <int G, bool OBL> STAGE void rne(CModel m, CLayout L, const Env &e)
{
	double *f = e.f;
	double *cacc = f + L.cacc, *cfrc = f + L.cfrc_body, *cdd = f + L.cdof_dot, *qvel = f + L.qvel;
	const int lane = e.lane;
	const bool grav = !(m.disableflags & MJB_DSBL_GRAVITY);
	SPROF_BEGIN();
	// lane = body: cacc = -gravity + sum of cdof_dot_d qvel_d over the dofs that move the body, then the body's own
	// inertial force  I a + v x* (I v)   (cfrc_body holds the per-body force, not its subtree sum)
	[[maybe_unused]] double rkeep[6] = { 0, 0, 0, 0, 0, 0 };  // (dense kernels: the lane's own body force stays in registers for phase 2)
	const bool anc_list = !OBL && m.dofanc_max > 0 && m.nbody <= G;
	const bool sub_mm = G == 64 && !OBL && m.sub_nt > 0;
	[[maybe_unused]] SubtreeA SA;
	[[maybe_unused]] int dbody = 0;
	if (sub_mm) {  // (operands of the backward pass's product, fetched while the forward pass runs)
		MJB_KEEP_BRANCH();
		SA = subtree_fetch(m, lane);
		dbody = m.dof_bodyid[lane < m.nv ? (int)lane : 0];
	}
	for (int b = lane; b < m.nbody; b += G) {
		double a[6] = { 0, 0, 0, grav ? -f[L.gravity] : 0.0, grav ? -f[L.gravity + 1] : 0.0, grav ? -f[L.gravity + 2] : 0.0 };
		if (anc_list) {  // (the body's ancestor dofs from its list, as in com_vel)
			MJB_KEEP_BRANCH();
			const mjb_i4 a4 = reinterpret_cast<const mjb_i4 MJB_AS4 *>(m.body_dofanc)[b];
			unsigned long long alo = ((unsigned long long)(unsigned int)a4[1] << 32) | (unsigned int)a4[0];
			unsigned long long ahi = ((unsigned long long)(unsigned int)a4[3] << 32) | (unsigned int)a4[2];
#pragma nounroll
			for (int k0 = 0; k0 < m.dofanc_max; k0 += 4) {
				double qd[4], cd[4][6];
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const int d = list_next(alo, ahi), dc = d != 0xFF ? d : 0;
					qd[q] = qvel[dc];
					for (int c = 0; c < 6; c++) cd[q][c] = cdd[6 * dc + c];
					if (d == 0xFF) qd[q] = 0.0;
				}
#pragma unroll
				for (int q = 0; q < 4; q++)
					for (int c = 0; c < 6; c++) a[c] += cd[q][c] * qd[q];
			}
		} else {
			MJB_KEEP_BRANCH();
			const unsigned int lo = OBL ? e.lc.dmlo : (unsigned int)m.body_dofmask[2 * b], hi = OBL ? e.lc.dmhi : (unsigned int)m.body_dofmask[2 * b + 1];
#pragma unroll 3
			for (int d = 0; d < m.nv; d++) {
				const double qd = maskbit(lo, hi, d) ? qvel[d] : 0.0;
				for (int c = 0; c < 6; c++) a[c] += cdd[6 * d + c] * qd;
			}
		}
		st6(cacc + 6 * b, a);
		double r[6];
		if (b == 0) {
			for (int k = 0; k < 6; k++) r[k] = 0;
		} else {
			double I[10], v[6], t[6], t1[6];
			ld10(I, f + L.cinert + 10 * b);
			ld6(v, f + L.cvel + 6 * b);
			mul_inert_vec(r, I, a);
			mul_inert_vec(t, I, v);
			cross_force(t1, v, t);
			for (int k = 0; k < 6; k++) r[k] += t1[k];
		}
		st6(cfrc + 6 * b, r);
		if constexpr (OBL && G == 16)
			for (int k = 0; k < 6; k++) rkeep[k] = r[k];
	}
	if constexpr (OBL && G == 16) {
		// lane = dof, one body per lane (nbody <= 16): the forces of the bodies dof d moves are summed straight out of the body
		// lanes' registers, each by a DPP row broadcast, in body order -- no LDS round trip between the two phases (4.2 k -> ~1.5 k
		// cycles of the 7 k this stage took on config 2)
		double acc[6] = { 0, 0, 0, 0, 0, 0 };
		const unsigned int blo = e.lc.d_bmlo;
		static_for<15>([&](auto bc) {
			constexpr int b = decltype(bc)::value + 1;
			if (b < m.nbody) {
				MJB_KEEP_BRANCH();
				const bool on = ((blo >> b) & 1u) != 0;
#pragma unroll
				for (int c = 0; c < 6; c++) {
					const double v = row_bcast16<b>(rkeep[c]);
					acc[c] += on ? v : 0.0;
				}
			}
		});
		if (lane < m.nv) {
			double a[6];
			ld6(a, f + L.cdof + 6 * lane);
			f[L.qfrc_bias + lane] = dot6r(a, acc);
		}
		gsync<G>();
		return;
	}
	gsync<G>();
	SPROF(27);
	if (sub_mm) {
		MJB_KEEP_BRANCH();
		// mj_rne's backward pass (nothing goes to the world body), then qfrc_bias_d = cdof_d . subtree force of dof d's body
		subtree_sum<6, true>(m, cfrc, lane, SA);
		gsync<G>();
		for (int d = lane; d < m.nv; d += G) {
			double a[6], acc[6];
			ld6(acc, cfrc + 6 * (d < G ? dbody : m.dof_bodyid[d]));
			ld6(a, f + L.cdof + 6 * d);
			f[L.qfrc_bias + d] = dot6r(a, acc);
		}
		gsync<G>();
		SPROF(28);
		return;
	}
	// lane = dof: qfrc_bias_d = cdof_d . (sum of the forces of the bodies that dof d moves)
	for (int d = lane; d < m.nv; d += G) {
		double acc[6] = { 0, 0, 0, 0, 0, 0 };
		// bodies moved by dof d: the transpose of the ancestor-dof masks, one 64-bit word per dof (per-lane constant)
		const unsigned int blo = OBL ? e.lc.d_bmlo : (unsigned int)m.dof_bodymask[2 * d], bhi = OBL ? e.lc.d_bmhi : (unsigned int)m.dof_bodymask[2 * d + 1];
#pragma unroll 3
		for (int b = 1; b < m.nbody; b++) {
			const bool on = maskbit(blo, bhi, b);
			for (int c = 0; c < 6; c++) {
				const double v = cfrc[6 * b + c];
				acc[c] += on ? v : 0.0;
			}
		}
		double a[6];
		ld6(a, f + L.cdof + 6 * d);
		f[L.qfrc_bias + d] = dot6r(a, acc);
	}
	gsync<G>();
	SPROF(28);
}

// ------------------------------------------------------------------------------------------------
// A15 sensors, one sensor per lane
// ------------------------------------------------------------------------------------------------
DEVI void frame_of(CModel m, CLayout L, const double *f, int objtype, int id, const double **pos,
                   const double **mat, double *quat)
{
	double bq[4], lq[4];
	switch (objtype) {
	case MJB_OBJ_BODY:
		*pos = f + L.xipos + 3 * id; *mat = f + L.ximat + 9 * id;
		ld4(bq, f + L.xquat + 4 * id);
		ldc4(lq, m.body_iquat + 4 * id);
		qmul(quat, bq, lq);
		break;
	case MJB_OBJ_XBODY:
		*pos = f + L.xpos + 3 * id; *mat = f + L.xmat + 9 * id;
		ld4(quat, f + L.xquat + 4 * id);
		break;
	case MJB_OBJ_GEOM:
		*pos = f + L.geom_xpos + 3 * id; *mat = f + L.geom_xmat + 9 * id;
		ld4(bq, f + L.xquat + 4 * m.geom_bodyid[id]);
		ldc4(lq, m.geom_quat + 4 * id);
		qmul(quat, bq, lq);
		break;
	default:
		*pos = f + L.site_xpos + 3 * id; *mat = f + L.site_xmat + 9 * id;
		ld4(bq, f + L.xquat + 4 * m.site_bodyid[id]);
		ldc4(lq, m.site_quat + 4 * id);
		qmul(quat, bq, lq);
	}
}

// com-based spatial motion vector (cvel or cacc) of the object's body moved to the object's frame origin
DEVI void object_motion(CModel m, CLayout L, const double *f, int field_off, int objtype, int id, double *res, bool local)
{
	const double *pos, *mat;
	double q[4], v[6], np[3], op[3], dif[3], cr[3];
	const int body = objtype == MJB_OBJ_GEOM ? m.geom_bodyid[id] : (objtype == MJB_OBJ_SITE ? m.site_bodyid[id] : id);
	frame_of(m, L, f, objtype, id, &pos, &mat, q);
	ld6(v, f + field_off + 6 * body);
	ld3(np, pos);
	ld3(op, f + L.subtree_com + 3 * m.body_rootid[body]);
	dif[0] = np[0] - op[0]; dif[1] = np[1] - op[1]; dif[2] = np[2] - op[2];
	cross3(cr, dif, v);
	double lin[3] = { v[3] - cr[0], v[4] - cr[1], v[5] - cr[2] };
	if (local) {
		double M[9];
		ld9(M, mat);
		matTvec3(res, M, v);
		matTvec3(res + 3, M, lin);
	} else {
		res[0] = v[0]; res[1] = v[1]; res[2] = v[2];
		res[3] = lin[0]; res[4] = lin[1]; res[5] = lin[2];
	}
}
DEVI void object_velocity(CModel m, CLayout L, const double *f, int objtype, int id, double *res, bool local)
{
	object_motion(m, L, f, L.cvel, objtype, id, res, local);
}
// mj_objectAcceleration: cacc moved to the object + the rotating-frame term omega x v
DEVI void object_acceleration(CModel m, CLayout L, const double *f, int objtype, int id, double *res, bool local)
{
	double vel[6], cr[3];
	object_motion(m, L, f, L.cacc, objtype, id, res, local);
	object_motion(m, L, f, L.cvel, objtype, id, vel, local);
	cross3(cr, vel, vel + 3);
	res[3] += cr[0]; res[4] += cr[1]; res[5] += cr[2];
}

// mj_contactForce: force / torque of contact c in its own frame (normal first); zero if it has no rows
DEVI void contact_force(CModel m, CLayout L, const double *f, const int *fi, int c, double *res)
{
	for (int k = 0; k < 6; k++) res[k] = 0;
	const int adr = fi[L.contact_efc_address + c], dim = fi[L.contact_dim + c];
	if (adr < 0) return;
	if (dim == 1) {
		res[0] = f[L.efc_force + adr];
	} else if (m.cone == MJB_CONE_ELLIPTIC) {
		for (int k = 0; k < 6; k++)
			if (k < dim) res[k] = f[L.efc_force + adr + k];
	} else {
		for (int k = 0; k < 10; k++)
			if (k < 2 * (dim - 1)) res[0] += f[L.efc_force + adr + k];
		for (int k = 0; k < 5; k++)
			if (k < dim - 1)
				res[1 + k] = (f[L.efc_force + adr + 2 * k] - f[L.efc_force + adr + 2 * k + 1]) * f[L.contact_friction + 5 * c + k];
	}
}

// (torque, force) about `oldpos` re-expressed about `newpos`
DEVI void move_force(double *res, const double *vec, const double *newpos, const double *oldpos)
{
	const double dif[3] = { newpos[0] - oldpos[0], newpos[1] - oldpos[1], newpos[2] - oldpos[2] };
	double cr[3];
	cross3(cr, dif, vec + 3);
	res[0] = vec[0] - cr[0]; res[1] = vec[1] - cr[1]; res[2] = vec[2] - cr[2];
	res[3] = vec[3]; res[4] = vec[4]; res[5] = vec[5];
}

// ------------------------------------------------------------------------------------------------
// mj_rnePostConstraint (only for models with touch / accelerometer / force / torque / frame*acc sensors): cacc with
// qacc, cfrc_ext = xfrc_applied + contact forces, cfrc_int = subtree sum of (inertial force - cfrc_ext).  Flat mask
// sums like rne: lane = contact (world wrench), lane = body (cacc, own force), lane = body (subtree sum).
// ------------------------------------------------------------------------------------------------
// (rarely used: kept out of line so that it costs the common kernels neither registers nor instruction-cache lines)
template <int G> __device__ __attribute__((noinline)) void rne_post(CModel m, CLayout L, const EnvLite e, bool use_xfrc)
{
	double *f = e.f;
	int *fi = e.fi;
	const int lane = e.lane, ncon = m.nconmax > 0 ? fi[L.ncon] : 0;
	double *wr = f + L.cwrench;  // [ncon][6] world (torque, force) of every contact at its position
	for (int c = lane; c < ncon; c += G) {
		double lf[6], fr[9], w[6];
		contact_force(m, L, f, fi, c, lf);
		ld9(fr, f + L.contact_frame + 9 * c);
		matTvec3(w + 3, fr, lf);
		matTvec3(w, fr, lf + 3);
		st6(wr + 6 * c, w);
	}
	gsync<G>();
	const bool grav = !(m.disableflags & MJB_DSBL_GRAVITY);
	for (int b = lane; b < m.nbody; b += G) {
		double ext[6] = { 0, 0, 0, 0, 0, 0 }, com[3];
		ld3(com, f + L.subtree_com + 3 * m.body_rootid[b]);
		if (b > 0 && use_xfrc) {
			const double *x = f + L.xfrc_applied + 6 * b;
			const double cf[6] = { x[3], x[4], x[5], x[0], x[1], x[2] };
			double ip[3], r[6];
			ld3(ip, f + L.xipos + 3 * b);
			move_force(r, cf, com, ip);
			for (int k = 0; k < 6; k++) ext[k] += r[k];
		}
		for (int c = 0; c < ncon; c++) {
			if (b == 0 || fi[L.contact_efc_address + c] < 0) continue;
			const int b1 = m.geom_bodyid[fi[L.contact_geom + 2 * c]], b2 = m.geom_bodyid[fi[L.contact_geom + 2 * c + 1]];
			if (b1 != b && b2 != b) continue;
			double w[6], cp[3], r[6];
			ld6(w, wr + 6 * c);
			ld3(cp, f + L.contact_pos + 3 * c);
			move_force(r, w, com, cp);
			const double sg = (b2 == b ? 1.0 : 0.0) - (b1 == b ? 1.0 : 0.0);
			for (int k = 0; k < 6; k++) ext[k] += sg * r[k];
		}
		st6(f + L.cfrc_ext + 6 * b, ext);
		const unsigned int lo = (unsigned int)m.body_dofmask[2 * b], hi = (unsigned int)m.body_dofmask[2 * b + 1];
		double a[6] = { 0, 0, 0, grav ? -f[L.gravity] : 0.0, grav ? -f[L.gravity + 1] : 0.0, grav ? -f[L.gravity + 2] : 0.0 };
		for (int d = 0; d < m.nv; d++) {
			const bool on = maskbit(lo, hi, d);
			const double qv = on ? f[L.qvel + d] : 0.0, qa = on ? f[L.qacc + d] : 0.0;
			for (int c = 0; c < 6; c++) a[c] += f[L.cdof_dot + 6 * d + c] * qv + f[L.cdof + 6 * d + c] * qa;
		}
		st6(f + L.cacc + 6 * b, a);
		double own[6] = { 0, 0, 0, 0, 0, 0 };
		if (b > 0) {
			double I[10], v[6], t[6], t1[6];
			ld10(I, f + L.cinert + 10 * b);
			ld6(v, f + L.cvel + 6 * b);
			mul_inert_vec(own, I, a);
			mul_inert_vec(t, I, v);
			cross_force(t1, v, t);
			for (int k = 0; k < 6; k++) own[k] += t1[k] - ext[k];
		}
		st6(f + L.cfrc_body + 6 * b, own);
	}
	gsync<G>();
	for (int b = lane; b < m.nbody; b += G) {
		const unsigned int lo = (unsigned int)m.body_submask[2 * b], hi = (unsigned int)m.body_submask[2 * b + 1];
		double acc[6] = { 0, 0, 0, 0, 0, 0 };
		for (int i = 1; i < m.nbody; i++) {
			const bool on = maskbit(lo, hi, i);
			for (int k = 0; k < 6; k++) {
				const double v = f[L.cfrc_body + 6 * i + k];
				acc[k] += on ? v : 0.0;
			}
		}
		st6(f + L.cfrc_int + 6 * b, acc);
	}
	gsync<G>();
}

// does the ray p + s dir (s >= 0) meet the site volume (sphere / box)?  -- the touch sensor's zone test
DEVI bool ray_hits_site(CModel m, CLayout L, const double *f, int site, const double *p, const double *dir)
{
	double M[9], rel[3], lp[3], ld[3];
	ld9(M, f + L.site_xmat + 9 * site);
	for (int k = 0; k < 3; k++) rel[k] = p[k] - f[L.site_xpos + 3 * site + k];
	matTvec3(lp, M, rel);
	matTvec3(ld, M, dir);
	const double sz[3] = { m.site_size[3 * site], m.site_size[3 * site + 1], m.site_size[3 * site + 2] };
	if (m.site_type[site] == MJB_GEOM_SPHERE) {
		const double b = dot3(lp, ld), c = dot3(lp, lp) - sz[0] * sz[0], a = dot3(ld, ld);
		if (c <= 0) return true;
		const double det = b * b - a * c;
		return det >= 0 && -b + sqrt(det) >= 0 && a > 0;
	}
	double tmin = 0, tmax = 1e300;
	for (int k = 0; k < 3; k++) {
		if (fabs(ld[k]) < MJB_MINVAL) {
			if (fabs(lp[k]) > sz[k]) return false;
			continue;
		}
		double t1 = (-sz[k] - lp[k]) / ld[k], t2 = (sz[k] - lp[k]) / ld[k];
		if (t1 > t2) { const double sw = t1; t1 = t2; t2 = sw; }
		if (t1 > tmin) tmin = t1;
		if (t2 < tmax) tmax = t2;
		if (tmin > tmax) return false;
	}
	return true;
}

// mju_rayGeom for the engine's primitives (engine_ray.c; oracle/mjo_smooth.c ray_geom): distance along pnt + x vec, -1 for no hit
DEVI double ray_quad(double a, double b, double c, double &x0, double &x1)
{
	double det = b * b - a * c;
	if (det < MJB_MINVAL) { x0 = x1 = -1; return -1; }
	det = sqrt(det);
	x0 = (-b - det) / a;
	x1 = (-b + det) / a;
	return x0 >= 0 ? x0 : (x1 >= 0 ? x1 : -1.0);
}
DEVI double ray_geom(const double *pos, const double *mat, const double *size, const double *pnt, const double *vec, int type)
{
	const double dif[3] = { pnt[0] - pos[0], pnt[1] - pos[1], pnt[2] - pos[2] };
	double lp[3], lv[3], x0, x1;
	matTvec3(lp, mat, dif);
	matTvec3(lv, mat, vec);
	if (type == MJB_GEOM_PLANE) {
		if (lv[2] > -MJB_MINVAL) return -1;
		const double x = -lp[2] / lv[2];
		if (x < 0) return -1;
		const double p0 = lp[0] + x * lv[0], p1 = lp[1] + x * lv[1];
		return ((size[0] <= 0 || fabs(p0) <= size[0]) && (size[1] <= 0 || fabs(p1) <= size[1])) ? x : -1.0;
	}
	if (type == MJB_GEOM_SPHERE) return ray_quad(dot3(lv, lv), dot3(lv, lp), dot3(lp, lp) - size[0] * size[0], x0, x1);
	if (type == MJB_GEOM_CAPSULE) {
		double x = -1;
		const double sol = ray_quad(lv[0] * lv[0] + lv[1] * lv[1], lv[0] * lp[0] + lv[1] * lp[1], lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0], x0, x1);
		if (sol >= 0 && fabs(lp[2] + sol * lv[2]) <= size[1]) x = sol;
		for (int side = 1; side >= -1; side -= 2) {
			const double ld[3] = { lp[0], lp[1], lp[2] - side * size[1] };
			ray_quad(dot3(lv, lv), dot3(lv, ld), dot3(ld, ld) - size[0] * size[0], x0, x1);
			if (x0 >= 0 && side * (lp[2] + x0 * lv[2]) >= size[1] && (x < 0 || x0 < x)) x = x0;
			if (x1 >= 0 && side * (lp[2] + x1 * lv[2]) >= size[1] && (x < 0 || x1 < x)) x = x1;
		}
		return x;
	}
	if (type == MJB_GEOM_BOX) {
		double x = -1;
		for (int i = 0; i < 3; i++) {
			if (fabs(lv[i]) <= MJB_MINVAL) continue;
			const int j = (i + 1) % 3, k = (i + 2) % 3;
			for (int side = -1; side <= 1; side += 2) {
				const double sol = (side * size[i] - lp[i]) / lv[i];
				if (sol >= 0 && fabs(lp[j] + sol * lv[j]) <= size[j] && fabs(lp[k] + sol * lv[k]) <= size[k] && (x < 0 || sol < x)) x = sol;
			}
		}
		return x;
	}
	return -1;
}

template <int G, bool CACHE = false> STAGE void sensors(CModel m, CLayout L, CState st, const Env &e, int stage, int compact)
{
	if (m.disableflags & MJB_DSBL_SENSOR) return;
	const int ncopy = m.sens_ncopy[stage - 1], nslow = m.sens_nslow[stage - 1];
	if (ncopy == 0 && nslow == 0) return;
	double *f = e.f;
	// plain copies: host-resolved {dst, src} pairs (table of the layout in use)
	if (CACHE && ncopy <= 2 * G) {  // (the pairs of this launch's layout sit in the lane's registers)
#pragma unroll
		for (int q = 0; q < 2; q++) {
			const int dst = e.lc.sc_dst[stage - 1][q];
			if (dst >= 0) f[L.sensordata + dst] = f[e.lc.sc_src[stage - 1][q]];
		}
	} else {
		const int tb = (3 * compact + stage - 1) * m.sens_ncopy_max;
		for (int t = e.lane; t < ncopy; t += G) {
			const int dst = m.sens_copy[2 * (tb + t)], src = m.sens_copy[2 * (tb + t) + 1];
			f[L.sensordata + dst] = f[src];
		}
	}
	for (int t = e.lane; t < nslow; t += G) {
		const int i = m.sens_slow[(stage - 1) * (m.nsensor ? m.nsensor : 1) + t];
		const int type = m.sensor_type[i], id = m.sensor_objid[i], ot = m.sensor_objtype[i];
		const int rid = m.sensor_refid[i], rt = m.sensor_reftype[i];
		double out[4] = { 0, 0, 0, 0 };
		bool real = true;
		switch (type) {
		case MJB_SENS_JOINTPOS: out[0] = f[L.qpos + m.jnt_qposadr[id]]; break;
		case MJB_SENS_ACTUATORPOS: out[0] = f[L.actuator_length + id]; break;
		case MJB_SENS_BALLQUAT:
			ld4(out, f + L.qpos + m.jnt_qposadr[id]);
			normalize4(out);
			real = false;
			break;
		case MJB_SENS_FRAMEPOS: case MJB_SENS_FRAMEQUAT: case MJB_SENS_FRAMEXAXIS: case MJB_SENS_FRAMEYAXIS:
		case MJB_SENS_FRAMEZAXIS: {
			const double *pos, *mat, *rpos = nullptr, *rmat = nullptr;
			double q[4], rq[4], RM[9];
			frame_of(m, L, f, ot, id, &pos, &mat, q);
			if (rid >= 0) {
				frame_of(m, L, f, rt, rid, &rpos, &rmat, rq);
				ld9(RM, rmat);
			}
			if (type == MJB_SENS_FRAMEPOS) {
				double p[3];
				ld3(p, pos);
				if (rid < 0) {
					out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
				} else {
					double rp[3];
					ld3(rp, rpos);
					double dif[3] = { p[0] - rp[0], p[1] - rp[1], p[2] - rp[2] };
					matTvec3(out, RM, dif);
				}
			} else if (type == MJB_SENS_FRAMEQUAT) {
				if (rid < 0) {
					out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
				} else {
					double neg[4] = { rq[0], -rq[1], -rq[2], -rq[3] };
					qmul(out, neg, q);
				}
				real = false;
			} else {
				const int c = type - MJB_SENS_FRAMEXAXIS;
				double ax[3] = { mat[c], mat[3 + c], mat[6 + c] };
				if (rid < 0) {
					out[0] = ax[0]; out[1] = ax[1]; out[2] = ax[2];
				} else {
					matTvec3(out, RM, ax);
				}
				real = false;
			}
			break;
		}
		case MJB_SENS_SUBTREECOM: ld3(out, f + L.subtree_com + 3 * id); break;
		case MJB_SENS_CLOCK: out[0] = f[L.time]; break;
		case MJB_SENS_JOINTVEL: out[0] = f[L.qvel + m.jnt_dofadr[id]]; break;
		case MJB_SENS_ACTUATORVEL: out[0] = f[L.actuator_velocity + id]; break;
		case MJB_SENS_BALLANGVEL: ld3(out, f + L.qvel + m.jnt_dofadr[id]); break;
		case MJB_SENS_VELOCIMETER: case MJB_SENS_GYRO: {
			double xv[6];
			object_velocity(m, L, f, MJB_OBJ_SITE, id, xv, true);
			const int o = type == MJB_SENS_GYRO ? 0 : 3;
			out[0] = xv[o]; out[1] = xv[o + 1]; out[2] = xv[o + 2];
			break;
		}
		case MJB_SENS_FRAMELINVEL: case MJB_SENS_FRAMEANGVEL: {
			double xv[6];
			const int o = type == MJB_SENS_FRAMELINVEL ? 3 : 0;
			object_velocity(m, L, f, ot, id, xv, false);
			if (rid >= 0) {
				const double *pos, *mat, *rpos, *rmat;
				double q[4], rq[4], rv[6], RM[9];
				frame_of(m, L, f, ot, id, &pos, &mat, q);
				frame_of(m, L, f, rt, rid, &rpos, &rmat, rq);
				object_velocity(m, L, f, rt, rid, rv, false);
				for (int k = 0; k < 6; k++) xv[k] -= rv[k];
				if (type == MJB_SENS_FRAMELINVEL) {
					double rel[3] = { pos[0] - rpos[0], pos[1] - rpos[1], pos[2] - rpos[2] }, cr[3];
					cross3(cr, rel, rv);
					xv[3] += cr[0]; xv[4] += cr[1]; xv[5] += cr[2];
				}
				ld9(RM, rmat);
				matTvec3(out, RM, xv + o);
			} else {
				out[0] = xv[o]; out[1] = xv[o + 1]; out[2] = xv[o + 2];
			}
			break;
		}
		case MJB_SENS_ACTUATORFRC: out[0] = f[L.actuator_force + id]; break;
		case MJB_SENS_TENDONPOS: out[0] = f[L.ten_length + id]; break;
		case MJB_SENS_TENDONVEL: out[0] = f[L.ten_velocity + id]; break;
		case MJB_SENS_ACCELEROMETER: {
			double acc[6];
			object_acceleration(m, L, f, MJB_OBJ_SITE, id, acc, true);
			out[0] = acc[3]; out[1] = acc[4]; out[2] = acc[5];
			break;
		}
		case MJB_SENS_FORCE: case MJB_SENS_TORQUE: {
			const int body = m.site_bodyid[id];
			double ci[6], sp[3], com[3], r[6], SM[9];
			ld6(ci, f + L.cfrc_int + 6 * body);
			ld3(sp, f + L.site_xpos + 3 * id);
			ld3(com, f + L.subtree_com + 3 * m.body_rootid[body]);
			move_force(r, ci, sp, com);
			ld9(SM, f + L.site_xmat + 9 * id);
			matTvec3(out, SM, type == MJB_SENS_FORCE ? r + 3 : r);
			break;
		}
		case MJB_SENS_TOUCH: {
			const int body = m.site_bodyid[id];
			const int ncon = m.nconmax > 0 ? e.fi[L.ncon] : 0;
			double tot = 0;
			for (int c = 0; c < ncon; c++) {
				const int b1 = m.geom_bodyid[e.fi[L.contact_geom + 2 * c]], b2 = m.geom_bodyid[e.fi[L.contact_geom + 2 * c + 1]];
				if (e.fi[L.contact_efc_address + c] < 0 || (body != b1 && body != b2)) continue;
				double lf[6], cp[3];
				contact_force(m, L, f, e.fi, c, lf);
				if (lf[0] <= 0) continue;
				const double sg = body == b2 ? -1.0 : 1.0;
				const double ray[3] = { sg * f[L.contact_frame + 9 * c], sg * f[L.contact_frame + 9 * c + 1], sg * f[L.contact_frame + 9 * c + 2] };
				ld3(cp, f + L.contact_pos + 3 * c);
				if (ray_hits_site(m, L, f, id, cp, ray)) tot += lf[0];
			}
			out[0] = tot;
			break;
		}
		case MJB_SENS_FRAMELINACC: case MJB_SENS_FRAMEANGACC: {
			double acc[6];
			object_acceleration(m, L, f, ot, id, acc, false);
			const int o = type == MJB_SENS_FRAMELINACC ? 3 : 0;
			out[0] = acc[o]; out[1] = acc[o + 1]; out[2] = acc[o + 2];
			break;
		}
		// ---- round 6: limit sensors -- mj_sensorPos / Vel / Acc report efc_pos - efc_margin, efc_vel, efc_force of the FIRST limit row of the joint / tendon,
		// zero without one.  Position and velocity are evaluated from the limit's definition (mj_instantiateLimit: lower side first, dist = value - range[0] or
		// range[1] - value, a row while dist < margin; its Jacobian is -side on the joint's dof): on the fused frames the rows are built after these stages run.
		case MJB_SENS_JOINTLIMITPOS: case MJB_SENS_JOINTLIMITVEL: case MJB_SENS_TENDONLIMITPOS: case MJB_SENS_TENDONLIMITVEL: {
			const bool jn = type == MJB_SENS_JOINTLIMITPOS || type == MJB_SENS_JOINTLIMITVEL, pos = type == MJB_SENS_JOINTLIMITPOS || type == MJB_SENS_TENDONLIMITPOS;
			const bool lim = m.nefcmax > 0 && !(m.disableflags & (MJB_DSBL_LIMIT | MJB_DSBL_CONSTRAINT)) &&
			                 (jn ? (m.jnt_limited[id] != 0 && m.jnt_type[id] >= MJB_JNT_SLIDE) : (m.tendon_limited[id] != 0));
			if (lim) {
				const double value = jn ? f[L.qpos + m.jnt_qposadr[id]] : f[L.ten_length + id];
				const double vel = jn ? f[L.qvel + m.jnt_dofadr[id]] : f[L.ten_velocity + id];
				const double margin = jn ? m.jnt_margin[id] : m.tendon_margin[id];
				const double r0 = jn ? m.jnt_range[2 * id] : m.tendon_range[2 * id], r1 = jn ? m.jnt_range[2 * id + 1] : m.tendon_range[2 * id + 1];
				if (value - r0 < margin) out[0] = pos ? value - r0 - margin : vel;
				else if (r1 - value < margin) out[0] = pos ? r1 - value - margin : -vel;
			}
			break;
		}
		case MJB_SENS_JOINTLIMITFRC: case MJB_SENS_TENDONLIMITFRC: {
			const int want = type == MJB_SENS_JOINTLIMITFRC ? MJB_CNSTR_LIMIT_JOINT : MJB_CNSTR_LIMIT_TENDON;
			const int ne = m.nefcmax > 0 ? e.fi[L.nefc] : 0;
			for (int r = 0; r < ne; r++)
				if (e.fi[L.efc_type + r] == want && e.fi[L.efc_id + r] == id) {
					out[0] = f[L.efc_force + r];
					break;
				}
			break;
		}
		case MJB_SENS_JOINTACTFRC: out[0] = f[L.qfrc_actuator + m.jnt_dofadr[id]]; break;
		case MJB_SENS_MAGNETOMETER: {  // the global magnetic flux in the site's frame
			double SM[9];
			const double mg[3] = { m.magnetic[0], m.magnetic[1], m.magnetic[2] };
			ld9(SM, f + L.site_xmat + 9 * id);
			matTvec3(out, SM, mg);
			break;
		}
		case MJB_SENS_RANGEFINDER: {  // mj_ray along the site's z axis: every visible geom but those of the site's body; -1: nothing hit
			double SM[9], sp[3], dist = -1;
			ld9(SM, f + L.site_xmat + 9 * id);
			ld3(sp, f + L.site_xpos + 3 * id);
			const double rv[3] = { SM[2], SM[5], SM[8] };
			const int skip = m.site_bodyid[id];
			for (int g = 0; g < m.ngeom; g++) {
				if (m.geom_bodyid[g] == skip || m.geom_rgba[4 * g + 3] == 0) continue;
				double gp[3], gm[9], gs[3];
				ld3(gp, f + L.geom_xpos + 3 * g);
				ld9(gm, f + L.geom_xmat + 9 * g);
				for (int k = 0; k < 3; k++) gs[k] = st.env_geom_size ? st.env_geom_size[((size_t)e.env * m.ngeom + g) * 3 + k] : m.geom_size[3 * g + k];
				const int gt = st.env_geom_type ? st.env_geom_type[(size_t)e.env * m.ngeom + g] : m.geom_type[g];
				const double x = ray_geom(gp, gm, gs, sp, rv, gt);
				if (x >= 0 && (x < dist || dist < 0)) dist = x;
			}
			out[0] = dist;
			break;
		}
		// ---- mj_subtreeVel's results in closed form: v_c = sum m_b v_b / M over the subtree, L = sum [ I_b w_b + m_b (x_b - c) x (v_b - v_c) ] about the subtree's
		// com c; a body's com velocity from cvel (taken at the root's subtree com o): v_b = v + w x (x_b - o); I_b w from cinert (about o): I_o w - m d x (w x d), d = x_b - o
		case MJB_SENS_SUBTREELINVEL: case MJB_SENS_SUBTREEANGMOM: {
			double c[3], vc[3] = { 0, 0, 0 }, mtot = 0;
			ld3(c, f + L.subtree_com + 3 * id);
			for (int pass = 0; pass < (type == MJB_SENS_SUBTREEANGMOM ? 2 : 1); pass++) {
				for (int b = id; b < m.nbody; b++) {
					bool in_sub = false;
					for (int a = b; a >= id; a = m.body_parentid[a]) {
						if (a == id) { in_sub = true; break; }
						if (a == 0) break;
					}
					if (!in_sub) continue;
					double ci[10], cv[6], o[3];
					ld10(ci, f + L.cinert + 10 * b);
					ld6(cv, f + L.cvel + 6 * b);
					ld3(o, f + L.subtree_com + 3 * m.body_rootid[b]);
					const double mass = ci[9];
					if (!(mass > 0)) continue;
					const double d[3] = { ci[6] / mass, ci[7] / mass, ci[8] / mass };
					double wxd[3], vb[3];
					cross3(wxd, cv, d);
					for (int k = 0; k < 3; k++) vb[k] = cv[3 + k] + wxd[k];
					if (pass == 0) {
						for (int k = 0; k < 3; k++) vc[k] += mass * vb[k];
						mtot += mass;
					} else {
						// I_o w: cinert's rotational block (xx yy zz xy xz yz)
						const double Iw[3] = { ci[0] * cv[0] + ci[3] * cv[1] + ci[4] * cv[2], ci[3] * cv[0] + ci[1] * cv[1] + ci[5] * cv[2], ci[4] * cv[0] + ci[5] * cv[1] + ci[2] * cv[2] };
						double dx[3], t[3], rel[3], dv[3], cr[3];
						cross3(t, d, wxd);
						for (int k = 0; k < 3; k++) { rel[k] = o[k] + d[k] - c[k]; dv[k] = vb[k] - vc[k]; }
						cross3(cr, rel, dv);
						(void)dx;
						for (int k = 0; k < 3; k++) out[k] += Iw[k] - mass * t[k] + mass * cr[k];
					}
				}
				if (pass == 0) {
					const double inv = 1.0 / fmax(MJB_MINVAL, mtot);
					for (int k = 0; k < 3; k++) vc[k] *= inv;
					if (type == MJB_SENS_SUBTREELINVEL) { out[0] = vc[0]; out[1] = vc[1]; out[2] = vc[2]; }
				}
			}
			break;
		}
		default: break;
		}
		const double cutoff = m.sensor_cutoff[i];
		const int dim = m.sensor_dim[i];
		double *dst = f + L.sensordata + m.sensor_adr[i];
		for (int k = 0; k < 4; k++) {
			if (k >= dim) break;
			double v = out[k];
			if (cutoff > 0 && real) {
				if (type == MJB_SENS_TOUCH || type == MJB_SENS_RANGEFINDER) v = v > cutoff ? cutoff : v;  // (mjDATATYPE_POSITIVE)
				else v = v < -cutoff ? -cutoff : (v > cutoff ? cutoff : v);
			}
			dst[k] = v;
		}
	}
	gsync<G>();
}

// ------------------------------------------------------------------------------------------------
// A12 actuation and smooth acceleration
// ------------------------------------------------------------------------------------------------
template <int G, bool CACHE = false> STAGE void fwd_actuation(CModel m, CLayout L, const Env &e)
{
	double *f = e.f;
	const bool off = m.nu == 0 || (m.disableflags & MJB_DSBL_ACTUATION);
	// one dof per lane: forces of the actuators driving it (host-built CSR lists, ascending actuator id)
	for (int d = e.lane; d < m.nv; d += G) {
		double acc = 0;
		if constexpr (CACHE) {
			if (e.lc.a_n <= 1) {  // at most one actuator on this dof: its constants sit in the lane's registers (same expressions)
				if (e.lc.a_n == 1) {
					const int i = e.lc.a_id, fl = e.lc.a_flags;
					double force = 0;
					if (!off) {
						double ctrl = f[L.ctrl + i];
						if (fl & 1) ctrl = ctrl < e.lc.a_clo ? e.lc.a_clo : (ctrl > e.lc.a_chi ? e.lc.a_chi : ctrl);
						const double len = f[L.actuator_length + i], vel = f[L.actuator_velocity + i];
						double gain = e.lc.a_g[0], bias = 0;
						if (fl & 2) gain = e.lc.a_g[0] + e.lc.a_g[1] * len + e.lc.a_g[2] * vel;
						if (fl & 4) bias = e.lc.a_b[0] + e.lc.a_b[1] * len + e.lc.a_b[2] * vel;
						force = gain * ctrl + bias;
						if (fl & 8) force = force < e.lc.a_flo ? e.lc.a_flo : (force > e.lc.a_fhi ? e.lc.a_fhi : force);
						acc += e.lc.a_gear * force;
					}
					f[L.actuator_force + i] = force;
				}
				f[L.qfrc_actuator + d] = acc;
				continue;
			}
		}
		const int t0 = m.dof_act_adr[d], t1 = m.dof_act_adr[d + 1];
		for (int t = t0; t < t1; t++) {
			const int i = m.dof_act_id[t];
			double force = 0;
			if (!off) {
				double ctrl = f[L.ctrl + i];
				if (m.actuator_ctrllimited[i] && !(m.disableflags & MJB_DSBL_CLAMPCTRL)) {
					const double lo = m.actuator_ctrlrange[2 * i], hi = m.actuator_ctrlrange[2 * i + 1];
					ctrl = ctrl < lo ? lo : (ctrl > hi ? hi : ctrl);
				}
				const double len = f[L.actuator_length + i], vel = f[L.actuator_velocity + i];
				double gain = m.actuator_gainprm[3 * i], bias = 0;
				if (m.actuator_gaintype[i] == MJB_GAIN_AFFINE)
					gain = m.actuator_gainprm[3 * i] + m.actuator_gainprm[3 * i + 1] * len + m.actuator_gainprm[3 * i + 2] * vel;
				if (m.actuator_biastype[i] == MJB_BIAS_AFFINE)
					bias = m.actuator_biasprm[3 * i] + m.actuator_biasprm[3 * i + 1] * len + m.actuator_biasprm[3 * i + 2] * vel;
				double input = ctrl;
				if (m.na > 0) {  // a stateful actuator's gain multiplies its activation (mj_fwdActuation)
					const int ja = m.actuator_actadr[i];
					if (ja >= 0) input = f[L.act + ja];
				}
				force = gain * input + bias;
				if (m.actuator_forcelimited[i]) {
					const double lo = m.actuator_forcerange[2 * i], hi = m.actuator_forcerange[2 * i + 1];
					force = force < lo ? lo : (force > hi ? hi : force);
				}
				acc += m.dof_act_mom[t] * force;
			}
			f[L.actuator_force + i] = force;
		}
		f[L.qfrc_actuator + d] = acc;
	}
	if (m.na > 0) {
		// act_dot of the stateful actuators, from the clamped ctrl: integrator ctrl, filter (ctrl - act) / max(mjMINVAL, dynprm[0])
		for (int i = e.lane; i < m.nu; i += G) {
			const int ja = m.actuator_actadr[i];
			if (ja < 0) continue;
			double ctrl = f[L.ctrl + i], dot = 0;
			if (m.actuator_ctrllimited[i] && !(m.disableflags & MJB_DSBL_CLAMPCTRL)) {
				const double lo = m.actuator_ctrlrange[2 * i], hi = m.actuator_ctrlrange[2 * i + 1];
				ctrl = ctrl < lo ? lo : (ctrl > hi ? hi : ctrl);
			}
			if (!off) {
				const double tau = m.actuator_dynprm[3 * i];
				dot = m.actuator_dyntype[i] == MJB_DYN_INTEGRATOR ? ctrl : (ctrl - f[L.act + ja]) / (tau > MJB_MINVAL ? tau : MJB_MINVAL);
			}
			f[L.act_dot + ja] = dot;
		}
	}
	gsync<G>();
}

// mj_advance, activations: act += h act_dot, clamped to actrange when the actuator is actlimited
template <int G> DEVI void advance_act(CModel m, double *act, const double *act0, const double *act_dot, double h, int lane)
{
	for (int i = lane; i < m.nu; i += G) {
		const int ja = m.actuator_actadr[i];
		if (ja < 0) continue;
		double a = act0[ja] + h * act_dot[ja];
		if (m.actuator_actlimited[i]) {
			const double lo = m.actuator_actrange[2 * i], hi = m.actuator_actrange[2 * i + 1];
			a = a < lo ? lo : (a > hi ? hi : a);
		}
		act[ja] = a;
	}
}

template <int G, int DENSE, bool TRI32 = false> STAGE void fwd_acceleration(CModel m, CLayout L, const Env &e, bool use_xfrc)
{
	double *f = e.f;
	for (int d = e.lane; d < m.nv; d += G) {
		double v = f[L.qfrc_passive + d] - f[L.qfrc_bias + d];
		v += f[L.qfrc_applied + d];
		v += f[L.qfrc_actuator + d];
		if (use_xfrc) {
			// Cartesian wrenches at body coms, projected with the cdof Jacobian, in body order
			const int bd = m.dof_bodyid[d];
			double cd[6];
			ld6(cd, f + L.cdof + 6 * d);
			for (int b = 1; b < m.nbody; b++) {
				const double *x = f + L.xfrc_applied + 6 * b;
				if (x[0] == 0 && x[1] == 0 && x[2] == 0 && x[3] == 0 && x[4] == 0 && x[5] == 0) continue;
				// is dof d on the path from b to the root?  (bd must be an ancestor-or-self of b)
				int a = b;
				while (a > bd) a = m.body_parentid[a];
				if (a != bd) continue;
				double ip[3], rc[3];
				ld3(ip, f + L.xipos + 3 * b);
				ld3(rc, f + L.subtree_com + 3 * m.body_rootid[b]);
				double off[3] = { ip[0] - rc[0], ip[1] - rc[1], ip[2] - rc[2] }, jp[3];
				cross3(jp, cd, off);
				jp[0] += cd[3]; jp[1] += cd[4]; jp[2] += cd[5];
				v += dot3(jp, x) + dot3(cd, x + 3);
			}
		}
		f[L.qfrc_smooth + d] = v;
		f[L.qacc_smooth + d] = v;
		f[L.eulerx + d] = v;  // rhs of Euler's implicit-damping solve when no constraint force can be added
	}
	gsync<G>();
	const bool dual = m.eulerdamp && m.nefcmax == 0;
	if constexpr (DENSE > 0)
		solve_dense16<G, DENSE>(m, e, f + L.qacc_smooth, f + L.qLD, f + L.qLDiagInv, f + L.eulerx, f + L.qH, f + L.qHdi, dual, e.dadr,
		                        f + L.solvescr);
	else if constexpr (DENSE < 0) {  // constrained kernel: a small system is solved by lanes 0-15 of the wavefront
		if (m.nv <= 16) {
			int dl[16];
			dadr_load(e, L, dl);
			solve_dense16<G, 16>(m, e, f + L.qacc_smooth, f + L.qLD, f + L.qLDiagInv, f + L.eulerx, f + L.qH, f + L.qHdi, dual, dl,
			                     f + L.solvescr);
		} else if (TRI32 && m.nv <= 32 && !dual) {
			MJB_KEEP_BRANCH();
			solve_tri32<G>(m, e, f + L.qacc_smooth, f + L.qLD, f + L.qLDiagInv, f + L.tri);
		} else
			solve2<G>(m, e, f + L.qacc_smooth, f + L.qLD, f + L.qLDiagInv, f + L.eulerx, f + L.qH, f + L.qHdi, dual);
	} else
		solve2<G>(m, e, f + L.qacc_smooth, f + L.qLD, f + L.qLDiagInv, f + L.eulerx, f + L.qH, f + L.qHdi, dual);
}

#include "mjb_constraint.h"

// A13 constraint solve when the model has no constraint rows: the unconstrained acceleration is the answer
template <int G> STAGE void fwd_constraint(CModel m, CLayout L, const Env &e)
{
	double *f = e.f;
	for (int d = e.lane; d < m.nv; d += G) {
		const double a = f[L.qacc_smooth + d];
		f[L.qacc + d] = a;
		f[L.qacc_warmstart + d] = a;
		f[L.qfrc_constraint + d] = 0;
	}
	if (e.lane == 0) {
		e.fi[L.ncon] = 0;
		e.fi[L.nefc] = 0;
		e.fi[L.solver_iter] = 0;
	}
	gsync<G>();
}

// ------------------------------------------------------------------------------------------------
// A16 semi-implicit Euler with implicit joint damping
// ------------------------------------------------------------------------------------------------
template <int G, bool CAN16, bool TRI32 = false, bool JC = false, bool PRE = false> STAGE void euler(CModel m, CLayout L, const Env &e)
{
	double *f = e.f;
	const double dt = m.timestep[0];
	double *x = f + L.eulerx;
	if (m.eulerdamp && m.nefcmax == 0) {
		// (M + h B) x = qfrc_smooth was already solved next to qacc_smooth (fwd_acceleration, dual solve)
	} else if (PRE && m.eulerdamp && m.nv <= 16) {
		// ... or next to qacc, behind the PGS stage (forward_rest, MJB_PGS_PRESOLVE)
	} else if (m.eulerdamp) {
		// (M + h B) x = qfrc_smooth + qfrc_constraint, factor qH prepared next to qLD in fwd_position
		for (int d = e.lane; d < m.nv; d += G) x[d] = f[L.qfrc_smooth + d] + f[L.qfrc_constraint + d];
		gsync<G>();
		if constexpr (CAN16) {
			if (m.nv <= 16) {
				int dl[16];
				dadr_load(e, L, dl);
				solve_dense16<G, 16>(m, e, x, f + L.qH, f + L.qHdi, x, f + L.qH, f + L.qHdi, false, dl, f + L.solvescr);
			} else if (TRI32 && m.nv <= 32) {
				MJB_KEEP_BRANCH();
				solve_tri32<G>(m, e, x, f + L.qH, f + L.qHdi, f + L.tri);
			} else
				solve<G>(m, e, x, f + L.qH, f + L.qHdi);
		} else
			solve<G>(m, e, x, f + L.qH, f + L.qHdi);
	} else {
		for (int d = e.lane; d < m.nv; d += G) x[d] = f[L.qacc + d];
		gsync<G>();
	}
	for (int d = e.lane; d < m.nv; d += G) f[L.qvel + d] += dt * x[d];
	gsync<G>();
	for (int j = e.lane; j < m.njnt; j += G) {
		const int jt = JC ? e.lc.j_type : m.jnt_type[j];  // (JC: the lane's joint constants sit in registers, njnt <= G)
		int pa = JC ? e.lc.j_qa : m.jnt_qposadr[j], va = JC ? e.lc.j_da : m.jnt_dofadr[j];
		if (jt == MJB_JNT_HINGE || jt == MJB_JNT_SLIDE) {
			f[L.qpos + pa] += dt * f[L.qvel + va];
		} else {
			if (jt == MJB_JNT_FREE) {
				for (int k = 0; k < 3; k++) f[L.qpos + pa + k] += dt * f[L.qvel + va + k];
				pa += 3;
				va += 3;
			}
			double q[4], w[3];
			ld4(q, f + L.qpos + pa);
			ld3(w, f + L.qvel + va);
			quat_integrate(q, w, dt);
			st4(f + L.qpos + pa, q);
		}
	}
	if (m.na > 0) advance_act<G>(m, f + L.act, f + L.act, f + L.act_dot, dt, e.lane);
	if (e.lane == 0) f[L.time] += dt;
	gsync<G>();
}

// mj_RungeKutta(m, d, 4) around the step loop's ONE copy of the forward stages (oracle/mjo_smooth.c mjo_rk4 has the scheme): called after
// evaluation rk = 0 .. 3 of a step, it folds F_rk = (qvel, qacc) into the weighted sums and either sets the state of evaluation
// rk + 1 (X0 advanced by h with a_{rk+1} F_rk, positions through the quaternion-aware integration) or, after the last one,
// advances X0 by h with the sums.  L.rk: q0 [nq] | v0 [nv] | sum B qvel [nv] | sum B qacc [nv] | warmstart [nv] | sensordata [S] | t0 | act0 [na] | sum B act_dot [na].
template <int G> STAGE void rk4_stage(CModel m, CLayout L, const Env &e, int rk)
{
	double *f = e.f;
	const int nq = m.nq, nv = m.nv, ns = m.nsensordata;
	double *q0 = f + L.rk, *v0 = q0 + nq, *accv = v0 + nv, *acca = accv + nv, *w0 = acca + nv, *sens = w0 + nv, *t0 = sens + ns;
	double *a0 = t0 + 1, *acct = a0 + m.na;
	const double h = m.timestep[0];
	const double B = (rk == 0 || rk == 3) ? 1.0 / 6.0 : 1.0 / 3.0;
	if (rk == 0) {
		for (int k = e.lane; k < nq; k += G) q0[k] = f[L.qpos + k];
		for (int k = e.lane; k < ns; k += G) sens[k] = f[L.sensordata + k];  // (the sub-stage evaluations skip the sensors in the reference)
		if (e.lane == 0) t0[0] = f[L.time];
	}
	for (int d = e.lane; d < nv; d += G) {
		if (rk == 0) v0[d] = f[L.qvel + d];
		accv[d] = (rk == 0 ? 0.0 : accv[d]) + B * f[L.qvel + d];
		acca[d] = (rk == 0 ? 0.0 : acca[d]) + B * f[L.qacc + d];
	}
	for (int k = e.lane; k < m.na; k += G) {
		if (rk == 0) a0[k] = f[L.act + k];
		acct[k] = (rk == 0 ? 0.0 : acct[k]) + B * f[L.act_dot + k];
	}
	gsync<G>();
	const bool last = rk == 3;
	const double a = rk == 2 ? 1.0 : 0.5;
	if (m.na > 0) {  // activations: X_i = act0 + h a F_{i-1} unclamped; the final advance clamps (mj_advance)
		if (last) advance_act<G>(m, f + L.act, a0, acct, h, e.lane);
		else
			for (int k = e.lane; k < m.na; k += G) f[L.act + k] = a0[k] + h * (0.0 + a * f[L.act_dot + k]);
	}
	// positions first: they read the velocity of evaluation rk (a F_rk) or the weighted sum, before qvel is overwritten
	for (int j = e.lane; j < m.njnt; j += G) {
		const int jt = m.jnt_type[j];
		int pa = m.jnt_qposadr[j], va = m.jnt_dofadr[j];
		if (jt == MJB_JNT_HINGE || jt == MJB_JNT_SLIDE) {
			const double v = last ? accv[va] : 0.0 + a * f[L.qvel + va];
			f[L.qpos + pa] = q0[pa] + h * v;
		} else {
			if (jt == MJB_JNT_FREE) {
				for (int k = 0; k < 3; k++) {
					const double v = last ? accv[va + k] : 0.0 + a * f[L.qvel + va + k];
					f[L.qpos + pa + k] = q0[pa + k] + h * v;
				}
				pa += 3;
				va += 3;
			}
			double q[4], w[3];
			ld4(q, q0 + pa);
			for (int k = 0; k < 3; k++) w[k] = last ? accv[va + k] : 0.0 + a * f[L.qvel + va + k];
			quat_integrate(q, w, h);
			st4(f + L.qpos + pa, q);
		}
	}
	gsync<G>();
	for (int d = e.lane; d < nv; d += G) {
		f[L.qvel + d] = v0[d] + h * (last ? acca[d] : 0.0 + a * f[L.qacc + d]);
		if (!last) f[L.qacc_warmstart + d] = w0[d];  // every evaluation of the step starts from the warmstart the step came in with
	}
	if (last) {
		for (int k = e.lane; k < ns; k += G) f[L.sensordata + k] = sens[k];
	}
	gsync<G>();
	// (mj_RungeKutta evaluates stage i at d->time = t0 + c_i h, c = 1/2, 1/2, 1: what a control / passive callback fired from that
	//  evaluation reads; nothing in the engine's own arithmetic depends on it)
	if (e.lane == 0) f[L.time] = t0[0] + (rk < 2 ? 0.5 : 1.0) * h;
	gsync<G>();
}

// ------------------------------------------------------------------------------------------------
// state <-> HBM, frame <-> HBM workspace
// ------------------------------------------------------------------------------------------------
template <int G> DEVI void copy_in(double *dst, const double *src, int n, int lane)
{
	for (int k = lane; k < n; k += G) dst[k] = src[k];
}
template <int G> DEVI void copy_out(double *dst, const double *src, int n, int lane)
{
	for (int k = lane; k < n; k += G) dst[k] = src[k];
}

template <int G> STAGE void load_state(CModel m, CLayout L, CState s, const Env &e)
{
	const size_t env = (size_t)e.env;
	copy_in<G>(e.f + L.qpos, s.qpos + env * m.nq, m.nq, e.lane);
	copy_in<G>(e.f + L.qvel, s.qvel + env * m.nv, m.nv, e.lane);
	copy_in<G>(e.f + L.act, s.act + env * m.na, m.na, e.lane);
	copy_in<G>(e.f + L.ctrl, s.ctrl + env * m.nu, m.nu, e.lane);
	copy_in<G>(e.f + L.qacc_warmstart, s.qacc_warmstart + env * m.nv, m.nv, e.lane);
	copy_in<G>(e.f + L.qfrc_applied, s.qfrc_applied + env * m.nv, m.nv, e.lane);
	if (s.use_xfrc) copy_in<G>(e.f + L.xfrc_applied, s.xfrc_applied + env * 6 * m.nbody, 6 * m.nbody, e.lane);
	copy_in<G>(e.f + L.ctrlnoise, s.ctrlnoise + env * m.nu, m.nu, e.lane);
	copy_in<G>(e.f + L.mocap_pos, s.mocap_pos + env * 3 * m.nmocap, 3 * m.nmocap, e.lane);
	copy_in<G>(e.f + L.mocap_quat, s.mocap_quat + env * 4 * m.nmocap, 4 * m.nmocap, e.lane);
	if (e.lane == 0) e.f[L.time] = s.time[env];
	// this env's gravity / geom friction: the model's values or the per-env overrides (mjb_set_env_*)
	for (int k = e.lane; k < 3; k += G) e.f[L.gravity + k] = s.env_gravity ? s.env_gravity[env * 3 + k] : m.gravity[k];
	if (m.nconmax > 0 && L.gfriction >= 0)
		for (int k = e.lane; k < 3 * m.ngeom; k += G)
			e.f[L.gfriction + k] = s.env_geom_friction ? s.env_geom_friction[env * 3 * m.ngeom + k] : m.geom_friction[k];
	// equality parameters (active | data | solref | solimp): the model's or this env's (mjb_set_env_equality)
	for (int k = e.lane; k < 19 * m.neq; k += G) {
		const int q = k / 19, j = k - 19 * q;
		double v;
		if (s.env_equality) v = s.env_equality[env * 19 * m.neq + k];
		else if (j == 0) v = m.eq_active[q] ? 1.0 : 0.0;
		else if (j < 12) v = m.eq_data[11 * q + j - 1];
		else if (j < 14) v = m.eq_solref[2 * q + j - 12];
		else v = m.eq_solimp[5 * q + j - 14];
		e.f[L.eqparam + k] = v;
	}
}

template <int G> STAGE void store_state(CModel m, CLayout L, CState s, const Env &e)
{
	const size_t env = (size_t)e.env;
	copy_out<G>(s.qpos + env * m.nq, e.f + L.qpos, m.nq, e.lane);
	copy_out<G>(s.qvel + env * m.nv, e.f + L.qvel, m.nv, e.lane);
	copy_out<G>(s.act + env * m.na, e.f + L.act, m.na, e.lane);
	copy_out<G>(s.ctrl + env * m.nu, e.f + L.ctrl, m.nu, e.lane);
	copy_out<G>(s.qacc_warmstart + env * m.nv, e.f + L.qacc_warmstart, m.nv, e.lane);
	copy_out<G>(s.qacc + env * m.nv, e.f + L.qacc, m.nv, e.lane);
	copy_out<G>(s.sensordata + env * m.nsensordata, e.f + L.sensordata, m.nsensordata, e.lane);
	copy_out<G>(s.ctrlnoise + env * m.nu, e.f + L.ctrlnoise, m.nu, e.lane);
	if (e.lane == 0) s.time[env] = e.f[L.time];
	if (m.enableflags & MJB_ENBL_ENERGY) copy_out<G>(s.energy + env * 2, e.f + L.energy, 2, e.lane);
}

// mj_checkPos / mj_checkVel / mj_checkAcc: NaN or |x| > mjMAXVAL -> flag in the int frame.  Returns 0 (fine), 1 (array a is
// bad) or 2 (only array b is bad): mj_step checks qpos first and resets on it, so a bad qpos hides a bad qvel.
template <int G> DEVI int any_bad(const Env &e, CLayout L, const double *a, int na, const double *b, int nb, const bool keep = false)
{
	// (keep: mj_checkAcc runs BEHIND the solver -- mjData.solver_iter is the step's iteration count when mj_step returns, so the word is put back)
	int *flag = e.fi + L.solver_iter;  // reused as a transient flag; rewritten by fwd_constraint
	const int old = *flag;
	if (e.lane == 0) *flag = 0;
	gsync<G>();
	bool bad_a = false, bad_b = false;
	for (int k = e.lane; k < na; k += G) bad_a |= !(a[k] == a[k]) || fabs(a[k]) > MJB_MAXVAL;
	for (int k = e.lane; k < nb; k += G) bad_b |= !(b[k] == b[k]) || fabs(b[k]) > MJB_MAXVAL;
	if (bad_b) *flag = 2;
	if (bad_a) *flag = 1;  // (lanes of a group run in lockstep: this store lands after the one above)
	gsync<G>();
	const int r = *flag;
	if (keep) {
		gsync<G>();
		if (e.lane == 0) *flag = old;
	}
	return r;
}

template <int G> __device__ __attribute__((noinline)) void reset_frame_state(CModel m, CLayout L, CState s, const EnvLite e, int warning)
{
	double *f = e.f;
	for (int k = e.lane; k < L.nstate; k += G) f[k] = 0;  // state prefix starts at offset 0
	gsync<G>();
	for (int k = e.lane; k < m.nq; k += G) f[L.qpos + k] = m.qpos0[k];
	for (int b = e.lane; b < m.nbody; b += G) {
		const int mid = m.body_mocapid[b];
		if (mid < 0) continue;
		for (int k = 0; k < 3; k++) f[L.mocap_pos + 3 * mid + k] = m.body_pos[3 * b + k];
		for (int k = 0; k < 4; k++) f[L.mocap_quat + 4 * mid + k] = m.body_quat[4 * b + k];
	}
	if (e.lane == 0) atomicAdd(s.nwarn + warning, 1ull);  // mjData.warning[mjWARN_BADQPOS / BADQVEL / BADQACC].number
	gsync<G>();
}

// ------------------------------------------------------------------------------------------------
// pipeline pieces
// ------------------------------------------------------------------------------------------------
// Every stage sees the model / layout through a freshly laundered copy of the parameter pointer: values derived
// from them (frame addresses, per-lane table entries) then live inside one stage only instead of being hoisted to
// the top of the step and kept in registers -- or spilled -- across all stages.  MJB_LAUNDER=0 disables it.
#ifndef MJB_LAUNDER
#define MJB_LAUNDER 1
#endif
#ifndef MJB_LAUNDER_DENSE
#define MJB_LAUNDER_DENSE 1  // (0: let the dense kernels hoist across stages -- measured no faster on MI355X)
#endif
template <bool ON> DEVI const KernelParams MJB_AS4 *launder_params(const KernelParams MJB_AS4 *p)
{
	if (MJB_LAUNDER && ON) asm volatile("" : "+s"(p));
	return p;
}
// (the lane index too: values derived from it -- lane * stride addresses, lane == k predicates of the unrolled register code -- are
//  otherwise hoisted out of the step loop to the top of the kernel and, at 256 registers, spilled there and reloaded from scratch at
//  every use: a trip to HBM in place of one integer instruction)
#define MJB_LAUNDER_LANE(e) do { } while (0)  // (the lane index launders itself: LaneId)
#define VIEW(P, compact, ...)                                              \
	do {                                                                   \
		const KernelParams MJB_AS4 *Pq_ = launder_params<MJB_LAUNDER_HERE>(P); \
		CModel m = Pq_->m;                                                 \
		CLayout L = (compact) ? Pq_->Lc : Pq_->L;                          \
		CState s = Pq_->s;                                                 \
		(void)s;                                                           \
		MJB_LAUNDER_LANE(e);                                               \
		__VA_ARGS__;                                                       \
	} while (0)

template <int G, int CON, int DENSE> DEVI void forward_first(const KernelParams MJB_AS4 *P, const Env &e, int compact)
{
	constexpr bool MJB_LAUNDER_HERE = MJB_LAUNDER_DENSE || DENSE == 0;
	[[maybe_unused]] CState s = P->s;  // (profiling macros)
	PROF_BEGIN();
	MJB_REP(0) VIEW(P, compact, kinematics<G, (G == 64 || DENSE != 0), (DENSE != 0)>(m, L, s, e));
	PROF(0);
	MJB_REP(1) VIEW(P, compact, com_pos<G, (DENSE != 0)>(m, L, e));
	PROF(1);
	MJB_REP(2) {  // (crb + factorisation together: the lean frame factorises M + h B in place)
	VIEW(P, compact, crb<G, (DENSE != 0)>(m, L, e));
	PROF(2);
	if constexpr (DENSE)
		VIEW(P, compact, factor_dense16<G, DENSE>(m, e, e.f + L.qM, e.f + L.qLD, e.f + L.qLDiagInv, e.f + L.MhB, e.f + L.qH,
		                                   e.f + L.qHdi, m.eulerdamp != 0, e.dadr, e.f + L.crbbuf));
	else if constexpr (CON != 0) {
		if (P->m.nv <= 16)
			VIEW(P, compact, {
				int dl[16];
				dadr_load(e, L, dl);
				factor_dense16<G, 16>(m, e, e.f + L.qM, e.f + L.qLD, e.f + L.qLDiagInv, e.f + L.MhB, e.f + L.qH, e.f + L.qHdi,
				                      m.eulerdamp != 0, dl, e.f + L.crbbuf);
			});
		else if ((CON >= 2 && CON <= 4) && P->m.nv <= 32)  // (the 512-register Newton kernels, like TRI32 below)
			VIEW(P, compact, {
				if (m.flv_n > 0) {
					MJB_KEEP_BRANCH();
					factor_levels<G>(m, e, e.f + L.qM, e.f + L.qLD, e.f + L.qLDiagInv, e.f + L.MhB, e.f + L.qH, e.f + L.qHdi, m.eulerdamp != 0);
				} else {
					MJB_KEEP_BRANCH();
					factor_dense32<G>(m, e, e.f + L.qM, e.f + L.qLD, e.f + L.qLDiagInv, e.f + L.MhB, e.f + L.qH, e.f + L.qHdi, m.eulerdamp != 0);
				}
			});
		else
			VIEW(P, compact, factor2<G>(m, e, e.f + L.qM, e.f + L.qLD, e.f + L.qLDiagInv, e.f + L.MhB, e.f + L.qH, e.f + L.qHdi,
			                            m.eulerdamp != 0));
	} else
		VIEW(P, compact, factor2<G>(m, e, e.f + L.qM, e.f + L.qLD, e.f + L.qLDiagInv, e.f + L.MhB, e.f + L.qH, e.f + L.qHdi,
		                            m.eulerdamp != 0));
	}
	PROF(3);
	if constexpr (CON) {
		MJB_REP(16) VIEW(P, compact, collision<G>(m, L, s, e));
		PROF(16);
	}
	// Constraint rows (make_constraint + J M^-1 + reference accelerations) read positions, contacts and qvel only and nothing
	// before the solver reads them: on the compact frame of the fused step they are built AFTER the velocity stage, where
	// efc_J can overlay what the position / velocity stages no longer need (mjb_api.hip, compute_layout); on the full frame
	// (mjb_forward / mjb_step1, whose callers may look at efc_* / contacts between the halves) in MuJoCo's place.  ONE copy of
	// the stages in the instruction stream: a two-trip loop picks the trip they run in.
	const int con_trip = (CON != 0 && compact) ? 1 : 0;
#pragma nounroll
	for (int trip = 0; trip < (CON != 0 ? 2 : 1); trip++) {
		if constexpr (CON != 0) {
			if (trip == con_trip) {
				PROF_BEGIN();
				MJB_REP(17) {
				VIEW(P, compact, make_constraint<G, CON>(m, L, s, e));
				PROF(17);
				if constexpr (CON == 1 || CON == 5 || CON == 9) {
					// (plain PGS, nv <= 16: the rows of B = J M^-1 are solved for inside the PGS stage, in registers)
					if (P->m.nv > 16) VIEW(P, compact, project_constraint<G, CON>(m, L, e));
					else if constexpr (CON == 5) VIEW(P, compact, project_constraint_dense16<G, CON>(m, L, e));
				}
				VIEW(P, compact, reference_constraint<G, CON>(m, L, s, e));
				}
				PROF(18);
			}
			if (trip) break;
		}
		PROF_BEGIN();
		MJB_REP(4) VIEW(P, compact, transmission<G, (DENSE != 0)>(m, L, e));
		VIEW(P, compact, sensors<G, (DENSE != 0)>(m, L, s, e, MJB_STAGE_POS, compact));
		PROF(4);
		MJB_REP(5) VIEW(P, compact, com_vel<G, (DENSE != 0)>(m, L, e));
		PROF(5);
		MJB_REP(6) VIEW(P, compact, passive<G, (DENSE != 0)>(m, L, e));
		PROF(6);
		MJB_REP(7) VIEW(P, compact, rne<G, (DENSE != 0)>(m, L, e));
		PROF(7);
		VIEW(P, compact, sensors<G, (DENSE != 0)>(m, L, s, e, MJB_STAGE_VEL, compact));
		PROF(8);
	}
}

template <int G, int CON, int DENSE> DEVI void forward_rest(const KernelParams MJB_AS4 *P, const Env &e, int compact)
{
	constexpr bool MJB_LAUNDER_HERE = MJB_LAUNDER_DENSE || DENSE == 0;
	[[maybe_unused]] CState s = P->s;  // (profiling macros)
	PROF_BEGIN();
	MJB_REP(9) VIEW(P, compact, fwd_actuation<G, (DENSE != 0)>(m, L, e));
	PROF(9);
	MJB_REP(10) VIEW(P, compact, fwd_acceleration<G, (CON != 0 ? -1 : DENSE), (CON >= 2 && CON <= 4)>(m, L, e, s.use_xfrc != 0));  // (TRI32: the Newton kernels -- in the PGS ones its 124 registers bring back the spill-before-exec-restore pattern)
	PROF(10);
	if constexpr (CON == 4 && G == 64) {
		// up to 256 rows.  The fused step's frame holds the first L.jrows (= 64) rows of efc_J: an env-step within that runs the
		// one-row-per-lane solver on it, one beyond reads J from the env's block in HBM; the full frame (mjb_forward / mjb_step1 /
		// mjb_step2) holds all of J
		VIEW(P, compact, {
			const int ne = __builtin_amdgcn_readfirstlane(e.fi[L.nefc]);
			if (s.rowstat != nullptr && e.lane == 0) {  // what the host's wide-frame policy reads after the launch
				atomicAdd(s.rowstat, 1ull);
				if (ne > 64) atomicAdd(s.rowstat + 1, 1ull);
				if (ne > 128) atomicAdd(s.rowstat + 2, 1ull);
			}
			if (L.jrows >= m.nefcmax && ne > 64) {
				MJB_KEEP_BRANCH();
				fwd_constraint_newton<G, 4>(m, L, e);
			} else if (ne <= L.jrows && ne <= 64) {  // (the full frame too: the same instantiation as the fused step's common case)
				MJB_KEEP_BRANCH();
				fwd_constraint_newton<G, 1>(m, L, e);
			} else if (ne <= L.jrows && ne <= 128) {  // the wide fused frame: up to 128 rows in LDS, two per lane
				MJB_KEEP_BRANCH();
				fwd_constraint_newton<G, 2>(m, L, e);
			} else {
				MJB_KEEP_BRANCH();
				fwd_constraint_newton<G, 4, false, true>(m, L, e, s.efc_Jg + (size_t)e.env * s.efc_Jg_stride);
			}
		});
	} else if constexpr (CON >= 2 && CON <= 3 && G == 64) {
		VIEW(P, compact, fwd_constraint_newton<G, (CON == 2 ? 1 : 2)>(m, L, e));
	} else if constexpr (CON >= 6 && CON <= 8 && G == 64) {
		VIEW(P, compact, fwd_constraint_newton<G, (CON == 6 ? 1 : (CON == 7 ? 2 : 4)), true>(m, L, e));
	} else if constexpr ((CON == 1 || CON == 5 || CON == 9) && G == 64) {
		if constexpr (CON == 5) {  // elliptic cone blocks: rows of B in LDS (the block code leaves no registers for them)
			VIEW(P, compact, fwd_constraint_pgs<G, true, false, CON>(m, L, s, e));
		} else {
			if (P->m.nv <= 16) {
				VIEW(P, compact, fwd_constraint_pgs<G, false, true, CON>(m, L, s, e));
				if constexpr (MJB_PGS_PRESOLVE) MJB_REP(25) {
					// qacc = qacc_smooth + M^-1 qfrc_constraint, and -- under implicit joint damping -- Euler's (M + h B)^-1 (qfrc_smooth + qfrc_constraint)
					// beside it: one dual substitution (both factors were built together in fwd_position); Euler finds its vector solved
					VIEW(P, compact, {
						double *f = e.f;
						const bool damp = m.eulerdamp != 0;
						if (e.lane < m.nv) {
							const double c = f[L.qfrc_constraint + e.lane];
							f[L.qacc + e.lane] = c;
							f[L.eulerx + e.lane] = f[L.qfrc_smooth + e.lane] + c;
						}
						gsync<G>();
						int dl[16];
						dadr_load(e, L, dl);
						solve_dense16<G, 16>(m, e, f + L.qacc, f + L.qLD, f + L.qLDiagInv, f + L.eulerx, f + L.qH, f + L.qHdi, damp, dl, f + L.solvescr);
						if (e.lane < m.nv) {
							const double a = f[L.qacc_smooth + e.lane] + f[L.qacc + e.lane];
							f[L.qacc + e.lane] = a;
							f[L.qacc_warmstart + e.lane] = a;
						}
						gsync<G>();
					});
				}
			} else
				VIEW(P, compact, fwd_constraint_pgs_ldsB<G, false, CON>(m, L, s, e));
		}
	} else {
		VIEW(P, compact, fwd_constraint<G>(m, L, e));
	}
	PROF(11);
	if constexpr (CON != 0) {  // workload statistics (mjb_set_stats): what this evaluation asked of the solver -- fire-and-forget atomics
		if (P->s.stats != nullptr) {
			MJB_KEEP_BRANCH();
			VIEW(P, compact, {
				if (e.lane == 0) {
					const int ne = e.fi[L.nefc], nc = m.nconmax > 0 ? e.fi[L.ncon] : 0;
					unsigned long long *st = P->s.stats;
					atomicAdd(st, 1ull);
					atomicAdd(st + 1, (unsigned long long)e.fi[L.solver_iter]);
					atomicAdd(st + 2 + (ne < 256 ? ne : 256), 1ull);
					atomicAdd(st + 259 + (nc < 128 ? nc : 128), 1ull);
				}
			});
		}
	}
	VIEW(P, compact, if (m.need_rnepost) rne_post<G>(m, L, lite(e), s.use_xfrc != 0));
	VIEW(P, compact, sensors<G, (DENSE != 0)>(m, L, s, e, MJB_STAGE_ACC, compact));
	PROF(12);
}

// ---- angles:: helpers of the reference's ros_control bridge (ROS package `angles`, a dependency absent from
// /root/reference; restated from its published header angles/angles.h) ----
DEVI double angdist(double from, double to)  // shortest_angular_distance
{
	const double two_pi = 6.283185307179586476925;
	double a = fmod(fmod(to - from, two_pi) + two_pi, two_pi);  // normalize_angle_positive
	if (a > 0.5 * two_pi) a -= two_pi;
	return a;
}
DEVI double two_pi_complement(double a)
{
	const double two_pi = 6.283185307179586476925;
	if (a > two_pi || a < -two_pi) a = fmod(a, two_pi);
	if (a < 0) return two_pi + a;
	if (a > 0) return -two_pi + a;
	return two_pi;
}
DEVI bool find_min_max_delta(double from, double left, double right, double &dmin, double &dmax)
{
	const double pi = 3.14159265358979323846;
	const double d0 = angdist(from, left), d1 = angdist(from, right), d2 = two_pi_complement(d0), d3 = two_pi_complement(d1);
	if (d0 == 0) {
		dmin = d0;
		dmax = fmax(d1, d3);
		return true;
	}
	if (d1 == 0) {
		dmax = d1;
		dmin = fmin(d0, d2);
		return true;
	}
	double lo = d0, lo2 = d2, hi = d1, hi2 = d3;
	if (d2 < lo) { lo = d2; lo2 = d0; }
	if (d3 > hi) { hi = d3; hi2 = d1; }
	if (lo <= hi2 || hi >= lo2) {
		dmin = hi2;
		dmax = lo2;
		return left == -pi && right == pi;
	}
	dmin = lo;
	dmax = hi;
	return true;
}
DEVI double angdist_with_limits(double from, double to, double left, double right)  // shortest_angular_distance_with_limits
{
	const double two_pi = 6.283185307179586476925;
	double dmin = -two_pi, dmax = two_pi, tmin = -two_pi, tmax = two_pi;
	const bool inside = find_min_max_delta(from, left, right, dmin, dmax);
	const double delta = angdist(from, to), comp = two_pi_complement(delta);
	if (inside) {
		if (delta >= dmin && delta <= dmax) return delta;
		if (comp >= dmin && comp <= dmax) return comp;
		find_min_max_delta(to, left, right, tmin, tmax);
		if (fabs(tmin) < fabs(tmax)) return fmax(delta, comp);
		if (fabs(tmin) > fabs(tmax)) return fmin(delta, comp);
		return fabs(delta) < fabs(comp) ? delta : comp;
	}
	find_min_max_delta(to, left, right, tmin, tmax);
	if (fabs(dmin) < fabs(dmax)) return fmin(delta, comp);
	if (fabs(dmin) > fabs(dmax)) return fmax(delta, comp);
	return fabs(delta) < fabs(comp) ? delta : comp;
}

// mj_energyPos / mj_energyVel (mjENBL_ENERGY): potential = -sum m g.xipos + joint / tendon spring energy, kinetic =
// 0.5 qvel' M qvel.  Evaluated for the LAST step of a launch only (intermediate values are not observable), by one lane:
// a few hundred dependent LDS reads once per launch.
template <int G> __device__ __attribute__((noinline)) void energy(CModel m, CLayout L, const EnvLite e)
{
	double *f = e.f;
	if (e.lane == 0) {
		double pe = 0, ke = 0;
		if (!(m.disableflags & MJB_DSBL_GRAVITY))
			for (int b = 1; b < m.nbody; b++)
				pe -= MP_BODY_MASS(m, e, b) * (f[L.gravity] * f[L.xipos + 3 * b] + f[L.gravity + 1] * f[L.xipos + 3 * b + 1] +
				                               f[L.gravity + 2] * f[L.xipos + 3 * b + 2]);
		if (!(m.disableflags & MJB_DSBL_PASSIVE)) {
			for (int j = 0; j < m.njnt; j++) {
				const double k = m.jnt_stiffness[j];
				if (k == 0) continue;
				int pa = m.jnt_qposadr[j];
				const int jt = m.jnt_type[j];
				if (jt == MJB_JNT_FREE) {
					for (int c = 0; c < 3; c++) {
						const double dq = f[L.qpos + pa + c] - m.qpos_spring[pa + c];
						pe += 0.5 * k * dq * dq;
					}
					pa += 3;
				}
				if (jt == MJB_JNT_FREE || jt == MJB_JNT_BALL) {
					double q[4], qs[4], dif[3];
					ld4(q, f + L.qpos + pa);
					normalize4(q);
					ldc4(qs, m.qpos_spring + pa);
					quat_sub(dif, q, qs);
					pe += 0.5 * k * dot3(dif, dif);
				} else {
					const double dq = f[L.qpos + pa] - m.qpos_spring[pa];
					pe += 0.5 * k * dq * dq;
				}
			}
			for (int t = 0; t < m.ntendon; t++) {
				const double dl = f[L.ten_length + t] - m.tendon_lengthspring[t];
				pe += 0.5 * m.tendon_stiffness[t] * dl * dl;
			}
		}
		for (int en = 0; en < m.nM; en++) {
			const int i = m.M_rowdof[en], j = m.M_coldof[en];
			ke += (i == j ? 0.5 : 1.0) * f[L.qM + en] * f[L.qvel + i] * f[L.qvel + j];
		}
		f[L.energy] = pe;
		f[L.energy + 1] = ke;
	}
	gsync<G>();
}

// device-side DefaultRobotHWSim::writeSim (include/mjb.h, mjb_hwsim_*): lane = controlled joint
// ros::Time(double).toNSec(): sec = floor(t), nsec = round((t - sec) 1e9)
DEVI long long ros_ns(double t)
{
	const double sec = floor(t);
	return (long long)sec * 1000000000LL + (long long)floor((t - sec) * 1e9 + 0.5);
}

template <int G> __device__ __attribute__((noinline)) void hwsim_write(CModel m, CLayout L, const HwSim MJB_AS4 &hw, const EnvLite e)
{
	double *f = e.f;
	double dt = m.timestep[0];
	const bool estop = hw.estop != 0;
	// MujocoRosControlPlugin::controlCallback around writeSim (mujoco_ros_control_plugin.cpp:153-194; oracle/mjo_hwsim.c
	// mjo_hwsim_control_callback): the stamps are per env, every lane derives the same decisions from them
	const bool cadence = hw.period_ns > 0;
	double *cad = cadence ? hw.cad + (size_t)e.env * (2 + 2 * hw.n) : nullptr;
	bool update = false;
	long long lu = 0, lw = 0, t = 0;
	if (cadence) {
		t = ros_ns(f[L.time]);
		lu = (long long)cad[0];
		lw = (long long)cad[1];
		if (t < lu) {  // the time went backwards (reset): both stamps re-armed (:160-169)
			lu = t;
			lw = t;
		}
		const long long sim_period = t - lu;
		update = sim_period >= hw.period_ns || (lu == 0 && sim_period != 0);  // (:171-176; nothing happens at t = 0)
		if (update) lu = t;
		const bool write = lu != 0 && t > lw;                                 // (:190-193)
		if (update) {  // readSim (default_robot_hw_sim.cpp:229-245): the joint state the PIDs see until the next update
			for (int k = e.lane; k < hw.n; k += G) {
				const int j = hw.joint[k];
				const double position = f[L.qpos + m.jnt_qposadr[j]];
				double *jp = cad + 2 + k;
				jp[0] = hw.kind[k] == MJB_HW_PRISMATIC ? position : jp[0] + angdist(jp[0], position);
				jp[hw.n] = f[L.qvel + m.jnt_dofadr[j]];
			}
		}
		gsync<G>();  // (every lane has read the stamps)
		if (e.lane == 0) {
			cad[0] = (double)lu;
			cad[1] = (double)(write ? t : lw);
		}
		if (!write) {
			gsync<G>();
			return;
		}
		dt = 1e-9 * (double)(t - lw);
	}
	for (int k = e.lane; k < hw.n; k += G) {
		const int j = hw.joint[k], method = hw.method[k], qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
		const size_t at = (size_t)e.env * hw.n + k;
		const double *gn = hw.gains + 8 * k;
		const double pos = cadence ? cad[2 + k] : f[L.qpos + qa], vel = cadence ? cad[2 + hw.n + k] : f[L.qvel + da];
		const double cpos = estop ? hw.cmd_hold[at] : hw.cmd_pos[at];
		double error = 0;
		bool pid = false;
		switch (method) {
		case MJB_HW_EFFORT: f[L.qfrc_applied + da] = estop ? 0.0 : hw.cmd_eff[at]; break;
		case MJB_HW_POSITION:
			f[L.qpos + qa] = cpos;
			f[L.qvel + da] = 0;
			f[L.qfrc_applied + da] = 0;
			break;
		case MJB_HW_VELOCITY:
			f[L.qvel + da] = estop ? 0.0 : hw.cmd_vel[at];
			f[L.qfrc_applied + da] = 0;
			break;
		case MJB_HW_POSITION_PID: {
			const int kind = hw.kind[k];
			if (kind == MJB_HW_REVOLUTE) {
				// position command saturated to the joint limits (pj_sat_interface_.enforceLimits, :263), then the error the
				// reference takes with angles::shortest_angular_distance_with_limits (:289-291)
				const bool lim = gn[7] > gn[6];
				const double c = lim ? fmin(fmax(cpos, gn[6]), gn[7]) : cpos;
				error = lim ? angdist_with_limits(pos, c, gn[6], gn[7]) : c - pos;
			} else if (kind == MJB_HW_CONTINUOUS) {
				error = angdist(pos, cpos);
			} else {
				error = cpos - pos;
			}
			pid = true;
			break;
		}
		case MJB_HW_VELOCITY_PID:
			error = estop ? -vel : hw.cmd_vel[at] - vel;
			pid = true;
			break;
		default: break;
		}
		if (pid) {
			double ierr = hw.pid[2 * at], last = hw.pid[2 * at + 1];
			const double derr = (error - last) / dt;
			ierr += dt * error;
			if (hw.antiwindup[k] && gn[1] != 0) {
				const double lo = gn[4] / fabs(gn[1]), hi = gn[3] / fabs(gn[1]);
				ierr = fmin(fmax(ierr, lo), hi);
			}
			double iterm = gn[1] * ierr;
			if (!hw.antiwindup[k]) iterm = fmin(fmax(iterm, gn[4]), gn[3]);
			double cmd = gn[0] * error + iterm + gn[2] * derr;
			if (gn[5] > 0) cmd = fmin(fmax(cmd, -gn[5]), gn[5]);
			hw.pid[2 * at] = ierr;
			hw.pid[2 * at + 1] = error;
			f[L.qfrc_applied + da] = cmd;
		}
	}
	gsync<G>();
}

// (zpre: the lane's normal of this step was fetched from the launch's pre-generated buffer a step ago -- mjb_noise_kernel, same
//  function, same key: the ~2 k cycles of Philox + Box-Muller leave the step's dependent chain)
template <int G> STAGE void ctrl_noise(CModel m, CLayout L, CNoise nz, const Env &e,
                                      unsigned int step, bool zpre = false, double zval = 0)
{
	if (zpre) {  // (a real branch: as a select the compiler evaluates the generator speculatively)
		MJB_KEEP_BRANCH();
		if (e.lane < m.nu) {
			const int i = e.lane;
			const double v = nz.rate * e.f[L.ctrlnoise + i] + nz.scale * zval;
			e.f[L.ctrlnoise + i] = v;
			e.f[L.ctrl + i] = v;
		}
	} else {
		MJB_KEEP_BRANCH();
		for (int i = e.lane; i < m.nu; i += G) {
			const double z = philox_normal(nz.seed, (unsigned long long)(nz.env_offset + e.env), step, (unsigned int)i);
			const double v = nz.rate * e.f[L.ctrlnoise + i] + nz.scale * z;
			e.f[L.ctrlnoise + i] = v;
			e.f[L.ctrl + i] = v;
		}
	}
	gsync<G>();
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// CON: 0 = model without constraint rows; 1 = PGS (5 = PGS with elliptic cone blocks), 2 / 3 / 4 = Newton with 1 / 2 / 4 rows per lane,
// 6 / 7 / 8 = CG with 1 / 2 / 4 rows per lane (the Newton solver without its Hessian) (collision / rows / solver stages compiled in;
// one env per wavefront) -- separate kernels keep each instruction stream and register budget small.
// Constrained kernels get the full 512-register budget (one wave per SIMD; most frames limit the CU to 1 - 4 envs anyway):
// few spills, and room for the register-resident AR rows / Hessian rows of the solvers.  CON == 9 is the plain PGS step again
// under a 256-register cap, picked when EIGHT lean frames fit one CU's LDS: two waves per SIMD hide each other's dependent
// chains, which is worth more than the spills cost (config 3: +31 % measured).
// ROCm 7.2's LLVM can place a spill ahead of an exec restore and lose lanes (tools/check_spill_exec.py, `make lint`
// guards every build): an earlier revision had to cap these kernels at 256 VGPRs because of it, and the CG variants (CON >= 6,
// not a BASELINE workload) still are -- at 512 the allocator produced exactly that pattern in the 2-rows-per-lane CG kernel.
template <int G, int CON, int DENSE>
#ifndef MJB_DEV_OCC
#define MJB_DEV_OCC 1
#endif
__global__ void __launch_bounds__(256, (CON >= 6 ? 2 : (CON ? MJB_DEV_OCC : (G == 64 ? 4 : (G == 32 ? 2 : 1)))))
    mjb_step_kernel(const KernelParams MJB_AS4 *__restrict__ P, const int mode_arg, const int nsteps,
                    const unsigned int step0, const int epb, const int frame_bytes, const int chunk, const int env_lo, const int env_hi)
{
	// [env_lo, env_hi): the envs this launch steps (the whole batch, or -- split steps of the host runtime -- the callback envs /
	// the rest; mjb_step1_prefix, mjb_step_rest, mjb_step2_prefix)
	// launch parameters live in device memory behind a constant-address-space pointer: every field is
	// fetched with a scalar load where it is used instead of pinning ~300 SGPRs for the whole kernel
	const DevModel MJB_AS4 &m = P->m;
	const int mode = mode_arg;
	const int compact = (mode == MJB_MODE_STEP && P->use_compact) ? P->use_compact : 0;  // 0 full frame, 1 fused, 2 wide fused (the host swapped it into P->Lc)
	const FrameLayout MJB_AS4 &L = compact ? P->Lc : P->L;
	const DevState MJB_AS4 &s = P->s;
	const NoiseCfg MJB_AS4 &nz = P->nz;
	constexpr bool MJB_LAUNDER_HERE = MJB_LAUNDER_DENSE || DENSE == 0;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int slot = threadIdx.x / G;
	Env e;
#ifdef MJB_PROFILE
	e.prof = s.prof;
	if (threadIdx.x == 0) mjb_prof_lds = ProfLds{ { 0, 0 }, { 0, 0 }, (unsigned int)s.prof_base, 0 };  // (every block: with the work queue env 0's items run wherever they land)
	__syncthreads();
#endif
	e.lane.mask = G - 1;  // (1-D blocks of whole wavefronts: lane in group = hardware lane id & (G - 1))
	if constexpr (DENSE) {
#pragma unroll
		for (int i = 0; i < 16; i++) e.dadr[i] = m.M_dense[16 * i + e.lane];
	}
	// (constrained kernels: the map lives packed in the frame's int area, see dadr_load)
	if constexpr (DENSE != 0) {  // (tried in the constrained kernels too: no gain, the PGS kernel loses 8 % to register pressure)
		const int b = e.lane < m.nbody ? e.lane : 0;
		LaneConst &c = e.lc;
		c.dmlo = (unsigned int)m.body_dofmask[2 * b]; c.dmhi = (unsigned int)m.body_dofmask[2 * b + 1];
		c.smlo = (unsigned int)m.body_submask[2 * b]; c.smhi = (unsigned int)m.body_submask[2 * b + 1];
		c.jntadr = m.body_rec2[4 * b]; c.jntnum = m.body_rec2[4 * b + 1]; c.simple = m.body_rec2[4 * b + 2];
		const int j = (b && c.jntnum > 0) ? c.jntadr : 0;
		c.jtype = m.njnt ? m.jnt_type[j] : 0;
		c.qa = m.njnt ? m.jnt_qposadr[j] : 0;
		c.q0 = m.njnt ? m.qpos0[c.qa] : 0.0;
		for (int k = 0; k < 3; k++) {
			c.bpos[k] = m.body_pos[3 * b + k];
			c.ipos[k] = m.body_ipos[3 * b + k];
			c.jaxis[k] = m.njnt ? m.jnt_axis[3 * j + k] : 0.0;
			c.jpos[k] = m.njnt ? m.jnt_pos[3 * j + k] : 0.0;
		}
		for (int k = 0; k < 4; k++) {
			c.bquat[k] = m.body_quat[4 * b + k];
			c.iquat[k] = m.body_iquat[4 * b + k];
		}
		c.rootid = m.body_rootid[b];
		c.mass = m.body_mass[b];
		c.stmass = m.body_subtreemass[b];
		for (int k = 0; k < 3; k++) c.inertia[k] = m.body_inertia[3 * b + k];
		const int dd = e.lane < m.nv ? e.lane : 0;
		c.d_body = m.nv ? m.dof_bodyid[dd] : 0;
		c.d_parent = m.body_parentid[c.d_body];
		const int dj = m.nv ? m.dof_jntid[dd] : 0;
		c.d_zero = (m.nv && m.jnt_type[dj] == MJB_JNT_FREE && dd - m.jnt_dofadr[dj] < 3) ? 1 : 0;
		c.d_simple = (m.nv && m.dof_jstart[dd] == m.body_dofadr[c.d_body]) ? 1 : 0;
		c.d_bmlo = m.nv ? (unsigned int)m.dof_bodymask[2 * dd] : 0u;
		c.d_bmhi = m.nv ? (unsigned int)m.dof_bodymask[2 * dd + 1] : 0u;
		{  // lane = actuator
			const int i = (m.nu && e.lane < m.nu) ? (int)e.lane : 0;
			const int tj = (m.nu && m.actuator_trntype[i] == MJB_TRN_JOINT) ? m.actuator_trnid[2 * i] : 0;  // (tendon transmissions do not use this cache: m.act_tendon)
			c.u_qa = m.nu ? m.jnt_qposadr[tj] : 0;
			c.u_da = m.nu ? m.jnt_dofadr[tj] : 0;
			c.u_gear = m.nu ? m.actuator_gear[6 * i] : 0.0;
		}
		{  // lane = dof: its actuator
			const int t0 = m.nv ? m.dof_act_adr[dd] : 0, t1 = m.nv ? m.dof_act_adr[dd + 1] : 0;
			c.a_n = (m.na > 0 && t1 > t0) ? 2 : t1 - t0;  // (stateful actuators: the table walk, which knows about activations)
			const int i = c.a_n > 0 ? m.dof_act_id[t0] : 0;
			c.a_id = i;
			const bool has = c.a_n > 0 && m.nu > 0;
			c.a_flags = 0;
			if (has && m.actuator_ctrllimited[i] && !(m.disableflags & MJB_DSBL_CLAMPCTRL)) c.a_flags |= 1;
			if (has && m.actuator_gaintype[i] == MJB_GAIN_AFFINE) c.a_flags |= 2;
			if (has && m.actuator_biastype[i] == MJB_BIAS_AFFINE) c.a_flags |= 4;
			if (has && m.actuator_forcelimited[i]) c.a_flags |= 8;
			c.a_clo = has ? m.actuator_ctrlrange[2 * i] : 0.0;
			c.a_chi = has ? m.actuator_ctrlrange[2 * i + 1] : 0.0;
			for (int k = 0; k < 3; k++) {
				c.a_g[k] = has ? m.actuator_gainprm[3 * i + k] : 0.0;
				c.a_b[k] = has ? m.actuator_biasprm[3 * i + k] : 0.0;
			}
			c.a_flo = has ? m.actuator_forcerange[2 * i] : 0.0;
			c.a_fhi = has ? m.actuator_forcerange[2 * i + 1] : 0.0;
			c.a_gear = has ? m.dof_act_mom[t0] : 0.0;
		}
		{  // lane = joint
			const int jj = (m.njnt && e.lane < m.njnt) ? (int)e.lane : 0;
			c.j_type = m.njnt ? m.jnt_type[jj] : 0;
			c.j_qa = m.njnt ? m.jnt_qposadr[jj] : 0;
			c.j_da = m.njnt ? m.jnt_dofadr[jj] : 0;
			c.j_stiff = m.njnt ? m.jnt_stiffness[jj] : 0.0;
			c.j_spring = m.njnt ? m.qpos_spring[c.j_qa] : 0.0;
			c.j_damp = (m.njnt && m.nv) ? m.dof_damping[c.j_da] : 0.0;
		}
		{
			for (int r = 0; r < 6; r++) {  // (unconditional load at a clamped row + select: no divergent branch around a load into the cache)
				const int rr = r < m.kin_rounds + 2 ? r : m.kin_rounds + 1;
				const int v = m.body_anc[rr * m.nbody + b];
				c.anc[r] = (b != 0 && r < m.kin_rounds + 2) ? v : 0;
			}
			const int jj = (m.njnt && e.lane < m.njnt) ? (int)e.lane : 0;
			c.j_body = m.njnt ? m.jnt_bodyid[jj] : 0;
			c.j_root = m.body_rootid[c.j_body];
			const int nitem = m.njnt + m.ngeom + m.nsite, it = (int)e.lane;
			c.k_kind = -1;
			c.k_id = c.k_body = c.k_same = 0;
			for (int k = 0; k < 3; k++) c.k_pos[k] = 0;
			for (int k = 0; k < 4; k++) c.k_quat[k] = 0;
			if (nitem <= G && it < nitem) {
				if (it < m.njnt) {
					c.k_kind = 0;
					c.k_id = it;
					c.k_body = m.body_rec[4 * m.jnt_bodyid[it]];
					c.k_same = m.jnt_type[it] == MJB_JNT_FREE ? 1 : 0;
				} else {
					const bool isg = it < m.njnt + m.ngeom;
					const int id = isg ? it - m.njnt : it - m.njnt - m.ngeom;
					c.k_kind = isg ? 1 : 2;
					c.k_id = id;
					c.k_body = isg ? m.geom_bodyid[id] : m.site_bodyid[id];
					c.k_same = isg ? m.geom_sameframe[id] : m.site_sameframe[id];
					for (int k = 0; k < 3; k++) c.k_pos[k] = isg ? m.geom_pos[3 * id + k] : m.site_pos[3 * id + k];
					for (int k = 0; k < 4; k++) c.k_quat[k] = isg ? m.geom_quat[4 * id + k] : m.site_quat[4 * id + k];
				}
			} else if (nitem <= G) {
				c.k_kind = 3;  // (a lane without an item)
			}
			for (int q = 0; q < 3; q++) {
				const int en = (int)e.lane + G * q;
				const bool has = m.nM <= 3 * G && en < m.nM;
				const int i = has ? m.M_rowdof[en] : 0, j = has ? m.M_coldof[en] : 0;
				c.q_row[q] = has ? i : (m.nM <= 3 * G ? -1 : -2);
				c.q_col[q] = j;
				c.q_arm[q] = (has && i == j) ? m.dof_armature[i] : 0.0;
				c.q_hd[q] = (has && i == j) ? m.timestep[0] * m.dof_damping_int[i] : 0.0;
			}
		}
		for (int st = 0; st < 3; st++) {  // the sensors' plain copies: {dst, src} pairs of this launch's layout, two per lane and stage
			const int tb = (3 * compact + st) * m.sens_ncopy_max, nc = m.sens_ncopy[st];
			for (int q = 0; q < 2; q++) {
				const int t = (int)e.lane + G * q;
				c.sc_dst[st][q] = t < nc ? m.sens_copy[2 * (tb + t)] : -1;
				c.sc_src[st][q] = t < nc ? m.sens_copy[2 * (tb + t) + 1] : 0;
			}
		}
	}
	e.f = reinterpret_cast<double *>(smem + (size_t)slot * frame_bytes);
	e.fi = reinterpret_cast<int *>(e.f + L.ndouble);

	// Constrained kernels, long fused launches (chunk > 0): the K steps of an env are cut into chunks and every (chunk, env) pair is
	// a work item handed out from a counter in chunk-major order; an item waits for its env's previous chunk (taken nenv items
	// earlier by a block that is running or done, so the wait cannot deadlock) and the state crosses HBM between chunks exactly
	// as it does between launches.  An env's cost varies 2x around the mean and a CU holds few envs: with two envs per slot the
	// slowest pair sets the launch time, with 2 * nchunk items per slot the slots even out (config 3: +7 %).
	const bool dyn = CON != 0 && G == 64 && chunk > 0 && mode == MJB_MODE_STEP && s.sched != nullptr;  // (whole-batch launches only)
	const int nchunk = dyn ? (nsteps + chunk - 1) / chunk : 1;
	// (otherwise) grid-stride over env groups so any batch size runs with a bounded grid
	for (int base = env_lo + blockIdx.x * epb;; base += gridDim.x * epb) {
		int item_chunk = 0;
		if (dyn) {
			int w = 0;
			if (e.lane == 0) w = atomicAdd(s.sched, 1);
			w = __builtin_amdgcn_readfirstlane(w);
			if (w >= s.nenv * nchunk) break;
			item_chunk = w / s.nenv;
			e.env = w - item_chunk * s.nenv;
			if (item_chunk > 0) {
				int *done = s.sched + 1 + e.env;
				while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < item_chunk)
					__builtin_amdgcn_s_sleep(16);
				__threadfence();  // (the state the previous chunk stored is read through the vector cache)
			}
		} else {
			if (base >= env_hi) break;
			e.env = base + slot;
		}
		if constexpr (G == 64) e.env = __builtin_amdgcn_readfirstlane(e.env);  // (one env per wavefront: keep the index and everything derived from it scalar)
		if (e.env >= env_hi) continue;  // whole group idles together (group == slot)
		if constexpr (DENSE == 0) e.mp = s.env_mass ? s.env_mass + (size_t)e.env * (7 * m.nbody + m.nv + m.ntendon + 1) : nullptr;
		else e.mp = nullptr;  // (batches with per-env masses never run the dense kernels)
		double *ws = s.frame_ws ? s.frame_ws + (size_t)e.env * s.frame_stride : nullptr;

		if (mode == MJB_MODE_STEP2 || (DENSE == 0 && CON != 9 && (mode == MJB_MODE_STEP21 || mode == MJB_MODE_RKMID || mode == MJB_MODE_RKLAST))) {
			// resume: full frame from the workspace, then the (possibly host-modified) state on top
			// (eight loads in flight per lane: the copy is a chain of HBM round trips otherwise -- a split step of ONE callback env
			//  is pure latency, profiles/r03_callback_path.txt)
#pragma unroll 8
			for (int k = e.lane; k < L.ndouble; k += G) e.f[k] = ws[k];
			int *wsi = reinterpret_cast<int *>(ws + L.ndouble);
#pragma unroll 8
			for (int k = e.lane; k < L.nint; k += G) e.fi[k] = wsi[k];
			gsync<G>();
		} else {
			// derived part of the frame starts undefined; zero it once so dumps are deterministic
			if (mode != MJB_MODE_STEP)
				for (int k = e.lane; k < L.ndouble; k += G) e.f[k] = 0;
			for (int k = e.lane; k < L.nint; k += G) e.fi[k] = 0;
			gsync<G>();
		}
		if constexpr (CON != 0) {
			if (m.nv <= 16 && e.lane < 16) {
				for (int q = 0; q < 4; q++) {
					unsigned int w = 0;
					for (int r = 0; r < 4; r++) {
						const int a = m.M_dense[16 * (4 * q + r) + e.lane];
						w |= (unsigned int)(a < 0 ? 255 : a) << (8 * r);
					}
					e.fi[L.dadr + 4 * e.lane + q] = (int)w;
				}
			}
		}
		load_state<G>(m, L, s, e);
		gsync<G>();

		// One copy of every stage in the instruction stream (the whole kernel must stay I-cache
		// resident): modes only switch stage groups on and off.
		// (MJB_MODE_STEP21: the second half of one split step and the first half of the next in one launch -- two trips of the step
		//  loop, trip 0 = STEP2's stages, trip 1 = STEP1's; the flags below are trip-invariant in every other mode)
		// (not in the dense kernels nor in the 256-register PGS variant, which only ever runs fused launches: both lose 1 - 2.5 % to the
		//  extra mode; the host picks the generic / 512-register kernels for this launch)
		const int st0 = item_chunk * chunk;  // first step of this work item (0 unless the launch is chunked)
		const int nst = mode == MJB_MODE_STEP ? (dyn ? (nsteps - st0 < chunk ? nsteps - st0 : chunk) : nsteps) : ((DENSE == 0 && CON != 9 && mode == MJB_MODE_STEP21) ? 2 : 1);
		// (an RK4 step cut at its callback points: this launch starts at evaluation rk0 -- whose first half the previous launch ran --
		//  and stops after ONE rk4_stage, with the next evaluation's first half done)
		// ctrl noise from the launch's pre-generated buffer (one value per lane: nu <= G), first value fetched here
		int zhalf_i = -1;
		if (mode == MJB_MODE_STEP && nz.enabled && s.zbuf != nullptr && m.nu <= G) {
			if (s.zinfo[0] == step0 && (int)s.zinfo[1] == nsteps && (int)s.zinfo[2] == s.nenv) zhalf_i = 0;
			else if (s.zinfo[4] == step0 && (int)s.zinfo[5] == nsteps && (int)s.zinfo[6] == s.nenv) zhalf_i = 1;
		}
		const bool zpre = zhalf_i >= 0;
		const double *const zb = s.zbuf + (zhalf_i > 0 ? s.zhalf : 0ull);
		double znext = 0;
		if (zpre && e.lane < m.nu) znext = zb[((size_t)st0 * s.nenv + e.env) * m.nu + e.lane];
		const bool rksplit = DENSE == 0 && CON != 9 && (mode == MJB_MODE_RKMID || mode == MJB_MODE_RKLAST);
		const int rk0 = rksplit ? nsteps : 0;
#pragma nounroll
		for (int st = 0; st < nst; st++) {
			// (the launch mode re-laundered per step: the flags derived from it -- a dozen scalar masks -- are invariants of the step loop
			//  otherwise, hoisted to the top of the kernel and kept in spilled SGPRs)
			int mode = mode_arg;
			asm volatile("" : "+s"(mode));
			const bool combo = DENSE == 0 && CON != 9 && mode == MJB_MODE_STEP21;
			const bool checks = mode != MJB_MODE_FORWARD;
			const bool rkmode = mode == MJB_MODE_RKMID || mode == MJB_MODE_RKLAST;
			const bool do_first = combo ? st == 1 : (mode != MJB_MODE_STEP2 && !rkmode), do_rest = combo ? st == 0 : mode != MJB_MODE_STEP1;
			const bool do_euler = combo ? st == 0 : (mode == MJB_MODE_STEP || mode == MJB_MODE_STEP2 || rkmode);
			const bool hw_on = do_rest && checks && P->hw.n > 0;  // device-side DefaultRobotHWSim stage registered
			PROF_BEGIN();
			if (do_first && checks && nz.enabled) {
				ctrl_noise<G>(m, L, nz, e, step0 + (unsigned int)(st0 + st), zpre, znext);
				if (zpre && st + 1 < nst && e.lane < m.nu)  // next step's normal: a whole step of work hides the trip to HBM
					znext = zb[((size_t)(st0 + st + 1) * s.nenv + e.env) * m.nu + e.lane];
			}
			PROF(13);
			// attempt 1 only runs after mj_checkAcc found a bad qacc: reset, full forward, integrate
			bool rk4 = false;  // (the dense kernels integrate by Euler only: the host picks the generic ones for RK4)
			if constexpr (DENSE != 0) {
#pragma nounroll
			for (int attempt = 0; attempt < 2; attempt++) {
				if (do_first || attempt) {
					if (attempt == 0 && checks) {
						const int bad = any_bad<G>(e, L, e.f + L.qpos, m.nq, e.f + L.qvel, m.nv);
						if (bad) reset_frame_state<G>(m, L, s, lite(e), bad == 1 ? MJB_WARN_BADQPOS : MJB_WARN_BADQVEL);
					}
					if constexpr (MJB_DOUBLE_STAGE != 99) forward_first<G, CON, DENSE>(P, e, compact);  // (99: measurement build without the forward pass)
					if (st0 + st == (mode == MJB_MODE_STEP ? nsteps : nst) - 1 && P->m.enableflags & MJB_ENBL_ENERGY) VIEW(P, compact, energy<G>(m, L, lite(e)));
				}
				if (!do_rest) break;
				// device-side DefaultRobotHWSim::writeSim runs where the reference's control callback fires: after the position
				// and velocity stages, before actuation (mjcb_control inside mj_forward; mujoco_ros_control_plugin.cpp:153-194)
				if (hw_on) VIEW(P, compact, hwsim_write<G>(m, L, Pq_->hw, lite(e)));
				if constexpr (MJB_DOUBLE_STAGE != 99) forward_rest<G, CON, DENSE>(P, e, compact);
				if (attempt || !checks || !any_bad<G>(e, L, e.f + L.qacc, m.nv, e.f, 0, true)) break;
				reset_frame_state<G>(m, L, s, lite(e), MJB_WARN_BADQACC);
			}
			} else {
			// ... and, with <option integrator="RK4">, the three sub-stage evaluations of mj_RungeKutta run through the same loop (ONE
			// copy of the forward stages in the instruction stream): rk = evaluation index, rk4_stage() sets the next state
			rk4 = do_euler && P->m.integrator == MJB_INT_RK4;
#pragma nounroll
			for (int rk = rk0;;) {
#pragma nounroll
			for (int attempt = 0; attempt < 2; attempt++) {
				if (do_first || attempt || rk > rk0) {
					if (attempt == 0 && !rk && checks) {
						const int bad = any_bad<G>(e, L, e.f + L.qpos, m.nq, e.f + L.qvel, m.nv);
						if (bad) reset_frame_state<G>(m, L, s, lite(e), bad == 1 ? MJB_WARN_BADQPOS : MJB_WARN_BADQVEL);
					}
					if constexpr (MJB_DOUBLE_STAGE != 99) forward_first<G, CON, DENSE>(P, e, compact);  // (99: measurement build without the forward pass)
					if (st0 + st == (mode == MJB_MODE_STEP ? nsteps : nst) - 1 && P->m.enableflags & MJB_ENBL_ENERGY) VIEW(P, compact, energy<G>(m, L, lite(e)));
				}
				if (!do_rest) break;
				if (rk4 && !rk) {  // the warmstart the step came in with (the solvers save qacc as they finish)
					VIEW(P, compact, {
						for (int d = e.lane; d < m.nv; d += G) e.f[L.rk + m.nq + 3 * m.nv + d] = e.f[L.qacc_warmstart + d];
						gsync<G>();
					});
				}
				// device-side DefaultRobotHWSim::writeSim runs where the reference's control callback fires: after the position
				// and velocity stages, before actuation (mjcb_control inside mj_forward; mujoco_ros_control_plugin.cpp:153-194)
				// (RK4: once per step, at the step's own evaluation -- the PID state advances by one period; its forces stay for the sub-stages)
				if (hw_on && !rk) VIEW(P, compact, hwsim_write<G>(m, L, Pq_->hw, lite(e)));
				if constexpr (MJB_DOUBLE_STAGE != 99) forward_rest<G, CON, DENSE>(P, e, compact);
				if (attempt || rk || !checks || !any_bad<G>(e, L, e.f + L.qacc, m.nv, e.f, 0, true)) break;
				reset_frame_state<G>(m, L, s, lite(e), MJB_WARN_BADQACC);
			}
				if (!rk4 || !do_rest) break;
				VIEW(P, compact, rk4_stage<G>(m, L, e, rk));
				if (++rk == 4) break;
				if (rksplit) {  // the next evaluation's first half, then back to the host for its callbacks
					if constexpr (MJB_DOUBLE_STAGE != 99) forward_first<G, CON, DENSE>(P, e, compact);  // (99: measurement build without the forward pass)
					// (mjData.energy follows the evaluation, as in the fused step, mjb_step2_prefix and the oracle's mjo_step2_rk)
					if (P->m.enableflags & MJB_ENBL_ENERGY) VIEW(P, compact, energy<G>(m, L, lite(e)));
					break;
				}
			}
			}
			PROF(14);  // whole forward (incl. checks)
			if (do_euler && !rk4) VIEW(P, compact, euler<G, (CON != 0), (CON >= 2 && CON <= 4), (DENSE != 0), (MJB_PGS_PRESOLVE && (CON == 1 || CON == 9))>(m, L, e));
			PROF(15);
		}

		store_state<G>(m, L, s, e);
		if (P->hw.n > 0)  // the device-side hwsim stage writes qfrc_applied: keep mjData's view of it current
			copy_out<G>(s.qfrc_applied + (size_t)e.env * m.nv, e.f + L.qfrc_applied, m.nv, e.lane);
		if (ws && (mode != MJB_MODE_STEP || s.keep_frame)) {
#pragma unroll 8
			for (int k = e.lane; k < L.ndouble; k += G) ws[k] = e.f[k];
			int *wsi = reinterpret_cast<int *>(ws + L.ndouble);
#pragma unroll 8
			for (int k = e.lane; k < L.nint; k += G) wsi[k] = e.fi[k];
		}
		if (dyn) {  // publish the chunk: the state stores above first
			__threadfence();
			if (e.lane == 0) __hip_atomic_store(s.sched + 1 + e.env, item_chunk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		}
		gsync<G>();
	}
#ifdef MJB_PROFILE
	__syncthreads();
	if (threadIdx.x < 2 && s.prof && mjb_prof_lds.cnt[threadIdx.x] && mjb_prof_lds.base + threadIdx.x < 32) {
		atomicAdd(s.prof + mjb_prof_lds.base + threadIdx.x, mjb_prof_lds.sum[threadIdx.x]);
		atomicAdd(s.prof + 32 + mjb_prof_lds.base + threadIdx.x, (unsigned long long)mjb_prof_lds.cnt[threadIdx.x]);
	}
#endif
}

__global__ void mjb_reset_kernel(const KernelParams MJB_AS4 *__restrict__ P, const unsigned char *mask)
{
	const DevModel MJB_AS4 &m = P->m;
	const DevState MJB_AS4 &s = P->s;
	const int env = blockIdx.x * blockDim.x + threadIdx.x;
	if (env >= s.nenv) return;
	if (mask && !mask[env]) return;
	const size_t e = (size_t)env;
	for (int k = 0; k < m.nq; k++) s.qpos[e * m.nq + k] = m.qpos0[k];
	for (int k = 0; k < m.nv; k++) {
		s.qvel[e * m.nv + k] = 0;
		s.qacc_warmstart[e * m.nv + k] = 0;
		s.qfrc_applied[e * m.nv + k] = 0;
		s.qacc[e * m.nv + k] = 0;
	}
	for (int k = 0; k < m.na; k++) s.act[e * m.na + k] = 0;
	for (int k = 0; k < m.nu; k++) {
		s.ctrl[e * m.nu + k] = 0;
		s.ctrlnoise[e * m.nu + k] = 0;
	}
	for (int k = 0; k < 6 * m.nbody; k++) s.xfrc_applied[e * 6 * m.nbody + k] = 0;
	for (int k = 0; k < m.nsensordata; k++) s.sensordata[e * m.nsensordata + k] = 0;
	for (int b = 0; b < m.nbody; b++) {
		const int mid = m.body_mocapid[b];
		if (mid < 0) continue;
		for (int k = 0; k < 3; k++) s.mocap_pos[(e * m.nmocap + mid) * 3 + k] = m.body_pos[3 * b + k];
		for (int k = 0; k < 4; k++) s.mocap_quat[(e * m.nmocap + mid) * 4 + k] = m.body_quat[4 * b + k];
	}
	s.time[e] = 0;
	s.energy[2 * e] = s.energy[2 * e + 1] = 0;
}

template <int G, int CON, int DENSE = 0>
int launch_g(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int env_hi, int mode, int nsteps, unsigned int step0,
             int epb, void *stream, int chunk = 0)
{
	const int nenv = env_hi - env_lo;
	const int frame_bytes = ((L.ndouble * 8 + L.nint * 4) + 15) & ~15;
	const int maxlds = mjb_max_lds_bytes();
	int threads = epb * G;
	if (threads > 256) {
		epb = 256 / G;
		threads = epb * G;
	}
	while (epb > 1 && epb * frame_bytes > maxlds) {
		epb--;
		threads = epb * G;
	}
	size_t lds = (size_t)epb * frame_bytes;
	if ((int)lds > maxlds) return (int)hipErrorInvalidValue;
	{  // measurement knob: MJB_DEBUG_LDS_BYTES=<n> requests at least n bytes per block, i.e. caps the resident blocks per CU
		static const int floor_bytes = [] { const char *v = getenv("MJB_DEBUG_LDS_BYTES"); return v ? atoi(v) : 0; }();
		if (floor_bytes > (int)lds && floor_bytes <= maxlds) lds = floor_bytes;
	}
	auto kern = mjb_step_kernel<G, CON, DENSE>;
	hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
	                                     (int)lds);
	if (err != hipSuccess) return (int)err;
	int blocks = (nenv + epb - 1) / epb;
	const int maxblocks = 256 * 16;
	if (blocks > maxblocks) blocks = maxblocks;
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, (hipStream_t)stream,
	                   (const KernelParams MJB_AS4 *)Pdev, mode, nsteps, step0, epb, frame_bytes, CON != 0 ? chunk : 0, env_lo, env_hi);
	return (int)hipGetLastError();
}

}  // namespace

// The kernel variants are compiled in slices, one translation unit per slice (-DMJB_GROUP=0..4: minutes of device code
// generation run in parallel); without MJB_GROUP the whole file is one unit (profiling and development builds).
//   0: models without constraint rows + the dispatcher   1: PGS (1, 5)   2: Newton 1 / 2 rows per lane   3: Newton 4 rows   4: CG
//   5: the 256-register PGS variant (9) on its own: out-of-line helpers shared with the 512-register kernels would be compiled
//      for their budget and cost it its second wave per SIMD, or spills (760 instead of 576 in the same unit as variants 1 / 5)
#ifndef MJB_GROUP
#define MJB_GROUP -1
#endif
#define MJB_HAS_GROUP(g) (MJB_GROUP < 0 || MJB_GROUP == (g))
int mjb_launch_group1(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk);
int mjb_launch_group2(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk);
int mjb_launch_group3(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk);
int mjb_launch_group4(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk);
int mjb_launch_group5(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk);

#if !defined(MJB_DEV_ONLY_CON)
#if MJB_HAS_GROUP(1)
int mjb_launch_group1(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk)
{
	if (constrained == 5) return launch_g<64, 5>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
	return launch_g<64, 1>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
}
#endif
#if MJB_HAS_GROUP(5)
int mjb_launch_group5(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk)
{
	return launch_g<64, 9>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
}
#endif
#if MJB_HAS_GROUP(2)
int mjb_launch_group2(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk)
{
	if (constrained == 3) return launch_g<64, 3>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
	return launch_g<64, 2>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
}
#endif
#if MJB_HAS_GROUP(3)
int mjb_launch_group3(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk)
{
	return launch_g<64, 4>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
}
#endif
#if MJB_HAS_GROUP(4)
int mjb_launch_group4(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0, int epb, int constrained, void *stream, int chunk)
{
	if (constrained == 7) return launch_g<64, 7>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
	if (constrained == 8) return launch_g<64, 8>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
	return launch_g<64, 6>(Pdev, L, env_lo, nenv, mode, nsteps, step0, epb, stream, chunk);
}
#endif
#endif

#if MJB_HAS_GROUP(6)
// ---- the CONSTRAINT half of a split step: the smooth stages of this step ran in lane = env form (mjb_smooth_kernel.h) and left
// geom frames, cdof, both L'DL factors, qfrc_smooth and qacc_smooth in the env's hand-off record; here one env per wavefront takes them into its
// lean frame and runs A4 - A7, A13 and A16: collision, make_constraint, the reference accelerations, PGS (nv <= 16: rows of J M^-1 and AR in
// registers), Euler.  The stage functions and the frame are those of kernel variant 9; what is gone from the instruction stream is every smooth
// stage, the step loop and the work queue (one launch = one step).  flags bit 0: workload statistics on (mjb_set_stats).
// one env's constraint half on the wavefront's frame e (e.env set)
template <int CON> DEVI void cstep_env(const KernelParams MJB_AS4 *__restrict__ P, Env &e, const int flags)
{
	constexpr int G = 64;
	constexpr bool MJB_LAUNDER_HERE = true;
	const int compact = 1;
	VIEW(P, compact, {
		for (int k = e.lane; k < L.nint; k += G) e.fi[k] = 0;
		gsync<G>();
		if (e.lane < 16) {
			for (int q = 0; q < 4; q++) {
				unsigned int w = 0;
				for (int r = 0; r < 4; r++) {
					const int a = m.M_dense[16 * (4 * q + r) + e.lane];
					w |= (unsigned int)(a < 0 ? 255 : a) << (8 * r);
				}
				e.fi[L.dadr + 4 * e.lane + q] = (int)w;
			}
		}
		load_state<G>(m, L, s, e);
		// the smooth stages' results
		const HandoffLayout hl = mjb_handoff_layout(m.ngeom, m.nv, m.nbody, m.nM);
		const double *H = s.handoff + (size_t)e.env * (size_t)s.handoff_stride;
		double *f = e.f;
		copy_in<G>(f + L.geom_xpos, H + hl.geom_xpos, 3 * m.ngeom, e.lane);
		copy_in<G>(f + L.geom_xmat, H + hl.geom_xmat, 9 * m.ngeom, e.lane);
		copy_in<G>(f + L.cdof, H + hl.cdof, 6 * m.nv, e.lane);
		copy_in<G>(f + L.subtree_com, H + hl.subtree_com, 3 * m.nbody, e.lane);
		copy_in<G>(f + L.qLD, H + hl.qLD, m.nM, e.lane);
		copy_in<G>(f + L.qLDiagInv, H + hl.qLDiagInv, m.nv, e.lane);
		copy_in<G>(f + L.qH, H + hl.qH, m.nM, e.lane);
		copy_in<G>(f + L.qHdi, H + hl.qHdi, m.nv, e.lane);
		copy_in<G>(f + L.qfrc_smooth, H + hl.qfrc_smooth, m.nv, e.lane);
		copy_in<G>(f + L.qacc_smooth, H + hl.qacc_smooth, m.nv, e.lane);
		gsync<G>();
	});
	VIEW(P, compact, collision<G>(m, L, s, e));
	VIEW(P, compact, make_constraint<G, CON>(m, L, s, e));
	VIEW(P, compact, reference_constraint<G, CON>(m, L, s, e));
	VIEW(P, compact, fwd_constraint_pgs<G, false, true, CON>(m, L, s, e));
	// qacc = qacc_smooth + M^-1 qfrc_constraint, and -- under implicit joint damping -- Euler's (M + h B)^-1 (qfrc_smooth + qfrc_constraint)
	// beside it: one dual substitution (forward_rest, MJB_PGS_PRESOLVE)
	VIEW(P, compact, {
		double *f = e.f;
		const bool damp = m.eulerdamp != 0;
		if (e.lane < m.nv) {
			const double c = f[L.qfrc_constraint + e.lane];
			f[L.qacc + e.lane] = c;
			f[L.eulerx + e.lane] = f[L.qfrc_smooth + e.lane] + c;
		}
		gsync<G>();
		int dl[16];
		dadr_load(e, L, dl);
		solve_dense16<G, 16>(m, e, f + L.qacc, f + L.qLD, f + L.qLDiagInv, f + L.eulerx, f + L.qH, f + L.qHdi, damp, dl, f + L.solvescr);
		if (e.lane < m.nv) {
			const double a = f[L.qacc_smooth + e.lane] + f[L.qacc + e.lane];
			f[L.qacc + e.lane] = a;
			f[L.qacc_warmstart + e.lane] = a;
		}
		gsync<G>();
	});
	if (flags & 1) {  // workload statistics (mjb_set_stats)
		VIEW(P, compact, {
			if (s.stats != nullptr && e.lane == 0) {
				const int ne = e.fi[L.nefc], nc = m.nconmax > 0 ? e.fi[L.ncon] : 0;
				unsigned long long *st = s.stats;
				atomicAdd(st, 1ull);
				atomicAdd(st + 1, (unsigned long long)e.fi[L.solver_iter]);
				atomicAdd(st + 2 + (ne < 256 ? ne : 256), 1ull);
				atomicAdd(st + 259 + (nc < 128 ? nc : 128), 1ull);
			}
		});
	}
	// mj_checkAcc: a bad qacc resets the env; mj_step then runs the forward pass again on mj_resetData's state and advances it -- the result is the
	// same state for every env of the model (ctrl reads zero after the reset), computed once by the host: DevState::reset_step
	bool advanced = false;
	VIEW(P, compact, {
		if (any_bad<G>(e, L, e.f + L.qacc, m.nv, e.f, 0, true)) {
			MJB_KEEP_BRANCH();
			reset_frame_state<G>(m, L, s, lite(e), MJB_WARN_BADQACC);
			if (s.reset_step != nullptr) {
				copy_in<G>(e.f + L.qpos, s.reset_step, m.nq, e.lane);
				copy_in<G>(e.f + L.qvel, s.reset_step + m.nq, m.nv, e.lane);
				copy_in<G>(e.f + L.qacc_warmstart, s.reset_step + m.nq + m.nv, m.nv, e.lane);
				copy_in<G>(e.f + L.qacc, s.reset_step + m.nq + m.nv, m.nv, e.lane);
				if (e.lane == 0) e.f[L.time] = m.timestep[0];
			}
			gsync<G>();
			// (the reset reaches the arrays this kernel does not otherwise store)
			copy_out<G>(s.ctrl + (size_t)e.env * m.nu, e.f + L.ctrl, m.nu, e.lane);
			copy_out<G>(s.ctrlnoise + (size_t)e.env * m.nu, e.f + L.ctrlnoise, m.nu, e.lane);
			copy_out<G>(s.qfrc_applied + (size_t)e.env * m.nv, e.f + L.qfrc_applied, m.nv, e.lane);
			advanced = true;
		}
	});
	if (!advanced) VIEW(P, compact, euler<G, true, false, false, true>(m, L, e));
	VIEW(P, compact, {
		const size_t env = (size_t)e.env;
		copy_out<G>(s.qpos + env * m.nq, e.f + L.qpos, m.nq, e.lane);
		copy_out<G>(s.qvel + env * m.nv, e.f + L.qvel, m.nv, e.lane);
		copy_out<G>(s.qacc_warmstart + env * m.nv, e.f + L.qacc_warmstart, m.nv, e.lane);
		copy_out<G>(s.qacc + env * m.nv, e.f + L.qacc, m.nv, e.lane);
		if (e.lane == 0) s.time[env] = e.f[L.time];
	});
	gsync<G>();
}

template <int CON>
__global__ void __launch_bounds__(256, 2)
    mjb_cstep_kernel(const KernelParams MJB_AS4 *__restrict__ P, const int epb, const int frame_bytes, const int env_lo, const int env_hi, const int flags)
{
	constexpr int G = 64;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int slot = threadIdx.x / G;
	Env e;
#ifdef MJB_PROFILE
	e.prof = P->s.prof;
#endif
	e.lane.mask = G - 1;
	{
		const FrameLayout MJB_AS4 &L = P->Lc;
		e.f = reinterpret_cast<double *>(smem + (size_t)slot * frame_bytes);
		e.fi = reinterpret_cast<int *>(e.f + L.ndouble);
	}
	e.mp = nullptr;
	for (int base = env_lo + blockIdx.x * epb; base < env_hi; base += gridDim.x * epb) {
		e.env = __builtin_amdgcn_readfirstlane(base + slot);
		if (e.env >= env_hi) continue;
		cstep_env<CON>(P, e, flags);
	}
}

int mjb_launch_cstep(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int env_hi, int epb, int flags, void *stream)
{
	const int nenv = env_hi - env_lo;
	if (nenv <= 0) return 0;
	const int frame_bytes = ((L.ndouble * 8 + L.nint * 4) + 15) & ~15;
	if (epb > 4) epb = 4;
	if (epb < 1) epb = 1;
	const size_t lds = (size_t)epb * frame_bytes;
	if ((int)lds > 160 * 1024) return (int)hipErrorInvalidValue;
	auto kern = mjb_cstep_kernel<9>;
	if (lds > 65536) {  // (four frames of config 3: 80 KB.  Per device and size, not per launch: one launch = one step here)
		static std::mutex mu;
		static std::map<std::pair<int, int>, hipError_t> done;
		int dev = 0;
		(void)hipGetDevice(&dev);
		std::lock_guard<std::mutex> lock(mu);
		auto it = done.find({ dev, (int)lds });
		hipError_t err;
		if (it == done.end()) {
			err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			done[{ dev, (int)lds }] = err;
		} else
			err = it->second;
		if (err != hipSuccess) return (int)err;
	}
	int blocks = (nenv + epb - 1) / epb;
	if (blocks > 256 * 16) blocks = 256 * 16;
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(epb * 64), lds, (hipStream_t)stream, (const KernelParams MJB_AS4 *)Pdev, epb, frame_bytes, env_lo, env_hi, flags);
	return (int)hipGetLastError();
}
#endif  // MJB_HAS_GROUP(6)

#if MJB_HAS_GROUP(0)
int mjb_max_lds_bytes() { return 160 * 1024; }

int mjb_launch_step(const KernelParams *Pdev, const FrameLayout &L, int env_lo, int nenv, int mode, int nsteps, unsigned int step0,
                    int lanes_per_env, int envs_per_block, int constrained, int dense, void *stream)
{
	// (the headline kernels are instantiated first so that they sit at the start of the code object whatever happens to the
	//  size of the constrained ones: their absolute placement is worth ~2 % on config 2)
	const int chunk = constrained >> 8;  // (steps per work item of a chunked launch, 0 = one item per env; mjb_api.hip: launch)
	constrained &= 255;
#ifdef MJB_DEV_ONLY_CON  // development switch: compile ONE constrained kernel variant (seconds instead of minutes)
	return launch_g<64, MJB_DEV_ONLY_CON>(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, stream, chunk);
#else
	if (!constrained) {
		switch (lanes_per_env) {
		case 16:
			if (dense == 12) return launch_g<16, 0, 12>(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, stream);
			if (dense == 8) return launch_g<16, 0, 8>(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, stream);
			if (dense) return launch_g<16, 0, 16>(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, stream);
			return launch_g<16, 0>(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, stream);
		case 8: return launch_g<8, 0>(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, stream);
		case 32: return launch_g<32, 0>(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, stream);
		case 64: return launch_g<64, 0>(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, stream);
		default: return (int)hipErrorInvalidValue;
		}
	}
	if (lanes_per_env != 64) return (int)hipErrorInvalidValue;
	if (constrained == 2 || constrained == 3) return mjb_launch_group2(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, constrained, stream, chunk);
	if (constrained == 4) return mjb_launch_group3(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, constrained, stream, chunk);
	if (constrained >= 6 && constrained <= 8) return mjb_launch_group4(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, constrained, stream, chunk);
	if (constrained == 9) return mjb_launch_group5(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, constrained, stream, chunk);
	return mjb_launch_group1(Pdev, L, env_lo, nenv, mode, nsteps, step0, envs_per_block, constrained, stream, chunk);
#endif
}

// The standard normals the ctrl-noise injector (mujoco_env.cpp:469-481) draws during ONE fused launch, generated ahead of it: element
// (st, env, i) = philox_normal(seed, env_offset + env, step0 + st, i), the very call the step kernel makes when it has no buffer.
// A throughput kernel (one value per thread, ~600 k wavefronts for config 2's 4096 x 1000 x 9) in place of a 2 k-cycle link in every
// step's dependent chain.
__global__ void mjb_noise_kernel(const KernelParams MJB_AS4 *__restrict__ P, double *__restrict__ z, unsigned int *zinfo, const int nenv,
                                 const int nu, const int nsteps, const unsigned int step0)
{
	const NoiseCfg MJB_AS4 &nz = P->nz;
	const size_t n = (size_t)nsteps * nenv * nu;
	for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
		const unsigned int i = (unsigned int)(t % nu);
		const size_t q = t / nu;
		const int env = (int)(q % nenv);
		const unsigned int st = (unsigned int)(q / nenv);
		z[t] = philox_normal(nz.seed, (unsigned long long)(nz.env_offset + env), step0 + st, i);
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		zinfo[0] = step0;
		zinfo[1] = (unsigned int)nsteps;
		zinfo[2] = (unsigned int)nenv;
	}
}

int mjb_launch_noise(const KernelParams *Pdev, double *zbuf, unsigned int *zinfo, int nenv, int nu, int nsteps, unsigned int step0, void *stream)
{
	const size_t n = (size_t)nsteps * nenv * nu;
	const int threads = 256;
	size_t blocks = (n + threads - 1) / threads;
	if (blocks > 256 * 64) blocks = 256 * 64;
	hipLaunchKernelGGL(mjb_noise_kernel, dim3((unsigned int)blocks), dim3(threads), 0, (hipStream_t)stream, (const KernelParams MJB_AS4 *)Pdev, zbuf, zinfo,
	                   nenv, nu, nsteps, step0);
	return (int)hipGetLastError();
}

int mjb_launch_reset(const KernelParams *Pdev, int nenv, const unsigned char *mask_dev, void *stream)
{
	const int threads = 256;
	const int blocks = (nenv + threads - 1) / threads;
	hipLaunchKernelGGL(mjb_reset_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream,
	                   (const KernelParams MJB_AS4 *)Pdev, mask_dev);
	return (int)hipGetLastError();
}
#endif  // MJB_HAS_GROUP(0)
