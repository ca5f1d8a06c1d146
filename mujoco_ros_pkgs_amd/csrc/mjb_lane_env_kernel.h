// mjb_lane_env_kernel.h -- device code of the lane = env kernel (see mjb_lane_env.hip for what it is): a template over the model's integer
// structure `T` (a LeTopo_* struct: csrc/lane_env_topos.h, or the one mjb_lane_env.hip writes for hiprtc) and the LDS budget LP.
// Included by mjb_lane_env.hip (the compiled-in topologies) and by the source hiprtc compiles for any other eligible model.
#pragma once
#ifndef __HIPCC_RTC__  // (hiprtc brings its own runtime header)
#include <hip/hip_runtime.h>
#endif

#include "mjb_dev.h"
#include "mjb_math.h"

namespace mjb_le {

// compile-time loop: f(IC<0>{}), f(IC<1>{}), ...  (own three-line integer sequence: the header is also compiled by hiprtc, without <utility>)
template <int V> struct IC {
	static constexpr int value = V;
	constexpr operator int() const { return V; }
};
template <int... Is> struct ISeq {};
template <int N, int... Is> struct MkSeq : MkSeq<N - 1, N - 1, Is...> {};
template <int... Is> struct MkSeq<0, Is...> { using type = ISeq<Is...>; };
template <typename F, int... Is> DEVI void sfor_impl(F &&f, ISeq<Is...>) { (f(IC<Is>{}), ...); }
template <int N, typename F> DEVI void sfor(F &&f) { sfor_impl(f, typename MkSeq<N>::type{}); }

DEVI double frcp(double x)
{
	double r = __builtin_amdgcn_rcp(x);
	r = fma(fma(-x, r, 1.0), r, r);
	r = fma(fma(-x, r, 1.0), r, r);
	return r;
}

// compile-time queries on a topology
template <class T> struct Tq {
	// dof a is dof i or one of its ancestors
	static constexpr bool anc(int a, int i)
	{
		for (int j = i; j >= 0; j = T::dof_parentid[j])
			if (j == a) return true;
		return false;
	}
	static constexpr bool is_root(int b) { return b > 0 && T::body_rootid[b] == b; }
	// entry e of MuJoCo's sparse qM (dof_Madr[i] + k: dof i with its k-th ancestor, itself first) -> i, a
	static constexpr int ent_i(int e)
	{
		int i = 0;
		for (int d = 0; d < T::NV; d++)
			if (T::dof_Madr[d] <= e) i = d;
		return i;
	}
	static constexpr int ent_a(int e)
	{
		int a = ent_i(e);
		for (int k = T::dof_Madr[ent_i(e)]; k < e; k++) a = T::dof_parentid[a];
		return a;
	}
	// some sensor of the model needs the pose of this site / body
	static constexpr bool has_actuator_sensor()
	{
		for (int i = 0; i < T::NSENSOR; i++)
			if (T::sensor_type[i] == MJB_SENS_ACTUATORFRC) return true;
		return false;
	}
};

// 1 / sqrt(x): hardware seed + two Newton steps (x > 0)
DEVI double frsq(double x)
{
	double y = __builtin_amdgcn_rsq(x);
	y = y * fma(-0.5 * x * y, y, 1.5);
	y = y * fma(-0.5 * x * y, y, 1.5);
	return y;
}

// mju_normalize4 of a body's world quaternion: untouched within mjMINVAL of unit length.  (Its identity case for a vanishing quaternion cannot
// occur here: the argument is a product of the model's unit quaternions and of (cos, axis sin) hinge quaternions.)
DEVI void normalize4_sel(double *q)
{
	const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
	const double r = frsq(n2), n = n2 * r;
	const double s = (fabs(n - 1) > MJB_MINVAL) ? r : 1.0;
	q[0] *= s;
	q[1] *= s;
	q[2] *= s;
	q[3] *= s;
}

// A loaded value the optimiser must treat as already there: `cond ? k : load` otherwise becomes a per-lane branch around the load.
DEVI double pinv(double x)
{
	asm volatile("" : "+v"(x));
	return x;
}
// "This value has to be HERE": the wait for a prefetched scalar / LDS value is taken at this point, BEFORE the next prefetch is issued.
// LDS reads and scalar loads share one counter and scalar loads return out of order, so any wait is a wait for everything in
// flight: a region that issues its successor's fetches first and then touches its own data waits for both.
DEVI void touch_s(double x) { asm volatile("" ::"s"(x)); }
DEVI void touch_v(double x) { asm volatile("" ::"v"(x)); }
// ... and a whole half record: placed at the END of the region that issued its loads, it keeps every one of them inside that region
// (left alone, the loads of entries first used late in the next region are sunk there and waited for on the spot)
template <int N> DEVI void touch_rec(const double *h)  // (the N entries the sweep reads)
{
	asm volatile("" ::"s"(h[0]), "s"(h[1]), "s"(h[2]), "s"(h[3]), "s"(h[4]), "s"(h[5]), "s"(h[6]), "s"(h[7]), "s"(h[8]), "s"(h[9]));
	if constexpr (N > 10) asm volatile("" ::"s"(h[10]), "s"(h[11]), "s"(h[12]), "s"(h[13]));
}
DEVI double pins(double x)  // the same for a wave-uniform (scalar) value
{
	asm volatile("" : "+s"(x));
	return x;
}
// clamp by v_max / v_min, NaN passed through as the ternary chain of mj_fwdActuation passes it
DEVI double clampd(double c, double lo, double hi)
{
	const double v = fmin(fmax(c, lo), hi);
	return c != c ? c : v;
}

DEVI bool bad_val(double x) { return !(fabs(x) <= MJB_MAXVAL); }  // NaN or beyond mjMAXVAL: one unordered compare

// sin and cos of a joint half-angle, branch-free: k = round(x * 2/pi), r = x - k * pi/2 through three fma steps (pi/2 split into
// 53-bit pieces: exact to rounding while |x| < ~1e6 rad; beyond, the absolute error grows like |x| * 2^-53 * k-independent terms --
// a hinge wound up that far is not a simulation any more, and mj_check* only stops it at 1e10), then the fdlibm kernel polynomials on
// [-pi/4, pi/4] and a quadrant swap.  libm's sincos carries a divergent slow path for huge arguments; this kernel must not branch
// per lane (see the note on divergent control flow at the kernel).
DEVI void sincos_nb(double x, double *sn, double *cs)
{
	const double k = rint(x * 0.63661977236758134308);
	double r = fma(-k, 1.57079632679489655800e+00, x);
	r = fma(-k, 6.12323399573676603587e-17, r);
	r = fma(-k, -1.49738490485916983827e-33, r);
	const double z = r * r;
	// __kernel_sin / __kernel_cos (fdlibm), tail argument dropped (|r| <= pi/4 to rounding)
	const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
	             S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
	const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
	             C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
	const double ps = fma(fma(fma(fma(fma(S6, z, S5), z, S4), z, S3), z, S2), z, S1);
	const double sr = fma(z * r, ps, r);
	const double pc = fma(fma(fma(fma(fma(C6, z, C5), z, C4), z, C3), z, C2), z, C1);
	const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
	const int q = (int)k & 3;
	const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
	*sn = (q & 2) ? -s0 : s0;
	*cs = ((q + 1) & 2) ? -c0 : c0;
}

// LDS of a block (= one wavefront): pair slot q of lane l = the two doubles at (q * 64 + l) * 16 bytes -- one ds_read_b128 /
// ds_write_b128 per pair, conflict-free.  Slots [0, NV): (qpos_i, qvel_i); then three slots per body whose force the backward
// sweep reads (cfrc_body); then, while the budget LP lasts, five per body for its cinert (leaf-most bodies first): the wavefront's
// overflow space next to its registers.  LP = 40 pair slots (40 KB) when four wavefronts share a CU, 80 / 160 when the batch leaves
// a CU to two / one (the launcher picks): what does not fit stays in registers, i.e. mostly in their AGPR half, at four moves per
// double and round trip against one LDS instruction per PAIR and direction.
template <class T, int LP> struct Lds {
	using Q = Tq<T>;
	// the body's cfrc is consumed by the backward sweep (it, or an ancestor, carries a joint)
	static constexpr bool needed(int b)
	{
		for (int a = b; a > 0; a = T::body_parentid[a])
			if (T::body_jnt[a] >= 0) return true;
		return false;
	}
	static constexpr int slot(int b)  // first of the body's three pair slots
	{
		int n = 0;
		for (int a = 1; a < b; a++)
			if (needed(a)) n++;
		return T::NV + 3 * n;
	}
	static constexpr int cin_slot(int b)  // first of the body's five cinert slots, -1: the body's cinert stays in registers
	{
		int at = slot(T::NBODY);
		for (int a = T::NBODY - 1; a >= 1; a--) {
			if (!needed(a)) continue;
			if (at + 5 > LP) return -1;
			if (a == b) return at;
			at += 5;
		}
		return -1;
	}
	static constexpr int nslots()
	{
		int at = slot(T::NBODY);
		for (int a = T::NBODY - 1; a >= 1; a--)
			if (needed(a) && at + 5 <= LP) at += 5;
		return at;
	}
	static constexpr int bytes() { return nslots() * 64 * 16; }
};

struct alignas(16) Pair { double a, b; };

// ROLE: 0 = one wavefront runs the whole step of its 64 envs.  1 / 2 = the DUO form, two wavefronts of one workgroup (on two SIMDs of a CU) share
// the 64 envs of the block: the step's two independent halves -- what depends on qpos alone (poses, cinert, composite inertias, qM, both factors:
// role 1, "P") and what depends on qvel too (velocities, the bodies' forces, the force block, qfrc_smooth: role 2, "V") -- run side by side, V hands
// qfrc_smooth over through LDS, P solves and integrates, and hands the new state (which lives in LDS anyway) and the mj_check* verdicts back.  Both
// compute the poses, cdof and cinert (the shared prefix).  A lone wavefront issues one instruction every ~4 cycles whatever it is, so the step's
// length is its instruction count: ~6.2 k in one piece, ~max(P, V) + solve + Euler in two.  Pays while the batch leaves SIMDs idle (the launcher).
// rendezvous of a DUO block's two wavefronts: LDS traffic done, then the barrier.  (__syncthreads() also waits for the wavefront's outstanding GLOBAL
// loads -- V's ctrl-noise normals and qfrc_applied are fetched a sweep ahead of their use precisely so that nobody waits for HBM)
DEVI void le_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NV> struct DuoSlots { static constexpr int n = (NV + 1) / 2 + 1; };  // qfrc_smooth pairs + the mail slot
template <class T, int LP, int ROLE = 0>
DEVI void lane_env_body(const KernelParams MJB_AS4 *__restrict__ P, const int nsteps, const unsigned int step0, const int env_lo, const int env_hi,
                        unsigned char *const smem_le)
{
	constexpr int NB = T::NBODY, NV = T::NV, NU = T::NU;
	// (roles 3 / 4 = the PIPELINED duo: P computes every pose and cinert ONCE and passes them on body by body -- a workgroup barrier per body, a
	//  two-deep ring of (xpos, xmat) in LDS -- V follows one body behind with cdof, velocities and forces and passes cdof back for P's composite-inertia
	//  sweep.  Nothing is computed twice; needs every needed body's cinert and cdof in LDS: Duo2<T>.)
	// (roles 5 / 6 / 7 = the TRIO, three wavefronts per 64 envs: 5 = P runs the pose chain and nothing else; 6 = C follows it through the ring with cinert
	//  and cdof, then takes the composite-inertia sweep, qM, both factors, the solves and Euler; 7 = V as role 4, with its own cinert of every body.
	//  P's sweep is the step's critical chain: everything that is not a pose is off it.)
	constexpr bool DP = ROLE == 0 || ROLE == 1 || ROLE == 3 || ROLE == 6, DV = ROLE == 0 || ROLE == 2 || ROLE == 4 || ROLE == 7, DUO = ROLE != 0;  // this wavefront does the position half (inertias, factors, solves, Euler) / the velocity half
	constexpr bool PIPE = ROLE >= 3;
	constexpr bool POSE = ROLE <= 3 || ROLE == 5, RINGC = ROLE == 4 || ROLE == 6 || ROLE == 7;  // computes the poses / takes them from the ring
	constexpr bool EPOS = ROLE == 0 || ROLE == 1 || ROLE == 3 || ROLE == 5;                       // gathers mj_energyPos along its pose sweep
	constexpr bool SENSF = ROLE == 0 || ROLE == 2 || ROLE == 3 || ROLE == 5;  // frame sensors: who holds the poses (and, of two, who has the time)
	using Q = Tq<T>;
	constexpr int LPE = PIPE ? (1 << 20) : (DUO ? LP - DuoSlots<NV>::n : LP);
	using LD = Lds<T, LPE>;
	static_assert(LD::slot(T::NBODY) <= LPE, "lane = env kernel: state and forces of the topology need more LDS than this instantiation's budget");
	// (trio) body b's half-angle (sin, cos) comes from C: a hinge whose two predecessors in the sweep are needed bodies too (C publishes them two barriers ahead)
	constexpr auto SCUSE = [](int b) {
		if (b < 3 || b >= T::NBODY || T::body_jnt[b] < 0) return false;
		if (T::jnt_type[T::body_jnt[b]] != MJB_JNT_HINGE) return false;
		return LD::needed(b) && LD::needed(b - 1) && LD::needed(b - 2);
	};
	constexpr int RINGN = ROLE >= 5 ? 3 : 2;  // depth of the pose ring (the trio's V reads two bodies behind P)
	constexpr int LASTB = [] { for (int c = NB - 1; c >= 1; c--) if (LD::needed(c)) return c; return 0; }();  // the leaf the composite-inertia sweep starts at
	constexpr int RING = LD::nslots();                                              // (PIPE) 2 x 6 pair slots of the pose ring behind the solo layout
	constexpr int XS = PIPE ? RING + 6 * RINGN : LD::nslots(), MAIL = XS + (NV + 1) / 2, SC0 = MAIL + 1;  // (trio) SC0 + b: body b's half-angle (sin, cos), from C to P  // (DUO) pair slots of qfrc_smooth, and of P's verdicts for V
	const int lane_le = DUO ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
	Pair *const lp = reinterpret_cast<Pair *>(smem_le) + lane_le;  // pair slot q of this lane: lp[64 * q]
	// (a tail lane without an env keeps running on the last env's data and stores nothing: no divergent exit, the wave-uniform
	//  branches below stay uniform)
	const int env_raw = env_lo + (int)(blockIdx.x * (DUO ? 64u : blockDim.x)) + lane_le;  // (solo: blockDim.x = 64; fewer: a measurement knob, MJB_LANE_ENV_WAVE_LANES)
	const bool live = env_raw < env_hi;
	const int env = live ? env_raw : env_hi - 1;
	const size_t ev = (size_t)env;

	// ---- the env's state: (qpos, qvel) in LDS, the OU noise state in registers
	double cn[NU > 0 ? NU : 1];
	double time;
	bool badp_next = false, badv_next = false;  // mj_checkPos / mj_checkVel verdict on the state the next step starts from
	bool wasreset = false;  // mj_resetData ran inside this launch: ctrl / qfrc_applied read as zero from then on (the frame copy of the generic kernels)
	{
		const DevState MJB_AS4 &s = P->s;
		if constexpr (DP) {
			sfor<NV>([&](auto I) {
				const Pair s2{ s.qpos[ev * NV + I], s.qvel[ev * NV + I] };
				lp[64 * I] = s2;
				badp_next |= bad_val(s2.a);
				badv_next |= bad_val(s2.b);
			});
		}
		sfor<NU>([&](auto I) { cn[I] = DV ? s.ctrlnoise[ev * NU + I] : 0.0; });
		time = s.time[ev];
		if constexpr (DUO) {  // the load is "Euler of step -1": P's verdicts on the loaded state go to V by mail
			if constexpr (DP) lp[64 * MAIL] = Pair{ (double)((badp_next ? 2 : 0) | (badv_next ? 1 : 0)), 0.0 };
			le_barrier();
			if constexpr (!DP) {
				const int code = (int)lp[64 * MAIL].a;
				badp_next = (code & 2) != 0;
				badv_next = (code & 1) != 0;
			}
		}
	}
	const bool nz_on = P->nz.enabled != 0;
	// ctrl noise from the launch's pre-generated buffer when the host filled one for exactly this launch (mjb_api.hip: launch)
	int zhalf_i = -1;
	if (nz_on && P->s.zbuf != nullptr) {
		const unsigned int *zi = P->s.zinfo;
		if (zi[0] == step0 && (int)zi[1] == nsteps && (int)zi[2] == P->s.nenv) zhalf_i = 0;
		else if (zi[4] == step0 && (int)zi[5] == nsteps && (int)zi[6] == P->s.nenv) zhalf_i = 1;
	}
	const bool zpre = zhalf_i >= 0;

#ifdef MJB_LE_PROBE  // (measurement build: cycle stamps of the step's phases, summed over the launch, into the env-0 sensordata of the role)
	unsigned long long pk_t = 0, pk_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#define LE_PK0() pk_t = __builtin_readcyclecounter()
#define LE_PK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pk_acc[i] += n_ - pk_t; pk_t = n_; } while (0)
#else
#define LE_PK0() do { } while (0)
#define LE_PK(i) do { } while (0)
#endif
#pragma nounroll
	for (int st = 0; st < nsteps; st++) {
		LE_PK0();
		// (the parameter pointer laundered per step: model constants are re-fetched by scalar loads where they are used instead of
		//  being hoisted out of the step loop into ~700 SGPRs the wavefront does not have)
		const KernelParams MJB_AS4 *Pq = P;
		asm volatile("" : "+s"(Pq));
		const DevModel MJB_AS4 &m = Pq->m;
		const DevState MJB_AS4 &s = Pq->s;
		const bool last = st == nsteps - 1;
		// the model's numeric constants: one tape in consumption order (mjb_dev.h), a half record (64 bytes) per scalar load
		const LeTapeHdr MJB_AS4 *th = reinterpret_cast<const LeTapeHdr MJB_AS4 *>(m.le_tape);
		const LeTapeBody MJB_AS4 *tb = reinterpret_cast<const LeTapeBody MJB_AS4 *>(th + 1);
		const double dt = th->dt;

		// ---- H10: the reference's ctrl-noise injector (mujoco_env.cpp:469-481): this step's normals are FETCHED here and folded into
		// the OU state where the forces are assembled, after the root -> leaf sweep -- the trip to HBM hides behind the sweep
		double z[NU > 0 ? NU : 1];
		sfor<NU>([&](auto I) { z[I] = 0; });
		if (DV && nz_on) {
			if (zpre) {
				asm volatile("" ::: "memory");
				const double *zb = s.zbuf + (zhalf_i > 0 ? s.zhalf : 0ull) + ((size_t)st * s.nenv + ev) * NU;
				sfor<NU>([&](auto I) { z[I] = zb[I]; });
			} else {
				asm volatile("" ::: "memory");
				// (one copy of the generator in the instruction stream: the normals go through the cfrc slots, free at this point)
				const unsigned long long seed = Pq->nz.seed, genv = (unsigned long long)(Pq->nz.env_offset + env);
				double *zl = reinterpret_cast<double *>(smem_le) + 2 * 64 * NV + lane_le;
#pragma nounroll
				for (int i = 0; i < NU; i++) zl[64 * i] = philox_normal(seed, genv, step0 + (unsigned int)st, (unsigned int)i);
				sfor<NU>([&](auto I) { z[I] = zl[64 * I]; });
			}
		}
		bool rs = false;  // mj_resetData ran in THIS step (after the injector wrote ctrl: ctrl and the OU state read zero)

		// ---- mj_checkPos / mj_checkVel (qpos first: its reset hides a bad qvel)
		{
			// (the flags were computed where the state was last in registers: at the load, and in the previous step's mj_Euler)
			const bool badp = badp_next, badv = badv_next;
			// (NO per-lane branch anywhere in this kernel: with ~200 live doubles the register allocator spills around every join, and
			//  ROCm 7.2's LLVM places such spills ahead of the exec restore -- the parked lanes lose them, tools/check_spill_exec.py.
			//  A reset is a handful of selects under a wave-uniform test; every lane issues the counter's atomic, with 0 or 1.)
			const bool bad = badp || badv;
			if (__builtin_amdgcn_ballot_w64(bad)) {
				if constexpr (DP) {
					atomicAdd(s.nwarn + MJB_WARN_BADQPOS, (badp && live) ? 1ull : 0ull);
					atomicAdd(s.nwarn + MJB_WARN_BADQVEL, (!badp && badv && live) ? 1ull : 0ull);
					sfor<NV>([&](auto I) {
						const Pair o = lp[64 * I];
						const double oa = pinv(o.a), ob = pinv(o.b), q0 = pins(tb[T::jnt_bodyid[I]].qpos0);  // (evaluated before the selects, not inside them)
						lp[64 * I] = Pair{ bad ? q0 : oa, bad ? 0.0 : ob };
					});
				}
				time = bad ? 0.0 : time;
				wasreset = wasreset || bad;
				rs = bad;
				if constexpr (DUO) le_barrier();  // (both wavefronts hold the same verdicts: both are here) V reads the reset state
			}
		}

		double qacc[NV], qaccd[NV];  // M^-1 f, and the acceleration Euler advances with: (M + h B)^-1 f under implicit joint damping
		double en_pe = 0, en_ke = 0;
#pragma nounroll
		for (int attempt = 0; attempt < 2; attempt++) {
			// (the tape address laundered per trip: its loads are invariants of this two-trip loop, and the optimiser hoists every one of
			//  them -- ~350 doubles -- in front of it)
			{
				const double MJB_AS4 *tp = m.le_tape;
				asm volatile("" : "+s"(tp));
				th = reinterpret_cast<const LeTapeHdr MJB_AS4 *>(tp);
				tb = reinterpret_cast<const LeTapeBody MJB_AS4 *>(th + 1);
			}
			const LeTapeAct MJB_AS4 *const ta = reinterpret_cast<const LeTapeAct MJB_AS4 *>(tb + NB);
			const bool e_on = DP && last && (m.enableflags & MJB_ENBL_ENERGY);
			bool ep_on = e_on;
			if constexpr (EPOS != DP) ep_on = EPOS && last && (m.enableflags & MJB_ENBL_ENERGY);
			const bool eg_on = ep_on && !(m.disableflags & MJB_DSBL_GRAVITY);
			// (sensordata is an output of the LAUNCH: evaluated at its last step -- or, mjb_set_sensors_every_step, at every step as the generic kernels do:
			//  what A15 costs per step on this kernel is a bench line, other_configs.2_sensors_every_step)
			const bool sens_step = last || s.sens_every_step != 0;
			const bool sens_on = DV && sens_step && !(m.disableflags & MJB_DSBL_SENSOR);  // (a tail lane rewrites the last env's values)
			const bool sensf_on = SENSF && sens_step && !(m.disableflags & MJB_DSBL_SENSOR);
			double *sd = s.sensordata + ev * T::NSENSORDATA;
			double pe = 0;

			// ============ one sweep root -> leaf: A1 mj_kinematics, comPos (cinert, cdof), A8 comVel, A9 RNE's forward pass ============
			// Spatial quantities of a tree are taken about the origin of its root body instead of MuJoCo's subtree com (any common point
			// gives the same qM / qfrc_bias; the com would need every body's pose before the first inertia, i.e. a second sweep with
			// 15 doubles per body kept across).
			double xpos[NB][3], xquat[NB][4], xmat[NB][9];
			double cin[NB][10];  // cinert of the bodies that found no room in LDS
			double cdof[NV > 0 ? NV : 1][6];
			double cvel[NB][6], cacc[NB][6];
			double f[NV > 0 ? NV : 1];  // qfrc_passive + qfrc_applied + qfrc_actuator, then (- qfrc_bias) qfrc_smooth
			double grav[3];
			{
				const bool g_on = !(m.disableflags & MJB_DSBL_GRAVITY);
				for (int k = 0; k < 3; k++) grav[k] = g_on ? th->gravity[k] : 0.0;
			}
			const bool pas_on = !(m.disableflags & MJB_DSBL_PASSIVE);
			__builtin_amdgcn_sched_barrier(0);
			// (two scheduling regions per body, each fetching the NEXT region's half record at its top: the scalar loads of a region
			//  cannot be hoisted beyond it -- left alone, the compiler issues them bodies ahead and parks ~540 SGPRs in VGPR lanes)
			double hA[NB + 1][16], hB[NB][16];
			for (int k = 0; k < 14; k++) hA[1][k] = reinterpret_cast<const double MJB_AS4 *>(tb + 1)[k];
			// ... and the (qpos, qvel) pair of the next jointed body: LDS reads and scalar loads share one counter, so a read issued where
			// it is needed would wait for the record fetched beside it
			double qfa[NV > 0 ? NV : 1];  // qfrc_applied: fetched in the sweep's last region, read by the force block behind it
			Pair pq[NB + 1];
			{
				constexpr int j1 = [] { for (int c = 1; c < NB; c++) if (T::body_jnt[c] >= 0) return T::body_jnt[c]; return -1; }();
				if constexpr (j1 >= 0) pq[T::jnt_bodyid[j1]] = lp[64 * j1];
			}
			// (the trio's P) a hinge's half-angle sine / cosine one body AHEAD: the two polynomial chains depend on qpos alone, so they run beside the
			// previous body's pose chain (quaternion product, normalisation, rsqrt, matrix) and fill its dependency stalls instead of lengthening the
			// critical chain; the body's (qpos, qvel) pair and qpos0 are fetched one more region ahead for that
			[[maybe_unused]] double psn[NB + 2], pcs[NB + 2], q0n[NB + 2];
			[[maybe_unused]] Pair pqn[NB + 2], scq[NB + 2];
			if constexpr (ROLE == 5 && NB > 2) {
				if constexpr (T::body_jnt[2] >= 0) {
					if constexpr (T::jnt_type[T::body_jnt[2]] == MJB_JNT_HINGE) {
						pqn[2] = lp[64 * T::body_jnt[2]];
						q0n[2] = tb[2].qpos0;
					}
				}
			}
			// position-stage sensors on a body's frames (the last step's values are the launch's sensordata)
			auto frame_sensors = [&](auto Bq, const double *xipos) {
				constexpr int b = Bq;
				if (sensf_on) {
					sfor<T::NSENSOR>([&](auto S) {
						constexpr int i = S, type = T::sensor_type[i], ot = T::sensor_objtype[i], id = T::sensor_objid[i], adr = T::sensor_adr[i];
						if constexpr (type == MJB_SENS_FRAMEPOS || type == MJB_SENS_FRAMEQUAT) {
							constexpr int sb = ot == MJB_OBJ_SITE ? T::site_bodyid[id] : id;
							if constexpr (sb == b) {
								double o3[3], o4[4];
								if constexpr (ot == MJB_OBJ_SITE) {
									if constexpr (type == MJB_SENS_FRAMEPOS) {
										if constexpr (T::site_sameframe[id]) {
											for (int k = 0; k < 3; k++) o3[k] = xpos[b][k];
										} else {
											double sp[3], v[3];
											ldc3(sp, m.site_pos + 3 * id);
											matvec3(v, xmat[b], sp);
											for (int k = 0; k < 3; k++) o3[k] = v[k] + xpos[b][k];
										}
									} else {
										double sq[4];
										ldc4(sq, m.site_quat + 4 * id);
										qmul(o4, xquat[b], sq);
									}
								} else if constexpr (ot == MJB_OBJ_BODY) {
									if constexpr (type == MJB_SENS_FRAMEPOS) {
										for (int k = 0; k < 3; k++) o3[k] = xipos[k];
									} else {
										double iq[4];
										ldc4(iq, m.body_iquat + 4 * b);
										qmul(o4, xquat[b], iq);
									}
								} else {  // xbody
									for (int k = 0; k < 3; k++) o3[k] = xpos[b][k];
									for (int k = 0; k < 4; k++) o4[k] = xquat[b][k];
								}
								if constexpr (type == MJB_SENS_FRAMEPOS) {
									const double cut = m.sensor_cutoff[i];
									for (int k = 0; k < 3; k++) sd[adr + k] = cut > 0 ? clampd(o3[k], -cut, cut) : o3[k];
								} else {
									for (int k = 0; k < 4; k++) sd[adr + k] = o4[k];
								}
							}
						}
					});
				}
			};
			// (ring consumers) one body off the ring: its pose (position relative to the tree root's origin), cinert, cdof; V: velocities and the body's force
			auto consume = [&](auto Bq) {
				constexpr int b = Bq;
				constexpr int p = T::body_parentid[b], j = T::body_jnt[b];
				constexpr int ord = (LD::slot(b) - NV) / 3, rg = RING + 6 * (ord % RINGN), c0 = LD::cin_slot(b);
				static_assert(c0 >= 0, "pipelined forms: every needed body's cinert lives in LDS");
				double xp[3], xm[9], ci[10];
				{
					const Pair a0 = lp[64 * rg], a1 = lp[64 * (rg + 1)], a2 = lp[64 * (rg + 2)], a3 = lp[64 * (rg + 3)], a4 = lp[64 * (rg + 4)], a5 = lp[64 * (rg + 5)];
					xp[0] = a0.a; xp[1] = a0.b; xp[2] = a1.a; xm[0] = a1.b; xm[1] = a2.a; xm[2] = a2.b; xm[3] = a3.a; xm[4] = a3.b; xm[5] = a4.a; xm[6] = a4.b; xm[7] = a5.a; xm[8] = a5.b;
				}
				if constexpr ((ROLE == 4 && b == LASTB) || ROLE == 7) {
					// the LAST needed body's cinert came with its pose: P computes that one itself and goes from its last pose straight into the
					// composite-inertia sweep, which starts at this body -- it never waits for V's last phase
					for (int k = 0; k < 5; k++) {
						const Pair c = lp[64 * (c0 + k)];
						ci[2 * k] = c.a;
						ci[2 * k + 1] = c.b;
					}
				} else
				{
					// cinert about the tree root's origin, as in the fused sweep: X Ib X' + the com offset's terms; handed to P's composite-inertia sweep
					const LeTapeBody MJB_AS4 &tj = tb[b];
					double dif[3] = { xp[0], xp[1], xp[2] };
					if constexpr (!T::body_sameframe[b]) {
						const double ip[3] = { tj.ipos[0], tj.ipos[1], tj.ipos[2] };
						double v[3];
						matvec3(v, xm, ip);
						for (int k = 0; k < 3; k++) dif[k] += v[k];
					}
					const double mass = tj.mass;
					const double *X = xm;
					const double ixx = tj.ibody[0], iyy = tj.ibody[1], izz = tj.ibody[2], ixy = tj.ibody[3], ixz = tj.ibody[4], iyz = tj.ibody[5];
					double Tm[9];
					for (int rr = 0; rr < 3; rr++) {
						Tm[3 * rr + 0] = X[3 * rr] * ixx + X[3 * rr + 1] * ixy + X[3 * rr + 2] * ixz;
						Tm[3 * rr + 1] = X[3 * rr] * ixy + X[3 * rr + 1] * iyy + X[3 * rr + 2] * iyz;
						Tm[3 * rr + 2] = X[3 * rr] * ixz + X[3 * rr + 1] * iyz + X[3 * rr + 2] * izz;
					}
					ci[0] = Tm[0] * X[0] + Tm[1] * X[1] + Tm[2] * X[2] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
					ci[1] = Tm[3] * X[3] + Tm[4] * X[4] + Tm[5] * X[5] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
					ci[2] = Tm[6] * X[6] + Tm[7] * X[7] + Tm[8] * X[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
					ci[3] = Tm[0] * X[3] + Tm[1] * X[4] + Tm[2] * X[5] - mass * dif[0] * dif[1];
					ci[4] = Tm[0] * X[6] + Tm[1] * X[7] + Tm[2] * X[8] - mass * dif[0] * dif[2];
					ci[5] = Tm[3] * X[6] + Tm[4] * X[7] + Tm[5] * X[8] - mass * dif[1] * dif[2];
					ci[6] = mass * dif[0];
					ci[7] = mass * dif[1];
					ci[8] = mass * dif[2];
					ci[9] = mass;
					if constexpr (ROLE != 7) for (int k = 0; k < 5; k++) lp[64 * (c0 + k)] = Pair{ ci[2 * k], ci[2 * k + 1] };  // (for the composite-inertia sweep of P / of C itself)
				}
				[[maybe_unused]] double pv[6], pa[6];
				if constexpr (!DV) {
				} else if constexpr (p == 0 || !LD::needed(p)) {  // the world, or a jointless chain down from it: at rest
					for (int k = 0; k < 6; k++) pv[k] = 0;
					pa[0] = pa[1] = pa[2] = 0;
					for (int k = 0; k < 3; k++) pa[3 + k] = -grav[k];
				} else {
					for (int k = 0; k < 6; k++) { pv[k] = cvel[p][k]; pa[k] = cacc[p][k]; }
				}
				if constexpr (j >= 0) {
					[[maybe_unused]] double qv = 0;
					if constexpr (DV) qv = lp[64 * j].b;
					const LeTapeBody MJB_AS4 &tj = tb[b];
					const double ax[3] = { tj.jaxis[0], tj.jaxis[1], tj.jaxis[2] };
					double xaxis[3];
					matvec3(xaxis, xm, ax);
					double *cd = cdof[j];
					if constexpr (T::jnt_type[j] == MJB_JNT_SLIDE) {
						cd[0] = cd[1] = cd[2] = 0;
						for (int k = 0; k < 3; k++) cd[3 + k] = xaxis[k];
					} else {
						// (the anchor from the body's FINAL frame: xpos + xmat jnt_pos -- the point mj_kinematics' off-centre correction keeps fixed)
						const double jp[3] = { tj.jpos[0], tj.jpos[1], tj.jpos[2] };
						double xanch[3] = { xp[0], xp[1], xp[2] }, off[3];
						if (jp[0] != 0 || jp[1] != 0 || jp[2] != 0) {
							double v[3];
							matvec3(v, xm, jp);
							for (int k = 0; k < 3; k++) xanch[k] += v[k];
						}
						for (int k = 0; k < 3; k++) off[k] = -xanch[k];  // (root origin - anchor)
						for (int k = 0; k < 3; k++) cd[k] = xaxis[k];
						cross3(cd + 3, xaxis, off);
					}
					if constexpr (!DV) {
					} else if constexpr (p == 0 || !LD::needed(p)) {
						for (int k = 0; k < 6; k++) { cvel[b][k] = cd[k] * qv; cacc[b][k] = pa[k]; }
					} else {
						double cdd[6];
						cross_motion(cdd, pv, cd);
						for (int k = 0; k < 6; k++) { cvel[b][k] = pv[k] + cd[k] * qv; cacc[b][k] = pa[k] + cdd[k] * qv; }
					}
				} else if constexpr (DV) {
					for (int k = 0; k < 6; k++) { cvel[b][k] = pv[k]; cacc[b][k] = pa[k]; }
				}
				if constexpr (DV) {
					double cf[6], t0[6], t1[6];
					mul_inert_vec(cf, ci, cacc[b]);
					mul_inert_vec(t0, ci, cvel[b]);
					cross_force(t1, cvel[b], t0);
					constexpr int q0 = LD::slot(b);
					lp[64 * q0] = Pair{ cf[0] + t1[0], cf[1] + t1[1] };
					lp[64 * (q0 + 1)] = Pair{ cf[2] + t1[2], cf[3] + t1[3] };
					lp[64 * (q0 + 2)] = Pair{ cf[4] + t1[4], cf[5] + t1[5] };
				}
			};
			sfor<NB>([&](auto B) {
				constexpr int b = B;
				if constexpr (b > 0 && RINGC) {
					// ---- the pipelined V (and the trio's C): body b's pose and cinert come through LDS, one barrier per needed body.  The trio's V runs TWO bodies
					// behind P: it takes the cinert C computed one phase earlier instead of computing its own, and never paces the pose chain
					if constexpr (LD::needed(b)) {
						if constexpr (ROLE == 6) {
							// (the trio's C) a hinge's half-angle sine and cosine are a third of the pose chain's instructions and depend on qpos alone: C computes them
							// two bodies ahead of P, in the time it would wait at this barrier, and P picks them up from LDS
							constexpr int t2 = [] {
								int c = b, hops = 0;
								for (int q = b + 1; q < NB && hops < 2; q++) if (LD::needed(q)) { c = q; hops++; }
								return hops == 2 ? c : 0;
							}();
							if constexpr (t2 > 0 && SCUSE(t2)) {
								double sn, cs;
								const double qp2 = lp[64 * T::body_jnt[t2]].a;
								sincos_nb((qp2 - tb[t2].qpos0) * 0.5, &sn, &cs);
								lp[64 * (SC0 + t2)] = Pair{ sn, cs };
							}
						}
						LE_PK(0);
						le_barrier();
						LE_PK(2);
						if constexpr (ROLE != 7) consume(IC<b>{});
						else {
							constexpr int pb = [] { for (int c = b - 1; c >= 1; c--) if (LD::needed(c)) return c; return 0; }();
							if constexpr (pb > 0) consume(IC<pb>{});
						}
					}
					if constexpr (DV && b == 1) sfor<NV>([&](auto I) { qfa[I] = s.qfrc_applied[ev * NV + I]; });  // (read by the force block behind the sweep: a trip to HBM)
					__builtin_amdgcn_sched_barrier(0);
				}
				if constexpr (b > 0 && POSE) {
				constexpr int p = T::body_parentid[b], j = T::body_jnt[b], r = T::body_rootid[b];
				touch_s(hA[b][0]);
				if constexpr (j >= 0) touch_v(pq[b].a);
				for (int k = 0; k < 10; k++) hB[b][k] = reinterpret_cast<const double MJB_AS4 *>(tb + b)[16 + k];
				const double *const A = hA[b];  // pos[3] quat[4] jaxis[3] jpos[3] qpos0 stiffness spring
				if constexpr (ROLE == 5 && b == 1 && b + 1 < NB) {
					if constexpr (T::body_jnt[b + 1] >= 0) {
						if constexpr (T::jnt_type[T::body_jnt[b + 1]] == MJB_JNT_HINGE) sincos_nb((pqn[b + 1].a - q0n[b + 1]) * 0.5, &psn[b + 1], &pcs[b + 1]);
					}
				}
				double pos[3] = { A[0], A[1], A[2] }, quat[4] = { A[3], A[4], A[5], A[6] };
				if constexpr (p != 0) {
					double v[3], q[4];
					matvec3(v, xmat[p], pos);
					for (int k = 0; k < 3; k++) pos[k] = v[k] + xpos[p][k];
					qmul(q, xquat[p], quat);
					for (int k = 0; k < 4; k++) quat[k] = q[k];
				}
				[[maybe_unused]] double xaxis[3], xanch[3], qp = 0, qv = 0;
				[[maybe_unused]] bool offc = false;
				if constexpr (j >= 0) {
					qp = pq[b].a;
					qv = pq[b].b;
					// The joint's world axis is its local axis through the body's FINAL orientation (a hinge turns about it, a slide does not
					// turn), and a hinge's anchor stays where it was: the frame before the joint motion -- mj_kinematics' xaxis / xanchor
					// source -- is only needed for an off-centre anchor (jnt_pos != 0, wave-uniform).
					const double jp[3] = { A[10], A[11], A[12] };
					offc = jp[0] != 0 || jp[1] != 0 || jp[2] != 0;
					for (int k = 0; k < 3; k++) xanch[k] = pos[k];
					if (offc) {
						double M0[9], v[3];
						quat2mat_nocheck(M0, quat);
						matvec3(v, M0, jp);
						for (int k = 0; k < 3; k++) xanch[k] += v[k];
					}
					if constexpr (T::jnt_type[j] == MJB_JNT_HINGE) {
						double sn, cs, ql[4], q[4];
						if constexpr (ROLE == 5 && SCUSE(b)) { sn = scq[b].a; cs = scq[b].b; }  // (C's, fetched in the previous body's second region)
						else if constexpr (ROLE == 5 && b == 2) { sn = psn[b]; cs = pcs[b]; }  // (computed beside the first body's chain)
						else sincos_nb((qp - A[13]) * 0.5, &sn, &cs);
						ql[0] = cs; ql[1] = A[7] * sn; ql[2] = A[8] * sn; ql[3] = A[9] * sn;
						qmul(q, quat, ql);
						for (int k = 0; k < 4; k++) quat[k] = q[k];
					}
					if (ep_on && pas_on) {  // mj_energyPos: the joint spring
						const double dqs = qp - tb[b].spring;  // (last step only)
						pe += 0.5 * tb[b].stiffness * dqs * dqs;
					}
				}
				normalize4_sel(quat);
				for (int k = 0; k < 4; k++) xquat[b][k] = quat[k];
				quat2mat_nocheck(xmat[b], quat);
				if constexpr (j >= 0) {
					const double ax[3] = { A[7], A[8], A[9] };
					matvec3(xaxis, xmat[b], ax);
					if constexpr (T::jnt_type[j] == MJB_JNT_SLIDE) {
						const double dq = qp - A[13];
						for (int k = 0; k < 3; k++) pos[k] += xaxis[k] * dq;
					} else if (offc) {  // correct for off-centre rotation
						const double jp[3] = { A[10], A[11], A[12] };
						double v[3];
						matvec3(v, xmat[b], jp);
						for (int k = 0; k < 3; k++) pos[k] = xanch[k] - v[k];
					}
				}
				for (int k = 0; k < 3; k++) xpos[b][k] = pos[k];
				touch_rec<10>(hB[b]);
				__builtin_amdgcn_sched_barrier(0);
				// ---- second region: inertial frame, cinert, velocities, forces; the next body's pose half on its way
				touch_s(hB[b][0]);
				if constexpr (b + 1 < NB) for (int k = 0; k < 14; k++) hA[b + 1][k] = reinterpret_cast<const double MJB_AS4 *>(tb + b + 1)[k];
				{
					constexpr int nb = [] { for (int c = b + 1; c < NB; c++) if (T::body_jnt[c] >= 0) return c; return -1; }();
					if constexpr (nb >= 0) pq[nb] = lp[64 * T::body_jnt[nb]];
				}
				if constexpr (ROLE == 5 && SCUSE(b + 1)) scq[b + 1] = lp[64 * (SC0 + b + 1)];
				const double *const Bh = hB[b];  // ipos[3] ibody[6] mass damping armature hdamping
				// inertial frame
				double xipos[3];
				if constexpr (T::body_sameframe[b]) {
					for (int k = 0; k < 3; k++) xipos[k] = xpos[b][k];
				} else {
					double ip[3] = { Bh[0], Bh[1], Bh[2] }, v[3];
					matvec3(v, xmat[b], ip);
					for (int k = 0; k < 3; k++) xipos[k] = v[k] + xpos[b][k];
				}
				const double mass = Bh[9];
				if (eg_on) pe -= mass * (grav[0] * xipos[0] + grav[1] * xipos[1] + grav[2] * xipos[2]);
				frame_sensors(B, xipos);
				if constexpr (LD::needed(b)) {
					// cinert about the tree root's origin (mju_inertCom with that offset)
					[[maybe_unused]] double ci[10];
					if constexpr (ROLE <= 2 || (ROLE == 3 && b == LASTB)) {
						double dif[3];
						if constexpr (r == b) { dif[0] = xipos[0] - pos[0]; dif[1] = xipos[1] - pos[1]; dif[2] = xipos[2] - pos[2]; }
						else for (int k = 0; k < 3; k++) dif[k] = xipos[k] - xpos[r][k];
						// world inertia X Ib X' with the body-frame inertia matrix Ib = R(iquat) diag(inertia) R(iquat)' from the tape
						// (mju_inertCom builds the same matrix as ximat diag ximat', ximat = X R(iquat))
						const double *X = xmat[b];
						const double ixx = Bh[3], iyy = Bh[4], izz = Bh[5], ixy = Bh[6], ixz = Bh[7], iyz = Bh[8];
						double Tm[9];
						for (int rr = 0; rr < 3; rr++) {
							Tm[3 * rr + 0] = X[3 * rr] * ixx + X[3 * rr + 1] * ixy + X[3 * rr + 2] * ixz;
							Tm[3 * rr + 1] = X[3 * rr] * ixy + X[3 * rr + 1] * iyy + X[3 * rr + 2] * iyz;
							Tm[3 * rr + 2] = X[3 * rr] * ixz + X[3 * rr + 1] * iyz + X[3 * rr + 2] * izz;
						}
						double *res = ci;
						res[0] = Tm[0] * X[0] + Tm[1] * X[1] + Tm[2] * X[2] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
						res[1] = Tm[3] * X[3] + Tm[4] * X[4] + Tm[5] * X[5] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
						res[2] = Tm[6] * X[6] + Tm[7] * X[7] + Tm[8] * X[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
						res[3] = Tm[0] * X[3] + Tm[1] * X[4] + Tm[2] * X[5] - mass * dif[0] * dif[1];
						res[4] = Tm[0] * X[6] + Tm[1] * X[7] + Tm[2] * X[8] - mass * dif[0] * dif[2];
						res[5] = Tm[3] * X[6] + Tm[4] * X[7] + Tm[5] * X[8] - mass * dif[1] * dif[2];
						res[6] = mass * dif[0];
						res[7] = mass * dif[1];
						res[8] = mass * dif[2];
						res[9] = mass;
					}
					// parent's velocity / acceleration (world: zero velocity, -gravity)
					[[maybe_unused]] double pv[6], pa[6];
					if constexpr (!DV) {
					} else if constexpr (p == 0) {
						for (int k = 0; k < 6; k++) pv[k] = 0;
						pa[0] = pa[1] = pa[2] = 0;
						for (int k = 0; k < 3; k++) pa[3 + k] = -grav[k];
					} else if constexpr (!LD::needed(p)) {  // a jointless chain down from the world: at rest
						for (int k = 0; k < 6; k++) pv[k] = 0;
						pa[0] = pa[1] = pa[2] = 0;
						for (int k = 0; k < 3; k++) pa[3 + k] = -grav[k];
					} else {
						for (int k = 0; k < 6; k++) { pv[k] = cvel[p][k]; pa[k] = cacc[p][k]; }
					}
					if constexpr (ROLE == 5) {  // (the trio's P: poses only)
					} else if constexpr (j >= 0) {
						double *cd = cdof[j];
						if constexpr (T::jnt_type[j] == MJB_JNT_SLIDE) {
							cd[0] = cd[1] = cd[2] = 0;
							for (int k = 0; k < 3; k++) cd[3 + k] = xaxis[k];
						} else {
							double off[3];
							if constexpr (r == b) for (int k = 0; k < 3; k++) off[k] = pos[k] - xanch[k];
							else for (int k = 0; k < 3; k++) off[k] = xpos[r][k] - xanch[k];
							for (int k = 0; k < 3; k++) cd[k] = xaxis[k];
							cross3(cd + 3, xaxis, off);
						}
						if constexpr (!DV) {
						} else if constexpr (p == 0 || !LD::needed(p)) {
							// cdof_dot = cvel(parent) x cdof = 0
							for (int k = 0; k < 6; k++) { cvel[b][k] = cd[k] * qv; cacc[b][k] = pa[k]; }
						} else {
							double cdd[6];
							cross_motion(cdd, pv, cd);
							for (int k = 0; k < 6; k++) { cvel[b][k] = pv[k] + cd[k] * qv; cacc[b][k] = pa[k] + cdd[k] * qv; }
						}
					} else if constexpr (DV) {
						for (int k = 0; k < 6; k++) { cvel[b][k] = pv[k]; cacc[b][k] = pa[k]; }
					}
					if constexpr (DP && (ROLE != 3 || b == LASTB)) {  // the composite-inertia sweep reads it back (pipelined duo: the last body's, which V's force computation reads too)
						if constexpr (LD::cin_slot(b) >= 0) {
							constexpr int c0 = LD::cin_slot(b);
							for (int k = 0; k < 5; k++) lp[64 * (c0 + k)] = Pair{ ci[2 * k], ci[2 * k + 1] };
						} else {
							for (int k = 0; k < 10; k++) cin[b][k] = ci[k];
						}
					}
					if constexpr (DV) {
						// cfrc_body = cinert * cacc + cvel x* (cinert * cvel): parked in LDS for the backward sweep
						double cf[6], t0[6], t1[6];
						mul_inert_vec(cf, ci, cacc[b]);
						mul_inert_vec(t0, ci, cvel[b]);
						cross_force(t1, cvel[b], t0);
						constexpr int q0 = LD::slot(b);
						lp[64 * q0] = Pair{ cf[0] + t1[0], cf[1] + t1[1] };
						lp[64 * (q0 + 1)] = Pair{ cf[2] + t1[2], cf[3] + t1[3] };
						lp[64 * (q0 + 2)] = Pair{ cf[4] + t1[4], cf[5] + t1[5] };
					}
					if constexpr (ROLE == 3 || ROLE == 5) {  // the pose to V (and, of three, to C)
						constexpr int ord = (LD::slot(b) - NV) / 3, rg = RING + 6 * (ord % RINGN);
						const double *xm = xmat[b];
						double xp[3] = { 0, 0, 0 };  // relative to the tree root's origin: what cdof is taken about
						if constexpr (r != b) for (int k = 0; k < 3; k++) xp[k] = xpos[b][k] - xpos[r][k];
						lp[64 * rg] = Pair{ xp[0], xp[1] };
						lp[64 * (rg + 1)] = Pair{ xp[2], xm[0] };
						lp[64 * (rg + 2)] = Pair{ xm[1], xm[2] };
						lp[64 * (rg + 3)] = Pair{ xm[3], xm[4] };
						lp[64 * (rg + 4)] = Pair{ xm[5], xm[6] };
						lp[64 * (rg + 5)] = Pair{ xm[7], xm[8] };
						LE_PK(0);
						le_barrier();
						LE_PK(2);
					}
				}
				if constexpr (b + 1 < NB) touch_rec<14>(hA[b + 1]);
				if constexpr (DV && b == NB - 1) sfor<NV>([&](auto I) { qfa[I] = s.qfrc_applied[ev * NV + I]; });
				__builtin_amdgcn_sched_barrier(0);
				}
			});

			LE_PK(0);
			if constexpr (ROLE >= 5) {  // (the trio: C's cinert of the last body is in LDS; V takes that body now)
				le_barrier();
				if constexpr (ROLE == 7) consume(IC<LASTB>{});
			}
			LE_PK(1);
			// ============ A8 mj_passive, the injector's OU update, A12 mj_fwdActuation (joint transmission), qfrc_applied ============
			if constexpr (DV) {
				if (nz_on && attempt == 0) {
					const double rate = Pq->nz.rate, scale = Pq->nz.scale;
					sfor<NU>([&](auto I) { const double v = rate * cn[I] + scale * z[I]; cn[I] = rs ? 0.0 : v; });
				}
				double ctrl[NU > 0 ? NU : 1];
				if (nz_on) {
					sfor<NU>([&](auto I) { ctrl[I] = cn[I]; });
				} else {
					sfor<NU>([&](auto I) { const double c = pinv(s.ctrl[ev * NU + I]); ctrl[I] = wasreset ? 0.0 : c; });  // (load first: a select around a load becomes a per-lane branch)
				}
				Pair sq[NV > 0 ? NV : 1];
				sfor<NV>([&](auto I) {
					constexpr int j = I;
					sq[j] = lp[64 * j];
					const LeTapeBody MJB_AS4 &tj = tb[T::jnt_bodyid[j]];
					double pas = 0;
					if (pas_on) {
						pas = -tj.stiffness * (sq[j].a - tj.spring);
						pas -= tj.damping * sq[j].b;
					}
					f[j] = pas + (wasreset ? 0.0 : qfa[j]);
				});
				const bool act_on = !(m.disableflags & MJB_DSBL_ACTUATION);
				const bool clamp_on = !(m.disableflags & MJB_DSBL_CLAMPCTRL);
				sfor<NU>([&](auto U) {
					constexpr int i = U, j = T::act_jnt[i];
					double force = 0;
					const LeTapeAct MJB_AS4 &A = ta[i];
					const double gear = A.gear;
					if (act_on) {
						double c = ctrl[i];
						if constexpr (T::act_ctrllimited[i]) {
							if (clamp_on) c = clampd(c, A.ctrllo, A.ctrlhi);
						}
						const double len = sq[j].a * gear, vel = sq[j].b * gear;
						double gain = A.gain[0], bs = 0;
						if constexpr (T::act_gaintype[i] == MJB_GAIN_AFFINE) gain = gain + A.gain[1] * len + A.gain[2] * vel;
						if constexpr (T::act_biastype[i] == MJB_BIAS_AFFINE) bs = A.bias[0] + A.bias[1] * len + A.bias[2] * vel;
						force = gain * c + bs;
						if constexpr (T::act_forcelimited[i]) force = clampd(force, A.forcelo, A.forcehi);
						f[j] += gear * force;
					}
					if (sens_on) {
						sfor<T::NSENSOR>([&](auto S) {
							constexpr int q = S;
							if constexpr (T::sensor_objid[q] == i && (T::sensor_type[q] == MJB_SENS_ACTUATORFRC || T::sensor_type[q] == MJB_SENS_ACTUATORPOS || T::sensor_type[q] == MJB_SENS_ACTUATORVEL)) {
								double v = T::sensor_type[q] == MJB_SENS_ACTUATORFRC ? force : (T::sensor_type[q] == MJB_SENS_ACTUATORPOS ? sq[j].a * gear : sq[j].b * gear);
								const double cut = m.sensor_cutoff[q];
								sd[T::sensor_adr[q]] = cut > 0 ? clampd(v, -cut, cut) : v;
							}
						});
					}
				});
			}
			if (sens_on) {
				sfor<T::NSENSOR>([&](auto S) {
					constexpr int q = S, type = T::sensor_type[q];
					if constexpr (type == MJB_SENS_JOINTPOS || type == MJB_SENS_JOINTVEL || type == MJB_SENS_CLOCK) {
						constexpr int jj = T::sensor_objid[q] < 0 ? 0 : T::sensor_objid[q];
						const Pair s3 = lp[64 * jj];
						const double v = type == MJB_SENS_CLOCK ? time : (type == MJB_SENS_JOINTPOS ? s3.a : s3.b);
						const double cut = m.sensor_cutoff[q];
						sd[T::sensor_adr[q]] = cut > 0 ? clampd(v, -cut, cut) : v;
					}
				});
			}
			__builtin_amdgcn_sched_barrier(0);

			LE_PK(2);
			// ================= A2 mj_crb + A9 RNE backward pass, one sweep leaf -> root =================
			double qM[NV > 0 ? NV : 1][NV > 0 ? NV : 1];  // [i][a], a = i or an ancestor of i (the other entries never exist)
			double csum[NB][6], crbs[NB][10];             // forces / composite inertias of a body's children, summed as the sweep passes them
			double pcf[NB][6], pcb[NB][10], parm[NB];     // the NEXT body's force / cinert / joint armature, fetched one region ahead
			auto fetch_body = [&](auto Bn) {
				constexpr int nb = Bn;
				constexpr int q0 = LD::slot(nb);
				if constexpr (DP && T::body_jnt[nb] >= 0) parm[nb] = tb[nb].armature;
				if constexpr (DV) {
					const Pair c0 = lp[64 * q0], c1 = lp[64 * (q0 + 1)], c2 = lp[64 * (q0 + 2)];
					pcf[nb][0] = c0.a; pcf[nb][1] = c0.b; pcf[nb][2] = c1.a; pcf[nb][3] = c1.b; pcf[nb][4] = c2.a; pcf[nb][5] = c2.b;
				}
				if constexpr (DP && LD::cin_slot(nb) >= 0) {
					constexpr int cs = LD::cin_slot(nb);
					for (int k = 0; k < 5; k++) {
						const Pair c = lp[64 * (cs + k)];
						pcb[nb][2 * k] = c.a;
						pcb[nb][2 * k + 1] = c.b;
					}
				}
			};
			{
				constexpr int lastb = [] { for (int c = NB - 1; c >= 1; c--) if (LD::needed(c)) return c; return 0; }();
				if constexpr (lastb > 0) fetch_body(IC<lastb>{});
			}
			sfor<NB - 1>([&](auto Bi) {
				constexpr int b = NB - 1 - Bi;
				constexpr int p = T::body_parentid[b], j = T::body_jnt[b];
				if constexpr (LD::needed(b)) {
					if constexpr (DV) touch_v(pcf[b][0]);
					else if constexpr (DP && LD::cin_slot(b) >= 0) touch_v(pcb[b][0]);
					if constexpr (DP && j >= 0) touch_s(parm[b]);
					{
						constexpr int nb = [] { for (int c = b - 1; c >= 1; c--) if (LD::needed(c)) return c; return 0; }();
						if constexpr (nb > 0) fetch_body(IC<nb>{});
					}
					[[maybe_unused]] double cf[6] = { 0, 0, 0, 0, 0, 0 };
					if constexpr (DV) for (int k = 0; k < 6; k++) cf[k] = pcf[b][k];
					// (children have larger ids: their sums are complete; cset is a compile-time fact after unrolling)
					constexpr bool has_child = [] { for (int c = b + 1; c < NB; c++) if (T::body_parentid[c] == b && LD::needed(c)) return true; return false; }();
					if constexpr (DV && has_child) for (int k = 0; k < 6; k++) cf[k] += csum[b][k];
					[[maybe_unused]] double cb[10];  // composite inertia of the body (mj_crb)
					if constexpr (DP) {
						if constexpr (LD::cin_slot(b) >= 0) {
							for (int k = 0; k < 10; k++) cb[k] = pcb[b][k];
						} else {
							for (int k = 0; k < 10; k++) cb[k] = cin[b][k];
						}
						if constexpr (has_child) for (int k = 0; k < 10; k++) cb[k] += crbs[b][k];
					}
					if constexpr (j >= 0) {
						if constexpr (DV) f[j] -= dot6r(cdof[j], cf);
						if constexpr (DP) {
							double buf[6];
							mul_inert_vec(buf, cb, cdof[j]);
							sfor<NV>([&](auto A) {
								constexpr int a = A;
								if constexpr (Q::anc(a, j)) qM[j][a] = (a == j ? parm[b] : 0.0) + dot6r(cdof[a], buf);
							});
						}
					}
					if constexpr (p > 0 && LD::needed(p)) {
						constexpr bool first = [] { for (int c = b + 1; c < NB; c++) if (T::body_parentid[c] == p && LD::needed(c)) return false; return true; }();
						if constexpr (DV) {
							if constexpr (first) for (int k = 0; k < 6; k++) csum[p][k] = cf[k];
							else for (int k = 0; k < 6; k++) csum[p][k] += cf[k];
						}
						if constexpr (DP) {
							if constexpr (first) for (int k = 0; k < 10; k++) crbs[p][k] = cb[k];
							else for (int k = 0; k < 10; k++) crbs[p][k] += cb[k];
						}
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			});
			LE_PK(3);
			if constexpr (ROLE == 5) en_pe = pe;
			if constexpr (DUO && !DP) {
				// ---- V's half ends here: qfrc_smooth to P, then P's verdict on the step (the trio's P: the two rendezvous and the verdict)
				if constexpr (DV) sfor<(NV + 1) / 2>([&](auto K) {
					constexpr int k = K;
					lp[64 * (XS + k)] = Pair{ f[2 * k], 2 * k + 1 < NV ? f[2 * k + 1 < NV ? 2 * k + 1 : 0] : 0.0 };
				});
				LE_PK(4);
				le_barrier();  // (A) qfrc_smooth is in LDS; P's factors are ready
				LE_PK(5);
				le_barrier();  // (B) P has solved and either integrated or -- bad qacc -- reset its lanes
				const int code = (int)lp[64 * MAIL].a;
				LE_PK(6);
				if (!(__builtin_amdgcn_readfirstlane(code) & 8)) {  // (bit 3: P's ballot, the same in every lane)
					badp_next = (code & 2) != 0;
					badv_next = (code & 1) != 0;
					break;
				}
				const bool bada = (code & 4) != 0;  // mj_checkAcc reset this lane: the forward pass runs once more
				sfor<NU>([&](auto I) { cn[I] = bada ? 0.0 : cn[I]; });
				time = bada ? 0.0 : time;
				wasreset = wasreset || bada;
				continue;
			}
			if (e_on) {  // mj_energyVel: 0.5 qvel' M qvel (mj_energyPos was gathered along the sweep)
				double ke = 0, qv[NV > 0 ? NV : 1];
				sfor<NV>([&](auto I) { qv[I] = lp[64 * I].b; });
				sfor<NV>([&](auto I) {
					sfor<NV>([&](auto A) {
						constexpr int i = I, a = A;
						if constexpr (Q::anc(a, i)) ke += (a == i ? 0.5 : 1.0) * qM[i][a] * qv[i] * qv[a];
					});
				});
				if constexpr (EPOS) en_pe = pe;
				en_ke = ke;
			}

			// ================= A3 mj_factorM + A12 mj_fwdAcceleration; A16's (M + h B) factor and solve beside them =================
			// L'DL in place, pivots from the last dof up: row k scaled by 1 / D_k, then row i -= L_ki * (row k restricted to i's ancestors)
			const bool damp_on = m.eulerdamp != 0;
			constexpr bool FH = true;
			double qH[NV > 0 ? NV : 1][NV > 0 ? NV : 1], dinv[NV > 0 ? NV : 1], hinv[NV > 0 ? NV : 1];
			if constexpr (FH) {
				sfor<NV>([&](auto I) {
					sfor<NV>([&](auto A) {
						constexpr int i = I, a = A;
						if constexpr (Q::anc(a, i)) qH[i][a] = qM[i][a] + (a == i ? tb[T::jnt_bodyid[i]].hdamping : 0.0);
					});
				});
			}
			sfor<NV>([&](auto Ki) {
				constexpr int k = NV - 1 - Ki;
				dinv[k] = frcp(qM[k][k]);
				if constexpr (FH) hinv[k] = frcp(qH[k][k]);
				sfor<NV>([&](auto Ii) {
					constexpr int i = NV - 1 - Ii;  // ancestors of k, nearest first
					if constexpr (i < k && Q::anc(i, k)) {
						const double tm = qM[k][i] * dinv[k];
						[[maybe_unused]] double th = 0;
						if constexpr (FH) th = qH[k][i] * hinv[k];
						sfor<NV>([&](auto A) {
							constexpr int a = A;
							if constexpr (Q::anc(a, i)) {
								qM[i][a] -= tm * qM[k][a];
								if constexpr (FH) qH[i][a] -= th * qH[k][a];
							}
						});
						qM[k][i] = tm;
						if constexpr (FH) qH[k][i] = th;
					}
				});
			});
			// x = M^-1 f and y = (M + h B)^-1 f: L' sweep, D, L sweep
			double x[NV > 0 ? NV : 1], y[NV > 0 ? NV : 1];
			if constexpr (DUO && DP && NV > 0) {  // (the factors BEFORE the rendezvous: left alone the compiler sinks them behind it, next to the solves)
				touch_v(dinv[0]);
				if constexpr (FH) touch_v(hinv[0]);
			}
			LE_PK(4);
			if constexpr (DUO && DP) {
				le_barrier();  // (A) V's qfrc_smooth
				LE_PK(5);
				sfor<(NV + 1) / 2>([&](auto K) {
					constexpr int k = K;
					const Pair v = lp[64 * (XS + k)];
					f[2 * k] = v.a;
					if constexpr (2 * k + 1 < NV) f[2 * k + 1] = v.b;
				});
			}
			sfor<NV>([&](auto I) { x[I] = f[I]; y[I] = f[I]; });
			sfor<NV>([&](auto Ii) {
				constexpr int i = NV - 1 - Ii;
				sfor<NV>([&](auto A) {
					constexpr int a = A;
					if constexpr (a < i && Q::anc(a, i)) { x[a] -= qM[i][a] * x[i]; if constexpr (FH) y[a] -= qH[i][a] * y[i]; }
				});
			});
			sfor<NV>([&](auto I) { x[I] *= dinv[I]; if constexpr (FH) y[I] *= hinv[I]; });
			sfor<NV>([&](auto I) {
				constexpr int i = I;
				sfor<NV>([&](auto Ai) {
					constexpr int a = NV - 1 - Ai;  // nearest ancestor first, as mj_solveLD walks them
					if constexpr (a < i && Q::anc(a, i)) { x[i] -= qM[i][a] * x[a]; if constexpr (FH) y[i] -= qH[i][a] * y[a]; }
				});
			});
			sfor<NV>([&](auto I) { qacc[I] = x[I]; qaccd[I] = damp_on ? y[I] : x[I]; });

			// ---- mj_checkAcc: a bad qacc resets the env and the forward pass runs once more (mj_step)
			if (attempt) break;
			bool bada = false;
			sfor<NV>([&](auto I) { bada |= bad_val(qacc[I]); });
			if (!__builtin_amdgcn_ballot_w64(bada)) break;  // (wave-uniform: the rare second trip recomputes every lane; the others get the same values)
			atomicAdd(s.nwarn + MJB_WARN_BADQACC, (bada && live) ? 1ull : 0ull);
			sfor<NV>([&](auto I) {
				const Pair o = lp[64 * I];
				const double oa = pinv(o.a), ob = pinv(o.b), q0 = pins(tb[T::jnt_bodyid[I]].qpos0);
				lp[64 * I] = Pair{ bada ? q0 : oa, bada ? 0.0 : ob };
			});
			sfor<NU>([&](auto I) { cn[I] = bada ? 0.0 : cn[I]; });
			time = bada ? 0.0 : time;
			wasreset = wasreset || bada;
			if constexpr (DUO && DP) {
				lp[64 * MAIL] = Pair{ (double)(8 | (bada ? 4 : 0)), 0.0 };
				le_barrier();  // (B, retry) V runs its half again on the reset state
			}
		}
		if constexpr (ROLE == 5) {
			if (last && (m.enableflags & MJB_ENBL_ENERGY)) s.energy[2 * ev] = en_pe;  // (C stores the kinetic half)
		}
		if constexpr (DUO && !DP) {
			time += dt;
			continue;
		}

		if (last) {  // mj_advance's qacc_warmstart = qacc; mjData.energy of the launch's last step
			sfor<NV>([&](auto I) {
				s.qacc[ev * NV + I] = qacc[I];
				s.qacc_warmstart[ev * NV + I] = qacc[I];
			});
			if (m.enableflags & MJB_ENBL_ENERGY) {
				if constexpr (EPOS) s.energy[2 * ev] = en_pe;
				s.energy[2 * ev + 1] = en_ke;
			}
		}
		// ================= A16 mj_Euler =================
		badp_next = badv_next = false;
		sfor<NV>([&](auto I) {
			Pair s2 = lp[64 * I];
			s2.b += dt * qaccd[I];
			s2.a += dt * s2.b;
			lp[64 * I] = s2;
			badp_next |= bad_val(s2.a);  // the next step's mj_checkPos / mj_checkVel
			badv_next |= bad_val(s2.b);
		});
		time += dt;
		if constexpr (DUO && DP) {
			lp[64 * MAIL] = Pair{ (double)((badp_next ? 2 : 0) | (badv_next ? 1 : 0)), 0.0 };
			LE_PK(6);
			le_barrier();  // (B) the new state and its verdicts
			LE_PK(7);
		}
	}
#ifdef MJB_LE_PROBE
	if (env_raw == env_lo) for (int k = 0; k < 8; k++) P->s.sensordata[(ROLE == 4 || ROLE == 2 || ROLE == 6 ? 8 : (ROLE == 7 ? 16 : 0)) + k] = (double)pk_acc[k] / nsteps;
#endif

	// ---- the launch's state back to HBM (store_state of the generic kernels; sensordata went out from the last step)
	{  // (tail lanes store the last env's state a second time)
		const DevState MJB_AS4 &s = P->s;
		if constexpr (DP) {
			sfor<NV>([&](auto I) {
				const Pair s2 = lp[64 * I];
				s.qpos[ev * NV + I] = s2.a;
				s.qvel[ev * NV + I] = s2.b;
			});
			s.time[ev] = time;
		}
		if constexpr (DV) {
			sfor<NU>([&](auto I) { s.ctrlnoise[ev * NU + I] = cn[I]; });
			if (nz_on) sfor<NU>([&](auto I) { s.ctrl[ev * NU + I] = cn[I]; });
			else if (__builtin_amdgcn_ballot_w64(wasreset)) sfor<NU>([&](auto I) { const double c = pinv(s.ctrl[ev * NU + I]); s.ctrl[ev * NU + I] = wasreset ? 0.0 : c; });
		}
	}
}

// LDS of a DUO block: the solo layout under a budget reduced by the exchange slots, then those
template <class T, int LP> constexpr int duo_bytes() { return (Lds<T, LP - DuoSlots<T::NV>::n>::nslots() + DuoSlots<T::NV>::n) * 64 * 16; }

// ... and of a pipelined DUO block (roles 3 / 4): every needed body's cinert, every dof's cdof, the pose ring, the exchange slots
template <class T> constexpr int trio_bytes() { return (Lds<T, (1 << 20)>::nslots() + 18 + DuoSlots<T::NV>::n + T::NBODY) * 64 * 16; }
template <class T> constexpr int duo2_bytes() { return (Lds<T, (1 << 20)>::nslots() + 12 + DuoSlots<T::NV>::n) * 64 * 16; }
template <class T, int LP>
DEVI void lane_env_duo2(const KernelParams MJB_AS4 *__restrict__ P, const int nsteps, const unsigned int step0, const int env_lo, const int env_hi,
                        unsigned char *const smem_le)
{
	if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64) lane_env_body<T, LP, 3>(P, nsteps, step0, env_lo, env_hi, smem_le);
	else lane_env_body<T, LP, 4>(P, nsteps, step0, env_lo, env_hi, smem_le);
}

// the TRIO: wavefront 0 = the pose chain, 1 = inertias / factors / solves / Euler, 2 = velocities and forces (blockDim.x = 192; the pipelined duo's LDS layout)
template <class T, int LP>
DEVI void lane_env_trio(const KernelParams MJB_AS4 *__restrict__ P, const int nsteps, const unsigned int step0, const int env_lo, const int env_hi,
                        unsigned char *const smem_le)
{
	const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
	if (w == 0) lane_env_body<T, LP, 5>(P, nsteps, step0, env_lo, env_hi, smem_le);
	else if (w == 1) lane_env_body<T, LP, 6>(P, nsteps, step0, env_lo, env_hi, smem_le);
	else lane_env_body<T, LP, 7>(P, nsteps, step0, env_lo, env_hi, smem_le);
}

// the DUO kernel's body: wavefront 0 of the block takes the position half, wavefront 1 the velocity half (blockDim.x = 128)
template <class T, int LP>
DEVI void lane_env_duo(const KernelParams MJB_AS4 *__restrict__ P, const int nsteps, const unsigned int step0, const int env_lo, const int env_hi,
                       unsigned char *const smem_le)
{
	if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64) lane_env_body<T, LP, 1>(P, nsteps, step0, env_lo, env_hi, smem_le);
	else lane_env_body<T, LP, 2>(P, nsteps, step0, env_lo, env_hi, smem_le);
}

}  // namespace mjb_le
