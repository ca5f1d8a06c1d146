// mjb_api.hip — host side of the C-ABI declared in include/mjb.h (libmjb.so).
// Owns model validation, the device model blob, HBM state arrays, launches and host<->device copies.
// No CPU compute path exists here: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mjb_dev.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

#define HIP_TRY(expr)                                                                                   \
	do {                                                                                                \
		hipError_t _e = (expr);                                                                         \
		if (_e != hipSuccess) return fail(MJB_ENODEVICE, "%s: %s", #expr, hipGetErrorString(_e));     \
	} while (0)

struct FieldInfo {
	const char *name;
	const char *rows;
	const char *cols;  // literal or size name (DD2)
	int kind;          // 0 state, 1 derived double, 2 derived double (2 size names), 3 int
};

const FieldInfo kFields[] = {
#define MJB_DS(name, rows, cols) { #name, #rows, #cols, 0 },
#define MJB_DD(name, rows, cols) { #name, #rows, #cols, 1 },
#define MJB_DD2(name, rows, cols) { #name, #rows, #cols, 2 },
#define MJB_DI(name, rows, cols) { #name, #rows, #cols, 3 },
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
};

int size_by_name(const mjb_model_desc &d, const char *n)
{
#define MJB_SIZE(name) if (!strcmp(n, #name)) return d.name;
#define MJB_OPT_I(name)
#define MJB_OPT_D(name, k)
#define MJB_ARR_I(name, rows, cols)
#define MJB_ARR_D(name, rows, cols)
#include "../../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	if (!strcmp(n, "one")) return 1;
	return atoi(n);
}

}  // namespace

struct mjb_model {
	mjb_model_desc h{};               // host copy (pointers into hint / hdbl)
	std::vector<int> hint;            // all int arrays, concatenated
	std::vector<double> hdbl;         // all double arrays, concatenated
	std::vector<size_t> ioff, doff;   // offsets of each array inside hint / hdbl (declaration order)
	std::vector<int> M_rowdof, M_coldof, dof_depth, dof_jstart, body_rec, body_rec2, dof_rec, fac_ops, fac_beg, body_dofmask, body_submask, M_dense, body_anc, dof_bodymask, M_sym, body_dofanc, dof_rec2, jnt_rec, flv_hdr, flv_rec, flv_ent;
	std::vector<int> sens_copy, sens_slow, dof_act_adr, dof_act_id;
	std::vector<int> pair_i;       // [ncollpair][8]  per candidate pair: g1, g2, type1, type2, condim, friction rule, collision-function override, 0
	std::vector<double> pair_d;    // [ncollpair][24] size1[3] size2[3] margin gap rbound1 rbound2 solref[2] solimp[5] includemargin friction[3] tran pad[2]
	std::vector<double> sub_S;     // 0/1 subtree matrix as MFMA A operands (mjb_dev.h)
	std::vector<double> lim_d;     // [njnt + ntendon][24] limit items in pair_d's slots (mjb_dev.h)
	std::vector<double> dof_act_mom;  // moment arm of entry t of the per-dof actuator lists (dof_act_adr / dof_act_id)
	int act_tendon = 0;               // some actuator drives a tendon
	std::vector<double> damp_int;  // [nv] -diag(D) of the integrator's implicit matrix M + h diag(.): dof_damping (Euler) / implicitfast's constant velocity derivative
	std::vector<int> lim_i;        // [njnt + ntendon][4]
	int sens_ncopy[3] = { 0, 0, 0 }, sens_nslow[3] = { 0, 0, 0 }, sens_ncopy_max = 0;
	int eulerdamp = 0, maxdepth = 0, kin_rounds = 0, need_rnepost = 0, nfriction = 0, sub_nt = 0, dofanc_max = 0, flv_n = 0;
	int field_size[MJB_F_COUNT]{};
	FrameLayout L{}, Lc{};
	// kernel variant 4 (Newton, capacity > 128 rows): a second, WIDE fused frame that holds 128 rows of efc_J and of every per-row
	// array (two envs per CU instead of four).  A batch whose env-steps mostly exceed 64 rows runs its long launches on it: those steps
	// take the two-rows-per-lane solver on LDS instead of the four-rows-per-lane one on the env's block in HBM (launch(), wide policy).
	FrameLayout Lw{};
	bool has_wide = false;
	int le_topo = -1;  // compiled-in topology of the lane = env kernel the model matches (mjb_lane_env.hip); -1: none; -2: eligible, built by hiprtc on first use
	std::vector<double> le_tape;  // its constant tape (mjb_dev.h), empty without a topology
	int sm_topo = -1;  // compiled-in topology of the split step's smooth kernel (mjb_smooth.hip) when the model's constraint half fits mjb_cstep_kernel too; -1: none
};

struct mjb_batch {
	const mjb_model *model = nullptr;
	int device = 0, nenv = 0;
	void *blob = nullptr;  // device model blob
	DevModel dm{};
	DevState st{};
	FrameLayout L{};
	NoiseCfg nz{};
	hipStream_t stream = nullptr;
	bool own_stream = false;
	int lanes = 0, epb = 0;
	unsigned int step_counter = 0;
	bool frame_valid = false;
	int frame_hi = 0;             // the frame workspace is current for envs [0, frame_hi) (a prefix after mjb_step1_prefix)
	int split_ncb = -1;           // inside a split step of this prefix (mjb_step1_prefix .. mjb_step2_prefix)
	bool split_rest_done = false;
	hipStream_t rest_stream = nullptr;  // the fused launch of the non-callback envs runs beside the callback envs' second half
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	bool rest_pending = false;
	// wide-frame policy of kernel variant 4 (launch()): counters of the last long fused launch -- env-steps, those beyond 64 rows, those
	// beyond 128 -- copied to pinned host memory behind it; the next long launch waits for that copy (the launch has ended by then or is
	// about to) and picks its frame from them: deterministic for a given sequence of launches
	bool wide = false;
	unsigned long long *rowstat_dev = nullptr, *rowstat_host = nullptr;  // (64-bit: envs x steps of one launch may exceed 2^32)
	hipEvent_t ev_rowstat = nullptr;
	bool rowstat_pending = false;
	bool rowstat_ever = false;     // a sample of the CURRENT workload has been looked at (cleared by mjb_reset / a new qpos: the next long launch probes first)
	bool probe_now = false, decide_now = false;  // mjb_step's probe launch / the launch behind it (launch())
	unsigned char *mask_dev = nullptr;
	KernelParams *params_dev = nullptr;  // device copy of {dm, L, st, nz}
	double *metrics_dev = nullptr;       // [16] mjb_metrics
	unsigned long long *stats_dev = nullptr;  // [MJB_NSTATS] mjb_set_stats (st.stats points here while counting)
	int *pair_i_dev = nullptr;           // device address of the per-pair int records inside the blob (mjb_register_collision patches them)
	unsigned long long steps_taken = 0;  // steps since the batch was made (step_counter is the 32-bit Philox counter)
	bool params_dirty = true;
	// sensors-plugin equivalent (mjb_sensor_*): noise models (host mirror + device copy) and the packed messages
	std::vector<int> sens_flag;
	std::vector<double> sens_mean, sens_sigma;
	int *sens_flag_dev = nullptr;
	double *sens_mean_dev = nullptr, *sens_sigma_dev = nullptr;
	float *sens_value = nullptr, *sens_truth = nullptr;
	bool sens_dirty = true, sens_packed = false;
	double *pack_dev = nullptr;          // staging block of mjb_get_packed / mjb_set_packed
	size_t pack_cap = 0;
	double *env_geom_size = nullptr;
	int *env_geom_type = nullptr;
	double *env_gravity = nullptr, *env_geom_friction = nullptr, *env_equality = nullptr, *env_mass = nullptr;  // per-env model parameter overrides (mjb_set_env_*)
	// device-side DefaultRobotHWSim (mjb_hwsim_*)
	HwSim hw{};
	int *hw_ints = nullptr;        // joint | method | kind | antiwindup, [4][n]
	double *hw_gains = nullptr;    // [n][8]
	double *hw_cmd = nullptr;      // pos | vel | eff | hold, [4][nenv][n]
	double *hw_pid = nullptr;      // [nenv][n][2]
	double *hw_cad = nullptr;      // [nenv][2 + 2 n] controller cadence (mjb_hwsim_set_period)
	double *zbuf = nullptr;        // pre-generated ctrl-noise normals of one fused launch (launch())
	size_t zcap = 0;               // its capacity in doubles PER HALF
	bool zdouble = false;          // two halves allocated (side-stream speculation possible)
	size_t zfail = (size_t)-1;     // smallest total allocation (doubles) that failed: not retried
	int lane_env_mode = -1;        // mjb_set_lane_env: -1 automatic, 0 never, 1 whenever eligible
	// the split step (mjb_set_split_step): smooth half in lane = env form + constraint half per wavefront, alternating on a few streams of env slices
	int split_mode = -1;           // -1 automatic (whole-batch fused launches of >= MJB_SPLIT_MIN_ENVS envs), 0 never, 1 whenever eligible
	bool split_used = false;       // the last fused launch ran that way
	int split_slices_last = 0;
	std::vector<hipStream_t> split_streams;
	std::vector<hipEvent_t> split_events;
	hipEvent_t ev_split_fork = nullptr;
	double *handoff_dev = nullptr; // DevState::handoff
	double *reset_step_dev = nullptr;  // DevState::reset_step
	bool lane_env_used = false;    // the last fused launch ran the lane = env kernel
	bool le_unavailable = false;   // the model's topology had to be built by hiprtc and that failed (mjb_lane_env_jit_error): generic kernels from then on
	int noise_mode = 0;            // how the last fused launch got its ctrl-noise normals: 0 in-kernel, 1 same-stream, 2 side-stream (mjb_noise_mode)
	unsigned int *zinfo = nullptr; // two records (one per half of zbuf)
	bool zvalid = false;           // a record names a launch: cleared (on the stream) before any step launch that does not use the buffer
	hipStream_t noise_stream = nullptr;  // the NEXT launch's normals are generated here while the current launch runs
	hipEvent_t ev_noise = nullptr, ev_noise_go = nullptr;
	bool spec_valid = false;       // half `spec_half` holds (or will, once ev_noise fires) the normals of launch (spec_step0, spec_nsteps)
	int spec_half = 0, spec_nsteps = 0;
	unsigned int spec_step0 = 0;
};

namespace {

// Frame layout.  compact == false: every field owns its storage (what mjb_forward / mjb_step1 dump and mjb_get
// reads).  compact == true (fused mjb_step only): xfrc_applied is absent and fields whose lifetimes never
// overlap inside one step share storage --
//   region A: kinloc (kinematics)  |  crb + crbbuf (crb)  |  cacc + cfrc_body (rne)  |  eulerx (euler)
//   region B: ximat (kinematics -> comPos)  |  cvel + cdof_dot (comVel -> rne / vel sensors)
// which brings the Franka frame from 12.6 KB to 9.3 KB, i.e. from 12 to 16 resident envs per CU.
//
// compact AND constrained (nefcmax > 0) -- "lean": the fused step builds the constraint rows AFTER the velocity stage (nothing
// between collision and the solver reads them; the kernel defers make_constraint whenever it runs on this layout), so
//   * efc_J overlays everything that is dead by then (region U): regions A and B, cinert, geom_xpos / geom_xmat, xanchor / xaxis,
//     site_xquat and -- without equalities -- xmat / xquat   [only when no sensor needs rne_post, which re-reads them];
//   * row bookkeeping shrinks to what the solver reads: efc_pos / efc_margin / efc_KBIP / efc_vel are gone (efc_aref holds
//     K imp (pos - margin) and efc_b the damping gain until reference_constraint folds them into aref), efc_D is gone for PGS
//     (1 / R on the fly) and efc_R for the primal solvers;
//   * contact_solref / contact_solimp are gone (make_constraint reads the pair record the contact came from);
//   * the PGS triangle scratch sits on the dead contact arrays, the box - box clipping scratch at the start of U.
// jrows_req > 0: rows of efc_J the fused frame of kernel variant 4 holds (choose_layout picks it); 0 = the default
void compute_layout(mjb_model *M, FrameLayout &L, bool compact, int jrows_req = 0)
{
	const mjb_model_desc &d = M->h;
	int off = 0, ioff = 0, nstate = 0;
	int idx = 0;
	auto dim = [&](const FieldInfo &fi) {
		int r = size_by_name(d, fi.rows);
		int c = size_by_name(d, fi.cols);
		return r * c;
	};
	int *slots[] = {
#define MJB_DS(name, rows, cols) &L.name,
#define MJB_DD(name, rows, cols) &L.name,
#define MJB_DD2(name, rows, cols) &L.name,
#define MJB_DI(name, rows, cols) &L.name,
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	};
	bool body_sensor = false;
	for (int i = 0; i < d.nsensor; i++)
		if (d.sensor_objtype[i] == MJB_OBJ_BODY || d.sensor_reftype[i] == MJB_OBJ_BODY) body_sensor = true;
	const bool alias_b = compact && !body_sensor;
	bool need_post = false;
	for (int i = 0; i < d.nsensor; i++) {
		const int t = d.sensor_type[i];
		if (t == MJB_SENS_TOUCH || t == MJB_SENS_ACCELEROMETER || t == MJB_SENS_FORCE || t == MJB_SENS_TORQUE ||
		    t == MJB_SENS_FRAMELINACC || t == MJB_SENS_FRAMEANGACC)
			need_post = true;
	}
	M->need_rnepost = need_post ? 1 : 0;
	const bool lean = compact && d.nefcmax > 0;
	const bool u_ok = lean && !need_post;
	const bool primal = d.solver == MJB_SOL_NEWTON || d.solver == MJB_SOL_CG;
	const bool ell_con = d.cone == MJB_CONE_ELLIPTIC && d.nconmax > 0;
	// kernel variant 4 (Newton, more than 128 rows of capacity), fused frame: the first 64 rows of efc_J only; an env-step with
	// more rows keeps J in the env's block of DevState::efc_Jg (config 5 never has: what lets two of its envs share a CU's LDS)
	int jrows = d.nefcmax;
	if (compact && d.solver == MJB_SOL_NEWTON && d.nefcmax > 128) {
		jrows = jrows_req > 0 ? std::max(4, std::min(128, jrows_req)) : 64;
		if (const char *v = getenv("MJB_DEBUG_JROWS")) jrows = std::max(4, std::min(64, atoi(v)));  // test knob: force the HBM path
	}
	L.jrows = jrows;
	std::vector<int> u_members;  // field ids overlaid by efc_J, in placement order
	// "X layout" -- lean frames of the Newton kernels (nv <= 32: M lives in registers inside the solver).  One region X is shared by
	// three generations of data:
	//   X = [ D | R ],  D = cdof | qM | qLD | qLDiagInv | contact dist / pos / frame / includemargin   (dead once the solver has
	//                       gathered its rows of M: nothing after make_constraint / fwd_acceleration reads them)
	//                   R = position / velocity-stage scratch (regions A, B, the U members)           (dead at make_constraint)
	//   make_constraint .. solver:   efc_J | efc_D | efc_aref | efc_b | efc_force at the END of X
	//   fwd_acceleration, Euler:     the dense-triangle scratch (tri | solvescr | eulerx) right below efc_J, inside R
	//   solver only:                 nwt_H | nwt_vec | nwt_row | nwt_hc from the START of X (over D and the head of R)
	// (a jointlimitfrc / tendonlimitfrc sensor reads efc_force of its row after the solve: such a model keeps every row array whole in the frame)
	bool row_sensor = false;
	for (int i = 0; i < d.nsensor; i++) row_sensor = row_sensor || d.sensor_type[i] == MJB_SENS_JOINTLIMITFRC || d.sensor_type[i] == MJB_SENS_TENDONLIMITFRC;
	const bool xl = u_ok && !row_sensor && d.solver == MJB_SOL_NEWTON && d.nv <= 32;
	// ... and, when the frame holds only `jrows` rows of efc_J (kernel variant 4), every other per-row array is capped at the same
	// count: an env-step with more rows keeps all of its row data in the env's block of DevState::efc_Jg (mjb_dev.h, RowBlock).
	// Config 5: 53.0 -> 40.9 KB = four envs per CU, one per SIMD.
	const int rcap = (xl && jrows < d.nefcmax) ? jrows : d.nefcmax;
	L.rcap = rcap;
	L.hcrow = (rcap < d.nefcmax) ? 1 : 0;
	std::vector<int> d_members, p2_members;
	int fsize[MJB_F_COUNT];
	for (const FieldInfo &fi : kFields) {
		int n = dim(fi);
		if (idx == MJB_F_efc_AR) n = 0;  // (the PGS kernel keeps each row of AR in its lane's registers)
		if (idx == MJB_F_efc_frictionloss && M->nfriction == 0) n = 0;  // no dry-friction rows in this model
		if (idx == MJB_F_efc_B && d.solver != MJB_SOL_PGS) n = 0;  // the primal solvers need no J M^-1
		// (nv <= 16: the PGS kernel solves for its row of B in registers; the field only exists in the full layout, for mjb_get)
		const bool no_B = idx == MJB_F_efc_B && n > 0 && compact && d.nv <= 16 && !(d.cone == MJB_CONE_ELLIPTIC && d.nconmax > 0);
		if (!compact) M->field_size[idx] = n;
		if ((idx == MJB_F_cfrc_int || idx == MJB_F_cfrc_ext) && !need_post) n = 0;  // computed only when a sensor needs them
		// (with rne_post the true cacc is written between fwd_acceleration and Euler, whose rhs shares region A)
		const bool in_a = compact && (idx == MJB_F_crb || (idx == MJB_F_cacc && !need_post) || idx == MJB_F_cfrc_body);
		const bool in_b = alias_b && (idx == MJB_F_ximat || idx == MJB_F_cvel || idx == MJB_F_cdof_dot);
		if (compact && idx == MJB_F_xfrc_applied) n = 0;
		if (idx == MJB_F_efc_J && n > 0) n = jrows * d.nv;
		if (rcap < d.nefcmax && n > 0 && (idx == MJB_F_efc_D || idx == MJB_F_efc_aref || idx == MJB_F_efc_b || idx == MJB_F_efc_force ||
		                                  idx == MJB_F_efc_frictionloss || idx == MJB_F_efc_type || idx == MJB_F_efc_id))
			n = n / d.nefcmax * rcap;
		fsize[idx] = n;
		const bool gone = lean && (idx == MJB_F_efc_pos || idx == MJB_F_efc_margin || idx == MJB_F_efc_KBIP || idx == MJB_F_efc_vel ||
		                           (idx == MJB_F_efc_D && d.solver == MJB_SOL_PGS && !ell_con) || (idx == MJB_F_efc_R && primal) ||
		                           idx == MJB_F_contact_solref || idx == MJB_F_contact_solimp);
		const bool in_u = u_ok && (idx == MJB_F_efc_J || idx == MJB_F_cinert || idx == MJB_F_geom_xpos || idx == MJB_F_geom_xmat ||
		                           idx == MJB_F_xanchor || idx == MJB_F_xaxis || idx == MJB_F_site_xquat ||
		                           (d.neq == 0 && (idx == MJB_F_xmat || idx == MJB_F_xquat)));
		if (gone) {
			*slots[idx] = -1;  // kernels test the offset
			idx++;
			continue;
		}
		if (xl && (idx == MJB_F_cdof || idx == MJB_F_qM || idx == MJB_F_qLD || idx == MJB_F_qLDiagInv || idx == MJB_F_contact_dist ||
		           idx == MJB_F_contact_pos || idx == MJB_F_contact_frame || idx == MJB_F_contact_includemargin)) {
			d_members.push_back(idx);
			*slots[idx] = -1;  // placed below
			idx++;
			continue;
		}
		if (xl && (idx == MJB_F_efc_D || idx == MJB_F_efc_aref || idx == MJB_F_efc_b || idx == MJB_F_efc_force || idx == MJB_F_efc_frictionloss)) {
			p2_members.push_back(idx);
			*slots[idx] = -1;  // placed below
			idx++;
			continue;
		}
		if (in_u && !in_a && !in_b) {
			if (idx != MJB_F_efc_J) u_members.push_back(idx);
			*slots[idx] = -1;  // placed below
			idx++;
			continue;
		}
		if (no_B) {
			if (!compact) M->field_size[idx] = n;
			*slots[idx] = -1;  // kernels test L.efc_B >= 0
			idx++;
			continue;
		}
		if (fi.kind == 3) {
			*slots[idx] = ioff;
			ioff += n;
		} else if (in_a || in_b) {
			*slots[idx] = -1;  // placed below
		} else {
			*slots[idx] = off;
			off += n;
			if (fi.kind == 0) nstate = off;
		}
		idx++;
	}
	const bool newton = d.nefcmax > 0 && (d.solver == MJB_SOL_NEWTON || d.solver == MJB_SOL_CG);  // primal solvers
	const int off_before_nwt = off;
	L.nwt_M = off;
	off += (newton && d.nv > 32) ? d.nv * d.nv : 0;  // (nv <= 32: the solver keeps row `lane` of M in registers, host table M_sym)
	L.nwt_H = off;
	// (nv <= 32: the packed lower triangle + one dump slot, mjb_constraint.h MJB_HPACK; 16 < nv: at least chol_schur16's 16 x 17 scratch)
	off += newton ? (d.nv <= 32 ? std::max(d.nv * (d.nv + 1) / 2 + 1, d.nv > 16 ? 272 : 0) : d.nv * d.nv) : 0;
	L.nwt_vec = off;
	off += newton ? 5 * d.nv : 0;  // qacc | M qacc | grad | search | (CG: M^-1 grad)
	L.nwt_row = off;
	off += newton ? 3 * rcap : 0;
	L.nwt_hc = off;
	// cone blocks: the primal solvers size them by the model's largest contact dimension (config 5: condim 4 -> 16 doubles instead
	// of 36; at least 10, the line-search constants a contact parks there); PGS keeps its 6 x 6 blocks of AR
	L.hcd = 6;
	L.hcs = 36;
	if (newton) {
		int maxdim = 1;
		for (int p = 0; p < d.ncollpair; p++) maxdim = std::max(maxdim, M->pair_i[(size_t)8 * p + 4]);
		L.hcd = std::min(6, maxdim);
		// (blocks laid out by row on a frame with more than one row per lane: the line search parks its ten constants per contact in the contact's block, and a
		//  contact of dimension 3 owns 3 x hcd doubles -- nine at hcd = 3.  Round 5, tools/wide_dim3_check.py: the neighbour's first constant was overwritten)
		if (L.hcrow && rcap > 64 && L.hcd < 4) L.hcd = 4;
		L.hcs = std::max(L.hcd * L.hcd, 10);
	}
	// (hcrow: a contact's block sits at hcd * its first row -- rows, not contacts, bound the solver that runs on this frame)
	off += ((newton || (d.nefcmax > 0 && d.solver == MJB_SOL_PGS)) && d.cone == MJB_CONE_ELLIPTIC) ? (L.hcrow ? L.hcd * rcap : L.hcs * d.nconmax) : 0;  // (PGS: the contacts' blocks of AR)
	const int n_nwt = off - off_before_nwt;  // nwt_M | nwt_H | nwt_vec | nwt_row | nwt_hc
	if (xl) off = off_before_nwt;            // (X layout: they overlay the head of X, placed below)
	L.gravity = off;
	off += 3;
	// (lean frame: collision takes the mixed friction from the pair record, or from the per-env override in HBM)
	L.gfriction = lean ? -1 : off;
	off += (d.nconmax > 0 && !lean) ? 3 * d.ngeom : 0;
	L.eqparam = off;
	off += 19 * d.neq;
	L.cwrench = off;
	off += need_post ? 6 * d.nconmax : 0;
	L.MhB = off;
	off += d.nM;
	L.qH = lean ? L.MhB : off;  // (lean frame: factorised in place -- nothing reads M + h B after its factor exists)
	off += lean ? 0 : d.nM;
	L.qHdi = off;
	off += d.nv;
	L.rk = d.integrator == MJB_INT_RK4 ? off : -1;  // (persistent across the evaluations of one step)
	off += d.integrator == MJB_INT_RK4 ? d.nq + 4 * d.nv + d.nsensordata + 1 + 2 * d.na : 0;  // X0 | sums | warmstart | the step's sensordata | t0 | act0 | sum B act_dot
	// transient scratch of the constrained kernels: ntri doubles for the packed dense triangle of the L'DL factor (nv <= 16, PGS:
	// the J M^-1 rows; 16 < nv <= 32: the M^-1 solves of fwd_acceleration / Euler, solve_tri32), 128 for the box - box narrow phase
	const int ntri = d.nefcmax <= 0 ? 0 : ((d.nv <= 16 && d.solver == MJB_SOL_PGS) ? 128 : ((d.nv > 16 && d.nv <= 32) ? 496 : 0));
	const int nbb = d.nconmax > 0 ? 216 : 0;  // (MJB_BBSCR of mjb_constraint.h)
	const int n_kin = 7 * d.nbody, n_crb = 10 * d.nbody, n_buf = 6 * d.nv < 32 ? 32 : 6 * d.nv, n_c6 = 6 * d.nbody;  // (crbbuf doubles as the 32-double pivot-row scratch of the dense factor)
	if (compact && xl) {
		const int x0 = off;
		for (int id : d_members) {
			*slots[id] = off;
			off += fsize[id];
		}
		const int a0 = off;  // start of R
		L.kinloc = a0;
		L.crb = a0;
		L.crbbuf = a0 + n_crb;
		L.cacc = a0;
		L.cfrc_body = a0 + n_c6;
		L.bbscr = a0;
		{
			int sz = n_kin;
			if (nbb > sz) sz = nbb;
			if (n_crb + n_buf > sz) sz = n_crb + n_buf;
			if (2 * n_c6 > sz) sz = 2 * n_c6;
			if (d.nv > sz) sz = d.nv;
			off += sz;
		}
		if (alias_b) {
			const int b0 = off;
			L.ximat = b0;
			L.cvel = b0;
			L.cdof_dot = b0 + n_c6;
			int szb = 9 * d.nbody;
			if (n_c6 + 6 * d.nv > szb) szb = n_c6 + 6 * d.nv;
			off += szb;
		}
		for (int id : u_members) {
			*slots[id] = off;
			off += fsize[id];
		}
		int n_p2 = fsize[MJB_F_efc_J];
		for (int id : p2_members) n_p2 += fsize[id];
		const int n_scr = ntri + 32 + d.nv;
		// (the triangle scratch of fwd_acceleration / Euler: on the contact arrays at the end of D when they are large enough --
		//  dist | pos | frame | includemargin are contiguous and dead after make_constraint, while qM / qLD next to them are still
		//  read -- otherwise at the start of R)
		const bool tri_in_d = 14 * d.nconmax >= n_scr;
		int jstart = std::max(x0 + n_nwt, tri_in_d ? a0 : a0 + n_scr);  // the solver's arrays and the triangle scratch both end below efc_J, which never reaches into D (make_constraint reads cdof and the contacts while it writes J)
		if (jstart + n_p2 > off) off = jstart + n_p2;
		jstart = off - n_p2;                             // ... which sits at the end of X
		L.efc_J = jstart;
		{
			int o = jstart + fsize[MJB_F_efc_J];
			for (int id : p2_members) {
				*slots[id] = o;
				o += fsize[id];
			}
		}
		L.tri = tri_in_d ? L.contact_dist : jstart - n_scr;
		L.solvescr = L.tri + ntri;
		L.eulerx = L.solvescr + 32;
		{
			const int shift = x0 - off_before_nwt;
			L.nwt_M += shift; L.nwt_H += shift; L.nwt_vec += shift; L.nwt_row += shift; L.nwt_hc += shift;
		}
	} else if (compact) {
		const int a0 = off;
		L.kinloc = a0;
		L.crb = a0;
		L.crbbuf = a0 + n_crb;
		if (!need_post) L.cacc = a0;
		L.cfrc_body = a0 + n_c6;
		L.eulerx = a0;
		L.tri = a0 + 32;  // (alive inside the PGS stage / the M^-1 solves: cacc / cfrc_body are dead by then; Euler's vector sits below it)
		L.bbscr = a0;
		L.solvescr = L.crbbuf;
		int sz = n_kin;
		if (ntri > 0 && sz < 32 + ntri) sz = 32 + ntri;
		if (nbb > sz) sz = nbb;
		if (n_crb + n_buf > sz) sz = n_crb + n_buf;
		if (2 * n_c6 > sz) sz = 2 * n_c6;
		if (d.nv > sz) sz = d.nv;
		off += sz;
		if (alias_b) {
			const int b0 = off;
			L.ximat = b0;
			L.cvel = b0;
			L.cdof_dot = b0 + n_c6;
			int szb = 9 * d.nbody;
			if (n_c6 + 6 * d.nv > szb) szb = n_c6 + 6 * d.nv;
			off += szb;
		}
		if (u_ok) {
			for (int id : u_members) {
				*slots[id] = off;
				off += fsize[id];
			}
			L.efc_J = a0;
			if (off - a0 < fsize[MJB_F_efc_J]) off = a0 + fsize[MJB_F_efc_J];
			// (efc_J is alive from the end of the velocity stage to the end of the solver: what runs in between -- the dense
			//  M^-1 solve of fwd_acceleration, the PGS stage's triangle -- takes its scratch, and Euler its right-hand side, from
			//  the contact arrays nobody reads after make_constraint: dist | pos | frame | includemargin are contiguous, 14
			//  doubles per contact; friction, which the elliptic solvers read, stays intact)
			const int late = ntri + 32 + d.nv;
			int ls = off;
			if (14 * d.nconmax >= late) ls = L.contact_dist;
			else off += late;
			L.tri = ls;
			L.solvescr = ls + ntri;
			L.eulerx = L.solvescr + 32;
		}
	} else {
		L.kinloc = off;
		off += n_kin;
		L.crbbuf = off;
		off += n_buf;
		L.eulerx = off;
		off += d.nv;
		L.solvescr = L.crbbuf;
		L.tri = off;
		off += ntri;
		L.bbscr = off;
		off += nbb;
	}
	if (off & 1) off++;
	L.ndouble = off;
	L.iscratch = ioff;
	{
		int a = d.ncollpair, b = d.neq + d.nv + d.njnt + 2 * d.ntendon + d.nconmax;
		if (newton && rcap > a) a = rcap;  // (the primal solvers park one int of row metadata per constraint row here)
		ioff += a > b ? a : b;
	}
	L.dadr = ioff;
	ioff += (d.nefcmax > 0 && d.nv <= 16) ? 64 : 0;
	L.nint = (ioff + 1) & ~1;
	L.nstate = nstate;
}

int layout_bytes(const FrameLayout &L) { return ((L.ndouble * 8 + L.nint * 4) + 15) & ~15; }

// development knob MJB_DEBUG_LAYOUT: every field's offset in the fused frame, sorted (doubles, then ints)
void dump_layout(const mjb_model *M, const FrameLayout &L)
{
	const int *slots[] = {
#define MJB_DS(name, rows, cols) &L.name,
#define MJB_DD(name, rows, cols) &L.name,
#define MJB_DD2(name, rows, cols) &L.name,
#define MJB_DI(name, rows, cols) &L.name,
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	};
	std::vector<std::pair<int, std::string>> dd, ii;
	for (int i = 0; i < MJB_F_COUNT; i++)
		if (*slots[i] >= 0) (kFields[i].kind == 3 ? ii : dd).push_back({ *slots[i], kFields[i].name });
	const std::pair<int, const char *> extra[] = { { L.MhB, "MhB" }, { L.qH, "qH" }, { L.qHdi, "qHdi" }, { L.nwt_M, "nwt_M" }, { L.nwt_H, "nwt_H" },
		{ L.nwt_vec, "nwt_vec" }, { L.nwt_row, "nwt_row" }, { L.nwt_hc, "nwt_hc" }, { L.gravity, "gravity" }, { L.gfriction, "gfriction" },
		{ L.eqparam, "eqparam" }, { L.cwrench, "cwrench" }, { L.kinloc, "kinloc" }, { L.crbbuf, "crbbuf" }, { L.eulerx, "eulerx" },
		{ L.tri, "tri" }, { L.solvescr, "solvescr" }, { L.bbscr, "bbscr" } };
	for (auto &x : extra)
		if (x.first >= 0) dd.push_back({ x.first, std::string("*") + x.second });
	ii.push_back({ L.iscratch, "*iscratch" });
	ii.push_back({ L.dadr, "*dadr" });
	std::sort(dd.begin(), dd.end());
	std::sort(ii.begin(), ii.end());
	fprintf(stderr, "mjb fused layout: %d bytes, jrows %d, hcs %d, hcd %d, ndouble %d, nint %d, nstate %d\n doubles:", layout_bytes(L), L.jrows, L.hcs,
	        L.hcd, L.ndouble, L.nint, L.nstate);
	for (auto &x : dd) fprintf(stderr, " %s@%d", x.second.c_str(), x.first);
	fprintf(stderr, "\n ints:");
	for (auto &x : ii) fprintf(stderr, " %s@%d", x.second.c_str(), x.first);
	fprintf(stderr, "\n");
	(void)M;
}

// The fused frame of kernel variant 4 (Newton, capacity > 128 rows) keeps only `jrows` rows of efc_J in LDS (the rest, and all of J
// of an env-step with more rows, live in DevState::efc_Jg).  Residency beats the price of the HBM path (measured 10 % on the steps
// that take it): pick the row count that lets the most envs share a CU's LDS, and the largest such.
void choose_fused_layout(mjb_model *M)
{
	compute_layout(M, M->Lc, true);
	const mjb_model_desc &d = M->h;
	M->Lw = M->Lc;
	M->has_wide = false;
	if (d.solver == MJB_SOL_NEWTON && d.nefcmax > 128 && getenv("MJB_DEBUG_FRAME_ROWS")) {  // measurement knob: rows of J in THE fused frame
		compute_layout(M, M->Lc, true, atoi(getenv("MJB_DEBUG_FRAME_ROWS")));
		M->Lw = M->Lc;
		if (getenv("MJB_DEBUG_LAYOUT")) dump_layout(M, M->Lc);
		return;
	}
	if (!(d.solver == MJB_SOL_NEWTON && d.nefcmax > 128) || getenv("MJB_DEBUG_JROWS")) {
		if (getenv("MJB_DEBUG_LAYOUT")) dump_layout(M, M->Lc);
		return;
	}
	auto occ = [&](int jr) {
		FrameLayout t{};
		compute_layout(M, t, true, jr);
		// (gfx950 hands out its 160 KB of LDS in 128 granules of 1280 bytes -- measured: a 54000-byte frame runs two to a CU, a
		//  50272-byte one three; 512-register kernels: one wave per SIMD, four envs per CU at most)
		return std::min(4, 128 / ((layout_bytes(t) + 1279) / 1280));
	};
	const int best = occ(16);
	int pick = 64;
	if (best > occ(64))
		for (pick = 60; pick > 16 && occ(pick) < best; pick -= 4) {}
	compute_layout(M, M->Lc, true, pick);
	if (getenv("MJB_DEBUG_LAYOUT"))  // development knob
		dump_layout(M, M->Lc);
	// the wide frame: up to 128 rows (two per lane) -- the largest multiple of four with which THREE frames share a CU's LDS when at least
	// 96 rows do (the power grasp of config 5: p99 of nefc = 111, 112 rows fit three times; an env-step beyond them takes the HBM row block),
	// else 128 when two fit
	int wrows = 128;
	for (int jr = 124; jr >= 96; jr -= 4)
		if (occ(jr) >= 3) { wrows = jr; break; }
	if (const char *v = getenv("MJB_WIDE_ROWS")) wrows = std::max(68, std::min(128, atoi(v)));  // measurement knob
	compute_layout(M, M->Lw, true, wrows);
	const int wbytes = ((M->Lw.ndouble * 8 + M->Lw.nint * 4) + 15) & ~15;
	M->has_wide = M->Lw.jrows > M->Lc.jrows && 2 * ((wbytes + 1279) / 1280) <= 128;
	if (!M->has_wide) M->Lw = M->Lc;
	compute_layout(M, M->Lc, true, pick);  // (compute_layout also sets model-wide derived fields: leave them as the default frame's)
	if (getenv("MJB_DEBUG_LAYOUT") && M->has_wide) dump_layout(M, M->Lw);
}

// Sensors whose value is a plain copy of frame doubles (joint / actuator scalars, clock, subtree com, global
// frame positions) become {dst, src} pairs resolved against both layouts; the rest stay on the general path.
void build_sensor_tables(mjb_model *M)
{
	const mjb_model_desc &h = M->h;
	std::vector<std::pair<int, int>> cp[3][3];  // layouts: 0 full, 1 fused, 2 wide fused (== 1 when the model has none)
	std::vector<int> slow[3];
	for (int i = 0; i < h.nsensor; i++) {
		const int st = h.sensor_needstage[i] - 1;
		if (st < 0 || st > 2) continue;
		const int type = h.sensor_type[i], id = h.sensor_objid[i], ot = h.sensor_objtype[i];
		bool simple = h.sensor_cutoff[i] <= 0;
		for (int lay = 0; lay < 3 && simple; lay++) {
			const FrameLayout &L = lay == 0 ? M->L : (lay == 1 ? M->Lc : M->Lw);
			int src = -1, n = 1;
			switch (type) {
			case MJB_SENS_JOINTPOS: src = L.qpos + h.jnt_qposadr[id]; break;
			case MJB_SENS_JOINTVEL: src = L.qvel + h.jnt_dofadr[id]; break;
			case MJB_SENS_ACTUATORPOS: src = L.actuator_length + id; break;
			case MJB_SENS_ACTUATORVEL: src = L.actuator_velocity + id; break;
			case MJB_SENS_ACTUATORFRC: src = L.actuator_force + id; break;
			case MJB_SENS_CLOCK: src = L.time; break;
			case MJB_SENS_SUBTREECOM: src = L.subtree_com + 3 * id; n = 3; break;
			case MJB_SENS_BALLANGVEL: src = L.qvel + h.jnt_dofadr[id]; n = 3; break;
			case MJB_SENS_FRAMEQUAT:
				if (h.sensor_refid[i] >= 0 || (ot != MJB_OBJ_XBODY && ot != MJB_OBJ_SITE)) { simple = false; break; }
				n = 4;
				src = ot == MJB_OBJ_XBODY ? L.xquat + 4 * id : L.site_xquat + 4 * id;
				break;
			case MJB_SENS_FRAMEPOS:
				if (h.sensor_refid[i] >= 0) { simple = false; break; }
				n = 3;
				if (ot == MJB_OBJ_BODY) src = L.xipos + 3 * id;
				else if (ot == MJB_OBJ_XBODY) src = L.xpos + 3 * id;
				else if (ot == MJB_OBJ_GEOM) src = L.geom_xpos + 3 * id;
				else src = L.site_xpos + 3 * id;
				break;
			default: simple = false;
			}
			if (simple)
				for (int k = 0; k < n; k++) cp[lay][st].push_back({ h.sensor_adr[i] + k, src + k });
		}
		if (!simple) slow[st].push_back(i);
	}
	int mx = 0;
	for (int st = 0; st < 3; st++) {
		M->sens_ncopy[st] = (int)cp[0][st].size();
		M->sens_nslow[st] = (int)slow[st].size();
		if (M->sens_ncopy[st] > mx) mx = M->sens_ncopy[st];
	}
	M->sens_ncopy_max = mx;
	M->sens_copy.assign((size_t)3 * 3 * (mx ? mx : 1) * 2, 0);
	for (int lay = 0; lay < 3; lay++)
		for (int st = 0; st < 3; st++)
			for (size_t k = 0; k < cp[lay][st].size() && (int)k < M->sens_ncopy[st]; k++) {
				M->sens_copy[((size_t)(lay * 3 + st) * (mx ? mx : 1) + k) * 2] = cp[lay][st][k].first;
				M->sens_copy[((size_t)(lay * 3 + st) * (mx ? mx : 1) + k) * 2 + 1] = cp[lay][st][k].second;
			}
	M->sens_slow.assign((size_t)3 * (h.nsensor ? h.nsensor : 1), 0);
	for (int st = 0; st < 3; st++)
		for (size_t k = 0; k < slow[st].size(); k++) M->sens_slow[(size_t)st * (h.nsensor ? h.nsensor : 1) + k] = slow[st][k];
}

template <typename T> T *dev_alloc(size_t n)
{
	void *p = nullptr;
	if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
	hipMemset(p, 0, (n ? n : 1) * sizeof(T));
	return static_cast<T *>(p);
}

}  // namespace

extern "C" {

const char *mjb_last_error(void) { return g_err.c_str(); }
int mjb_model_desc_size(void) { return (int)sizeof(mjb_model_desc); }
int mjb_version(void) { return MJB_VERSION; }

int mjb_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

mjb_model *mjb_compile(const mjb_model_desc *desc)
{
	if (!desc) {
		fail(MJB_EINVAL, "mjb_compile: null desc");
		return nullptr;
	}
	const mjb_model_desc &d = *desc;
	if (d.nq < 0 || d.nv < 0 || d.nbody < 1 || d.njnt < 0 || d.nu < 0 || d.nM < 0) {
		fail(MJB_EINVAL, "mjb_compile: negative size");
		return nullptr;
	}
	{  // activations: one variable per stateful actuator, in actuator order; integrator / filter dynamics
		int na = 0;
		for (int i = 0; i < d.nu; i++) {
			const int dt = d.actuator_dyntype[i];
			if (dt != MJB_DYN_NONE && dt != MJB_DYN_INTEGRATOR && dt != MJB_DYN_FILTER) {
				fail(MJB_EUNSUPPORTED, "mjb_compile: actuator dyntype (muscle / user dynamics) is not supported");
				return nullptr;
			}
			if (d.na > 0 && d.actuator_actadr[i] != (dt != MJB_DYN_NONE ? na : -1)) {
				fail(MJB_EINVAL, "mjb_compile: actuator_actadr must number the stateful actuators in order (-1: stateless)");
				return nullptr;
			}
			if (d.na > 0 && dt != MJB_DYN_NONE && d.actuator_actlimited[i] && !(d.actuator_actrange[2 * i] < d.actuator_actrange[2 * i + 1])) {
				fail(MJB_EINVAL, "mjb_compile: actlimited actuator with an empty actrange");
				return nullptr;
			}
			na += dt != MJB_DYN_NONE ? 1 : 0;
		}
		if (na != d.na) {
			fail(MJB_EINVAL, "mjb_compile: na does not match the number of stateful actuators");
			return nullptr;
		}
	}
	if (d.integrator != MJB_INT_EULER && d.integrator != MJB_INT_RK4 && d.integrator != MJB_INT_IMPLICITFAST) {
		fail(MJB_EUNSUPPORTED, "mjb_compile: integrators Euler, RK4 and implicitfast are implemented (implicit is refused)");
		return nullptr;
	}
	if (!(d.timestep[0] > 0)) {
		fail(MJB_EINVAL, "mjb_compile: timestep must be positive");
		return nullptr;
	}
	if (d.nv > 64 || d.nbody > 64) {
		fail(MJB_EUNSUPPORTED, "mjb_compile: nv = %d, nbody = %d: the tree stages use 64-bit ancestor / subtree masks "
		                       "(nv <= 64, nbody <= 64)", d.nv, d.nbody);
		return nullptr;
	}
	// every array the description declares must be there before anything below reads it
#define MJB_SIZE(name)                                                                    \
	if (d.name < 0) {                                                                     \
		fail(MJB_EINVAL, "mjb_compile: negative size %s", #name);                         \
		return nullptr;                                                                   \
	}
#define MJB_OPT_I(name)
#define MJB_OPT_D(name, n)
#define MJB_ARR_I(name, rows, cols)                                                       \
	if ((size_t)d.rows * (cols) && !d.name) {                                             \
		fail(MJB_EINVAL, "mjb_compile: array %s is NULL", #name);                         \
		return nullptr;                                                                   \
	}
#define MJB_ARR_D(name, rows, cols) MJB_ARR_I(name, rows, cols)
#include "../../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	for (int i = 0; i < d.nsensor; i++) {
		static const int ok[] = { MJB_SENS_MAGNETOMETER, MJB_SENS_RANGEFINDER, MJB_SENS_TOUCH, MJB_SENS_ACCELEROMETER, MJB_SENS_VELOCIMETER, MJB_SENS_GYRO, MJB_SENS_FORCE,
			                      MJB_SENS_TORQUE, MJB_SENS_JOINTPOS, MJB_SENS_JOINTVEL, MJB_SENS_TENDONPOS, MJB_SENS_TENDONVEL,
			                      MJB_SENS_ACTUATORPOS, MJB_SENS_ACTUATORVEL, MJB_SENS_ACTUATORFRC, MJB_SENS_BALLQUAT,
			                      MJB_SENS_BALLANGVEL, MJB_SENS_FRAMEPOS, MJB_SENS_FRAMEQUAT, MJB_SENS_FRAMEXAXIS,
			                      MJB_SENS_FRAMEYAXIS, MJB_SENS_FRAMEZAXIS, MJB_SENS_FRAMELINVEL, MJB_SENS_FRAMEANGVEL,
			                      MJB_SENS_FRAMELINACC, MJB_SENS_FRAMEANGACC, MJB_SENS_SUBTREECOM, MJB_SENS_CLOCK, MJB_SENS_JOINTLIMITPOS, MJB_SENS_JOINTLIMITVEL,
			                      MJB_SENS_JOINTLIMITFRC, MJB_SENS_TENDONLIMITPOS, MJB_SENS_TENDONLIMITVEL, MJB_SENS_TENDONLIMITFRC, MJB_SENS_SUBTREELINVEL,
			                      MJB_SENS_SUBTREEANGMOM, MJB_SENS_JOINTACTFRC };
		bool found = false;
		for (int t : ok) found = found || t == d.sensor_type[i];
		if (!found) {
			fail(MJB_EUNSUPPORTED, "mjb_compile: sensor %d has type %d, which is not implemented", i, d.sensor_type[i]);
			return nullptr;
		}
	}
	for (int i = 0; i < d.neq; i++) {
		const int t = d.eq_type[i], a = d.eq_obj1id[i], b = d.eq_obj2id[i];
		const bool body_ok = a >= 0 && a < d.nbody && b >= 0 && b < d.nbody;
		const bool jnt_ok = a >= 0 && a < d.njnt && b >= -1 && b < d.njnt;
		const bool ten_ok = a >= 0 && a < d.ntendon && b >= -1 && b < d.ntendon;
		if (!((t == MJB_EQ_CONNECT || t == MJB_EQ_WELD) ? body_ok : ((t == MJB_EQ_JOINT && jnt_ok) || (t == MJB_EQ_TENDON && ten_ok)))) {
			fail(MJB_EUNSUPPORTED, "mjb_compile: equality %d: type %d with objects (%d, %d) is not supported "
			                       "(connect / weld between bodies, joint between hinge / slide joints)", i, t, a, b);
			return nullptr;
		}
	}
	if (d.neq > 0 && d.nefcmax <= 0) {
		fail(MJB_EINVAL, "mjb_compile: equality constraints need nefcmax > 0");
		return nullptr;
	}
	if (d.nefcmax > 0 || d.nconmax > 0) {
		if (d.solver != MJB_SOL_PGS && d.solver != MJB_SOL_NEWTON && d.solver != MJB_SOL_CG) {
			fail(MJB_EUNSUPPORTED, "mjb_compile: unknown solver (PGS = 0, CG = 1, Newton = 2)");
			return nullptr;
		}
		const int rowcap = d.solver != MJB_SOL_PGS ? 256 : ((d.cone == MJB_CONE_ELLIPTIC && d.nconmax > 0) ? 64 : 128);
		if (d.nefcmax > rowcap || d.nv > 64) {
			fail(MJB_EUNSUPPORTED, "mjb_compile: one env per wavefront: nv <= 64 and nefcmax <= 128 (PGS; 64 with elliptic cones) / "
			                       "256 (Newton / CG, up to 4 rows per lane); got nefcmax = %d, nv = %d", d.nefcmax, d.nv);
			return nullptr;
		}
		if (!(d.meaninertia[0] > 0)) {
			fail(MJB_EINVAL, "mjb_compile: meaninertia must be positive");
			return nullptr;
		}
	}
	mjb_model *M = new (std::nothrow) mjb_model;
	if (!M) {
		fail(MJB_ENOMEM, "mjb_compile: out of memory");
		return nullptr;
	}
	M->h = d;
	// copy arrays (declaration order)
#define MJB_SIZE(name)
#define MJB_OPT_I(name)
#define MJB_OPT_D(name, n)
#define MJB_ARR_I(name, rows, cols)                                                       \
	{                                                                                     \
		size_t n = (size_t)d.rows * (cols);                                               \
		if (n && !d.name) {                                                               \
			fail(MJB_EINVAL, "mjb_compile: array %s is NULL", #name);                     \
			delete M;                                                                     \
			return nullptr;                                                               \
		}                                                                                 \
		M->ioff.push_back(M->hint.size());                                                \
		M->hint.insert(M->hint.end(), d.name, d.name + n);                                \
		if ((M->hint.size() & 1)) M->hint.push_back(0);                                   \
	}
#define MJB_ARR_D(name, rows, cols)                                                       \
	{                                                                                     \
		size_t n = (size_t)d.rows * (cols);                                               \
		if (n && !d.name) {                                                               \
			fail(MJB_EINVAL, "mjb_compile: array %s is NULL", #name);                     \
			delete M;                                                                     \
			return nullptr;                                                               \
		}                                                                                 \
		M->doff.push_back(M->hdbl.size());                                                \
		M->hdbl.insert(M->hdbl.end(), d.name, d.name + n);                                \
	}
#include "../../include/mjb_model_fields.def"
#undef MJB_ARR_I
#undef MJB_ARR_D
	// re-point the host copy
	{
		size_t ii = 0, di = 0;
#define MJB_ARR_I(name, rows, cols) M->h.name = M->hint.data() + M->ioff[ii++];
#define MJB_ARR_D(name, rows, cols) M->h.name = M->hdbl.data() + M->doff[di++];
#include "../../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	}
	const mjb_model_desc &h = M->h;
	// ---- every index table a kernel uses as an LDS / HBM offset must stay inside its target
	{
		auto in = [](int v, int lo, int hi) { return v >= lo && v < hi; };
		const char *bad = nullptr;
		for (int b = 0; b < h.nbody && !bad; b++) {
			if (!in(h.body_mocapid[b], -1, h.nmocap)) bad = "body_mocapid";
			if (!in(h.body_rootid[b], 0, h.nbody) || !in(h.body_weldid[b], 0, h.nbody)) bad = "body_rootid / body_weldid";
			if (h.body_jntnum[b] < 0 || h.body_dofnum[b] < 0 || (h.body_jntnum[b] && !in(h.body_jntadr[b], 0, h.njnt)) ||
			    h.body_jntnum[b] + (h.body_jntnum[b] ? h.body_jntadr[b] : 0) > h.njnt || (h.body_dofnum[b] && !in(h.body_dofadr[b], 0, h.nv)) ||
			    h.body_dofnum[b] + (h.body_dofnum[b] ? h.body_dofadr[b] : 0) > h.nv)
				bad = "body_jntadr / body_jntnum / body_dofadr / body_dofnum";
		}
		for (int j = 0; j < h.njnt && !bad; j++) {
			const int t = h.jnt_type[j], nq = t == MJB_JNT_FREE ? 7 : (t == MJB_JNT_BALL ? 4 : 1), nd = t == MJB_JNT_FREE ? 6 : (t == MJB_JNT_BALL ? 3 : 1);
			if (!in(t, 0, 4)) bad = "jnt_type";
			else if (!in(h.jnt_bodyid[j], 0, h.nbody)) bad = "jnt_bodyid";
			else if (h.jnt_qposadr[j] < 0 || h.jnt_qposadr[j] + nq > h.nq) bad = "jnt_qposadr";
			else if (h.jnt_dofadr[j] < 0 || h.jnt_dofadr[j] + nd > h.nv) bad = "jnt_dofadr";
		}
		for (int i = 0; i < h.nv && !bad; i++) {
			if (!in(h.dof_bodyid[i], 0, h.nbody)) bad = "dof_bodyid";
			if (!in(h.dof_jntid[i], 0, h.njnt)) bad = "dof_jntid";
			if (!in(h.dof_parentid[i], -1, i)) bad = "dof_parentid";
		}
		for (int g = 0; g < h.ngeom && !bad; g++) {
			if (!in(h.geom_bodyid[g], 0, h.nbody)) bad = "geom_bodyid";
			const int t = h.geom_type[g];
			if (t != MJB_GEOM_PLANE && t != MJB_GEOM_SPHERE && t != MJB_GEOM_CAPSULE && t != MJB_GEOM_BOX) bad = "geom_type (plane / sphere / capsule / box)";
			if (!in(h.geom_condim[g], 1, 7) || h.geom_condim[g] == 2 || h.geom_condim[g] == 5) bad = "geom_condim (1, 3, 4, 6)";
		}
		for (int p = 0; p < 2 * h.ncollpair && !bad; p++)
			if (!in(h.collpair_geom[p], 0, h.ngeom)) bad = "collpair_geom";
		for (int p = 0; p < h.ncollpair && !bad; p++)
			if (h.collpair_explicit[p] && h.collpair_condim[p] > 0 && h.collpair_condim[p] != 1 && h.collpair_condim[p] != 3 && h.collpair_condim[p] != 4 && h.collpair_condim[p] != 6)
				bad = "collpair_condim (1, 3, 4, 6)";
		for (int st = 0; st < h.nsite && !bad; st++)
			if (!in(h.site_bodyid[st], 0, h.nbody)) bad = "site_bodyid";
		for (int t = 0; t < h.ntendon && !bad; t++)
			if (h.tendon_num[t] < 0 || h.tendon_adr[t] < 0 || h.tendon_adr[t] + h.tendon_num[t] > h.nwrap) bad = "tendon_adr / tendon_num";
		for (int w = 0; w < h.nwrap && !bad; w++)
			if (!in(h.wrap_objid[w], 0, h.njnt) || h.jnt_type[h.wrap_objid[w]] < MJB_JNT_SLIDE) bad = "wrap_objid (hinge / slide joints)";
		auto objcount = [&](int ot) {
			switch (ot) {
			case MJB_OBJ_BODY: case MJB_OBJ_XBODY: return h.nbody;
			case MJB_OBJ_JOINT: return h.njnt;
			case MJB_OBJ_GEOM: return h.ngeom;
			case MJB_OBJ_SITE: return h.nsite;
			case MJB_OBJ_ACTUATOR: return h.nu;
			default: return 0;
			}
		};
		for (int i = 0; i < h.nsensor && !bad; i++) {
			const int t = h.sensor_type[i];
			int cnt;
			if (t == MJB_SENS_JOINTPOS || t == MJB_SENS_JOINTVEL || t == MJB_SENS_BALLQUAT || t == MJB_SENS_BALLANGVEL || t == MJB_SENS_JOINTACTFRC ||
			    (t >= MJB_SENS_JOINTLIMITPOS && t <= MJB_SENS_JOINTLIMITFRC))
				cnt = h.njnt;
			else if (t == MJB_SENS_TENDONPOS || t == MJB_SENS_TENDONVEL || (t >= MJB_SENS_TENDONLIMITPOS && t <= MJB_SENS_TENDONLIMITFRC)) cnt = h.ntendon;
			else if (t == MJB_SENS_ACTUATORPOS || t == MJB_SENS_ACTUATORVEL || t == MJB_SENS_ACTUATORFRC) cnt = h.nu;
			else if (t == MJB_SENS_SUBTREECOM || t == MJB_SENS_SUBTREELINVEL || t == MJB_SENS_SUBTREEANGMOM) cnt = h.nbody;
			else if (t == MJB_SENS_CLOCK) cnt = 1 << 30;
			else if (t >= MJB_SENS_FRAMEPOS && t <= MJB_SENS_FRAMEANGACC) cnt = objcount(h.sensor_objtype[i]);
			else cnt = h.nsite;  // touch, accelerometer, velocimeter, gyro, force, torque
			if (t != MJB_SENS_CLOCK && !in(h.sensor_objid[i], 0, cnt)) bad = "sensor_objid";
			else if (t >= MJB_SENS_JOINTLIMITPOS && t <= MJB_SENS_JOINTLIMITFRC && h.jnt_type[h.sensor_objid[i]] < MJB_JNT_SLIDE) bad = "sensor_objid (joint limit sensors: hinge / slide joints)";
			if (h.sensor_refid[i] >= 0 && !in(h.sensor_refid[i], 0, objcount(h.sensor_reftype[i]))) bad = "sensor_refid";
			if (!in(h.sensor_dim[i], 1, 5) || h.sensor_adr[i] < 0 || h.sensor_adr[i] + h.sensor_dim[i] > h.nsensordata) bad = "sensor_adr / sensor_dim";
			if (!in(h.sensor_needstage[i], 1, 4)) bad = "sensor_needstage";
		}
		if (bad) {
			fail(MJB_EINVAL, "mjb_compile: index table %s is out of range", bad);
			delete M;
			return nullptr;
		}
	}
	// ---- validation of the tree tables the kernels index with
	for (int b = 1; b < h.nbody; b++)
		if (h.body_parentid[b] < 0 || h.body_parentid[b] >= b) {
			fail(MJB_EINVAL, "mjb_compile: body_parentid[%d] must precede the body", b);
			delete M;
			return nullptr;
		}
	int nM = 0;
	for (int i = 0; i < h.nv; i++) {
		if (h.dof_parentid[i] >= i || h.dof_Madr[i] != nM) {
			fail(MJB_EINVAL, "mjb_compile: inconsistent dof_parentid / dof_Madr at dof %d", i);
			delete M;
			return nullptr;
		}
		int depth = 0;
		for (int j = i; j >= 0; j = h.dof_parentid[j]) {
			M->M_rowdof.push_back(i);
			M->M_coldof.push_back(j);
			depth++;
		}
		M->dof_depth.push_back(depth);
		if (depth > M->maxdepth) M->maxdepth = depth;
		nM += depth;
	}
	if (nM != h.nM) {
		fail(MJB_EINVAL, "mjb_compile: nM = %d does not match the dof tree (%d)", h.nM, nM);
		delete M;
		return nullptr;
	}
	for (int j = 0; j < h.njnt; j++) {
		int t = h.jnt_type[j];
		if (t < 0 || t > 3) {
			fail(MJB_EINVAL, "mjb_compile: bad jnt_type[%d]", j);
			delete M;
			return nullptr;
		}
	}
	for (int i = 0; i < h.nu; i++) {
		int j = h.actuator_trnid[2 * i];
		const bool okj = h.actuator_trntype[i] == MJB_TRN_JOINT && j >= 0 && j < h.njnt && h.jnt_type[j] >= MJB_JNT_SLIDE;
		const bool okt = h.actuator_trntype[i] == MJB_TRN_TENDON && j >= 0 && j < h.ntendon;
		if (!okj && !okt) {
			fail(MJB_EUNSUPPORTED, "mjb_compile: actuator %d: joint transmission on hinge / slide joints and fixed-tendon transmission are supported", i);
			delete M;
			return nullptr;
		}
	}
	// velocity-group start of each dof (hinge/slide: itself; ball: first of 3; free: first of each triple)
	M->dof_jstart.resize(h.nv);
	for (int dd = 0; dd < h.nv; dd++) {
		int j = h.dof_jntid[dd], t = h.jnt_type[j], da = h.jnt_dofadr[j];
		if (t == MJB_JNT_HINGE || t == MJB_JNT_SLIDE) M->dof_jstart[dd] = dd;
		else if (t == MJB_JNT_BALL) M->dof_jstart[dd] = da;
		else M->dof_jstart[dd] = (dd - da < 3) ? da : da + 3;
	}
	// packed per-body / per-dof records and the factorisation micro-program
	for (int b = 0; b < h.nbody; b++) {
		int rec[4] = { h.body_parentid[b], h.body_dofadr[b], h.body_dofnum[b], h.body_rootid[b] };
		int rec2[4] = { h.body_jntadr[b], h.body_jntnum[b], h.body_sameframe[b], h.body_weldid[b] };
		M->body_rec.insert(M->body_rec.end(), rec, rec + 4);
		M->body_rec2.insert(M->body_rec2.end(), rec2, rec2 + 4);
	}
	for (int dd = 0; dd < h.nv; dd++) {
		int rec[4] = { h.dof_Madr[dd], M->dof_depth[dd] - 1, h.dof_bodyid[dd], h.dof_parentid[dd] };
		M->dof_rec.insert(M->dof_rec.end(), rec, rec + 4);
	}
	// what the per-dof / per-joint phases of com_vel / com_pos look up, one 16-byte record each (the chains of dependent table reads --
	// dof -> joint -> type, joint -> body -> root -- cost a trip to L2 per link)
	for (int dd = 0; dd < h.nv; dd++) {
		const int j = h.dof_jntid[dd], bd = h.dof_bodyid[dd];
		int rec[4] = { (h.jnt_type[j] == MJB_JNT_FREE && dd - h.jnt_dofadr[j] < 3) ? 1 : 0, M->dof_jstart[dd] == h.body_dofadr[bd] ? 1 : 0,
			           h.body_parentid[bd], bd };
		M->dof_rec2.insert(M->dof_rec2.end(), rec, rec + 4);
	}
	for (int j = 0; j < h.njnt; j++) {
		const int bi = h.jnt_bodyid[j];
		int rec[4] = { bi, h.jnt_type[j], h.jnt_dofadr[j], h.body_rootid[bi] };
		M->jnt_rec.insert(M->jnt_rec.end(), rec, rec + 4);
	}
	if (M->dof_rec2.empty()) M->dof_rec2.assign(4, 0);
	if (M->jnt_rec.empty()) M->jnt_rec.assign(4, 0);
	for (int k = 0; k < h.nv; k++) {
		M->fac_beg.push_back((int)M->fac_ops.size() / 4);
		const int kk = h.dof_Madr[k], na = M->dof_depth[k] - 1;
		for (int a = 0; a < na; a++) {
			const int i = M->M_coldof[kk + 1 + a];
			for (int bb = a; bb < na; bb++) {
				int op[4] = { h.dof_Madr[i] + (bb - a), kk + 1 + a, kk + 1 + bb, 0 };
				M->fac_ops.insert(M->fac_ops.end(), op, op + 4);
			}
		}
	}
	M->fac_beg.push_back((int)M->fac_ops.size() / 4);
	// the same updates scheduled by LEVELS of the elimination tree (= the dof tree): pivots of equal height touch disjoint rows of
	// their own and only meet in the entries of common ancestors, so a level runs in one round: one lane per contribution
	// LD[dst] -= LD[srcA] / D_pivot * LD[srcB], the contributions to one entry combined by its first lane in descending pivot order.
	// flv_hdr [level][4] = { slots, longest list of slot 0 / 1 / 2 };  flv_rec [level][3][64][4] (below).  flv_n == 0 (a level with
	// more than 192 contributions, a list longer than 6, more than 64 levels): the kernels keep the pivot-by-pivot factorisation
	{
		std::vector<int> hgt((size_t)(h.nv > 0 ? h.nv : 1), 0);
		int nlev = 0, nseg = 0;
		bool flv_ok = true;
		for (int k = h.nv - 1; k >= 0; k--) {
			const int pk = h.dof_parentid[k];
			if (pk >= 0 && hgt[pk] < hgt[k] + 1) hgt[pk] = hgt[k] + 1;
			if (hgt[k] + 1 > nlev) nlev = hgt[k] + 1;
		}
		for (int lev = 0; lev < nlev; lev++) {
			std::vector<int> dsts;
			std::vector<std::vector<int>> lists;
			for (int k = h.nv - 1; k >= 0; k--) {
				if (hgt[k] != lev) continue;
				const int kk = h.dof_Madr[k], na = M->dof_depth[k] - 1;
				for (int a = 0; a < na; a++) {
					const int i = M->M_coldof[kk + 1 + a];
					for (int bb = a; bb < na; bb++) {
						const int dst = h.dof_Madr[i] + (bb - a);
						size_t at = 0;
						while (at < dsts.size() && dsts[at] != dst) at++;
						if (at == dsts.size()) {
							dsts.push_back(dst);
							lists.emplace_back();
						}
						lists[at].push_back(kk + 1 + a);
						lists[at].push_back(kk + 1 + bb);
						lists[at].push_back(k);
					}
				}
			}
			// a level's CONTRIBUTIONS at fixed places, one 16-byte word per lane in up to three slots: { dst | (row + 1 of a diagonal
			// dst) << 16, valid | owner << 1 | list length << 8, srcA | srcB << 16, pivot }.  The contributions to one entry sit in
			// consecutive lanes of one 16-lane row, the entry's owner first: it collects the others' products by DPP row shifts.
			// The lists go first (they must not straddle a row), the single contributions fill up.
			if (dsts.empty()) continue;
			std::vector<int> W;
			int tm[3] = { 0, 0, 0 };
			auto put = [&](size_t it) {
				const int d = dsts[it], n = (int)lists[it].size() / 3;
				const int drow1 = M->M_rowdof[d] == M->M_coldof[d] ? M->M_rowdof[d] + 1 : 0;
				while ((int)(W.size() / 4) % 16 + n > 16) W.insert(W.end(), 4, 0);
				for (int t = 0; t < n; t++) {
					const int slot = (int)(W.size() / 4) / 64;
					if (slot < 3 && n > tm[slot]) tm[slot] = n;
					int r[4] = { d | (drow1 << 16), 1 | (t == 0 ? 2 : 0) | (n << 8), lists[it][3 * t] | (lists[it][3 * t + 1] << 16), lists[it][3 * t + 2] };
					W.insert(W.end(), r, r + 4);
				}
			};
			for (size_t it = 0; it < dsts.size(); it++) {
				if (lists[it].size() / 3 > 6) flv_ok = false;
				else if (lists[it].size() / 3 > 1) put(it);
			}
			for (size_t it = 0; it < dsts.size(); it++)
				if (lists[it].size() / 3 == 1) put(it);
			const int nslot = ((int)(W.size() / 4) + 63) / 64;
			if (nslot > 3) flv_ok = false;
			W.resize((size_t)3 * 64 * 4, 0);
			int hdr[4] = { nslot, tm[0], tm[1], tm[2] };
			M->flv_hdr.insert(M->flv_hdr.end(), hdr, hdr + 4);
			M->flv_rec.insert(M->flv_rec.end(), W.begin(), W.end());
			nseg++;
		}
		if (h.nM >= 65536 || nseg > 64) flv_ok = false;
		if (!flv_ok) {
			nseg = 0;
			M->flv_hdr.clear();
			M->flv_rec.clear();
		}
		// (three empty levels behind the last: the kernel's look-ahead fetches need no bounds test)
		M->flv_rec.insert(M->flv_rec.end(), (size_t)3 * (3 * 64 * 4), 0);
		M->flv_n = nseg;
		// per qM entry: its row, and whether it is the diagonal one (the first and the last round of factor_levels)
		M->flv_ent.assign((size_t)((h.nM + 255) / 256 * 256 + 256), 0);
		for (int en = 0; en < h.nM; en++) M->flv_ent[en] = M->M_rowdof[en] | (M->M_rowdof[en] == M->M_coldof[en] ? 0x10000 : 0);
		if (M->flv_hdr.empty()) M->flv_hdr.assign(4, 0);
		if (M->flv_rec.empty()) M->flv_rec.assign(4, 0);
	}
	// ancestor-dof bit masks per body (contact Jacobians)
	M->body_dofmask.assign((size_t)2 * h.nbody, 0);
	for (int b = 1; b < h.nbody; b++) {
			int bb = b;
			while (bb > 0 && h.body_dofnum[bb] == 0) bb = h.body_parentid[bb];
			if (bb == 0) continue;
			for (int i = h.body_dofadr[bb] + h.body_dofnum[bb] - 1; i >= 0; i = h.dof_parentid[i])
				M->body_dofmask[2 * b + (i >> 5)] |= (int)(1u << (i & 31));
		}
	// dense address map of the joint-space inertia for the register-resident factor / solve (nv <= 16)
	M->M_dense.assign(256, -1);
	if (h.nv <= 16)
		for (int en = 0; en < h.nM; en++) M->M_dense[16 * M->M_rowdof[en] + M->M_coldof[en]] = en;
	// symmetric dense address map (nv <= 32): qM address of entry (i, j) = (j, i), -1 where the tree leaves a zero.  The primal
	// solvers keep row `lane` of M in registers (fwd_constraint_newton) instead of a dense nv x nv copy in the frame.
	M->M_sym.assign(1024, -1);
	if (h.nv <= 32)
		for (int en = 0; en < h.nM; en++) {
			M->M_sym[32 * M->M_rowdof[en] + M->M_coldof[en]] = en;
			M->M_sym[32 * M->M_coldof[en] + M->M_rowdof[en]] = en;
		}
	// ancestors at distance 2^r for the pointer-jumping kinematics (0 = world or beyond the root)
	{
		int maxd = 0;
		std::vector<int> depth((size_t)h.nbody, 0);
		for (int b = 1; b < h.nbody; b++) {
			depth[b] = depth[h.body_parentid[b]] + 1;
			if (depth[b] > maxd) maxd = depth[b];
		}
		M->kin_rounds = 0;
		while ((1 << M->kin_rounds) < maxd) M->kin_rounds++;
		M->body_anc.assign((size_t)(M->kin_rounds + 2) * h.nbody, 0);
		for (int b = 1; b < h.nbody; b++) M->body_anc[b] = h.body_parentid[b];
		for (int r = 1; r < M->kin_rounds + 2; r++)
			for (int b = 1; b < h.nbody; b++) {
				const int a = M->body_anc[(size_t)(r - 1) * h.nbody + b];
				M->body_anc[(size_t)r * h.nbody + b] = a ? M->body_anc[(size_t)(r - 1) * h.nbody + a] : 0;
			}
	}
	// ancestor dof lists for the root-to-leaf sums
	{
		M->body_dofanc.assign((size_t)4 * h.nbody, -1);
		bool aok = h.nv <= 255 && h.nv > 0;
		int amax = 0;
		for (int b = 1; b < h.nbody && aok; b++) {
			int n = 0;
			for (int i = 0; i < h.nv && i < 64; i++)
				if ((M->body_dofmask[2 * b + (i >> 5)] >> (i & 31)) & 1) {
					if (n >= 16) { aok = false; break; }
					unsigned int w = (unsigned int)M->body_dofanc[4 * b + (n >> 2)];
					w = (w & ~(0xFFu << (8 * (n & 3)))) | ((unsigned int)i << (8 * (n & 3)));
					M->body_dofanc[4 * b + (n >> 2)] = (int)w;
					n++;
				}
			if (n > amax) amax = n;
		}
		if (h.nv > 64) aok = false;
		M->dofanc_max = aok ? amax : 0;
	}
	// subtree membership masks (subtree com, composite inertia)
	M->body_submask.assign((size_t)2 * h.nbody, 0);
	for (int b = 0; b < h.nbody && h.nbody <= 64; b++)
		for (int a = b;; a = h.body_parentid[a]) {
			M->body_submask[2 * a + (b >> 5)] |= (int)(1u << (b & 31));
			if (a == 0) break;
		}
	M->sub_nt = h.nbody <= 16 ? 1 : (h.nbody <= 32 ? 2 : 0);
	M->sub_S.assign((size_t)(M->sub_nt > 0 ? M->sub_nt * 4 * M->sub_nt * 64 : 1), 0.0);
	for (int t = 0; t < M->sub_nt; t++)
		for (int k = 0; k < 4 * M->sub_nt; k++)
			for (int l = 0; l < 64; l++) {
				const int a = 16 * t + (l & 15), b = 4 * k + (l >> 4);
				if (a < h.nbody && b < h.nbody && ((M->body_submask[2 * a + (b >> 5)] >> (b & 31)) & 1))
					M->sub_S[((size_t)t * 4 * M->sub_nt + k) * 64 + l] = 1.0;
			}
	M->dof_bodymask.assign((size_t)2 * (h.nv > 0 ? h.nv : 1), 0);
	for (int b = 1; b < h.nbody; b++)
		for (int i = 0; i < h.nv; i++)
			if ((M->body_dofmask[2 * b + (i >> 5)] >> (i & 31)) & 1) M->dof_bodymask[2 * i + (b >> 5)] |= (int)(1u << (b & 31));
	M->le_topo = mjb_lane_env_match(&h);
	if (M->le_topo < 0 && mjb_lane_env_eligible(&h)) M->le_topo = MJB_LE_TOPO_JIT;
	// the split step (smooth half in lane = env form, constraint half one env per wavefront): the constraint half is kernel variant 9's -- plain PGS,
	// pyramidal / frictionless contacts, nv <= 16, no stage that needs mj_rnePostConstraint
	M->sm_topo = -1;
	if (h.nefcmax > 0 && h.solver == MJB_SOL_PGS && !(h.cone == MJB_CONE_ELLIPTIC && h.nconmax > 0) && h.nv <= 16 && !M->need_rnepost && h.nefcmax <= 128)
		M->sm_topo = mjb_smooth_match(&h);
	if (M->le_topo != MJB_LE_TOPO_NONE || M->sm_topo >= 0) {
		M->le_tape.assign(mjb_lane_env_tape_doubles(&h), 0.0);
		mjb_lane_env_tape(&h, M->le_tape.data());
	}
	M->eulerdamp = 0;
	M->damp_int.assign(h.dof_damping, h.dof_damping + h.nv);
	if (h.integrator == MJB_INT_IMPLICITFAST) {
		// mj_implicit, mjINT_IMPLICITFAST: qH = M - h D, D = mjd_passive_vel + mjd_actuator_vel; a model constant on the diagonal here
		for (int t = 0; t < h.ntendon; t++)
			if (h.tendon_damping[t] != 0) {
				fail(MJB_EUNSUPPORTED, "mjb_compile: integrator implicitfast with tendon damping is not supported");
				delete M;
				return nullptr;
			}
		for (int i = 0; i < h.nv; i++) M->damp_int[i] = (h.disableflags & MJB_DSBL_PASSIVE) ? 0.0 : h.dof_damping[i];
		for (int i = 0; i < h.nu; i++) {
			if (h.actuator_gaintype[i] == MJB_GAIN_AFFINE && h.actuator_gainprm[3 * i + 2] != 0) {
				fail(MJB_EUNSUPPORTED, "mjb_compile: integrator implicitfast with a velocity term in an affine actuator gain is not supported");
				delete M;
				return nullptr;
			}
			if (h.disableflags & MJB_DSBL_ACTUATION) continue;
			const double bv = h.actuator_biastype[i] == MJB_BIAS_AFFINE ? h.actuator_biasprm[3 * i + 2] : 0.0, g = h.actuator_gear[6 * i];
			if (h.actuator_trntype[i] == MJB_TRN_TENDON) {
				if (bv != 0) {  // (moment' bv moment couples the tendon's joints: not diagonal)
					fail(MJB_EUNSUPPORTED, "mjb_compile: integrator implicitfast with a velocity-dependent actuator on a tendon is not supported");
					delete M;
					return nullptr;
				}
				continue;
			}
			M->damp_int[h.jnt_dofadr[h.actuator_trnid[2 * i]]] -= g * g * bv;
		}
		for (int i = 0; i < h.nv; i++)
			if (M->damp_int[i] != 0) M->eulerdamp = 1;
	} else if (!(h.disableflags & MJB_DSBL_EULERDAMP))
		for (int i = 0; i < h.nv; i++)
			if (h.dof_damping[i] > 0) M->eulerdamp = 1;
	M->nfriction = 0;
	if (!(h.disableflags & MJB_DSBL_FRICTIONLOSS))
		for (int i = 0; i < h.nv; i++)
			if (h.dof_frictionloss[i] > 0) M->nfriction++;
	if (!(h.disableflags & MJB_DSBL_FRICTIONLOSS))
		for (int t = 0; t < h.ntendon; t++)
			if (h.tendon_frictionloss[t] > 0) M->nfriction++;
	// per-pair records of the static candidate list: everything mj_collision / mj_contactParam derive from the two geoms'
	// constants, resolved once (friction stays per env: rule 0 = max of both geoms, 1 = geom1, 2 = geom2)
	M->pair_i.assign((size_t)8 * (h.ncollpair > 0 ? h.ncollpair : 1), 0);
	M->pair_d.assign((size_t)24 * (h.ncollpair > 0 ? h.ncollpair : 1), 0.0);
	for (int p = 0; p < h.ncollpair; p++) {
		const int g1 = h.collpair_geom[2 * p], g2 = h.collpair_geom[2 * p + 1];
		int *pi = M->pair_i.data() + 8 * p;
		double *pd = M->pair_d.data() + 24 * p;
		pi[0] = g1; pi[1] = g2; pi[2] = h.geom_type[g1]; pi[3] = h.geom_type[g2];
		pi[6] = MJB_COLFUNC_DEFAULT;  // collision-function override (mjb_register_collision patches the batch's copy)
		pi[7] = 0;
		for (int k = 0; k < 3; k++) {
			pd[k] = h.geom_size[3 * g1 + k];
			pd[3 + k] = h.geom_size[3 * g2 + k];
		}
		const double margin = std::max(h.geom_margin[g1], h.geom_margin[g2]), gap = std::max(h.geom_gap[g1], h.geom_gap[g2]);
		pd[6] = margin; pd[7] = gap; pd[8] = h.geom_rbound[g1]; pd[9] = h.geom_rbound[g2];
		double *solref = pd + 10, *solimp = pd + 12;
		const int pr1 = h.geom_priority[g1], pr2 = h.geom_priority[g2];
		if (pr1 != pr2) {  // mj_contactParam: the geom with the higher priority decides
			const int g = pr1 > pr2 ? g1 : g2;
			pi[4] = h.geom_condim[g];
			pi[5] = pr1 > pr2 ? 1 : 2;
			for (int k = 0; k < 2; k++) solref[k] = h.geom_solref[2 * g + k];
			for (int k = 0; k < 5; k++) solimp[k] = h.geom_solimp[5 * g + k];
		} else {
			pi[4] = std::max(h.geom_condim[g1], h.geom_condim[g2]);
			pi[5] = 0;
			const double s1 = h.geom_solmix[g1], s2 = h.geom_solmix[g2];
			double mix;
			if (s1 >= 1e-15 && s2 >= 1e-15) mix = s1 / (s1 + s2);
			else if (s1 < 1e-15 && s2 < 1e-15) mix = 0.5;
			else if (s1 < 1e-15) mix = 0.0;
			else mix = 1.0;
			const double r10 = h.geom_solref[2 * g1], r20 = h.geom_solref[2 * g2];
			for (int k = 0; k < 2; k++) {
				const double a = h.geom_solref[2 * g1 + k], b = h.geom_solref[2 * g2 + k];
				solref[k] = (r10 > 0 && r20 > 0) ? mix * a + (1 - mix) * b : std::min(a, b);
			}
			for (int k = 0; k < 5; k++) solimp[k] = mix * h.geom_solimp[5 * g1 + k] + (1 - mix) * h.geom_solimp[5 * g2 + k];
		}
		pd[17] = margin - gap;
		pd[21] = h.body_invweight0[2 * h.geom_bodyid[g1]] + h.body_invweight0[2 * h.geom_bodyid[g2]];
		for (int k = 0; k < 3; k++) {  // mj_contactParam's friction of the MODEL's geoms (per-env overrides are mixed on the device)
			const double a = h.geom_friction[3 * g1 + k], b = h.geom_friction[3 * g2 + k];
			pd[18 + k] = pi[5] == 0 ? std::max(a, b) : (pi[5] == 1 ? a : b);
		}
		if (h.collpair_explicit[p]) {
			// <contact><pair>: what the pair states goes on top of the geoms' mix (collpair_param: friction[5] solref[2] solimp[5] margin gap, NaN = not stated)
			const double *pp = h.collpair_param + 14 * p;
			if (h.collpair_condim[p] > 0) pi[4] = h.collpair_condim[p];
			if (!std::isnan(pp[5])) { solref[0] = pp[5]; solref[1] = pp[6]; }
			if (!std::isnan(pp[7])) for (int k = 0; k < 5; k++) solimp[k] = pp[7 + k];
			if (!std::isnan(pp[12])) pd[6] = pp[12];
			if (!std::isnan(pp[13])) pd[7] = pp[13];
			pd[17] = pd[6] - pd[7];
			if (!std::isnan(pp[0])) {  // the pair's five friction numbers: tangent 1, spin, roll 1 in the slots of the geoms' three; tangent 2, roll 2 in the pads
				pi[5] = 4;
				pd[18] = pp[0]; pd[19] = pp[2]; pd[20] = pp[3]; pd[22] = pp[1]; pd[23] = pp[4];
			}
		}
	}
	{  // limit records (see mjb_dev.h)
		const int nl = h.njnt + h.ntendon;
		M->lim_d.assign((size_t)24 * (nl > 0 ? nl : 1), 0.0);
		M->lim_i.assign((size_t)4 * (nl > 0 ? nl : 1), 0);
		for (int j = 0; j < h.njnt; j++) {
			double *r = M->lim_d.data() + 24 * j;
			int *ri = M->lim_i.data() + 4 * j;
			ri[0] = !h.jnt_limited[j] ? 0 : (h.jnt_type[j] >= MJB_JNT_SLIDE ? 1 : (h.jnt_type[j] == MJB_JNT_BALL ? 2 : 0));  // (2: a ball joint's angle limit)
			ri[1] = h.jnt_qposadr[j];
			ri[2] = h.jnt_dofadr[j];
			r[0] = h.jnt_range[2 * j]; r[1] = h.jnt_range[2 * j + 1]; r[6] = h.jnt_margin[j];
			r[10] = h.jnt_solref[2 * j]; r[11] = h.jnt_solref[2 * j + 1];
			for (int k = 0; k < 5; k++) r[12 + k] = h.jnt_solimp[5 * j + k];
			r[21] = h.dof_invweight0[h.jnt_dofadr[j]];
		}
		for (int t = 0; t < h.ntendon; t++) {
			double *r = M->lim_d.data() + 24 * (h.njnt + t);
			int *ri = M->lim_i.data() + 4 * (h.njnt + t);
			ri[0] = h.tendon_limited[t] ? 1 : 0;
			ri[1] = t;
			r[0] = h.tendon_range[2 * t]; r[1] = h.tendon_range[2 * t + 1]; r[6] = h.tendon_margin[t];
			r[10] = h.tendon_solref_lim[2 * t]; r[11] = h.tendon_solref_lim[2 * t + 1];
			for (int k = 0; k < 5; k++) r[12 + k] = h.tendon_solimp_lim[5 * t + k];
			r[21] = h.tendon_invweight0[t];
		}
	}
	compute_layout(M, M->L, false);
	choose_fused_layout(M);
	build_sensor_tables(M);
	// actuators per dof (CSR, ascending actuator id)
	M->dof_act_adr.assign((size_t)h.nv + 1, 0);
	// (one entry per (dof, actuator) with the actuator's moment arm on that dof: gear for a joint transmission, gear * coefficient for every
	//  joint of a fixed tendon)
	M->act_tendon = 0;
	auto each_arm = [&](auto &&fn) {
		for (int i = 0; i < h.nu; i++) {
			const int id = h.actuator_trnid[2 * i];
			if (h.actuator_trntype[i] == MJB_TRN_TENDON) {
				M->act_tendon = 1;
				for (int w = h.tendon_adr[id]; w < h.tendon_adr[id] + h.tendon_num[id]; w++)
					fn(h.jnt_dofadr[h.wrap_objid[w]], i, h.actuator_gear[6 * i] * h.wrap_prm[w]);
			} else
				fn(h.jnt_dofadr[id], i, h.actuator_gear[6 * i]);
		}
	};
	each_arm([&](int dof, int, double) { M->dof_act_adr[dof + 1]++; });
	for (int dd = 0; dd < h.nv; dd++) M->dof_act_adr[dd + 1] += M->dof_act_adr[dd];
	M->dof_act_id.assign((size_t)M->dof_act_adr[h.nv], 0);
	M->dof_act_mom.assign((size_t)M->dof_act_adr[h.nv], 0.0);
	{
		std::vector<int> fill(M->dof_act_adr.begin(), M->dof_act_adr.end() - 1);
		each_arm([&](int dof, int i, double arm) {
			M->dof_act_id[fill[dof]] = i;
			M->dof_act_mom[fill[dof]++] = arm;
		});
	}
	{
		const int frame_bytes = ((M->L.ndouble * 8 + M->L.nint * 4) + 15) & ~15;
		if (frame_bytes > mjb_max_lds_bytes()) {
			fail(MJB_EUNSUPPORTED, "mjb_compile: the per-env working set (%d bytes) exceeds one CU's LDS (%d bytes); "
			                       "lower nconmax / nefcmax", frame_bytes, mjb_max_lds_bytes());
			delete M;
			return nullptr;
		}
	}
	g_err.clear();
	return M;
}

void mjb_free_model(mjb_model *m) { delete m; }

int mjb_field_size(const mjb_model *m, int field)
{
	if (!m || field < 0 || field >= MJB_F_COUNT) return fail(MJB_EINVAL, "mjb_field_size: bad argument");
	return m->field_size[field];
}
int mjb_field_is_int(int field) { return field >= 0 && field < MJB_F_COUNT && kFields[field].kind == 3; }
int mjb_field_is_state(int field) { return field >= 0 && field < MJB_F_COUNT && kFields[field].kind == 0; }
const char *mjb_field_name(int field) { return field >= 0 && field < MJB_F_COUNT ? kFields[field].name : ""; }
int mjb_frame_doubles(const mjb_model *m) { return m ? m->L.ndouble + m->L.nint / 2 : fail(MJB_EINVAL, "null model"); }
int mjb_frame_bytes(const mjb_model *m, int fused)
{
	if (!m) return fail(MJB_EINVAL, "null model");
	const FrameLayout &L = fused == 2 ? m->Lw : (fused ? m->Lc : m->L);  // (2: the wide fused frame of kernel variant 4, == 1 when the model has none)
	return ((L.ndouble * 8 + L.nint * 4) + 15) & ~15;
}
int mjb_frame_offset(const mjb_model *m, int field, int fused)
{
	if (!m || field < 0 || field >= MJB_F_COUNT) return fail(MJB_EINVAL, "bad model / field");
	const FrameLayout &L = fused ? m->Lc : m->L;
	const int *slots[] = {
#define MJB_DS(name, rows, cols) &L.name,
#define MJB_DD(name, rows, cols) &L.name,
#define MJB_DD2(name, rows, cols) &L.name,
#define MJB_DI(name, rows, cols) &L.name,
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	};
	return *slots[field];
}

void mjb_free_batch(mjb_batch *b)
{
	if (!b) return;
	hipSetDevice(b->device);
	if (b->stream) hipStreamSynchronize(b->stream);
#define MJB_DS(name, rows, cols) if (b->st.name) hipFree(b->st.name);
#define MJB_DD(name, rows, cols)
#define MJB_DD2(name, rows, cols)
#define MJB_DI(name, rows, cols)
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	if (b->st.frame_ws) hipFree(b->st.frame_ws);
	if (b->st.nwarn) hipFree(b->st.nwarn);
	if (b->stats_dev) hipFree(b->stats_dev);
	if (b->st.pgs_B) hipFree(b->st.pgs_B);
	if (b->st.efc_Jg) hipFree(b->st.efc_Jg);
	if (b->rowstat_dev) hipFree(b->rowstat_dev);
	if (b->handoff_dev) hipFree(b->handoff_dev);
	if (b->reset_step_dev) hipFree(b->reset_step_dev);
	for (hipStream_t st : b->split_streams) hipStreamDestroy(st);
	for (hipEvent_t ev : b->split_events) hipEventDestroy(ev);
	if (b->ev_split_fork) hipEventDestroy(b->ev_split_fork);
	if (b->rowstat_host) hipHostFree(b->rowstat_host);
	if (b->ev_rowstat) hipEventDestroy(b->ev_rowstat);
	if (b->pack_dev) hipFree(b->pack_dev);
	if (b->rest_stream) {
		hipStreamSynchronize(b->rest_stream);
		hipStreamDestroy(b->rest_stream);
		hipEventDestroy(b->ev_fork);
		hipEventDestroy(b->ev_join);
	}
	if (b->st.sched) hipFree(b->st.sched);
	if (b->metrics_dev) hipFree(b->metrics_dev);
	if (b->st.prof) hipFree(b->st.prof);
	if (b->blob) hipFree(b->blob);
	if (b->mask_dev) hipFree(b->mask_dev);
	if (b->params_dev) hipFree(b->params_dev);
	if (b->env_gravity) hipFree(b->env_gravity);
	if (b->env_geom_friction) hipFree(b->env_geom_friction);
	if (b->env_geom_size) hipFree(b->env_geom_size);
	if (b->env_geom_type) hipFree(b->env_geom_type);
	if (b->env_equality) hipFree(b->env_equality);
	if (b->env_mass) hipFree(b->env_mass);
	if (b->hw_ints) hipFree(b->hw_ints);
	if (b->hw_gains) hipFree(b->hw_gains);
	if (b->hw_cmd) hipFree(b->hw_cmd);
	if (b->hw_pid) hipFree(b->hw_pid);
	if (b->hw_cad) hipFree(b->hw_cad);
	if (b->noise_stream) {
		hipStreamSynchronize(b->noise_stream);
		hipStreamDestroy(b->noise_stream);
		hipEventDestroy(b->ev_noise);
		hipEventDestroy(b->ev_noise_go);
	}
	if (b->zbuf) hipFree(b->zbuf);
	if (b->zinfo) hipFree(b->zinfo);
	if (b->sens_flag_dev) hipFree(b->sens_flag_dev);
	if (b->sens_mean_dev) hipFree(b->sens_mean_dev);
	if (b->sens_sigma_dev) hipFree(b->sens_sigma_dev);
	if (b->sens_value) hipFree(b->sens_value);
	if (b->sens_truth) hipFree(b->sens_truth);
	if (b->own_stream && b->stream) hipStreamDestroy(b->stream);
	delete b;
}

mjb_batch *mjb_make_batch(const mjb_model *M, int nenv, int device)
{
	if (!M || nenv <= 0) {
		fail(MJB_EINVAL, "mjb_make_batch: bad argument");
		return nullptr;
	}
	int ndev = mjb_device_count();
	if (ndev <= 0) {
		fail(MJB_ENODEVICE, "mjb_make_batch: no HIP device available (this engine has no CPU fallback)");
		return nullptr;
	}
	if (device < 0 || device >= ndev) {
		fail(MJB_EINVAL, "mjb_make_batch: device %d out of range (%d devices)", device, ndev);
		return nullptr;
	}
	if (hipSetDevice(device) != hipSuccess) {
		fail(MJB_ENODEVICE, "mjb_make_batch: hipSetDevice(%d) failed", device);
		return nullptr;
	}
	mjb_batch *b = new (std::nothrow) mjb_batch;
	if (!b) {
		fail(MJB_ENOMEM, "mjb_make_batch: out of memory");
		return nullptr;
	}
	b->model = M;
	b->device = device;
	b->nenv = nenv;
	b->L = M->L;
	if (const char *v = getenv("MJB_LANE_ENV")) {  // default lane_env_mode of new batches (include/mjb.h, mjb_set_lane_env)
		const int k = atoi(v);
		if (k >= -1 && k <= 1) b->lane_env_mode = k;
	}
	const mjb_model_desc &h = M->h;
	// ---- device model blob: [ints | doubles | derived int tables]
	size_t ni = M->hint.size(), nd = M->hdbl.size();
	size_t nt = M->M_rowdof.size() + M->M_coldof.size() + M->dof_depth.size() + M->dof_jstart.size() +
	            M->body_rec.size() + M->body_rec2.size() + M->dof_rec.size() + M->fac_ops.size() + M->fac_beg.size() +
	            M->body_dofmask.size() + M->body_submask.size() + M->M_dense.size() + M->M_sym.size() + M->body_anc.size() + M->dof_bodymask.size() + M->body_dofanc.size() + M->dof_rec2.size() + M->jnt_rec.size() + M->flv_hdr.size() + M->flv_rec.size() + M->flv_ent.size() + M->sens_copy.size() + M->sens_slow.size() + M->dof_act_adr.size() +
	            M->dof_act_id.size() + M->pair_i.size() + M->lim_i.size() + 104;
	size_t bytes_i = ((ni + nt) * sizeof(int) + 15) & ~size_t(15);
	// (the lane = env tape starts on a 64-byte boundary of the blob: wide scalar loads)
	const size_t o_damp = nd + M->pair_d.size() + M->lim_d.size() + M->sub_S.size();
	const size_t o_mom = o_damp + M->damp_int.size();
	size_t o_tape = o_mom + M->dof_act_mom.size();
	while ((bytes_i + o_tape * sizeof(double)) % 64) o_tape++;
	size_t bytes = bytes_i + (o_tape + M->le_tape.size()) * sizeof(double) + 16;
	if (hipMalloc(&b->blob, bytes) != hipSuccess) {
		fail(MJB_ENOMEM, "mjb_make_batch: hipMalloc(model blob) failed");
		mjb_free_batch(b);
		return nullptr;
	}
	std::vector<unsigned char> hostblob(bytes, 0);
	int *hi = reinterpret_cast<int *>(hostblob.data());
	double *hd = reinterpret_cast<double *>(hostblob.data() + bytes_i);
	if (ni) memcpy(hi, M->hint.data(), ni * sizeof(int));
	size_t t0 = ni;
	auto put = [&](const std::vector<int> &v) {
		t0 = (t0 + 3) & ~size_t(3);  // 16-byte aligned tables (wide scalar loads)
		size_t at = t0;
		if (!v.empty()) memcpy(hi + t0, v.data(), v.size() * sizeof(int));
		t0 += v.size();
		return at;
	};
	size_t o_row = put(M->M_rowdof), o_col = put(M->M_coldof), o_dep = put(M->dof_depth), o_js = put(M->dof_jstart);
	size_t o_br = put(M->body_rec), o_br2 = put(M->body_rec2), o_dr = put(M->dof_rec), o_fo = put(M->fac_ops),
	       o_fb = put(M->fac_beg), o_dm = put(M->body_dofmask), o_sm = put(M->body_submask), o_md = put(M->M_dense), o_ms = put(M->M_sym), o_an = put(M->body_anc), o_db = put(M->dof_bodymask), o_sc = put(M->sens_copy), o_ss = put(M->sens_slow),
	       o_aa = put(M->dof_act_adr), o_ai = put(M->dof_act_id), o_pi = put(M->pair_i), o_li = put(M->lim_i), o_da = put(M->body_dofanc), o_dr2 = put(M->dof_rec2), o_jr = put(M->jnt_rec), o_fh = put(M->flv_hdr), o_fr = put(M->flv_rec), o_fe = put(M->flv_ent);
	if (nd) memcpy(hd, M->hdbl.data(), nd * sizeof(double));
	memcpy(hd + nd, M->pair_d.data(), M->pair_d.size() * sizeof(double));
	memcpy(hd + nd + M->pair_d.size(), M->lim_d.data(), M->lim_d.size() * sizeof(double));
	memcpy(hd + nd + M->pair_d.size() + M->lim_d.size(), M->sub_S.data(), M->sub_S.size() * sizeof(double));
	if (!M->damp_int.empty()) memcpy(hd + o_damp, M->damp_int.data(), M->damp_int.size() * sizeof(double));
	if (!M->dof_act_mom.empty()) memcpy(hd + o_mom, M->dof_act_mom.data(), M->dof_act_mom.size() * sizeof(double));
	if (!M->le_tape.empty()) memcpy(hd + o_tape, M->le_tape.data(), M->le_tape.size() * sizeof(double));
	if (hipMemcpy(b->blob, hostblob.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
		fail(MJB_ENODEVICE, "mjb_make_batch: model upload failed");
		mjb_free_batch(b);
		return nullptr;
	}
	int *di = reinterpret_cast<int *>(b->blob);
	double *dd = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(b->blob) + bytes_i);
	DevModel &dm = b->dm;
	{
		size_t ii = 0, dj = 0;
#define MJB_SIZE(name) dm.name = h.name;
#define MJB_OPT_I(name) dm.name = h.name;
#define MJB_OPT_D(name, n) for (int k = 0; k < n; k++) dm.name[k] = h.name[k];
#define MJB_ARR_I(name, rows, cols) dm.name = (mjb_ciptr)(di + M->ioff[ii++]);
#define MJB_ARR_D(name, rows, cols) dm.name = (mjb_cdptr)(dd + M->doff[dj++]);
#include "../../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	}
	dm.M_rowdof = (mjb_ciptr)(di + o_row);
	dm.M_coldof = (mjb_ciptr)(di + o_col);
	dm.dof_depth = (mjb_ciptr)(di + o_dep);
	dm.dof_jstart = (mjb_ciptr)(di + o_js);
	dm.body_rec = (mjb_ciptr)(di + o_br);
	dm.body_rec2 = (mjb_ciptr)(di + o_br2);
	dm.dof_rec = (mjb_ciptr)(di + o_dr);
	dm.fac_ops = (mjb_ciptr)(di + o_fo);
	dm.fac_beg = (mjb_ciptr)(di + o_fb);
	dm.body_dofmask = (mjb_ciptr)(di + o_dm);
	dm.body_submask = (mjb_ciptr)(di + o_sm);
	dm.M_dense = (mjb_ciptr)(di + o_md);
	dm.M_sym = (mjb_ciptr)(di + o_ms);
	dm.body_anc = (mjb_ciptr)(di + o_an);
	dm.kin_rounds = M->kin_rounds;
	dm.body_dofanc = (mjb_ciptr)(di + o_da);
	dm.dofanc_max = M->dofanc_max;
	dm.dof_rec2 = (mjb_ciptr)(di + o_dr2);
	dm.jnt_rec = (mjb_ciptr)(di + o_jr);
	dm.flv_hdr = (mjb_ciptr)(di + o_fh);
	dm.flv_rec = (mjb_ciptr)(di + o_fr);
	dm.flv_n = M->flv_n;
	dm.flv_ent = (mjb_ciptr)(di + o_fe);
	dm.dof_bodymask = (mjb_ciptr)(di + o_db);
	dm.need_rnepost = M->need_rnepost;
	dm.sens_copy = (mjb_ciptr)(di + o_sc);
	dm.sens_slow = (mjb_ciptr)(di + o_ss);
	dm.dof_act_adr = (mjb_ciptr)(di + o_aa);
	dm.dof_act_id = (mjb_ciptr)(di + o_ai);
	dm.pair_i = (mjb_ciptr)(di + o_pi);
	b->pair_i_dev = di + o_pi;
	dm.pair_d = (mjb_cdptr)(dd + nd);
	dm.lim_d = (mjb_cdptr)(dd + nd + M->pair_d.size());
	dm.sub_S = (mjb_cdptr)(dd + nd + M->pair_d.size() + M->lim_d.size());
	dm.dof_damping_int = (mjb_cdptr)(dd + o_damp);
	dm.dof_act_mom = (mjb_cdptr)(dd + o_mom);
	dm.act_tendon = M->act_tendon;
	dm.sub_nt = M->sub_nt;
	dm.le_tape = M->le_tape.empty() ? (mjb_cdptr) nullptr : (mjb_cdptr)(dd + o_tape);
	dm.lim_i = (mjb_ciptr)(di + o_li);
	for (int k = 0; k < 3; k++) {
		dm.sens_ncopy[k] = M->sens_ncopy[k];
		dm.sens_nslow[k] = M->sens_nslow[k];
	}
	dm.sens_ncopy_max = M->sens_ncopy_max ? M->sens_ncopy_max : 1;
	dm.eulerdamp = M->eulerdamp;
	dm.nfriction = M->nfriction;
	dm.maxdepth = M->maxdepth;

	// ---- state arrays
	DevState &s = b->st;
	s.nenv = nenv;
	bool ok = true;
	int fidx = 0;
#define MJB_DS(name, rows, cols)                                                        \
	s.name = dev_alloc<double>((size_t)nenv * M->field_size[fidx]);                     \
	ok = ok && s.name;                                                                  \
	fidx++;
#define MJB_DD(name, rows, cols) fidx++;
#define MJB_DD2(name, rows, cols) fidx++;
#define MJB_DI(name, rows, cols) fidx++;
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	s.nwarn = dev_alloc<unsigned long long>(MJB_NWARNING);
	ok = ok && s.nwarn;
	s.stats = nullptr;
	s.prof = dev_alloc<unsigned long long>(64);
	ok = ok && s.prof;
	s.frame_ws = nullptr;
	s.frame_stride = b->L.ndouble + b->L.nint / 2;
	s.use_xfrc = 0;
	s.env_gravity = nullptr;
	s.env_geom_friction = nullptr;
	s.env_geom_size = nullptr;
	s.env_geom_type = nullptr;
	s.env_equality = nullptr;
	s.env_mass = nullptr;
	s.pgs_B = nullptr;
	s.efc_Jg = nullptr;
	s.efc_Jg_stride = 0;
	s.handoff = nullptr;
	s.handoff_stride = 0;
	s.reset_step = nullptr;
	s.sens_every_step = 0;
	for (int k = 0; k < 64; k++) s.colfunc[k] = MJB_COLFUNC_DEFAULT;
	s.sched = nullptr;
	if (h.nefcmax > 0) {  // constrained kernels: work queue of the chunked fused launches
		s.sched = dev_alloc<int>((size_t)nenv + 1);
		ok = ok && s.sched;
	}
	if (h.solver == MJB_SOL_NEWTON && h.nefcmax > 128) {  // kernel variant 4: row data of the env-steps beyond the fused frame's 64 rows
		// (stride: the kernels index the block with the cone-block stride of the frame THEY run on -- the wide frame of an all-condim-3 model has
		//  hcs 16 where the default frame has 10: size for the largest of the three layouts)
		s.efc_Jg_stride = mjb_rowblock_doubles(h.nefcmax, h.nv, h.nconmax, std::max(M->L.hcs, std::max(M->Lc.hcs, M->Lw.hcs)));
		s.efc_Jg = dev_alloc<double>((size_t)nenv * s.efc_Jg_stride);
		ok = ok && s.efc_Jg;
		if (M->has_wide) {  // row counters of the wide-frame policy (launch())
			b->rowstat_dev = dev_alloc<unsigned long long>(4);
			ok = ok && b->rowstat_dev && hipHostMalloc((void **)&b->rowstat_host, 4 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess &&
			     hipEventCreateWithFlags(&b->ev_rowstat, hipEventDisableTiming) == hipSuccess;
			if (ok) {
				memset(b->rowstat_host, 0, 4 * sizeof(unsigned long long));
				ok = hipMemset(b->rowstat_dev, 0, 4 * sizeof(unsigned long long)) == hipSuccess;
			}
		}
	}
	s.rowstat = nullptr;  // (set by launch() for the launches that count)
	if (h.solver == MJB_SOL_PGS && h.nv <= 16 && h.nefcmax > 64) {
		s.pgs_B = dev_alloc<double>((size_t)nenv * h.nefcmax * h.nv);  // (elliptic PGS never exceeds 64 rows: mjb_compile)
		ok = ok && s.pgs_B;
	}
	s.keep_frame = 0;
	s.prof_base = 0;
	if (!ok) {
		fail(MJB_ENOMEM, "mjb_make_batch: hipMalloc(state) failed for %d envs", nenv);
		mjb_free_batch(b);
		return nullptr;
	}
	if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
		fail(MJB_ENODEVICE, "mjb_make_batch: hipStreamCreate failed");
		mjb_free_batch(b);
		return nullptr;
	}
	b->own_stream = true;
	mjb_set_launch(b, 0, 0);
	if (mjb_reset(b, nullptr) != MJB_OK || mjb_synchronize(b) != MJB_OK) {
		mjb_free_batch(b);
		return nullptr;
	}
	g_err.clear();
	return b;
}

int mjb_nenv(const mjb_batch *b) { return b ? b->nenv : fail(MJB_EINVAL, "null batch"); }

int mjb_set_launch(mjb_batch *b, int lanes_per_env, int envs_per_block)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	const bool constrained = b->model->h.nefcmax > 0;
	if (lanes_per_env == 0) lanes_per_env = constrained ? 64 : 16;
	if (constrained && lanes_per_env != 64)
		return fail(MJB_EINVAL, "mjb_set_launch: models with constraint rows run one env per wavefront (lanes_per_env = 64)");
	if (lanes_per_env != 8 && lanes_per_env != 16 && lanes_per_env != 32 && lanes_per_env != 64)
		return fail(MJB_EINVAL, "mjb_set_launch: lanes_per_env must be 8, 16, 32 or 64");
	if (envs_per_block <= 0) envs_per_block = 64 / lanes_per_env;  // one wavefront per workgroup
	if (envs_per_block * lanes_per_env > 256) envs_per_block = 256 / lanes_per_env;
	b->lanes = lanes_per_env;
	b->epb = envs_per_block;
	return MJB_OK;
}

static int ensure_ws(mjb_batch *b)
{
	if (b->st.frame_ws) return MJB_OK;
	b->st.frame_ws = dev_alloc<double>((size_t)b->nenv * b->st.frame_stride);
	if (!b->st.frame_ws) return fail(MJB_ENOMEM, "frame workspace allocation failed");
	b->params_dirty = true;
	return MJB_OK;
}

// (re)upload the launch parameters when anything in them changed; ordered on the batch's stream
static int sync_params(mjb_batch *b)
{
	if (!b->params_dev) {
		HIP_TRY(hipMalloc((void **)&b->params_dev, sizeof(KernelParams)));
		b->params_dirty = true;
	}
	if (b->params_dirty) {
		KernelParams kp;
		kp.m = b->dm;
		kp.L = b->L;
		kp.Lc = b->wide ? b->model->Lw : b->model->Lc;
		kp.use_compact = (b->st.use_xfrc || b->st.keep_frame) ? 0 : (b->wide ? 2 : 1);  // (the value doubles as the layout's index into the host-built copy tables)
		kp.pad0 = 0;
		kp.s = b->st;
		kp.nz = b->nz;
		kp.hw = b->hw;
		// pageable source: the copy is staged before the call returns, so the local may go out of scope
		HIP_TRY(hipStreamSynchronize(b->stream));
		// (the fused launch of the non-callback envs, mjb_step_rest, runs on its own non-blocking stream and reads *params_dev lazily
		//  through scalar loads -- use_compact, use_xfrc, the layouts: an upload between mjb_step_rest and the second half, e.g. the
		//  first pushViews that carries a non-zero xfrc_applied, must not overtake it)
		if (b->rest_pending && b->rest_stream) HIP_TRY(hipStreamSynchronize(b->rest_stream));
		if (b->spec_valid && b->noise_stream) HIP_TRY(hipStreamSynchronize(b->noise_stream));  // (the side-stream noise generator reads nz)
		HIP_TRY(hipMemcpy(b->params_dev, &kp, sizeof kp, hipMemcpyHostToDevice));
		b->params_dirty = false;
	}
	return MJB_OK;
}

// whole-batch operations on the batch's stream wait for a fused launch of the non-callback envs that is still pending on the rest
// stream (a split step abandoned between mjb_step_rest and its second half -- an error, a reset raised inside a callback)
static int join_rest(mjb_batch *b)
{
	if (b->rest_pending) {
		HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_join, 0));
		b->rest_pending = false;
	}
	return MJB_OK;
}

// 0: no constraint rows; 1: PGS (5: PGS with elliptic contacts); 2 / 3 / 4: Newton with 1 / 2 / 4 rows per lane; 6 / 7 / 8: CG
static int kernel_variant(const mjb_model_desc &h)
{
	if (h.nefcmax <= 0) return 0;
	if (h.solver == MJB_SOL_PGS) return (h.cone == MJB_CONE_ELLIPTIC && h.nconmax > 0) ? 5 : 1;
	return (h.solver == MJB_SOL_CG ? 4 : 0) + (h.nefcmax <= 64 ? 2 : (h.nefcmax <= 128 ? 3 : 4));
}

static bool stream_capturing(hipStream_t st)
{
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	return cs != hipStreamCaptureStatusNone;
}

// ---- the split step: per step, the smooth half of every env of a slice in lane = env form (mjb_smooth_kernel.h), then the slice's
// constraint half one env per wavefront (mjb_cstep_kernel).  The smooth kernel is a latency (a few dozen wavefronts, ~20 us), the constraint kernel is
// the throughput: the batch is cut into slices that alternate the two on streams of their own, so one slice's smooth half runs beside the others'
// constraint halves.  A slice's launches are ordered by its stream; slices share nothing (no data-path exchange between envs).
static int split_slices(const mjb_batch *b, int nenv)
{
	static const int forced = [] { const char *v = getenv("MJB_SPLIT_SLICES"); return v ? atoi(v) : 0; }();  // measurement knob
	// (measured on config 3, MI355X, profiles/r06_split_step.txt: two slices are best from 32 768 envs up -- 33.2 M env-steps/s against the fused kernel's
	//  27.5 M; more slices mean shorter launches, and a launch ends with its slowest env)
	int n = forced > 0 ? forced : 2;
	const int waves = (nenv + 63) / 64;
	if (n > waves) n = waves;
	if (n > 16) n = 16;
	(void)b;
	return n < 1 ? 1 : n;
}

// the state one step after mj_resetData (mj_checkAcc's reset inside mjb_cstep_kernel), from a one-env batch on the generic kernels
static int split_reset_state(mjb_batch *b)
{
	const mjb_model_desc &h = b->model->h;
	mjb_batch *t = mjb_make_batch(b->model, 1, b->device);
	if (!t) return MJB_ENOMEM;
	t->split_mode = 0;
	std::vector<double> buf((size_t)h.nq + 2 * h.nv);
	int rc = mjb_reset(t, nullptr);
	if (rc == MJB_OK) rc = mjb_step(t, 1);
	if (rc == MJB_OK) rc = mjb_get(t, MJB_F_qpos, 0, 1, buf.data());
	if (rc == MJB_OK) rc = mjb_get(t, MJB_F_qvel, 0, 1, buf.data() + h.nq);
	if (rc == MJB_OK) rc = mjb_get(t, MJB_F_qacc_warmstart, 0, 1, buf.data() + h.nq + h.nv);
	mjb_free_batch(t);
	if (rc != MJB_OK) return rc;
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipMalloc((void **)&b->reset_step_dev, buf.size() * sizeof(double)));
	HIP_TRY(hipMemcpy(b->reset_step_dev, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice));
	return MJB_OK;
}

static int split_prepare(mjb_batch *b)
{
	const mjb_model_desc &h = b->model->h;
	if (!b->reset_step_dev) {
		int rc = split_reset_state(b);
		if (rc) return rc;
		b->st.reset_step = b->reset_step_dev;
		b->params_dirty = true;
	}
	if (!b->handoff_dev) {
		const HandoffLayout hl = mjb_handoff_layout(h.ngeom, h.nv, h.nbody, h.nM);
		b->handoff_dev = dev_alloc<double>((size_t)b->nenv * hl.stride);
		if (!b->handoff_dev) return fail(MJB_ENOMEM, "split step: hand-off records");
		HIP_TRY(hipMemset(b->handoff_dev, 0, (size_t)b->nenv * hl.stride * sizeof(double)));
		b->st.handoff = b->handoff_dev;
		b->st.handoff_stride = hl.stride;
		b->params_dirty = true;
	}
	if (!b->ev_split_fork) HIP_TRY(hipEventCreateWithFlags(&b->ev_split_fork, hipEventDisableTiming));
	return MJB_OK;
}

static int launch_split(mjb_batch *b, int nsteps, int env_lo, int env_hi, hipStream_t stream)
{
	const int nenv = env_hi - env_lo;
	const int ns = split_slices(b, nenv);
	while ((int)b->split_streams.size() < ns) {
		hipStream_t st = nullptr;
		hipEvent_t ev = nullptr;
		HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
		b->split_streams.push_back(st);
		HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		b->split_events.push_back(ev);
	}
	const FrameLayout &Lc = b->model->Lc;
	const int cflags = b->st.stats ? 1 : 0;
	// slices of whole wavefronts of the smooth kernel
	const int waves = (nenv + 63) / 64;
	HIP_TRY(hipEventRecord(b->ev_split_fork, stream));
	for (int k = 0; k < ns; k++) {
		const int w0 = (int)((long long)waves * k / ns), w1 = (int)((long long)waves * (k + 1) / ns);
		const int lo = env_lo + 64 * w0, hi = std::min(env_hi, env_lo + 64 * w1);
		if (hi <= lo) continue;
		hipStream_t st = ns == 1 ? stream : b->split_streams[k];
		if (ns > 1) HIP_TRY(hipStreamWaitEvent(st, b->ev_split_fork, 0));
		for (int i = 0; i < nsteps; i++) {
			int rc = mjb_launch_smooth(b->params_dev, b->model->sm_topo, lo, hi, b->step_counter + (unsigned int)i, i == nsteps - 1 ? 1 : 0, st);
			if (rc == 0) rc = mjb_launch_cstep(b->params_dev, Lc, lo, hi, b->epb, cflags, st);
			if (rc != 0) return fail(MJB_ENODEVICE, "split step launch failed: %s", hipGetErrorString((hipError_t)rc));
		}
		if (ns > 1) {
			HIP_TRY(hipEventRecord(b->split_events[k], st));
			HIP_TRY(hipStreamWaitEvent(stream, b->split_events[k], 0));
		}
	}
	b->split_slices_last = ns;
	return MJB_OK;
}

static int launch(mjb_batch *b, int mode, int nsteps, int env_lo = 0, int env_hi = -1, hipStream_t on = nullptr)
{
	if (env_hi < 0) env_hi = b->nenv;
	hipStream_t stream = on ? on : b->stream;
	const bool whole = env_lo == 0 && env_hi == b->nenv;
	HIP_TRY(hipSetDevice(b->device));
	const bool compact = mode == MJB_MODE_STEP && !b->st.use_xfrc && !b->st.keep_frame;
	int variant = kernel_variant(b->model->h);
	// Which fused frame (kernel variant 4 with a wide frame): a long whole-batch launch looks at the row counters of the previous one.
	// More than a quarter of the env-steps beyond 64 rows: the wide frame (128 rows in LDS, two envs per CU) -- measured on the
	// power-grasp workload of config 5 (94 % beyond 64 rows): 0.72 -> 2.41 M env-steps/s; fewer than 5 %: back to the default frame
	// (four envs per CU: the light workload runs 7.8 M on it, 4.2 M on the wide one).  MJB_WIDE_FRAME=0 / 1 pins the choice.
	bool count_rows = false;
	if (variant == 4 && b->model->has_wide && b->rowstat_dev) {
		static const int pinned = [] { const char *v = getenv("MJB_WIDE_FRAME"); return v ? atoi(v) : -1; }();
		bool want = b->wide;
		if (pinned >= 0) want = pinned != 0;
		else if (compact && whole && (nsteps >= 100 || b->probe_now || b->decide_now) && !stream_capturing(stream)) {  // (under stream capture nothing can be waited for: the frame stays)
			if (b->rowstat_pending) {
				// (the newest COMPLETED sample: a launch behind a running counting launch does not wait for it -- back-to-back launches stay
				//  enqueued ahead of the device -- except right behind mjb_step's probe launch, whose whole point is this decision)
				// Default: WAIT for the previous counting launch's sample (it has ended or is about to) -- the choice is then a function of the
				// sequence of launches alone, and a rollout reproduces bit for bit.  MJB_WIDE_POLICY_NOWAIT=1: take it only if it has landed
				// (back-to-back launches stay enqueued ahead of the device; which launch switches frames then depends on timing, results agree to
				// rounding).  Right behind mjb_step's probe launch the wait is the point.
				static const bool nowait = [] { const char *v = getenv("MJB_WIDE_POLICY_NOWAIT"); return v && *v == '1'; }();
				const hipError_t q = (b->decide_now || !nowait) ? hipEventSynchronize(b->ev_rowstat) : hipEventQuery(b->ev_rowstat);
				if (q == hipSuccess) {
					b->rowstat_pending = false;
					b->rowstat_ever = true;
					const double tot = (double)b->rowstat_host[0], gt64 = (double)b->rowstat_host[1];
					if (tot > 0) {
						if (!b->wide && gt64 > 0.25 * tot) want = true;
						else if (b->wide && gt64 < 0.05 * tot) want = false;
					}
				} else if (q != hipErrorNotReady)
					return fail(MJB_ENODEVICE, "hipEventQuery(row counters): %s", hipGetErrorString(q));
				else
					(void)hipGetLastError();
			}
			count_rows = !b->rowstat_pending;  // (one sample in flight at a time: the pinned words belong to it)
		}
		if (want != b->wide) {
			b->wide = want;
			b->params_dirty = true;
		}
	}
	// The split step for plain-PGS models whose smooth stages have a lane = env kernel (config 3): fused launches of the whole batch (or, forced, any range)
	bool use_split = false;
	if (mode == MJB_MODE_STEP && compact && variant == 1 && b->model->sm_topo >= 0 && b->split_mode != 0 && b->hw.n == 0 && !b->env_mass && !b->env_gravity &&
	    !b->env_equality && 8 * mjb_frame_bytes(b->model, 1) <= mjb_max_lds_bytes() && !stream_capturing(stream)) {
		// (automatic mode: OFF unless MJB_SPLIT_MIN_ENVS names a batch size.  Measured on config 3 at 32 768 envs, profiles/r06_split_step.txt: +21 % in the
		//  rollout's light-contact phase, +12 % over steps 1000 - 4000, -5 % over steps 500 - 2000 -- a launch pair per step ends with the slice's slowest env,
		//  and the heavy phase has the long PGS tails.  Not a rule to apply behind the user's back.)
		static const int min_envs = [] { const char *v = getenv("MJB_SPLIT_MIN_ENVS"); return v ? atoi(v) : 0; }();
		use_split = b->split_mode == 1 || (min_envs > 0 && whole && b->nenv >= min_envs);
		if (use_split) {
			int src = split_prepare(b);
			if (src) return src;
		}
	}
	b->split_used = use_split;
	{  // the kernels count rows (three atomics per env-step on one address) only in the launches whose counters are read
		unsigned long long *rs = count_rows ? b->rowstat_dev : nullptr;
		if (b->st.rowstat != rs) {
			b->st.rowstat = rs;
			b->params_dirty = true;
		}
	}
	int prc = sync_params(b);
	if (prc) return prc;
	if (use_split) {
		if (b->zvalid) {  // (the pre-generated ctrl-noise buffer names an older launch: this path draws its normals in the smooth kernel)
			HIP_TRY(hipMemsetAsync(b->zinfo, 0xff, 8 * sizeof(unsigned int), stream));
			b->zvalid = false;
		}
		b->lane_env_used = false;
		b->noise_mode = 0;
		return launch_split(b, nsteps, env_lo, env_hi, stream);
	}
	const int fused_id = b->wide ? 2 : 1;
	// plain PGS on the lean frame: when eight envs fit one CU's LDS, the 256-register build runs two waves per SIMD
	if (variant == 1 && compact && 8 * mjb_frame_bytes(b->model, 1) <= mjb_max_lds_bytes()) variant = 9;
	{
		static const int forced = [] { const char *v = getenv("MJB_DEBUG_VARIANT"); return v ? atoi(v) : -1; }();  // measurement knob
		if ((forced == 1 || forced == 9) && (variant == 1 || variant == 9)) variant = forced;  // (only the two builds of the same PGS step are interchangeable)
		if ((mode == MJB_MODE_STEP21 || mode == MJB_MODE_RKMID || mode == MJB_MODE_RKLAST) && variant == 9) variant = 1;  // (the 256-register build carries no chained-step / cut-RK4 mode)
	}
	// long fused launches of the constrained kernels hand out (chunk of steps, env) work items dynamically (mjb_step.hip)
	// ... unless every env gets a slot of its own (envs <= resident envs per CU x CUs): then the queue has nothing to even out, an
	// item taken by a wave whose env's previous chunk is still running elsewhere only waits, and the launch is fastest with each
	// env's steps run back to back on its slot.  Measured on config 5 (4 envs per CU = 1024 slots, MI355X), chunked / unchunked:
	// 1024 envs x 100 steps 4.04 / 4.72 M env-steps/s, x 1000 steps 4.20 / 5.00 M, 768 envs 3.40 / 3.75 M; 1280 envs 5.32 / 3.38 M.
	bool own_slot = false;
	if (compact && variant != 0) {
		int ncu = 0;  // (of THIS batch's device: a process may drive several)
		if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, b->device) != hipSuccess) ncu = 0;
		const int granules = (mjb_frame_bytes(b->model, fused_id) + 1279) / 1280;  // (gfx950: 128 LDS granules of 1280 bytes per CU)
		const int occ = std::min(variant == 9 ? 8 : 4, 128 / std::max(1, granules));  // (512-register kernels: one wave per SIMD)
		own_slot = ncu > 0 && b->nenv <= occ * ncu;
	}
	int chunk = 0;
	static const int forced_chunk = [] { const char *v = getenv("MJB_DEBUG_CHUNK"); return v ? atoi(v) : 0; }();  // measurement knob (also overrides own_slot)
	if (mode == MJB_MODE_STEP && variant != 0 && nsteps >= 100 && b->st.sched && whole && (!own_slot || forced_chunk > 0)) {
		static const bool off = [] { const char *v = getenv("MJB_DEBUG_NO_CHUNKS"); return v && *v && *v != '0'; }();  // measurement knob
		if (!off) {
			const int forced = forced_chunk;
			// (measured on MI355X: 10 - 40 steps per item are equally good on config 3; config 5 -- three envs per CU, a few heavy envs
			//  on the critical path -- likes them finer: 3 - 5 steps per item 3.85 M, 10 steps 3.79 M, 20 steps 3.63 M)
			const bool newton = variant >= 2 && variant <= 4;
			chunk = forced > 0 ? forced : (newton ? std::max(5, (nsteps + 19) / 20) : std::max(10, (nsteps + 15) / 16));
			chunk = std::min(chunk, 65535);
			HIP_TRY(hipMemsetAsync(b->st.sched, 0, ((size_t)b->nenv + 1) * sizeof(int), stream));
		}
	}
	// ctrl noise of a long fused launch: its normals are generated by a throughput kernel (mjb_noise_kernel) instead of inside every
	// step's dependent chain, when they fit the budget (MJB_NOISE_PREGEN_MB = the TOTAL allocation, default 2048; 0: always inside
	// the step kernel -- same values either way).  Unconstrained kernels (one wave per SIMD, 320 VGPRs: a second kernel finds room) get
	// a buffer of two halves: this launch reads one while the other is filled, on a side stream, for the launch expected next (same
	// length, the step counter this one ends at); a launch that finds its normals ready only waits for that event, anything else
	// generates them on its own stream first.  The constrained kernels fill the register file (256 / 512 VGPRs at two / one waves per
	// SIMD): a side-stream generator only finds a slot when the step launch drains (rocprofv3 showed it spanning the whole 130 - 180 ms
	// launch), so they -- and launches whose two halves exceed the budget -- generate one half on their own stream.
	bool zuse = false;
	int zhalf_now = 0;
	b->noise_mode = 0;
	if (mode == MJB_MODE_STEP && whole && b->nz.enabled && nsteps >= 16 && b->model->h.nu > 0 && b->model->h.nu <= b->lanes) {
		static const size_t cap_doubles = [] { const char *v = getenv("MJB_NOISE_PREGEN_MB"); return (size_t)(v ? atol(v) : 2048) * (1u << 20) / sizeof(double); }();
		const size_t need = (size_t)nsteps * b->nenv * b->model->h.nu;
		const bool want_double = variant == 0 && 2 * need <= cap_doubles;
		const size_t total = want_double ? 2 * need : need;
		if (total <= cap_doubles && total < b->zfail) {
			if (need > b->zcap || (want_double && !b->zdouble)) {
				HIP_TRY(hipStreamSynchronize(stream));
				if (b->noise_stream) HIP_TRY(hipStreamSynchronize(b->noise_stream));
				if (b->zbuf) hipFree(b->zbuf);
				b->spec_valid = false;
				b->zbuf = dev_alloc<double>(total);
				if (!b->zbuf) {
					(void)hipGetLastError();
					b->zfail = total;  // (remembered: a launch of this size or larger generates in the kernel from now on)
				}
				b->zcap = b->zbuf ? need : 0;
				b->zdouble = b->zbuf && want_double;
				if (!b->zinfo) b->zinfo = dev_alloc<unsigned int>(8);
				if (b->zinfo) HIP_TRY(hipMemset(b->zinfo, 0xff, 8 * sizeof(unsigned int)));
				b->zvalid = false;
				if (b->zdouble && !b->noise_stream) {
					HIP_TRY(hipStreamCreateWithFlags(&b->noise_stream, hipStreamNonBlocking));
					HIP_TRY(hipEventCreateWithFlags(&b->ev_noise, hipEventDisableTiming));
					HIP_TRY(hipEventCreateWithFlags(&b->ev_noise_go, hipEventDisableTiming));
				}
				b->st.zbuf = b->zbuf;
				b->st.zinfo = b->zbuf ? b->zinfo : nullptr;
				b->st.zhalf = (unsigned long long)b->zcap;
				b->params_dirty = true;
				prc = sync_params(b);
				if (prc) return prc;
			}
			if (b->zbuf && b->zinfo) {
				if (b->spec_valid && b->spec_step0 == b->step_counter && b->spec_nsteps == nsteps) {
					zhalf_now = b->spec_half;
					HIP_TRY(hipStreamWaitEvent(stream, b->ev_noise, 0));  // generated while the previous launch ran
					b->noise_mode = 2;
				} else {
					if (b->spec_valid) HIP_TRY(hipStreamWaitEvent(stream, b->ev_noise, 0));  // (a generator still writing its half: let it finish first)
					zhalf_now = 0;
					HIP_TRY(hipMemsetAsync(b->zinfo + 4, 0xff, 4 * sizeof(unsigned int), stream));
					int nrc = mjb_launch_noise(b->params_dev, b->zbuf, b->zinfo, b->nenv, b->model->h.nu, nsteps, b->step_counter, stream);
					if (nrc != 0) return fail(MJB_ENODEVICE, "noise kernel launch failed: %s", hipGetErrorString((hipError_t)nrc));
					b->noise_mode = 1;
				}
				b->spec_valid = false;
				zuse = b->zvalid = true;
			}
		}
	}
	if (!zuse && b->zvalid) {  // (a launch that generates its normals itself must not match a record left by an earlier one)
		if (b->spec_valid) HIP_TRY(hipStreamWaitEvent(stream, b->ev_noise, 0));
		b->spec_valid = false;
		HIP_TRY(hipMemsetAsync(b->zinfo, 0xff, 8 * sizeof(unsigned int), stream));
		b->zvalid = false;
	}
	const bool zspec = zuse && b->zdouble;
	if (zspec) HIP_TRY(hipEventRecord(b->ev_noise_go, stream));  // (everything before this launch -- the last reader of the other half -- is done)
	if (count_rows) HIP_TRY(hipMemsetAsync(b->rowstat_dev, 0, 4 * sizeof(unsigned long long), stream));
	// The lane = env kernel (mjb_lane_env.hip) for fused launches of a model whose topology is compiled in: one env per lane, no frame.
	// Per-env model overrides, the device hwsim stage, xfrc_applied and frame dumps keep the generic kernels.
	bool use_le = false;
	if (mode == MJB_MODE_STEP && compact && variant == 0 && b->model->le_topo != MJB_LE_TOPO_NONE && !b->le_unavailable && b->lane_env_mode != 0 && b->hw.n == 0 && !b->env_mass &&
	    !b->env_gravity && !b->st.stats) {
		static const int min_envs = [] { const char *v = getenv("MJB_LANE_ENV_MIN_ENVS"); return v ? atoi(v) : 4096; }();
		use_le = b->lane_env_mode == 1 || env_hi - env_lo >= min_envs;  // (also the fused launch of a split step's non-callback envs, mjb_step_rest)
	}
	b->lane_env_used = use_le;
	int rc;
	if (use_le) {
		rc = mjb_launch_lane_env(b->params_dev, b->model->le_topo, &b->model->h, b->nenv, env_lo, env_hi, nsteps, b->step_counter, stream);
		if (rc == MJB_LE_UNAVAILABLE) {  // (no hiprtc / no kernel header / compile error: remembered, the generic kernel runs -- mjb_lane_env_info says why)
			b->le_unavailable = true;
			b->lane_env_used = use_le = false;
		}
	}
	if (!use_le)
	rc = mjb_launch_step(b->params_dev, compact ? (b->wide ? b->model->Lw : b->model->Lc) : b->L, env_lo, env_hi, mode, nsteps, b->step_counter, b->lanes,
	                         b->epb, variant | (chunk << 8), (b->lanes == 16 && !b->env_mass && mode != MJB_MODE_STEP21 && b->model->h.integrator == MJB_INT_EULER && b->model->h.nefcmax <= 0 && b->model->h.nv <= 16 && b->model->h.nbody <= 16 && b->model->h.nu <= 16 && b->model->h.njnt <= 16) ? (b->model->h.nv <= 8 ? 8 : (b->model->h.nv <= 12 ? 12 : 16)) : 0, stream);
	if (rc != 0) return fail(MJB_ENODEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
	if (count_rows) {  // this launch's row counters, to pinned host memory behind it
		HIP_TRY(hipMemcpyAsync(b->rowstat_host, b->rowstat_dev, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipEventRecord(b->ev_rowstat, stream));
		b->rowstat_pending = true;
	}
	if (zspec) {  // the next launch's normals, into the half this one does not read, beside it
		const int other = 1 - zhalf_now;
		HIP_TRY(hipStreamWaitEvent(b->noise_stream, b->ev_noise_go, 0));
		int nrc = mjb_launch_noise(b->params_dev, b->zbuf + (size_t)other * b->zcap, b->zinfo + 4 * other, b->nenv, b->model->h.nu, nsteps,
		                           b->step_counter + (unsigned int)nsteps, b->noise_stream);
		if (nrc == 0) {
			HIP_TRY(hipEventRecord(b->ev_noise, b->noise_stream));
			b->spec_valid = true;
			b->spec_half = other;
			b->spec_step0 = b->step_counter + (unsigned int)nsteps;
			b->spec_nsteps = nsteps;
		}
	}
	return MJB_OK;
}

int mjb_step(mjb_batch *b, int nsteps)
{
	if (!b || nsteps < 0) return fail(MJB_EINVAL, "mjb_step: bad argument");
	if (nsteps == 0) return MJB_OK;
	int jrc = join_rest(b);
	if (jrc) return jrc;
	b->split_ncb = -1;  // (an open split step is abandoned)
	// Kernel variant 4 with a wide fused frame: a long launch on a workload nobody has looked at yet (a new batch, after mjb_reset or a new
	// qpos) starts with a short PROBE launch that counts this state's rows, and picks its frame from that -- the first rollout of a
	// high-contact batch used to run on the default frame (r05: 1461 ms against 288 ms in steady state).  A launch cut in two is the same
	// rollout (the state crosses HBM between launches exactly as between the chunks of one).
	int rc = MJB_OK;
	if (nsteps >= 100 && b->rowstat_dev && b->model->has_wide && !b->rowstat_ever && !b->rowstat_pending && !b->st.use_xfrc && !b->st.keep_frame) {
		const int probe = 4;
		b->probe_now = true;
		rc = launch(b, MJB_MODE_STEP, probe);
		b->probe_now = false;
		if (rc != MJB_OK) return rc;
		b->step_counter += (unsigned int)probe;
		b->steps_taken += (unsigned long long)probe;
		nsteps -= probe;
		b->decide_now = b->rowstat_pending;
	}
	rc = launch(b, MJB_MODE_STEP, nsteps);
	b->decide_now = false;
	if (rc == MJB_OK) {
		b->step_counter += (unsigned int)nsteps;
		b->steps_taken += (unsigned long long)nsteps;
		b->frame_valid = b->st.keep_frame != 0;
		b->frame_hi = b->nenv;
	}
	return rc;
}

int mjb_set_keep_frame(mjb_batch *b, int on)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	if (on) {
		int rc = ensure_ws(b);
		if (rc) return rc;
	}
	if ((b->st.keep_frame != 0) != (on != 0)) {
		b->st.keep_frame = on ? 1 : 0;
		b->params_dirty = true;
		b->frame_valid = false;
	}
	return MJB_OK;
}

int mjb_step1(mjb_batch *b)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	int rc = ensure_ws(b);
	if (rc) return rc;
	rc = join_rest(b);
	if (rc) return rc;
	rc = launch(b, MJB_MODE_STEP1, 1);
	if (rc == MJB_OK) {
		b->frame_valid = true;
		b->frame_hi = b->nenv;
	}
	return rc;
}

int mjb_step2(mjb_batch *b)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	if (!b->frame_valid || !b->st.frame_ws || b->frame_hi < b->nenv) return fail(MJB_EINVAL, "mjb_step2 without a preceding mjb_step1");
	int rc = launch(b, MJB_MODE_STEP2, 1);
	if (rc == MJB_OK) {
		b->step_counter += 1;
		b->steps_taken += 1;
	}
	return rc;
}

// ---- split step for a PREFIX of the batch (the host runtime's callback envs): envs [0, ncb) are stepped in two halves around
// the control-callback point, envs [ncb, nenv) take the SAME step as one fused launch (same Philox step counter, same results as
// a whole-batch launch: every env's arithmetic is independent of which launch carries it).  Order of calls for one step:
//   mjb_step1_prefix(ncb) -> [copy the callback envs' fields out] -> mjb_step_rest(ncb) -> [host callbacks run while it executes]
//   -> [copy their writes back] -> mjb_step2_prefix(ncb)   (advances the step counter)
int mjb_step1_prefix(mjb_batch *b, int ncb)
{
	if (!b || ncb < 0 || ncb > b->nenv) return fail(MJB_EINVAL, "mjb_step1_prefix: bad argument");
	// A split step abandoned AFTER mjb_step_rest: the rest envs already took a step the step counter never recorded (they would redraw
	// the same Philox step), and their launch may still be running on the rest stream.  Only mjb_step2_prefix / mjb_step21_prefix /
	// mjb_step2_rk_prefix (finish it) or mjb_reset / mjb_step (abandon it, whole batch) may follow; a split abandoned BEFORE
	// mjb_step_rest is simply restarted here.
	if (b->split_ncb >= 0 && b->split_rest_done && b->split_ncb < b->nenv)
		return fail(MJB_EINVAL, "mjb_step1_prefix: a split step of prefix %d is still open and its other envs already took the step "
		                        "(finish it with mjb_step2_prefix, or abandon it with mjb_reset / mjb_step)", b->split_ncb);
	int rc = join_rest(b);
	if (rc) return rc;
	rc = ensure_ws(b);
	if (rc) return rc;
	b->split_ncb = ncb;
	b->split_rest_done = false;
	if (ncb == 0) return MJB_OK;
	rc = launch(b, MJB_MODE_STEP1, 1, 0, ncb);
	if (rc == MJB_OK) {
		b->frame_valid = true;
		b->frame_hi = ncb;
	}
	return rc;
}

int mjb_step_rest(mjb_batch *b, int ncb)
{
	if (!b || ncb != b->split_ncb) return fail(MJB_EINVAL, "mjb_step_rest: not inside a split step of this prefix (mjb_step1_prefix first)");
	if (b->split_rest_done) return fail(MJB_EINVAL, "mjb_step_rest: called twice in one split step");
	b->split_rest_done = true;
	if (ncb >= b->nenv) return MJB_OK;
	// on its own stream, forked from the batch's stream here and joined in mjb_step2_prefix: the two groups of envs are independent,
	// so the callback envs' second half does not queue behind this launch
	HIP_TRY(hipSetDevice(b->device));
	if (!b->rest_stream) {
		HIP_TRY(hipStreamCreateWithFlags(&b->rest_stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
	}
	int prc = sync_params(b);  // (uploads, if any, on the batch's stream BEFORE the fork)
	if (prc) return prc;
	HIP_TRY(hipEventRecord(b->ev_fork, b->stream));
	HIP_TRY(hipStreamWaitEvent(b->rest_stream, b->ev_fork, 0));
	int rc = launch(b, MJB_MODE_STEP, 1, ncb, b->nenv, b->rest_stream);
	HIP_TRY(hipEventRecord(b->ev_join, b->rest_stream));
	b->rest_pending = true;
	return rc;
}

int mjb_step2_prefix(mjb_batch *b, int ncb)
{
	if (!b || ncb != b->split_ncb) return fail(MJB_EINVAL, "mjb_step2_prefix without the matching mjb_step1_prefix");
	int rc = MJB_OK;
	if (!b->split_rest_done) rc = mjb_step_rest(b, ncb);  // (a caller that skipped it: the rest must not miss the step)
	if (rc != MJB_OK) return rc;
	if (ncb > 0) {
		if (!b->frame_valid || !b->st.frame_ws || b->frame_hi < ncb) return fail(MJB_EINVAL, "mjb_step2_prefix without a preceding mjb_step1_prefix");
		rc = launch(b, MJB_MODE_STEP2, 1, 0, ncb);
	}
	if (b->rest_pending) {  // join: whatever follows on the batch's stream sees the rest's step too
		HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_join, 0));
		b->rest_pending = false;
	}
	if (rc == MJB_OK) {
		b->step_counter += 1;
		b->steps_taken += 1;
		b->split_ncb = -1;
	}
	return rc;
}

int mjb_step21_prefix(mjb_batch *b, int ncb)
{
	if (!b || ncb != b->split_ncb) return fail(MJB_EINVAL, "mjb_step21_prefix without the matching mjb_step1_prefix");
	int rc = MJB_OK;
	if (!b->split_rest_done) rc = mjb_step_rest(b, ncb);
	if (rc != MJB_OK) return rc;
	if (ncb > 0) {
		if (!b->frame_valid || !b->st.frame_ws || b->frame_hi < ncb) return fail(MJB_EINVAL, "mjb_step21_prefix without a preceding mjb_step1_prefix");
		// (trip 1 of the launch -- the next step's first half -- draws its ctrl noise at step_counter + 1, as mjb_step1_prefix would
		//  after mjb_step2_prefix advanced the counter)
		rc = launch(b, MJB_MODE_STEP21, 1, 0, ncb);
	}
	if (b->rest_pending) {  // join: whatever follows on the batch's stream sees the rest's step too
		HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_join, 0));
		b->rest_pending = false;
	}
	if (rc == MJB_OK) {
		b->step_counter += 1;
		b->steps_taken += 1;
		b->split_ncb = ncb;  // ... and the next split step is open: its first half has run
		b->split_rest_done = false;
		b->frame_valid = true;
		b->frame_hi = ncb;
	}
	return rc;
}

// ---- the second half of an RK4 step cut at the callback points of its four evaluations (mj_RungeKutta runs mj_forwardSkip, and with
// it mjcb_passive / mjcb_control, once per evaluation: plugin_utils.h:119-125 is why lastStageCallback exists).  After
// mjb_step1_prefix (evaluation 0's first half) and the host's callbacks:  rk = 0, 1, 2 finishes evaluation rk, folds it into the
// weighted sums, sets the state of evaluation rk + 1 and runs ITS first half -- the host's callbacks then see that evaluation's view
// (time = t0 + c h); rk = 3 finishes evaluation 3, advances the state and closes the step like mjb_step2_prefix.  The envs beyond
// the prefix take the whole RK4 step as one fused launch (mjb_step_rest, issued with rk = 0 if the caller did not).
int mjb_step2_rk_prefix(mjb_batch *b, int ncb, int rk)
{
	if (!b || ncb != b->split_ncb || rk < 0 || rk > 3) return fail(MJB_EINVAL, "mjb_step2_rk_prefix without the matching mjb_step1_prefix, or evaluation index outside 0..3");
	if (b->model->h.integrator != MJB_INT_RK4) return fail(MJB_EINVAL, "mjb_step2_rk_prefix: the model's integrator is not RK4");
	int rc = MJB_OK;
	if (!b->split_rest_done) rc = mjb_step_rest(b, ncb);
	if (rc != MJB_OK) return rc;
	if (ncb > 0) {
		if (!b->frame_valid || !b->st.frame_ws || b->frame_hi < ncb) return fail(MJB_EINVAL, "mjb_step2_rk_prefix without a preceding mjb_step1_prefix");
		rc = launch(b, rk == 3 ? MJB_MODE_RKLAST : MJB_MODE_RKMID, rk, 0, ncb);  // (the evaluation index travels as the kernel's nsteps)
	}
	if (rc != MJB_OK || rk < 3) return rc;
	rc = join_rest(b);
	if (rc == MJB_OK) {
		b->step_counter += 1;
		b->steps_taken += 1;
		b->split_ncb = -1;
	}
	return rc;
}

int mjb_forward(mjb_batch *b)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	int rc = ensure_ws(b);
	if (rc) return rc;
	rc = join_rest(b);
	if (rc) return rc;
	rc = launch(b, MJB_MODE_FORWARD, 1);
	if (rc == MJB_OK) {
		b->frame_valid = true;
		b->frame_hi = b->nenv;
	}
	return rc;
}

int mjb_reset(mjb_batch *b, const uint8_t *mask)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	HIP_TRY(hipSetDevice(b->device));
	const unsigned char *md = nullptr;
	if (mask) {
		if (!b->mask_dev) HIP_TRY(hipMalloc((void **)&b->mask_dev, (size_t)b->nenv));
		HIP_TRY(hipMemcpyAsync(b->mask_dev, mask, (size_t)b->nenv, hipMemcpyHostToDevice, b->stream));
		md = b->mask_dev;
	}
	int prc = join_rest(b);
	if (prc) return prc;
	b->split_ncb = -1;  // (an open split step is abandoned: mjb_step1_prefix starts the next one)
	prc = sync_params(b);
	if (prc) return prc;
	int rc = mjb_launch_reset(b->params_dev, b->nenv, md, b->stream);
	if (rc != 0) return fail(MJB_ENODEVICE, "reset launch failed: %s", hipGetErrorString((hipError_t)rc));
	if (mask) HIP_TRY(hipStreamSynchronize(b->stream));
	b->frame_valid = false;
	if (!mask) b->rowstat_ever = false;  // (a new workload: the next long launch probes its rows first)
	return MJB_OK;
}

static double *state_ptr(mjb_batch *b, int field)
{
	switch (field) {
#define MJB_DS(name, rows, cols) case MJB_F_##name: return b->st.name;
#define MJB_DD(name, rows, cols)
#define MJB_DD2(name, rows, cols)
#define MJB_DI(name, rows, cols)
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	default: return nullptr;
	}
}

static int frame_offset(const mjb_batch *b, int field)
{
	const int *slots[] = {
#define MJB_DS(name, rows, cols) &b->L.name,
#define MJB_DD(name, rows, cols) &b->L.name,
#define MJB_DD2(name, rows, cols) &b->L.name,
#define MJB_DI(name, rows, cols) &b->L.name,
#include "../../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	};
	return *slots[field];
}

static int check_range(mjb_batch *b, int field, int lo, int hi)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	if (field < 0 || field >= MJB_F_COUNT) return fail(MJB_EINVAL, "bad field id %d", field);
	if (lo < 0 || hi > b->nenv || lo > hi) return fail(MJB_ERANGE, "env range [%d,%d) outside [0,%d)", lo, hi, b->nenv);
	return MJB_OK;
}

int mjb_get(mjb_batch *b, int field, int env_lo, int env_hi, double *host)
{
	int rc = check_range(b, field, env_lo, env_hi);
	if (rc) return rc;
	if (kFields[field].kind == 3) return fail(MJB_EINVAL, "field %s is an int field; use mjb_get_int", kFields[field].name);
	const int n = b->model->field_size[field];
	if (n == 0 || env_lo == env_hi) return MJB_OK;
	if (!host) return fail(MJB_EINVAL, "null host buffer");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (kFields[field].kind == 0) {
		double *p = state_ptr(b, field);
		HIP_TRY(hipMemcpy(host, p + (size_t)env_lo * n, (size_t)(env_hi - env_lo) * n * sizeof(double),
		                  hipMemcpyDeviceToHost));
		return MJB_OK;
	}
	if (!b->frame_valid || !b->st.frame_ws || env_hi > b->frame_hi)
		return fail(MJB_EINVAL, "derived field %s is only readable after mjb_forward / mjb_step1 / mjb_step2",
		            kFields[field].name);
	const int off = frame_offset(b, field);
	HIP_TRY(hipMemcpy2D(host, (size_t)n * sizeof(double), b->st.frame_ws + (size_t)env_lo * b->st.frame_stride + off,
	                    (size_t)b->st.frame_stride * sizeof(double), (size_t)n * sizeof(double),
	                    (size_t)(env_hi - env_lo), hipMemcpyDeviceToHost));
	return MJB_OK;
}

int mjb_get_int(mjb_batch *b, int field, int env_lo, int env_hi, int *host)
{
	int rc = check_range(b, field, env_lo, env_hi);
	if (rc) return rc;
	if (kFields[field].kind != 3) return fail(MJB_EINVAL, "field %s is not an int field", kFields[field].name);
	const int n = b->model->field_size[field];
	if (n == 0 || env_lo == env_hi) return MJB_OK;
	if (!host) return fail(MJB_EINVAL, "null host buffer");
	if (!b->frame_valid || !b->st.frame_ws || env_hi > b->frame_hi)
		return fail(MJB_EINVAL, "int field %s is only readable after mjb_forward / mjb_step1 / mjb_step2",
		            kFields[field].name);
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	const int off = frame_offset(b, field);
	const int *base = reinterpret_cast<const int *>(b->st.frame_ws + (size_t)env_lo * b->st.frame_stride + b->L.ndouble) + off;
	HIP_TRY(hipMemcpy2D(host, (size_t)n * sizeof(int), base, (size_t)b->st.frame_stride * sizeof(double),
	                    (size_t)n * sizeof(int), (size_t)(env_hi - env_lo), hipMemcpyDeviceToHost));
	return MJB_OK;
}

int mjb_set(mjb_batch *b, int field, int env_lo, int env_hi, const double *host)
{
	int rc = check_range(b, field, env_lo, env_hi);
	if (rc) return rc;
	const int n = b->model->field_size[field];
	if (n == 0 || env_lo == env_hi) return MJB_OK;
	if (!host) return fail(MJB_EINVAL, "null host buffer");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (kFields[field].kind == 0) {
		double *p = state_ptr(b, field);
		HIP_TRY(hipMemcpy(p + (size_t)env_lo * n, host, (size_t)(env_hi - env_lo) * n * sizeof(double),
		                  hipMemcpyHostToDevice));
		if (field == MJB_F_xfrc_applied && !b->st.use_xfrc) {
			b->st.use_xfrc = 1;
			b->params_dirty = true;
		}
		if (field == MJB_F_qpos && env_lo == 0 && env_hi == b->nenv) b->rowstat_ever = false;  // (wide-frame policy: a new workload)
		return MJB_OK;
	}
	if (field != MJB_F_qfrc_passive)
		return fail(MJB_EINVAL, "field %s is derived and cannot be set (only state fields and qfrc_passive can)",
		            kFields[field].name);
	if (!b->frame_valid || !b->st.frame_ws || env_hi > b->frame_hi)
		return fail(MJB_EINVAL, "qfrc_passive can only be modified between mjb_step1 and mjb_step2");
	const int off = frame_offset(b, field);
	HIP_TRY(hipMemcpy2D(b->st.frame_ws + (size_t)env_lo * b->st.frame_stride + off,
	                    (size_t)b->st.frame_stride * sizeof(double), host, (size_t)n * sizeof(double),
	                    (size_t)n * sizeof(double), (size_t)(env_hi - env_lo), hipMemcpyHostToDevice));
	return MJB_OK;
}

int mjb_register_collision(mjb_batch *b, int geom_type1, int geom_type2, int func)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	auto known = [](int t) { return t == MJB_GEOM_PLANE || t == MJB_GEOM_SPHERE || t == MJB_GEOM_CAPSULE || t == MJB_GEOM_BOX; };
	if (!known(geom_type1) || !known(geom_type2)) return fail(MJB_EINVAL, "mjb_register_collision: geom types must be plane / sphere / capsule / box");
	if (func < MJB_COLFUNC_DEFAULT || func > MJB_COLFUNC_SPHERES) return fail(MJB_EINVAL, "mjb_register_collision: unknown function %d", func);
	const mjb_model *M = b->model;
	const int np = M->h.ncollpair;
	if (np == 0) return MJB_OK;
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	const int lo = geom_type1 < geom_type2 ? geom_type1 : geom_type2, hi = geom_type1 < geom_type2 ? geom_type2 : geom_type1;
	if (b->st.colfunc[8 * lo + hi] != func) {  // the table by current types (read when per-env geom types are in play)
		b->st.colfunc[8 * lo + hi] = func;
		b->params_dirty = true;
	}
	for (int p = 0; p < np; p++) {  // (pairs are stored with type1 <= type2)
		if (M->pair_i[8 * p + 2] != lo || M->pair_i[8 * p + 3] != hi) continue;
		HIP_TRY(hipMemcpy(b->pair_i_dev + 8 * p + 6, &func, sizeof(int), hipMemcpyHostToDevice));
	}
	return MJB_OK;
}

int mjb_get_many(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, double *const *host)
{
	if (!b || n < 0 || (n && (!fields || !host))) return fail(MJB_EINVAL, "mjb_get_many: bad argument");
	HIP_TRY(hipSetDevice(b->device));
	// every argument is checked BEFORE the first copy is enqueued: a refusal never leaves transfers in flight
	for (int k = 0; k < n; k++) {
		const int field = fields[k];
		int rc = check_range(b, field, env_lo, env_hi);
		if (rc) return rc;
		if (kFields[field].kind == 3) return fail(MJB_EINVAL, "field %s is an int field; use mjb_get_int", kFields[field].name);
		if (b->model->field_size[field] == 0 || env_lo == env_hi) continue;
		if (!host[k]) return fail(MJB_EINVAL, "null host buffer");
		if (kFields[field].kind != 0 && (!b->frame_valid || !b->st.frame_ws || env_hi > b->frame_hi))
			return fail(MJB_EINVAL, "derived field %s is only readable after mjb_forward / mjb_step1 / mjb_step2", kFields[field].name);
	}
	hipError_t err = hipSuccess;
	for (int k = 0; k < n && err == hipSuccess; k++) {
		const int field = fields[k], sz = b->model->field_size[field];
		if (sz == 0 || env_lo == env_hi) continue;
		if (kFields[field].kind == 0)
			err = hipMemcpyAsync(host[k], state_ptr(b, field) + (size_t)env_lo * sz, (size_t)(env_hi - env_lo) * sz * sizeof(double),
			                     hipMemcpyDeviceToHost, b->stream);
		else
			err = hipMemcpy2DAsync(host[k], (size_t)sz * sizeof(double), b->st.frame_ws + (size_t)env_lo * b->st.frame_stride + frame_offset(b, field),
			                       (size_t)b->st.frame_stride * sizeof(double), (size_t)sz * sizeof(double), (size_t)(env_hi - env_lo),
			                       hipMemcpyDeviceToHost, b->stream);
	}
	const hipError_t serr = hipStreamSynchronize(b->stream);  // (also after a failed enqueue: the copies before it are drained)
	if (err != hipSuccess) return fail(MJB_ENODEVICE, "mjb_get_many: %s", hipGetErrorString(err));
	if (serr != hipSuccess) return fail(MJB_ENODEVICE, "mjb_get_many: %s", hipGetErrorString(serr));
	return MJB_OK;
}

// (asynchronous on the batch's stream: the host buffers must stay untouched until the next synchronising call -- mjb_synchronize,
//  mjb_get*, a blocking mjb_step; page-locked buffers make these true DMA transfers.  MujocoEnv::commitData synchronises.)
int mjb_set_many(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, const double *const *host)
{
	if (!b || n < 0 || (n && (!fields || !host))) return fail(MJB_EINVAL, "mjb_set_many: bad argument");
	HIP_TRY(hipSetDevice(b->device));
	for (int k = 0; k < n; k++) {
		const int field = fields[k];
		int rc = check_range(b, field, env_lo, env_hi);
		if (rc) return rc;
		if (b->model->field_size[field] == 0 || env_lo == env_hi) continue;
		if (!host[k]) return fail(MJB_EINVAL, "null host buffer");
		if (kFields[field].kind != 0) {
			if (field != MJB_F_qfrc_passive)
				return fail(MJB_EINVAL, "field %s is derived and cannot be set (only state fields and qfrc_passive can)", kFields[field].name);
			if (!b->frame_valid || !b->st.frame_ws || env_hi > b->frame_hi) return fail(MJB_EINVAL, "qfrc_passive can only be modified between mjb_step1 and mjb_step2");
		}
	}
	hipError_t err = hipSuccess;
	for (int k = 0; k < n && err == hipSuccess; k++) {
		const int field = fields[k], sz = b->model->field_size[field];
		if (sz == 0 || env_lo == env_hi) continue;
		if (kFields[field].kind == 0) {
			err = hipMemcpyAsync(state_ptr(b, field) + (size_t)env_lo * sz, host[k], (size_t)(env_hi - env_lo) * sz * sizeof(double),
			                     hipMemcpyHostToDevice, b->stream);
			if (field == MJB_F_xfrc_applied && !b->st.use_xfrc) {
				b->st.use_xfrc = 1;
				b->params_dirty = true;
			}
		} else {
			err = hipMemcpy2DAsync(b->st.frame_ws + (size_t)env_lo * b->st.frame_stride + frame_offset(b, field),
			                       (size_t)b->st.frame_stride * sizeof(double), host[k], (size_t)sz * sizeof(double), (size_t)sz * sizeof(double),
			                       (size_t)(env_hi - env_lo), hipMemcpyHostToDevice, b->stream);
		}
	}
	if (err != hipSuccess) {
		hipStreamSynchronize(b->stream);  // drain what was enqueued before the failure
		return fail(MJB_ENODEVICE, "mjb_set_many: %s", hipGetErrorString(err));
	}
	return MJB_OK;
}

// ---- several fields in ONE transfer --------------------------------------------------------------------------------------
// A callback round of the host runtime moves ~25 fields of the callback envs to the host and ~8 back (mujoco_env.cpp stepBurst);
// one hipMemcpy(2D)Async per field is ~8 us of driver time each, 350 us per split step against ~50 us of kernels
// (profiles/r02_callback_path.txt).  Here a small kernel gathers the fields into one device block laid out field after field
// ([env][dim] each, in the order given) and ONE copy moves the block; the reverse for the writes.
namespace {
struct PackDesc {
	double *src;       // env-major source: state array, or the frame workspace + the field's offset
	long long stride;  // doubles from one env to the next in the source
	int dim;           // doubles per env
	long long dst;     // offset of the field's block inside the packed block
};
struct PackArgs {
	PackDesc d[40];
	int n, nenv;
};
__global__ void mjb_pack_kernel(PackArgs a, double *block, int env_lo, int unpack)
{
	const int f = blockIdx.y;
	if (f >= a.n) return;
	const PackDesc d = a.d[f];
	const long long total = (long long)a.nenv * d.dim;
	for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
		const long long e = t / d.dim, k = t - e * d.dim;
		double *s = d.src + (env_lo + e) * d.stride + k;
		if (unpack) *s = block[d.dst + t];
		else block[d.dst + t] = *s;
	}
}
}  // namespace

static int packed_transfer(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, double *host_block, bool to_host, const char *who)
{
	if (!b || n < 0 || n > 40 || (n && (!fields || !host_block))) return fail(MJB_EINVAL, "%s: bad argument (at most 40 fields)", who);
	HIP_TRY(hipSetDevice(b->device));
	PackArgs a{};
	long long total = 0;
	int maxdim = 1;
	for (int k = 0; k < n; k++) {
		const int field = fields[k];
		int rc = check_range(b, field, env_lo, env_hi);
		if (rc) return rc;
		if (kFields[field].kind == 3) return fail(MJB_EINVAL, "%s: field %s is an int field", who, kFields[field].name);
		const int sz = b->model->field_size[field];
		PackDesc &d = a.d[a.n];
		if (kFields[field].kind == 0) {
			d.src = state_ptr(b, field);
			d.stride = sz;
		} else {
			if (!to_host && field != MJB_F_qfrc_passive)
				return fail(MJB_EINVAL, "field %s is derived and cannot be set (only state fields and qfrc_passive can)", kFields[field].name);
			if (!b->frame_valid || !b->st.frame_ws || env_hi > b->frame_hi)
				return fail(MJB_EINVAL, "derived field %s is only accessible after mjb_forward / mjb_step1 / mjb_step2", kFields[field].name);
			d.src = b->st.frame_ws + frame_offset(b, field);
			d.stride = b->st.frame_stride;
		}
		d.dim = sz;
		d.dst = total;
		total += (long long)(env_hi - env_lo) * sz;
		if (sz > 0) a.n++;
		if (sz > maxdim) maxdim = sz;
		if (!to_host && field == MJB_F_xfrc_applied && !b->st.use_xfrc) {
			b->st.use_xfrc = 1;
			b->params_dirty = true;
		}
	}
	if (total == 0 || env_lo == env_hi) return MJB_OK;
	a.nenv = env_hi - env_lo;
	if ((long long)b->pack_cap < total) {
		if (b->pack_dev) hipFree(b->pack_dev);
		b->pack_dev = dev_alloc<double>((size_t)total);
		if (!b->pack_dev) {
			b->pack_cap = 0;
			return fail(MJB_ENOMEM, "%s: staging allocation failed", who);
		}
		b->pack_cap = (size_t)total;
	}
	const int threads = 256;
	int bx = (int)std::min<long long>(((long long)a.nenv * maxdim + threads - 1) / threads, 64);
	if (bx < 1) bx = 1;
	if (!to_host)
		HIP_TRY(hipMemcpyAsync(b->pack_dev, host_block, (size_t)total * sizeof(double), hipMemcpyHostToDevice, b->stream));
	hipLaunchKernelGGL(mjb_pack_kernel, dim3(bx, a.n), dim3(threads), 0, b->stream, a, b->pack_dev, env_lo, to_host ? 0 : 1);
	HIP_TRY(hipGetLastError());
	if (to_host) {
		HIP_TRY(hipMemcpyAsync(host_block, b->pack_dev, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, b->stream));
		HIP_TRY(hipStreamSynchronize(b->stream));
	}
	return MJB_OK;
}

int mjb_get_packed(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, double *host_block)
{
	return packed_transfer(b, n, fields, env_lo, env_hi, host_block, true, "mjb_get_packed");
}

int mjb_set_packed(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, const double *host_block)
{
	return packed_transfer(b, n, fields, env_lo, env_hi, const_cast<double *>(host_block), false, "mjb_set_packed");
}

int mjb_host_register(void *host, unsigned long long bytes)
{
	if (!host || !bytes) return fail(MJB_EINVAL, "mjb_host_register: bad argument");
	if (hipHostRegister(host, (size_t)bytes, hipHostRegisterPortable)  /* every device of a sharded batch may DMA from it */ != hipSuccess) {
		(void)hipGetLastError();
		return fail(MJB_ENODEVICE, "hipHostRegister failed");
	}
	return MJB_OK;
}

int mjb_host_unregister(void *host)
{
	if (!host) return MJB_OK;
	if (hipHostUnregister(host) != hipSuccess) (void)hipGetLastError();
	return MJB_OK;
}

void *mjb_device_ptr(mjb_batch *b, int field)
{
	if (!b || field < 0 || field >= MJB_F_COUNT || kFields[field].kind != 0) {
		fail(MJB_EINVAL, "mjb_device_ptr: not a state field");
		return nullptr;
	}
	return state_ptr(b, field);
}

int mjb_set_ctrl_noise(mjb_batch *b, double ctrl_noise_std, double ctrl_noise_rate, uint64_t seed, int64_t env_offset)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	if (ctrl_noise_std < 0) return fail(MJB_EINVAL, "ctrl_noise_std must be >= 0");
	const double dt = b->model->h.timestep[0];
	const double rate = std::exp(-dt / std::fmax(ctrl_noise_rate, 1e-15));
	b->nz.rate = rate;
	b->nz.scale = ctrl_noise_std * std::sqrt(1 - rate * rate);
	b->nz.seed = seed;
	b->nz.env_offset = env_offset;
	b->nz.enabled = ctrl_noise_std > 0;
	b->params_dirty = true;
	if (b->zinfo) {  // normals generated under the old key (also the ones a side-stream generator is still writing) are void
		HIP_TRY(hipSetDevice(b->device));
		if (b->noise_stream) HIP_TRY(hipStreamSynchronize(b->noise_stream));
		HIP_TRY(hipStreamSynchronize(b->stream));
		HIP_TRY(hipMemset(b->zinfo, 0xff, 8 * sizeof(unsigned int)));
		b->spec_valid = false;
		b->zvalid = false;
	}
	return MJB_OK;
}

int mjb_set_stats(mjb_batch *b, int on)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	HIP_TRY(hipSetDevice(b->device));
	int rc = join_rest(b);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (on) {
		if (!b->stats_dev) b->stats_dev = dev_alloc<unsigned long long>(MJB_NSTATS);
		if (!b->stats_dev) return fail(MJB_ENOMEM, "mjb_set_stats: allocation failed");
		HIP_TRY(hipMemset(b->stats_dev, 0, MJB_NSTATS * sizeof(unsigned long long)));
	}
	b->st.stats = on ? b->stats_dev : nullptr;
	b->params_dirty = true;
	return MJB_OK;
}

int mjb_get_stats(mjb_batch *b, unsigned long long *out)
{
	if (!b || !out) return fail(MJB_EINVAL, "mjb_get_stats: bad argument");
	if (!b->stats_dev) return fail(MJB_EINVAL, "mjb_get_stats before mjb_set_stats(b, 1)");
	HIP_TRY(hipSetDevice(b->device));
	int rc = join_rest(b);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(b->stream));
	HIP_TRY(hipMemcpy(out, b->stats_dev, MJB_NSTATS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return MJB_OK;
}

int mjb_noise_mode(const mjb_batch *b) { return b ? b->noise_mode : 0; }
int mjb_set_lane_env(mjb_batch *b, int mode)
{
	if (!b || mode < -1 || mode > 1) return fail(MJB_EINVAL, "mjb_set_lane_env: bad argument");
	b->lane_env_mode = mode;
	return MJB_OK;
}
int mjb_set_sensors_every_step(mjb_batch *b, int on)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	if ((b->st.sens_every_step != 0) != (on != 0)) {
		b->st.sens_every_step = on ? 1 : 0;
		b->params_dirty = true;
	}
	return MJB_OK;
}
int mjb_set_split_step(mjb_batch *b, int mode)
{
	if (!b || mode < -1 || mode > 1) return fail(MJB_EINVAL, "mjb_set_split_step: bad argument");
	b->split_mode = mode;
	return MJB_OK;
}
int mjb_split_step_info(const mjb_batch *b, int *used_last, int *slices)
{
	if (used_last) *used_last = b && b->split_used ? 1 : 0;
	if (slices) *slices = b ? b->split_slices_last : 0;
	return b ? b->model->sm_topo : -1;
}
int mjb_model_split_step(const mjb_model *m) { return m ? m->sm_topo : -1; }
const char *mjb_lane_env_error(void) { return mjb_lane_env_jit_error(); }
void mjb_lane_env_jit_counts(int *compiled, int *disk_hits) { mjb_lane_env_jit_stats(compiled, disk_hits); }
int mjb_model_lane_env(const mjb_model *m) { return m ? m->le_topo : -1; }
int mjb_lane_env_info(const mjb_batch *b, int *used_last)
{
	if (used_last) *used_last = b && b->lane_env_used ? 1 : 0;
	return b ? (b->le_unavailable ? -3 : b->model->le_topo) : -1;
}
int mjb_fused_frame(const mjb_batch *b) { return b && b->wide ? 2 : 1; }

void *mjb_get_stream(mjb_batch *b) { return b ? (void *)b->stream : nullptr; }

int mjb_set_stream(mjb_batch *b, void *hip_stream)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	HIP_TRY(hipSetDevice(b->device));
	if (b->stream) HIP_TRY(hipStreamSynchronize(b->stream));
	if (b->own_stream && b->stream) hipStreamDestroy(b->stream);
	b->stream = (hipStream_t)hip_stream;
	b->own_stream = false;
	return MJB_OK;
}

int mjb_synchronize(mjb_batch *b)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return MJB_OK;
}

int mjb_debug_profile(mjb_batch *b, unsigned long long *out64, int clear)
{
	if (!b || !out64) return fail(MJB_EINVAL, "mjb_debug_profile: bad argument");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	HIP_TRY(hipMemcpy(out64, b->st.prof, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	if (clear) HIP_TRY(hipMemset(b->st.prof, 0, 64 * sizeof(unsigned long long)));
	return MJB_OK;
}

int mjb_debug_profile_window(mjb_batch *b, int first_id)
{
	if (!b || first_id < 0 || first_id > 31) return fail(MJB_EINVAL, "mjb_debug_profile_window: bad argument");
	if (b->st.prof_base != first_id) {
		b->st.prof_base = first_id;
		b->params_dirty = true;
	}
	return MJB_OK;
}

int mjb_warning(mjb_batch *b, int which, unsigned long long *count)
{
	if (!b || !count || which < 0 || which >= MJB_NWARNING) return fail(MJB_EINVAL, "mjb_warning: bad argument");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	HIP_TRY(hipMemcpy(count, b->st.nwarn + which, sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return MJB_OK;
}

int mjb_warning_count(mjb_batch *b, unsigned long long *count)
{
	if (!b || !count) return fail(MJB_EINVAL, "mjb_warning_count: bad argument");
	unsigned long long w[MJB_NWARNING];
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	HIP_TRY(hipMemcpy(w, b->st.nwarn, sizeof w, hipMemcpyDeviceToHost));
	*count = w[MJB_WARN_BADQPOS] + w[MJB_WARN_BADQVEL] + w[MJB_WARN_BADQACC];
	return MJB_OK;
}

}  // extern "C"

// ---- aggregate metrics (mjb_metrics): one workgroup reduces the state arrays of the batch ----
__global__ void mjb_metrics_kernel(const double *qacc, const double *qvel, const double *time, const double *energy,
                                   const unsigned long long *nwarn, int nenv, int nv, double env_steps, double *out)
{
	__shared__ double red[4][256];
	double mq = 0, mv = 0, mt = 0, pe = 0, ke = 0;
	for (size_t k = threadIdx.x; k < (size_t)nenv * nv; k += blockDim.x) {
		mq = fmax(mq, fabs(qacc[k]));
		mv = fmax(mv, fabs(qvel[k]));
	}
	for (int e = threadIdx.x; e < nenv; e += blockDim.x) {
		mt = fmax(mt, time[e]);
		pe += energy[2 * e];
		ke += energy[2 * e + 1];
	}
	// (fmax drops NaNs: an env that went non-finite inside a launch has been reset before the state was stored)
	double v[5] = { mq, mv, mt, pe, ke };
	double res[5];
	for (int q = 0; q < 5; q++) {
		red[0][threadIdx.x] = v[q];
		__syncthreads();
		for (int s = blockDim.x / 2; s > 0; s >>= 1) {
			if ((int)threadIdx.x < s)
				red[0][threadIdx.x] = q < 3 ? fmax(red[0][threadIdx.x], red[0][threadIdx.x + s]) : red[0][threadIdx.x] + red[0][threadIdx.x + s];
			__syncthreads();
		}
		res[q] = red[0][0];
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		out[0] = env_steps;
		out[1] = (double)(nwarn[MJB_WARN_BADQPOS] + nwarn[MJB_WARN_BADQVEL] + nwarn[MJB_WARN_BADQACC]);
		out[2] = (double)nwarn[MJB_WARN_CONTACTFULL];
		out[3] = (double)nwarn[MJB_WARN_CNSTRFULL];
		out[4] = res[3];
		out[5] = res[4];
		out[6] = (double)nenv;
		out[7] = 0;
		out[8] = res[0];
		out[9] = res[1];
		out[10] = res[2];
		for (int k = 11; k < 16; k++) out[k] = 0;
	}
}

extern "C" {

void *mjb_metrics_device(mjb_batch *b)
{
	if (!b) {
		fail(MJB_EINVAL, "null batch");
		return nullptr;
	}
	if (hipSetDevice(b->device) != hipSuccess) return nullptr;
	if (!b->metrics_dev) {
		b->metrics_dev = dev_alloc<double>(16);
		if (!b->metrics_dev) {
			fail(MJB_ENOMEM, "mjb_metrics: allocation failed");
			return nullptr;
		}
	}
	hipLaunchKernelGGL(mjb_metrics_kernel, dim3(1), dim3(256), 0, b->stream, b->st.qacc, b->st.qvel, b->st.time, b->st.energy,
	                   b->st.nwarn, b->nenv, b->model->h.nv, (double)b->nenv * (double)b->steps_taken, b->metrics_dev);
	if (hipGetLastError() != hipSuccess) {
		fail(MJB_ENODEVICE, "mjb_metrics: launch failed");
		return nullptr;
	}
	return b->metrics_dev;
}

int mjb_metrics(mjb_batch *b, double *out16)
{
	if (!b || !out16) return fail(MJB_EINVAL, "mjb_metrics: bad argument");
	void *p = mjb_metrics_device(b);
	if (!p) return MJB_ENODEVICE;
	HIP_TRY(hipStreamSynchronize(b->stream));
	HIP_TRY(hipMemcpy(out16, p, 16 * sizeof(double), hipMemcpyDeviceToHost));
	return MJB_OK;
}

// ---- per-env model parameters (the subset that needs no mj_setConst) ----
static int env_param(mjb_batch *b, double **arr, const double **slot, int per_env, const double *model_vals, int env_lo, int env_hi,
                     const double *vals, const char *what)
{
	if (!b || !vals) return fail(MJB_EINVAL, "%s: bad argument", what);
	if (env_lo < 0 || env_hi > b->nenv || env_lo > env_hi) return fail(MJB_EINVAL, "%s: bad env range", what);
	if (per_env <= 0) return fail(MJB_EINVAL, "%s: the model has nothing to override", what);
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (!*arr) {  // first use: every env starts from the model's values
		*arr = dev_alloc<double>((size_t)b->nenv * per_env);
		if (!*arr) return fail(MJB_ENOMEM, "%s: allocation failed", what);
		std::vector<double> init((size_t)b->nenv * per_env);
		for (int e = 0; e < b->nenv; e++) memcpy(init.data() + (size_t)e * per_env, model_vals, (size_t)per_env * sizeof(double));
		HIP_TRY(hipMemcpy(*arr, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice));
		*slot = *arr;
		b->params_dirty = true;
	}
	HIP_TRY(hipMemcpy(*arr + (size_t)env_lo * per_env, vals, (size_t)(env_hi - env_lo) * per_env * sizeof(double), hipMemcpyHostToDevice));
	return MJB_OK;
}

int mjb_set_env_gravity(mjb_batch *b, int env_lo, int env_hi, const double *gravity)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	return env_param(b, &b->env_gravity, &b->st.env_gravity, 3, b->model->h.gravity, env_lo, env_hi, gravity, "mjb_set_env_gravity");
}

int mjb_set_env_geom_friction(mjb_batch *b, int env_lo, int env_hi, const double *friction)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	const mjb_model_desc &h = b->model->h;
	return env_param(b, &b->env_geom_friction, &b->st.env_geom_friction, h.nconmax > 0 ? 3 * h.ngeom : 0, h.geom_friction, env_lo,
	                 env_hi, friction, "mjb_set_env_geom_friction");
}

int mjb_set_env_geom_size(mjb_batch *b, int env_lo, int env_hi, const double *size)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	const mjb_model_desc &h = b->model->h;
	return env_param(b, &b->env_geom_size, &b->st.env_geom_size, h.nconmax > 0 ? 3 * h.ngeom : 0, h.geom_size, env_lo, env_hi, size,
	                 "mjb_set_env_geom_size");
}

int mjb_set_env_geom_type(mjb_batch *b, int env_lo, int env_hi, const int *type)
{
	if (!b || !type) return fail(MJB_EINVAL, "mjb_set_env_geom_type: bad argument");
	const mjb_model_desc &h = b->model->h;
	if (env_lo < 0 || env_hi > b->nenv || env_lo > env_hi) return fail(MJB_EINVAL, "mjb_set_env_geom_type: bad env range");
	if (h.nconmax <= 0 || h.ngeom <= 0) return fail(MJB_EINVAL, "mjb_set_env_geom_type: the model has nothing to override");
	for (size_t k = 0; k < (size_t)(env_hi - env_lo) * h.ngeom; k++)
		if (type[k] != MJB_GEOM_PLANE && type[k] != MJB_GEOM_SPHERE && type[k] != MJB_GEOM_CAPSULE && type[k] != MJB_GEOM_BOX &&
		    type[k] != 4 && type[k] != 5)  // (mjGEOM_ELLIPSOID / mjGEOM_CYLINDER are accepted as "no pair function": the geom yields no contacts)
			return fail(MJB_EUNSUPPORTED, "mjb_set_env_geom_type: geom type %d is neither a primitive of the engine nor ellipsoid / cylinder", type[k]);
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (!b->env_geom_type) {
		b->env_geom_type = dev_alloc<int>((size_t)b->nenv * h.ngeom);
		if (!b->env_geom_type) return fail(MJB_ENOMEM, "mjb_set_env_geom_type: allocation failed");
		std::vector<int> init((size_t)b->nenv * h.ngeom);
		for (int e = 0; e < b->nenv; e++) memcpy(init.data() + (size_t)e * h.ngeom, h.geom_type, (size_t)h.ngeom * sizeof(int));
		HIP_TRY(hipMemcpy(b->env_geom_type, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice));
		b->st.env_geom_type = b->env_geom_type;
		b->params_dirty = true;
	}
	HIP_TRY(hipMemcpy(b->env_geom_type + (size_t)env_lo * h.ngeom, type, (size_t)(env_hi - env_lo) * h.ngeom * sizeof(int), hipMemcpyHostToDevice));
	return MJB_OK;
}

int mjb_set_env_equality(mjb_batch *b, int env_lo, int env_hi, const double *params)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	const mjb_model_desc &h = b->model->h;
	std::vector<double> packed((size_t)19 * (h.neq > 0 ? h.neq : 1));
	for (int q = 0; q < h.neq; q++) {
		double *o = packed.data() + 19 * q;
		o[0] = h.eq_active[q] ? 1.0 : 0.0;
		for (int k = 0; k < 11; k++) o[1 + k] = h.eq_data[11 * q + k];
		for (int k = 0; k < 2; k++) o[12 + k] = h.eq_solref[2 * q + k];
		for (int k = 0; k < 5; k++) o[14 + k] = h.eq_solimp[5 * q + k];
	}
	return env_param(b, &b->env_equality, &b->st.env_equality, 19 * h.neq, packed.data(), env_lo, env_hi, params, "mjb_set_env_equality");
}

int mjb_env_mass_stride(const mjb_model *m)
{
	if (!m) return fail(MJB_EINVAL, "null model");
	return 7 * m->h.nbody + m->h.nv + m->h.ntendon + 1;
}

int mjb_set_env_mass_params(mjb_batch *b, int env_lo, int env_hi, const double *params)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	const mjb_model_desc &h = b->model->h;
	const int n = 7 * h.nbody + h.nv + h.ntendon + 1;
	std::vector<double> packed((size_t)n);
	double *o = packed.data();
	for (int i = 0; i < h.nbody; i++) o[i] = h.body_mass[i];
	for (int i = 0; i < h.nbody; i++) o[h.nbody + i] = h.body_subtreemass[i];
	for (int i = 0; i < 3 * h.nbody; i++) o[2 * h.nbody + i] = h.body_inertia[i];
	for (int i = 0; i < h.nv; i++) o[5 * h.nbody + i] = h.dof_invweight0[i];
	for (int i = 0; i < 2 * h.nbody; i++) o[5 * h.nbody + h.nv + i] = h.body_invweight0[i];
	for (int i = 0; i < h.ntendon; i++) o[7 * h.nbody + h.nv + i] = h.tendon_invweight0[i];
	o[7 * h.nbody + h.nv + h.ntendon] = h.meaninertia[0];
	return env_param(b, &b->env_mass, &b->st.env_mass, n, packed.data(), env_lo, env_hi, params, "mjb_set_env_mass_params");
}

// ---- mj_setConst for new body masses (callbacks.cpp:251-256, :582), host side, plain C++ ----
// What mj_setConst's set0 stage derives from the masses at qpos0: body_subtreemass, dof_invweight0 (diagonal of M^-1; joint-wise
// mean over the 3 dofs of a ball joint / each half of a free joint), body_invweight0 (mean translational / rotational diagonal of
// J M^-1 J' at the body's inertial frame; 0 for bodies welded to the world), tendon_invweight0 (J_t M^-1 J_t') and
// stat.meaninertia (mean diagonal of M).  M is assembled from body Jacobians (m Jp'Jp + Jr' R I R' Jr + armature), the same
// formulation as mujoco_ros_pkgs_amd/refdyn.py, which tests/test_setconst_cpp.py compares it with.
namespace {
struct SC {  // tiny dense helpers (row-major)
	static void quat2mat(const double *q, double *R)
	{
		const double w = q[0], x = q[1], y = q[2], z = q[3];
		R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
		R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
		R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
	}
	static void qmul(const double *a, const double *b, double *o)
	{
		const double r[4] = { a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
			                  a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0] };
		for (int k = 0; k < 4; k++) o[k] = r[k];
	}
	static void mv(const double *R, const double *v, double *o)
	{
		const double r[3] = { R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2], R[6] * v[0] + R[7] * v[1] + R[8] * v[2] };
		for (int k = 0; k < 3; k++) o[k] = r[k];
	}
	static void cross(const double *a, const double *b, double *o)
	{
		const double r[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
		for (int k = 0; k < 3; k++) o[k] = r[k];
	}
};
}  // namespace

int mjb_derive_mass_params(const mjb_model *model, const double *body_mass, const double *body_inertia, double *out)
{
	if (!model || !body_mass || !out) return fail(MJB_EINVAL, "mjb_derive_mass_params: bad argument");
	const mjb_model_desc &h = model->h;
	const int nb = h.nbody, nv = h.nv, nj = h.njnt, nt = h.ntendon;
	const double *inertia = body_inertia ? body_inertia : h.body_inertia;
	double *o_mass = out, *o_sub = out + nb, *o_inert = out + 2 * nb, *o_dof = out + 5 * nb, *o_body = out + 5 * nb + nv,
	       *o_ten = out + 7 * nb + nv, *o_mean = out + 7 * nb + nv + nt;
	for (int b = 0; b < nb; b++) {
		o_mass[b] = body_mass[b];
		o_sub[b] = body_mass[b];
		for (int k = 0; k < 3; k++) o_inert[3 * b + k] = inertia[3 * b + k];
		o_body[2 * b] = o_body[2 * b + 1] = 0;
	}
	for (int b = nb - 1; b > 0; b--) o_sub[h.body_parentid[b]] += o_sub[b];
	for (int i = 0; i < nv; i++) o_dof[i] = 0;
	for (int t = 0; t < nt; t++) o_ten[t] = 0;
	*o_mean = 1.0;
	if (nv == 0) return MJB_OK;
	// kinematics at qpos0
	std::vector<double> xpos(3 * nb, 0.0), xquat(4 * nb, 0.0), xmat(9 * nb, 0.0), xipos(3 * nb, 0.0), ximat(9 * nb, 0.0), xanchor(3 * std::max(1, nj), 0.0),
	    xaxis(3 * std::max(1, nj), 0.0);
	xquat[0] = 1;
	SC::quat2mat(&xquat[0], &xmat[0]);
	for (int b = 1; b < nb; b++) {
		const int p = h.body_parentid[b], ja = h.body_jntadr[b], jn = h.body_jntnum[b];
		double pos[3], q[4];
		if (jn == 1 && h.jnt_type[ja] == MJB_JNT_FREE) {
			const int qa = h.jnt_qposadr[ja];
			double nrm = 0;
			for (int k = 0; k < 4; k++) nrm += h.qpos0[qa + 3 + k] * h.qpos0[qa + 3 + k];
			nrm = std::sqrt(nrm);
			for (int k = 0; k < 3; k++) pos[k] = h.qpos0[qa + k];
			for (int k = 0; k < 4; k++) q[k] = h.qpos0[qa + 3 + k] / nrm;
			for (int k = 0; k < 3; k++) { xanchor[3 * ja + k] = pos[k]; xaxis[3 * ja + k] = h.jnt_axis[3 * ja + k]; }
		} else {
			double v[3];
			SC::mv(&xmat[9 * p], &h.body_pos[3 * b], v);
			for (int k = 0; k < 3; k++) pos[k] = xpos[3 * p + k] + v[k];
			SC::qmul(&xquat[4 * p], &h.body_quat[4 * b], q);
			for (int j = ja; j < ja + jn; j++) {
				double R[9], w[3];
				SC::quat2mat(q, R);
				SC::mv(R, &h.jnt_axis[3 * j], &xaxis[3 * j]);
				SC::mv(R, &h.jnt_pos[3 * j], w);
				for (int k = 0; k < 3; k++) xanchor[3 * j + k] = pos[k] + w[k];
				if (h.jnt_type[j] == MJB_JNT_BALL) {  // (hinge / slide sit at qpos0: no motion)
					const int qa = h.jnt_qposadr[j];
					double ql[4], nrm = 0;
					for (int k = 0; k < 4; k++) nrm += h.qpos0[qa + k] * h.qpos0[qa + k];
					nrm = std::sqrt(nrm);
					for (int k = 0; k < 4; k++) ql[k] = h.qpos0[qa + k] / nrm;
					SC::qmul(q, ql, q);
					SC::quat2mat(q, R);
					SC::mv(R, &h.jnt_pos[3 * j], w);
					for (int k = 0; k < 3; k++) pos[k] = xanchor[3 * j + k] - w[k];
				}
			}
		}
		double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
		for (int k = 0; k < 3; k++) xpos[3 * b + k] = pos[k];
		for (int k = 0; k < 4; k++) xquat[4 * b + k] = q[k] / nrm;
		SC::quat2mat(&xquat[4 * b], &xmat[9 * b]);
	}
	for (int b = 0; b < nb; b++) {
		double v[3], Ri[9];
		SC::mv(&xmat[9 * b], &h.body_ipos[3 * b], v);
		for (int k = 0; k < 3; k++) xipos[3 * b + k] = xpos[3 * b + k] + v[k];
		SC::quat2mat(&h.body_iquat[4 * b], Ri);
		for (int r = 0; r < 3; r++)
			for (int c = 0; c < 3; c++) {
				double a = 0;
				for (int k = 0; k < 3; k++) a += xmat[9 * b + 3 * r + k] * Ri[3 * k + c];
				ximat[9 * b + 3 * r + c] = a;
			}
	}
	// Jacobians of a point fixed to `body` (3 x nv each)
	auto jac = [&](int body, const double *point, std::vector<double> &jp, std::vector<double> &jr) {
		std::fill(jp.begin(), jp.end(), 0.0);
		std::fill(jr.begin(), jr.end(), 0.0);
		for (int b = body; b > 0; b = h.body_parentid[b])
			for (int j = h.body_jntadr[b]; j < h.body_jntadr[b] + h.body_jntnum[b]; j++) {
				const int d = h.jnt_dofadr[j], t = h.jnt_type[j];
				auto rot_dof = [&](int dof, const double *ax, const double *about) {
					const double off[3] = { point[0] - about[0], point[1] - about[1], point[2] - about[2] };
					double c[3];
					SC::cross(ax, off, c);
					for (int k = 0; k < 3; k++) { jr[(size_t)k * nv + dof] = ax[k]; jp[(size_t)k * nv + dof] = c[k]; }
				};
				if (t == MJB_JNT_HINGE) rot_dof(d, &xaxis[3 * j], &xanchor[3 * j]);
				else if (t == MJB_JNT_SLIDE) for (int k = 0; k < 3; k++) jp[(size_t)k * nv + d] = xaxis[3 * j + k];
				else {
					const int r0 = t == MJB_JNT_FREE ? d + 3 : d;
					if (t == MJB_JNT_FREE) for (int k = 0; k < 3; k++) jp[(size_t)k * nv + d + k] = 1.0;
					for (int k = 0; k < 3; k++) {
						const double ax[3] = { xmat[9 * b + k], xmat[9 * b + 3 + k], xmat[9 * b + 6 + k] };
						rot_dof(r0 + k, ax, t == MJB_JNT_FREE ? &xpos[3 * b] : &xanchor[3 * j]);
					}
				}
			}
	};
	std::vector<double> M((size_t)nv * nv, 0.0), jp((size_t)3 * nv), jr((size_t)3 * nv);
	for (int i = 0; i < nv; i++) M[(size_t)i * nv + i] = h.dof_armature[i];
	for (int b = 1; b < nb; b++) {
		const double mb = body_mass[b], *I = inertia + 3 * b;
		if (mb == 0 && I[0] == 0 && I[1] == 0 && I[2] == 0) continue;
		jac(b, &xipos[3 * b], jp, jr);
		double Iw[9];  // R diag(I) R'
		for (int r = 0; r < 3; r++)
			for (int c = 0; c < 3; c++) {
				double a = 0;
				for (int k = 0; k < 3; k++) a += ximat[9 * b + 3 * r + k] * I[k] * ximat[9 * b + 3 * c + k];
				Iw[3 * r + c] = a;
			}
		for (int i = 0; i < nv; i++)
			for (int j = 0; j < nv; j++) {
				double a = 0;
				for (int k = 0; k < 3; k++) a += mb * jp[(size_t)k * nv + i] * jp[(size_t)k * nv + j];
				for (int r = 0; r < 3; r++) {
					double t = 0;
					for (int c = 0; c < 3; c++) t += Iw[3 * r + c] * jr[(size_t)c * nv + j];
					a += jr[(size_t)r * nv + i] * t;
				}
				M[(size_t)i * nv + j] += a;
			}
	}
	double mean = 0;
	for (int i = 0; i < nv; i++) mean += M[(size_t)i * nv + i];
	*o_mean = mean / nv;
	// M^-1 by Gauss-Jordan with partial pivoting (nv <= 64)
	std::vector<double> A(M), Minv((size_t)nv * nv, 0.0);
	for (int i = 0; i < nv; i++) Minv[(size_t)i * nv + i] = 1.0;
	for (int c = 0; c < nv; c++) {
		int piv = c;
		for (int r = c + 1; r < nv; r++)
			if (std::fabs(A[(size_t)r * nv + c]) > std::fabs(A[(size_t)piv * nv + c])) piv = r;
		if (std::fabs(A[(size_t)piv * nv + c]) < 1e-300) return fail(MJB_EINVAL, "mjb_derive_mass_params: singular joint-space inertia (dof %d)", c);
		if (piv != c)
			for (int k = 0; k < nv; k++) { std::swap(A[(size_t)piv * nv + k], A[(size_t)c * nv + k]); std::swap(Minv[(size_t)piv * nv + k], Minv[(size_t)c * nv + k]); }
		const double d = 1.0 / A[(size_t)c * nv + c];
		for (int k = 0; k < nv; k++) { A[(size_t)c * nv + k] *= d; Minv[(size_t)c * nv + k] *= d; }
		for (int r = 0; r < nv; r++) {
			if (r == c) continue;
			const double f = A[(size_t)r * nv + c];
			if (f == 0) continue;
			for (int k = 0; k < nv; k++) { A[(size_t)r * nv + k] -= f * A[(size_t)c * nv + k]; Minv[(size_t)r * nv + k] -= f * Minv[(size_t)c * nv + k]; }
		}
	}
	auto quad = [&](const double *ja, const double *jb) {  // ja' M^-1 jb for two nv-vectors
		double a = 0;
		for (int i = 0; i < nv; i++) {
			if (ja[i] == 0) continue;
			double t = 0;
			for (int j = 0; j < nv; j++) t += Minv[(size_t)i * nv + j] * jb[j];
			a += ja[i] * t;
		}
		return a;
	};
	for (int b = 1; b < nb; b++) {
		if (h.body_weldid[b] == 0) continue;
		jac(b, &xipos[3 * b], jp, jr);
		double tr = 0, ro = 0;
		for (int k = 0; k < 3; k++) { tr += quad(&jp[(size_t)k * nv], &jp[(size_t)k * nv]); ro += quad(&jr[(size_t)k * nv], &jr[(size_t)k * nv]); }
		o_body[2 * b] = tr / 3;
		o_body[2 * b + 1] = ro / 3;
	}
	for (int j = 0; j < nj; j++) {
		const int d = h.jnt_dofadr[j], t = h.jnt_type[j];
		auto tr3 = [&](int a) { return (Minv[(size_t)a * nv + a] + Minv[(size_t)(a + 1) * nv + a + 1] + Minv[(size_t)(a + 2) * nv + a + 2]) / 3; };
		if (t == MJB_JNT_HINGE || t == MJB_JNT_SLIDE) o_dof[d] = Minv[(size_t)d * nv + d];
		else if (t == MJB_JNT_BALL) o_dof[d] = o_dof[d + 1] = o_dof[d + 2] = tr3(d);
		else {
			o_dof[d] = o_dof[d + 1] = o_dof[d + 2] = tr3(d);
			o_dof[d + 3] = o_dof[d + 4] = o_dof[d + 5] = tr3(d + 3);
		}
	}
	std::vector<double> jt((size_t)nv);
	for (int t = 0; t < nt; t++) {
		std::fill(jt.begin(), jt.end(), 0.0);
		for (int w = h.tendon_adr[t]; w < h.tendon_adr[t] + h.tendon_num[t]; w++) jt[h.jnt_dofadr[h.wrap_objid[w]]] += h.wrap_prm[w];
		o_ten[t] = quad(jt.data(), jt.data());
	}
	return MJB_OK;
}

int mjb_set_env_body_mass(mjb_batch *b, int env_lo, int env_hi, const double *body_mass, const double *body_inertia)
{
	if (!b || !body_mass) return fail(MJB_EINVAL, "mjb_set_env_body_mass: bad argument");
	if (env_lo < 0 || env_hi > b->nenv || env_lo > env_hi) return fail(MJB_EINVAL, "mjb_set_env_body_mass: bad env range");
	const mjb_model_desc &h = b->model->h;
	const int stride = mjb_env_mass_stride(b->model), n = env_hi - env_lo;
	std::vector<double> packed((size_t)std::max(1, n) * stride);
	// (the derivation is mj_setConst -- Jacobians of every body, an O(nbody nv^2) mass matrix and its inverse -- and a service call that
	//  sets one mass on the whole batch, setBodyStateCB with the default env range, hands over nenv identical rows under the physics
	//  mutex: derived once per DISTINCT row, the block of a row equal to its predecessor is copied)
	for (int e = 0; e < n; e++) {
		const double *bm = body_mass + (size_t)e * h.nbody, *bi = body_inertia ? body_inertia + (size_t)e * 3 * h.nbody : nullptr;
		double *dst = packed.data() + (size_t)e * stride;
		if (e > 0 && std::memcmp(bm, bm - h.nbody, sizeof(double) * h.nbody) == 0 &&
		    (!bi || std::memcmp(bi, bi - 3 * h.nbody, sizeof(double) * 3 * h.nbody) == 0)) {
			std::memcpy(dst, dst - stride, sizeof(double) * stride);
			continue;
		}
		const int rc = mjb_derive_mass_params(b->model, bm, bi, dst);
		if (rc != MJB_OK) return rc;
	}
	return mjb_set_env_mass_params(b, env_lo, env_hi, packed.data());
}

// ---- device-side DefaultRobotHWSim::writeSim (stage hwsim_write in mjb_step.hip) ----
int mjb_hwsim_configure(mjb_batch *b, int n, const mjb_hwsim_joint *joints)
{
	if (!b || n < 0 || (n > 0 && !joints)) return fail(MJB_EINVAL, "mjb_hwsim_configure: bad argument");
	const mjb_model_desc &h = b->model->h;
	for (int k = 0; k < n; k++) {
		const mjb_hwsim_joint &j = joints[k];
		if (j.joint < 0 || j.joint >= h.njnt || h.jnt_type[j.joint] < MJB_JNT_SLIDE)
			return fail(MJB_EINVAL, "mjb_hwsim_configure: entry %d: joint %d is not a hinge / slide joint of the model", k, j.joint);
		if (j.method < MJB_HW_EFFORT || j.method > MJB_HW_VELOCITY_PID || j.kind < MJB_HW_REVOLUTE || j.kind > MJB_HW_PRISMATIC)
			return fail(MJB_EINVAL, "mjb_hwsim_configure: entry %d: bad control method / joint kind", k);
	}
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (b->hw_ints) hipFree(b->hw_ints);
	if (b->hw_gains) hipFree(b->hw_gains);
	if (b->hw_cmd) hipFree(b->hw_cmd);
	if (b->hw_pid) hipFree(b->hw_pid);
	if (b->hw_cad) hipFree(b->hw_cad);
	b->hw_ints = nullptr; b->hw_gains = nullptr; b->hw_cmd = nullptr; b->hw_pid = nullptr; b->hw_cad = nullptr;
	b->hw = HwSim{};
	b->params_dirty = true;
	if (n == 0) return MJB_OK;
	std::vector<int> ints((size_t)4 * n);
	std::vector<double> gains((size_t)8 * n);
	for (int k = 0; k < n; k++) {
		const mjb_hwsim_joint &j = joints[k];
		ints[k] = j.joint; ints[n + k] = j.method; ints[2 * n + k] = j.kind; ints[3 * n + k] = j.antiwindup ? 1 : 0;
		const double g[8] = { j.p, j.i, j.d, j.i_max, j.i_min, j.effort_limit, j.lower, j.upper };
		for (int q = 0; q < 8; q++) gains[(size_t)8 * k + q] = g[q];
	}
	const size_t per = (size_t)b->nenv * n;
	b->hw_ints = dev_alloc<int>(ints.size());
	b->hw_gains = dev_alloc<double>(gains.size());
	b->hw_cmd = dev_alloc<double>(4 * per);
	b->hw_pid = dev_alloc<double>(2 * per);
	if (!b->hw_ints || !b->hw_gains || !b->hw_cmd || !b->hw_pid) return fail(MJB_ENOMEM, "mjb_hwsim_configure: allocation failed");
	HIP_TRY(hipMemcpy(b->hw_ints, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(b->hw_gains, gains.data(), gains.size() * sizeof(double), hipMemcpyHostToDevice));
	HIP_TRY(hipMemset(b->hw_cmd, 0, 4 * per * sizeof(double)));
	HIP_TRY(hipMemset(b->hw_pid, 0, 2 * per * sizeof(double)));
	b->hw.n = n;
	b->hw.estop = 0;
	b->hw.joint = b->hw_ints; b->hw.method = b->hw_ints + n; b->hw.kind = b->hw_ints + 2 * n; b->hw.antiwindup = b->hw_ints + 3 * n;
	b->hw.gains = b->hw_gains;
	b->hw.cmd_pos = b->hw_cmd; b->hw.cmd_vel = b->hw_cmd + per; b->hw.cmd_eff = b->hw_cmd + 2 * per; b->hw.cmd_hold = b->hw_cmd + 3 * per;
	b->hw.pid = b->hw_pid;
	return MJB_OK;
}

int mjb_hwsim_set_command(mjb_batch *b, int which, int env_lo, int env_hi, const double *cmd)
{
	if (!b || !cmd || which < 0 || which > 2) return fail(MJB_EINVAL, "mjb_hwsim_set_command: bad argument");
	if (b->hw.n <= 0) return fail(MJB_EINVAL, "mjb_hwsim_set_command before mjb_hwsim_configure");
	if (env_lo < 0 || env_hi > b->nenv || env_lo > env_hi) return fail(MJB_EINVAL, "mjb_hwsim_set_command: bad env range");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	const size_t per = (size_t)b->nenv * b->hw.n;
	HIP_TRY(hipMemcpy(b->hw_cmd + which * per + (size_t)env_lo * b->hw.n, cmd, (size_t)(env_hi - env_lo) * b->hw.n * sizeof(double),
	                  hipMemcpyHostToDevice));
	return MJB_OK;
}

void *mjb_hwsim_command_ptr(mjb_batch *b, int which)
{
	if (!b || which < 0 || which > 2 || b->hw.n <= 0) return nullptr;
	return b->hw_cmd + (size_t)which * b->nenv * b->hw.n;
}

// MujocoRosControlPlugin::controlCallback's cadence (mujoco_ros_control_plugin.cpp:153-194) for the device-side stage: with a control
// period the joint state the PIDs see is the one sampled at the last controller update (readSim, every control_period of sim
// time; first at the first non-zero time), writeSim runs at every step after it with period = time - last write, and nothing is
// read or written at t = 0.  control_period <= 0 switches back to a write at every step on the step's own state.
int mjb_hwsim_set_period(mjb_batch *b, double control_period)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	if (b->hw.n <= 0) return fail(MJB_EINVAL, "mjb_hwsim_set_period before mjb_hwsim_configure");
	// (a period below the timestep is accepted with a warning, as the reference's load() does, :100-105 -- ROS_WARN and carry on: the
	//  cadence test `sim_period >= control_period` then holds at every step, i.e. the controller updates every step)
	if (control_period > 0 && control_period < b->model->h.timestep[0])
		fprintf(stderr, "mjb_hwsim_set_period: desired controller update period (%g s) is faster than the simulation timestep (%g s)\n",
		        control_period, b->model->h.timestep[0]);
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	const int n = b->hw.n;
	if (control_period > 0) {
		const size_t per = 2 + 2 * (size_t)n;
		std::vector<double> init((size_t)b->nenv * per, 0.0);
		for (int e = 0; e < b->nenv; e++)
			for (int k = 0; k < n; k++) init[(size_t)e * per + 2 + k] = 1.0;  // joint_position_ starts at 1.0 (default_robot_hw_sim.cpp:129)
		if (!b->hw_cad) b->hw_cad = dev_alloc<double>(init.size());
		if (!b->hw_cad) return fail(MJB_ENOMEM, "mjb_hwsim_set_period: allocation failed");
		HIP_TRY(hipMemcpy(b->hw_cad, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice));
		const double sec = std::floor(control_period);
		b->hw.period_ns = (long long)sec * 1000000000LL + (long long)std::floor((control_period - sec) * 1e9 + 0.5);
		b->hw.cad = b->hw_cad;
	} else {
		b->hw.period_ns = 0;
		b->hw.cad = nullptr;
	}
	b->params_dirty = true;
	return MJB_OK;
}

int mjb_hwsim_estop(mjb_batch *b, int active)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	if (b->hw.n <= 0) return fail(MJB_EINVAL, "mjb_hwsim_estop before mjb_hwsim_configure");
	const int on = active ? 1 : 0;
	if (on == b->hw.estop) return MJB_OK;
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (on) {  // position-controlled joints hold the command they had when the e-stop came (default_robot_hw_sim.cpp:251-258)
		const size_t per = (size_t)b->nenv * b->hw.n;
		HIP_TRY(hipMemcpy(b->hw_cmd + 3 * per, b->hw_cmd, per * sizeof(double), hipMemcpyDeviceToDevice));
	}
	b->hw.estop = on;
	b->params_dirty = true;
	return MJB_OK;
}

// ---- sensors-plugin equivalent (kernel in mjb_sensor_pack.hip) ----
static int sensor_buffers(mjb_batch *b)
{
	const mjb_model_desc &h = b->model->h;
	const size_t ns = (size_t)(h.nsensor > 0 ? h.nsensor : 1), nd = (size_t)b->nenv * (size_t)(h.nsensordata > 0 ? h.nsensordata : 1);
	if (b->sens_flag.empty()) {
		b->sens_flag.assign(ns, 0);
		b->sens_mean.assign(3 * ns, 0.0);
		b->sens_sigma.assign(3 * ns, 0.0);
	}
	if (!b->sens_value) {
		b->sens_flag_dev = dev_alloc<int>(ns);
		b->sens_mean_dev = dev_alloc<double>(3 * ns);
		b->sens_sigma_dev = dev_alloc<double>(3 * ns);
		b->sens_value = dev_alloc<float>(nd);
		b->sens_truth = dev_alloc<float>(nd);
		if (!b->sens_flag_dev || !b->sens_mean_dev || !b->sens_sigma_dev || !b->sens_value || !b->sens_truth)
			return fail(MJB_ENOMEM, "sensor message buffers: allocation failed");
		b->sens_dirty = true;
	}
	return MJB_OK;
}

int mjb_sensor_set_noise(mjb_batch *b, int sensor, int set_flag, const double *mean3, const double *sigma3)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	const mjb_model_desc &h = b->model->h;
	if (sensor < 0 || sensor >= h.nsensor) return fail(MJB_EINVAL, "mjb_sensor_set_noise: no sensor %d", sensor);
	if (set_flag < 0 || set_flag > 7 || (set_flag && (!mean3 || !sigma3))) return fail(MJB_EINVAL, "mjb_sensor_set_noise: bad argument");
	HIP_TRY(hipSetDevice(b->device));
	int rc = sensor_buffers(b);
	if (rc) return rc;
	// registerNoiseModelsCB: the n-th SET bit of this request takes mean[n] / std[n]; bits accumulate (is_set |= flag)
	int idx = 0;
	for (int k = 0; k < 3; k++)
		if (set_flag & (1 << k)) {
			b->sens_mean[3 * sensor + idx] = mean3[idx];
			b->sens_sigma[3 * sensor + idx] = sigma3[idx];
			idx++;
		}
	if (set_flag == 0) b->sens_flag[sensor] = 0;  // (extension: a zero flag clears the model)
	else b->sens_flag[sensor] |= set_flag;
	b->sens_dirty = true;
	return MJB_OK;
}

int mjb_sensor_pack(mjb_batch *b, uint64_t seed)
{
	if (!b) return fail(MJB_EINVAL, "null batch");
	const mjb_model_desc &h = b->model->h;
	HIP_TRY(hipSetDevice(b->device));
	int rc = sensor_buffers(b);
	if (rc) return rc;
	rc = sync_params(b);
	if (rc) return rc;
	if (b->sens_dirty) {
		HIP_TRY(hipStreamSynchronize(b->stream));
		HIP_TRY(hipMemcpy(b->sens_flag_dev, b->sens_flag.data(), b->sens_flag.size() * sizeof(int), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(b->sens_mean_dev, b->sens_mean.data(), b->sens_mean.size() * sizeof(double), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(b->sens_sigma_dev, b->sens_sigma.data(), b->sens_sigma.size() * sizeof(double), hipMemcpyHostToDevice));
		b->sens_dirty = false;
	}
	rc = mjb_launch_sensor_pack(b->params_dev, b->nenv, h.nsensor, b->sens_flag_dev, b->sens_mean_dev, b->sens_sigma_dev, seed,
	                            b->nz.env_offset, b->step_counter, b->sens_value, b->sens_truth, b->stream);
	if (rc != 0) return fail(MJB_ENODEVICE, "sensor pack launch failed: %s", hipGetErrorString((hipError_t)rc));
	b->sens_packed = true;
	return MJB_OK;
}

int mjb_sensor_get(mjb_batch *b, int which, int env_lo, int env_hi, float *host)
{
	if (!b || !host || which < 0 || which > 1) return fail(MJB_EINVAL, "mjb_sensor_get: bad argument");
	if (!b->sens_packed) return fail(MJB_EINVAL, "mjb_sensor_get before mjb_sensor_pack");
	if (env_lo < 0 || env_hi > b->nenv || env_lo > env_hi) return fail(MJB_EINVAL, "mjb_sensor_get: bad env range");
	const size_t S = (size_t)b->model->h.nsensordata;
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	const float *src = (which == 0 ? b->sens_value : b->sens_truth) + (size_t)env_lo * S;
	HIP_TRY(hipMemcpy(host, src, (size_t)(env_hi - env_lo) * S * sizeof(float), hipMemcpyDeviceToHost));
	return MJB_OK;
}

void *mjb_sensor_device_ptr(mjb_batch *b, int which)
{
	if (!b || which < 0 || which > 1 || sensor_buffers(b) != MJB_OK) return nullptr;
	return which == 0 ? (void *)b->sens_value : (void *)b->sens_truth;
}

int mjb_time_steps(mjb_batch *b, int nsteps, int nlaunch, double *ms_per_launch)
{
	if (!b || nsteps <= 0 || nlaunch <= 0 || !ms_per_launch) return fail(MJB_EINVAL, "mjb_time_steps: bad argument");
	HIP_TRY(hipSetDevice(b->device));
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipEventRecord(e0, b->stream));
	for (int i = 0; i < nlaunch; i++) {
		int rc = mjb_step(b, nsteps);
		if (rc) return rc;
	}
	HIP_TRY(hipEventRecord(e1, b->stream));
	HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	*ms_per_launch = (double)ms / nlaunch;
	return MJB_OK;
}

}  // extern "C"
