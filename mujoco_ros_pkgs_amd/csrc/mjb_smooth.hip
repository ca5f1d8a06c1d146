// mjb_smooth.hip -- host side of the split step's smooth kernel (mjb_smooth_kernel.h): the compiled-in topologies (csrc/smooth_topos.h,
// tools/gen_lane_env_topo.py), the match of a model against them, the launch.  VERDICT r05 #1; SURVEY.md §8a rows A1 - A3, A8 - A9, A12.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include "mjb_smooth_kernel.h"
#include "smooth_topos.h"

namespace {

using namespace mjb_sm;

template <class T>
__global__ void __launch_bounds__(64) mjb_smooth_kernel(const KernelParams MJB_AS4 *__restrict__ P, const unsigned int step, const int flags, const int env_lo,
                                                        const int env_hi)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_sm[];
	smooth_lane_env<T>(P, step, flags, env_lo, env_hi, smem_sm);
}

template <class T> bool sm_matches(const mjb_model_desc &h)
{
	if (h.nbody != T::NBODY || h.nq != T::NQ || h.nv != T::NV || h.nu != T::NU || h.njnt != T::NJNT || h.ngeom != T::NGEOM || h.nsite != T::NSITE ||
	    h.nsensor != T::NSENSOR || h.nsensordata != T::NSENSORDATA || h.nM != T::NM)
		return false;
	for (int b = 0; b < h.nbody; b++) {
		const int jn = h.body_jntnum[b] == 1 ? h.body_jntadr[b] : -1;
		if (h.body_jntnum[b] > 1 || h.body_parentid[b] != T::body_parentid[b] || h.body_rootid[b] != T::body_rootid[b] || jn != T::body_jnt[b] ||
		    (h.body_sameframe[b] != 0) != (T::body_sameframe[b] != 0))
			return false;
	}
	for (int j = 0; j < h.njnt; j++)
		if (h.jnt_type[j] != T::jnt_type[j] || h.jnt_bodyid[j] != T::jnt_bodyid[j] || h.jnt_qposadr[j] != T::jnt_qposadr[j] || h.jnt_dofadr[j] != T::jnt_dofadr[j]) return false;
	for (int d = 0; d < h.nv; d++)
		if (h.dof_parentid[d] != T::dof_parentid[d] || h.dof_Madr[d] != T::dof_Madr[d]) return false;
	for (int g = 0; g < h.ngeom; g++)
		if (h.geom_bodyid[g] != T::geom_bodyid[g] || (h.geom_sameframe[g] != 0) != (T::geom_sameframe[g] != 0)) return false;
	for (int i = 0; i < h.nu; i++)
		if (h.actuator_trntype[i] != MJB_TRN_JOINT || h.actuator_dyntype[i] != MJB_DYN_NONE || h.actuator_trnid[2 * i] != T::act_jnt[i] ||
		    h.actuator_gaintype[i] != T::act_gaintype[i] || h.actuator_biastype[i] != T::act_biastype[i] ||
		    (h.actuator_ctrllimited[i] != 0) != (T::act_ctrllimited[i] != 0) || (h.actuator_forcelimited[i] != 0) != (T::act_forcelimited[i] != 0))
			return false;
	for (int i = 0; i < h.nsite; i++)
		if (h.site_bodyid[i] != T::site_bodyid[i] || (h.site_sameframe[i] != 0) != (T::site_sameframe[i] != 0)) return false;
	for (int i = 0; i < h.nsensor; i++)
		if (h.sensor_type[i] != T::sensor_type[i] || h.sensor_objtype[i] != T::sensor_objtype[i] || h.sensor_objid[i] != T::sensor_objid[i] ||
		    h.sensor_adr[i] != T::sensor_adr[i] || h.sensor_refid[i] >= 0)
			return false;
	// the hand-off offsets the generator wrote are the ones the constraint kernel computes (mjb_dev.h)
	const HandoffLayout hl = mjb_handoff_layout(h.ngeom, h.nv, h.nbody, h.nM);
	return hl.geom_xpos == T::H_GEOM_XPOS && hl.geom_xmat == T::H_GEOM_XMAT && hl.cdof == T::H_CDOF && hl.subtree_com == T::H_SUBTREE_COM && hl.qLD == T::H_QLD &&
	       hl.qLDiagInv == T::H_QLDIAGINV && hl.qH == T::H_QH && hl.qHdi == T::H_QHDI && hl.qfrc_smooth == T::H_QFRC_SMOOTH && hl.qacc_smooth == T::H_QACC_SMOOTH;
}

hipError_t sm_lds_attr(const void *fn, int bytes, int dev)
{
	if (bytes <= 65536) return hipSuccess;
	static std::mutex mu;
	static std::map<std::pair<const void *, int>, hipError_t> done;
	std::lock_guard<std::mutex> lock(mu);
	auto key = std::make_pair(fn, dev);
	auto it = done.find(key);
	if (it != done.end()) return it->second;
	const hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
	done[key] = r;
	return r;
}

}  // namespace

// Index of the compiled-in topology whose smooth stages the kernel covers, or -1.  (What the CONSTRAINT half needs of the model -- PGS with
// pyramidal cones, nv <= 16, the lean frame eight to a CU -- is the caller's test: mjb_api.hip, split_eligible.)
int mjb_smooth_match(const mjb_model_desc *h)
{
	if (!h || h->integrator != MJB_INT_EULER || h->nmocap > 0 || h->ntendon > 0 || h->neq > 0 || h->na > 0) return -1;
#define MJB_SM_X(id, T) \
	if (sm_matches<T>(*h)) return id;
	MJB_SM_TOPOS(MJB_SM_X)
#undef MJB_SM_X
	return -1;
}

const char *mjb_smooth_name(int topo)
{
#define MJB_SM_X(id, T) \
	if (topo == id) return T::name;
	MJB_SM_TOPOS(MJB_SM_X)
#undef MJB_SM_X
	return "";
}

int mjb_launch_smooth(const KernelParams *Pdev, int topo, int env_lo, int env_hi, unsigned int step, int flags, void *stream)
{
	const int n = env_hi - env_lo;
	if (n <= 0) return 0;
	int dev = 0;
	(void)hipGetDevice(&dev);
	const dim3 grid((unsigned int)((n + 63) / 64)), block(64);
#define MJB_SM_X(id, T)                                                                                                                  \
	if (topo == id) {                                                                                                                    \
		auto kern = mjb_smooth_kernel<T>;                                                                                                \
		constexpr int bytes = Sq<T>::bytes();                                                                                            \
		static_assert(bytes <= 160 * 1024, "smooth kernel: cdof, forces and inertias of the topology need more than a CU's LDS");       \
		const hipError_t attr = sm_lds_attr(reinterpret_cast<const void *>(kern), bytes, dev);                                           \
		if (attr != hipSuccess) return (int)attr;                                                                                        \
		hipLaunchKernelGGL(kern, grid, block, bytes, (hipStream_t)stream, (const KernelParams MJB_AS4 *)Pdev, step, flags, env_lo, env_hi); \
		return (int)hipGetLastError();                                                                                                   \
	}
	MJB_SM_TOPOS(MJB_SM_X)
#undef MJB_SM_X
	return (int)hipErrorInvalidValue;
}
