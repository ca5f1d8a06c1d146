// mjb_lane_env.hip -- the LANE = ENV form of the unconstrained fused step (SURVEY.md §7 "measure a 64-envs-per-wave (lane = env)
// variant for the smooth-dynamics phases and report both").
//
// mjb_step_kernel<16, 0, DENSE> spreads ONE env over 16 lanes: every stage is a chain of cross-lane exchanges (DPP, LDS rounds at
// ~110 cycles each) with a third of the lanes doing fp64 work, one wavefront per SIMD -- 42.8 k cycles per step whatever the batch
// size (profiles/r03_batch_size.txt: 222 - 230 M env-steps/s from 4096 to 65 536 envs).  Here a lane owns an env:
//   * no cross-lane traffic at all -- a step is one straight line of fp64 VALU work per wavefront of 64 envs, every fp64 issue slot
//     does 64 envs' worth of arithmetic;
//   * the model is the same for every lane, so every model constant is a SCALAR load into SGPRs (one per wavefront) and rides along
//     as the scalar operand of the VALU instruction that uses it;
//   * nothing per-env can be indexed at run time (a lane's "arrays" are registers), so the kernel is a template over the model's
//     INTEGER structure (csrc/lane_env_topos.h, tools/gen_lane_env_topo.py): parent ids, joint kinds and dof ancestry are
//     compile-time constants, every loop over bodies / dofs is unrolled, zero blocks of qM never exist.  Numeric constants stay
//     run-time data;
//   * the state of 64 envs lives in the wavefront's registers for all K fused steps (512 per lane at one wavefront per SIMD);
//     what does not fit spills to the AGPR half of the file and then to scratch -- `tools/kernel_meta.py` reports both.
// Same step as the oracle's mjo_step / mj_step (mujoco_env.cpp:498,552,593) for models the topology tables cover: hinge / slide
// trees without constraint rows, Euler with implicit joint damping, joint-transmission actuators, the sensors listed in
// gen_lane_env_topo.py, ctrl noise (mujoco_env.cpp:469-481), mj_check* resets, mjENBL_ENERGY.  Arithmetic follows MuJoCo's
// com-based formulation stage by stage (oracle/mjo_smooth.c); sums over bodies run in ascending instead of leaf-to-root order and
// reciprocals are Newton-refined hardware seeds, so results agree with the oracle to rounding, not bit for bit (tests/test_gpu_lane_env.py).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "mjb_lane_env_kernel.h"
#include "lane_env_topos.h"

namespace {

using namespace mjb_le;

template <class T, int LP>
__global__ void __launch_bounds__(64) mjb_lane_env_kernel(const KernelParams MJB_AS4 *__restrict__ P, const int nsteps, const unsigned int step0,
                                                          const int env_lo, const int env_hi)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_le[];
	lane_env_body<T, LP>(P, nsteps, step0, env_lo, env_hi, smem_le);
}

template <class T, int LP>
__global__ void __launch_bounds__(128) mjb_lane_env_duo_kernel(const KernelParams MJB_AS4 *__restrict__ P, const int nsteps, const unsigned int step0,
                                                               const int env_lo, const int env_hi)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_le[];
	lane_env_duo<T, LP>(P, nsteps, step0, env_lo, env_hi, smem_le);
}

template <class T>
__global__ void __launch_bounds__(128) mjb_lane_env_duo2_kernel(const KernelParams MJB_AS4 *__restrict__ P, const int nsteps, const unsigned int step0,
                                                                const int env_lo, const int env_hi)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_le[];
	lane_env_duo2<T, 160>(P, nsteps, step0, env_lo, env_hi, smem_le);
}

template <class T>
__global__ void __launch_bounds__(192) mjb_lane_env_trio_kernel(const KernelParams MJB_AS4 *__restrict__ P, const int nsteps, const unsigned int step0,
                                                                const int env_lo, const int env_hi)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_le[];
	lane_env_trio<T, 160>(P, nsteps, step0, env_lo, env_hi, smem_le);
}

template <class T> bool topo_matches(const mjb_model_desc &h)
{
	if (h.nbody != T::NBODY || h.nq != T::NQ || h.nv != T::NV || h.nu != T::NU || h.njnt != T::NJNT || h.nsite != T::NSITE ||
	    h.nsensor != T::NSENSOR || h.nsensordata != T::NSENSORDATA || h.nM != T::NM)
		return false;
	for (int b = 0; b < h.nbody; b++) {
		const int jn = h.body_jntnum[b] == 1 ? h.body_jntadr[b] : -1;
		if (h.body_jntnum[b] > 1 || h.body_parentid[b] != T::body_parentid[b] || h.body_rootid[b] != T::body_rootid[b] || jn != T::body_jnt[b] ||
		    (h.body_sameframe[b] != 0) != (T::body_sameframe[b] != 0))
			return false;
	}
	for (int j = 0; j < h.njnt; j++)
		if (h.jnt_type[j] != T::jnt_type[j] || h.jnt_bodyid[j] != T::jnt_bodyid[j] || h.jnt_qposadr[j] != j || h.jnt_dofadr[j] != j) return false;
	for (int d = 0; d < h.nv; d++)
		if (h.dof_parentid[d] != T::dof_parentid[d] || h.dof_Madr[d] != T::dof_Madr[d] || h.dof_jntid[d] != d) return false;
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
	return true;
}

}  // namespace

// Index of the compiled-in topology the model has, or -1 (mjb_compile; the generic kernels run every model).
int mjb_lane_env_match(const mjb_model_desc *h)
{
	if (!h || h->nefcmax > 0 || h->nconmax > 0 || h->integrator != MJB_INT_EULER || h->nmocap > 0 || h->ntendon > 0 || h->neq > 0 || h->na > 0 ||
	    h->nq != h->nv || h->njnt != h->nv)
		return -1;
#define MJB_LE_X(id, T) \
	if (topo_matches<T>(*h)) return id;
	MJB_LE_TOPOS(MJB_LE_X)
#undef MJB_LE_X
	return -1;
}

// ---- any other eligible model: the same template, compiled for the model's structure by hiprtc at its first fused launch ----
// (hiprtc is dlopen'ed: libmjb.so does not depend on it.  Whatever goes wrong -- no libhiprtc, the kernel header not next to the
//  library, a compile error -- the launch runs the generic kernel instead and the reason is kept for mjb_lane_env_info.)

// 1 when the model's structure fits the kernel (what tools/gen_lane_env_topo.py's eligible() accepts), whether or not a topology is compiled in
int mjb_lane_env_eligible(const mjb_model_desc *h)
{
	if (!h || h->nefcmax > 0 || h->nconmax > 0 || h->integrator != MJB_INT_EULER || h->nmocap > 0 || h->ntendon > 0 || h->neq > 0 || h->na > 0 ||
	    h->nq != h->nv || h->njnt != h->nv || h->nv < 1 || h->nbody > 24 || h->nv > 20)  // (size: ~22 doubles of state per body must fit the register file)
		return 0;
	for (int j = 0; j < h->njnt; j++)
		if ((h->jnt_type[j] != MJB_JNT_SLIDE && h->jnt_type[j] != MJB_JNT_HINGE) || h->jnt_qposadr[j] != j || h->jnt_dofadr[j] != j || h->dof_jntid[j] != j) return 0;
	for (int b = 0; b < h->nbody; b++)
		if (h->body_jntnum[b] > 1) return 0;
	for (int i = 0; i < h->nsensor; i++) {
		const int t = h->sensor_type[i], ot = h->sensor_objtype[i];
		const bool frame = t == MJB_SENS_FRAMEPOS || t == MJB_SENS_FRAMEQUAT;
		if (!(frame || t == MJB_SENS_JOINTPOS || t == MJB_SENS_JOINTVEL || t == MJB_SENS_ACTUATORFRC || t == MJB_SENS_ACTUATORPOS || t == MJB_SENS_ACTUATORVEL ||
		      t == MJB_SENS_CLOCK))
			return 0;
		if (frame && (h->sensor_refid[i] >= 0 || (ot != MJB_OBJ_BODY && ot != MJB_OBJ_XBODY && ot != MJB_OBJ_SITE))) return 0;
	}
	for (int i = 0; i < h->nu; i++)
		if (h->actuator_trntype[i] != MJB_TRN_JOINT || h->actuator_dyntype[i] != MJB_DYN_NONE) return 0;
	return 1;
}

namespace {

// the model's integer structure as the LeTopo struct the kernel template takes (the text doubles as the cache key)
std::string topo_source(const mjb_model_desc &h)
{
	std::string s = "struct LeTopo_rt {\n";
	auto scalar = [&](const char *n, int v) { s += std::string("\tstatic constexpr int ") + n + " = " + std::to_string(v) + ";\n"; };
	auto arr = [&](const char *n, int count, auto get) {
		s += std::string("\tstatic constexpr int ") + n + "[" + std::to_string(count > 0 ? count : 1) + "] = { ";
		for (int i = 0; i < (count > 0 ? count : 1); i++) s += (i ? ", " : "") + std::to_string(count > 0 ? (int)get(i) : 0);
		s += " };\n";
	};
	scalar("NBODY", h.nbody); scalar("NQ", h.nq); scalar("NV", h.nv); scalar("NU", h.nu); scalar("NJNT", h.njnt); scalar("NSITE", h.nsite);
	scalar("NSENSOR", h.nsensor); scalar("NSENSORDATA", h.nsensordata); scalar("NM", h.nM);
	arr("body_parentid", h.nbody, [&](int i) { return h.body_parentid[i]; });
	arr("body_rootid", h.nbody, [&](int i) { return h.body_rootid[i]; });
	arr("body_jnt", h.nbody, [&](int i) { return h.body_jntnum[i] == 1 ? h.body_jntadr[i] : -1; });
	arr("body_sameframe", h.nbody, [&](int i) { return h.body_sameframe[i] != 0; });
	arr("jnt_type", h.njnt, [&](int i) { return h.jnt_type[i]; });
	arr("jnt_bodyid", h.njnt, [&](int i) { return h.jnt_bodyid[i]; });
	arr("dof_parentid", h.nv, [&](int i) { return h.dof_parentid[i]; });
	arr("dof_Madr", h.nv, [&](int i) { return h.dof_Madr[i]; });
	arr("act_jnt", h.nu, [&](int i) { return h.actuator_trnid[2 * i]; });
	arr("act_gaintype", h.nu, [&](int i) { return h.actuator_gaintype[i]; });
	arr("act_biastype", h.nu, [&](int i) { return h.actuator_biastype[i]; });
	arr("act_ctrllimited", h.nu, [&](int i) { return h.actuator_ctrllimited[i] != 0; });
	arr("act_forcelimited", h.nu, [&](int i) { return h.actuator_forcelimited[i] != 0; });
	arr("site_bodyid", h.nsite, [&](int i) { return h.site_bodyid[i]; });
	arr("site_sameframe", h.nsite, [&](int i) { return h.site_sameframe[i] != 0; });
	arr("sensor_type", h.nsensor, [&](int i) { return h.sensor_type[i]; });
	arr("sensor_objtype", h.nsensor, [&](int i) { return h.sensor_objtype[i]; });
	arr("sensor_objid", h.nsensor, [&](int i) { return h.sensor_objid[i]; });
	arr("sensor_adr", h.nsensor, [&](int i) { return h.sensor_adr[i]; });
	s += "};\n";
	return s;
}

struct RtcApi {
	void *lib = nullptr;
	int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
	int (*compile)(void *, int, const char **) = nullptr;
	int (*log_size)(void *, size_t *) = nullptr;
	int (*log)(void *, char *) = nullptr;
	int (*code_size)(void *, size_t *) = nullptr;
	int (*code)(void *, char *) = nullptr;
	int (*destroy)(void **) = nullptr;
	bool ok = false;
};
RtcApi &rtc()
{
	static RtcApi a = [] {
		RtcApi r;
		for (const char *n : { "libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so" }) {
			r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
			if (r.lib) break;
		}
		if (!r.lib) return r;
		r.create = (decltype(r.create))dlsym(r.lib, "hiprtcCreateProgram");
		r.compile = (decltype(r.compile))dlsym(r.lib, "hiprtcCompileProgram");
		r.log_size = (decltype(r.log_size))dlsym(r.lib, "hiprtcGetProgramLogSize");
		r.log = (decltype(r.log))dlsym(r.lib, "hiprtcGetProgramLog");
		r.code_size = (decltype(r.code_size))dlsym(r.lib, "hiprtcGetCodeSize");
		r.code = (decltype(r.code))dlsym(r.lib, "hiprtcGetCode");
		r.destroy = (decltype(r.destroy))dlsym(r.lib, "hiprtcDestroyProgram");
		r.ok = r.create && r.compile && r.log_size && r.log && r.code_size && r.code && r.destroy;
		return r;
	}();
	return a;
}

struct JitKernel {
	hipModule_t mod = nullptr;
	hipFunction_t fn = nullptr;
	std::string error;  // non-empty: not available, why
};
std::mutex jit_mutex;
std::map<std::string, JitKernel> jit_cache;  // key: device | LDS budget | topology text
std::string jit_last_error;  // guarded by jit_err_mutex; mjb_lane_env_jit_error() hands out a thread-local copy
std::mutex jit_err_mutex;
void set_jit_error(const std::string &e)
{
	std::lock_guard<std::mutex> lock(jit_err_mutex);
	jit_last_error = e;
}
// gfx target of the device the kernel is built for (the in-tree build's $(ARCH) is gfx950; a hiprtc build follows the device it runs on)
std::string device_arch(int dev)
{
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, dev) != hipSuccess) return "gfx950";
	std::string a = p.gcnArchName;
	const size_t c = a.find(':');
	return c == std::string::npos ? a : a.substr(0, c);
}

// directory of the kernel header: next to libmjb.so (the in-tree build), or MJB_LANE_ENV_SRC
std::string source_dir()
{
	if (const char *v = getenv("MJB_LANE_ENV_SRC")) return v;
	Dl_info info;
	if (dladdr((const void *)&mjb_lane_env_eligible, &info) && info.dli_fname) {
		std::string p = info.dli_fname;
		const size_t k = p.rfind('/');
		return k == std::string::npos ? "." : p.substr(0, k);
	}
	return ".";
}

// ---- on-disk cache of the hiprtc builds: a new model's first eligible launch costs ~3 s of compilation per LDS budget and form; the code
// object is kept under $MJB_JIT_CACHE (default $XDG_CACHE_HOME/mjb_jit or ~/.cache/mjb_jit; MJB_JIT_CACHE=0 turns it off) in a file named by a hash of
// everything the build depends on: gfx arch, the TEXT of every header the source includes (the source fingerprint), LDS budget, form, the topology.
std::atomic<int> jit_compiled{ 0 }, jit_disk_hits{ 0 };
unsigned long long fnv1a(unsigned long long hsh, const std::string &t)
{
	for (unsigned char c : t) {
		hsh ^= c;
		hsh *= 1099511628211ull;
	}
	return hsh;
}
bool read_file(const std::string &path, std::string &out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return false;
	char buf[65536];
	size_t n;
	out.clear();
	while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
	fclose(f);
	return true;
}
std::string cache_dir()
{
	if (const char *v = getenv("MJB_JIT_CACHE")) return (*v == '0' && !v[1]) ? std::string() : std::string(v);
	if (const char *x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/mjb_jit";
	if (const char *hm = getenv("HOME")) if (*hm) return std::string(hm) + "/.cache/mjb_jit";
	return "/tmp/mjb_jit";
}
void make_dirs(const std::string &d)
{
	for (size_t k = 1; k <= d.size(); k++)
		if (k == d.size() || d[k] == '/') (void)mkdir(d.substr(0, k).c_str(), 0755);
}
// hash of the headers the generated source pulls in (a changed kernel header or data-field list invalidates every cached build)
bool source_fingerprint(const std::string &dir, unsigned long long &hsh)
{
	static const char *files[] = { "/mjb_lane_env_kernel.h", "/mjb_dev.h", "/mjb_math.h", "/../../include/mjb.h", "/../../include/mjb_model_fields.def",
		                           "/../../include/mjb_data_fields.def" };
	hsh = 1469598103934665603ull;
	std::string text;
	for (const char *f : files) {
		if (!read_file(dir + f, text)) return false;
		hsh = fnv1a(hsh, text);
	}
	return true;
}

const JitKernel &jit_get(const mjb_model_desc &h, int lp, int duo)  // duo: 0 solo, 1 two halves, 2 pipelined, 3 three wavefronts
{
	int dev = 0;
	(void)hipGetDevice(&dev);
	const std::string topo = topo_source(h);
	const std::string key = std::to_string(dev) + "|" + std::to_string(lp) + (duo == 3 ? "t|" : (duo == 2 ? "p|" : (duo ? "d|" : "|"))) + topo;
	std::lock_guard<std::mutex> lock(jit_mutex);
	auto it = jit_cache.find(key);
	if (it != jit_cache.end()) return it->second;
	JitKernel &k = jit_cache[key];
	static const bool off = [] { const char *v = getenv("MJB_LANE_ENV_JIT"); return v && *v == '0'; }();
	if (off) { k.error = "disabled (MJB_LANE_ENV_JIT=0)"; return k; }
	RtcApi &r = rtc();
	if (!r.ok) { k.error = "libhiprtc.so not found"; return k; }
	const std::string dir = source_dir();
	const std::string slp = std::to_string(lp);
	const std::string src = "#include \"mjb_lane_env_kernel.h\"\n" + topo +
	                        (duo == 3 ? "extern \"C\" __global__ void __launch_bounds__(192) le_rt(const KernelParams MJB_AS4 *P, int nsteps, unsigned int step0, int lo, int hi)\n{\n"
	                                    "\t__shared__ __attribute__((aligned(16))) unsigned char smem[mjb_le::trio_bytes<LeTopo_rt>()];\n"
	                                    "\tmjb_le::lane_env_trio<LeTopo_rt, 160>(P, nsteps, step0, lo, hi, smem);\n}\n"
	                         : duo == 2 ? "extern \"C\" __global__ void __launch_bounds__(128) le_rt(const KernelParams MJB_AS4 *P, int nsteps, unsigned int step0, int lo, int hi)\n{\n"
	                                    "\t__shared__ __attribute__((aligned(16))) unsigned char smem[mjb_le::duo2_bytes<LeTopo_rt>()];\n"
	                                    "\tmjb_le::lane_env_duo2<LeTopo_rt, 160>(P, nsteps, step0, lo, hi, smem);\n}\n"
	                         : duo ? "extern \"C\" __global__ void __launch_bounds__(128) le_rt(const KernelParams MJB_AS4 *P, int nsteps, unsigned int step0, int lo, int hi)\n{\n"
	                               "\t__shared__ __attribute__((aligned(16))) unsigned char smem[mjb_le::duo_bytes<LeTopo_rt, " + slp + ">()];\n"
	                               "\tmjb_le::lane_env_duo<LeTopo_rt, " + slp + ">(P, nsteps, step0, lo, hi, smem);\n}\n"
	                             : "extern \"C\" __global__ void __launch_bounds__(64) le_rt(const KernelParams MJB_AS4 *P, int nsteps, unsigned int step0, int lo, int hi)\n{\n"
	                               "\t__shared__ __attribute__((aligned(16))) unsigned char smem[mjb_le::Lds<LeTopo_rt, " + slp + ">::bytes()];\n"
	                               "\tmjb_le::lane_env_body<LeTopo_rt, " + slp + ">(P, nsteps, step0, lo, hi, smem);\n}\n");
	// the disk cache first
	std::string cpath;
	{
		const std::string cdir = cache_dir();
		unsigned long long fp = 0;
		if (!cdir.empty() && source_fingerprint(dir, fp)) {
			unsigned long long hk = fnv1a(fnv1a(fnv1a(fp, device_arch(dev)), src), "O3 fp-contract=fast no-machine-licm v1");
			char name[64];
			snprintf(name, sizeof name, "/le_%016llx.hsaco", hk);
			cpath = cdir + name;
			std::string blob;
			if (read_file(cpath, blob) && blob.size() > 64) {
				if (hipModuleLoadData(&k.mod, blob.data()) == hipSuccess && hipModuleGetFunction(&k.fn, k.mod, "le_rt") == hipSuccess) {
					jit_disk_hits++;
					return k;
				}
				(void)hipGetLastError();  // (a truncated or foreign file: rebuilt and overwritten below)
				k.mod = nullptr;
				k.fn = nullptr;
			}
		}
	}
	void *prog = nullptr;
	if (r.create(&prog, src.c_str(), "mjb_lane_env_rt.hip", 0, nullptr, nullptr) != 0) { k.error = "hiprtcCreateProgram failed"; return k; }
	const std::string i1 = "-I" + dir, i2 = "-I" + dir + "/../../include";
	const std::string archopt = "--offload-arch=" + device_arch(dev);
	const char *opts[] = { archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=fast", "-mllvm", "-disable-machine-licm", i1.c_str(), i2.c_str() };
	const int crc = r.compile(prog, (int)(sizeof opts / sizeof *opts), opts);
	if (crc != 0) {
		size_t n = 0;
		r.log_size(prog, &n);
		std::string log(n + 1, '\0');
		if (n) r.log(prog, &log[0]);
		k.error = "hiprtc compile failed: " + log.substr(0, 600);
		r.destroy(&prog);
		return k;
	}
	size_t n = 0;
	r.code_size(prog, &n);
	std::vector<char> code(n);
	r.code(prog, code.data());
	r.destroy(&prog);
	jit_compiled++;
	if (!cpath.empty()) {  // (written under a temporary name and renamed: a concurrent process never reads half a file)
		make_dirs(cpath.substr(0, cpath.rfind('/')));
		const std::string tmp = cpath + "." + std::to_string((long long)getpid()) + ".tmp";
		FILE *f = fopen(tmp.c_str(), "wb");
		if (f) {
			const bool okw = fwrite(code.data(), 1, code.size(), f) == code.size();
			fclose(f);
			if (!okw || rename(tmp.c_str(), cpath.c_str()) != 0) (void)remove(tmp.c_str());
		}
	}
	if (hipModuleLoadData(&k.mod, code.data()) != hipSuccess || hipModuleGetFunction(&k.fn, k.mod, "le_rt") != hipSuccess) {
		(void)hipGetLastError();
		k.error = "hipModuleLoadData / hipModuleGetFunction failed";
		k.mod = nullptr;
		k.fn = nullptr;
	}
	return k;
}

}  // namespace

namespace {
// CU count and the kernels' LDS attribute, cached per device (hipFuncSetAttribute acts on the current device's copy of the function)
int device_cus(int dev)
{
	static std::mutex mu;
	static std::map<int, int> cus;
	std::lock_guard<std::mutex> lock(mu);
	auto it = cus.find(dev);
	if (it != cus.end()) return it->second;
	int c = 0;
	if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 0;
	cus[dev] = c;
	return c;
}
hipError_t lds_attr_once(const void *fn, int bytes, int dev)
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

void mjb_lane_env_jit_stats(int *compiled, int *disk_hits)
{
	if (compiled) *compiled = jit_compiled.load();
	if (disk_hits) *disk_hits = jit_disk_hits.load();
}

const char *mjb_lane_env_jit_error(void)
{
	thread_local std::string copy;
	std::lock_guard<std::mutex> lock(jit_err_mutex);
	copy = jit_last_error;
	return copy.c_str();
}

// process-wide choice of the kernel's FORM (include/mjb.h): -1 = by batch size (default; MJB_LANE_ENV_DUO overrides), 0 / 1 / 2
static std::atomic<int> le_form_override{ -1 }, le_form_last{ -1 };
int mjb_lane_env_set_form(int form)
{
	return le_form_override.exchange((form >= 0 && form <= 3) ? form : -1);
}
int mjb_lane_env_last_form(void) { return le_form_last; }

size_t mjb_lane_env_tape_doubles(const mjb_model_desc *h)
{
	return (sizeof(LeTapeHdr) + (size_t)h->nbody * sizeof(LeTapeBody) + (size_t)h->nu * sizeof(LeTapeAct)) / sizeof(double);
}

void mjb_lane_env_tape(const mjb_model_desc *h, double *tape)
{
	LeTapeHdr *th = reinterpret_cast<LeTapeHdr *>(tape);
	LeTapeBody *tb = reinterpret_cast<LeTapeBody *>(th + 1);
	LeTapeAct *ta = reinterpret_cast<LeTapeAct *>(tb + h->nbody);
	th->dt = h->timestep[0];
	for (int k = 0; k < 3; k++) th->gravity[k] = h->gravity[k];
	for (int b = 0; b < h->nbody; b++) {
		LeTapeBody &t = tb[b];
		for (int k = 0; k < 3; k++) { t.pos[k] = h->body_pos[3 * b + k]; t.ipos[k] = h->body_ipos[3 * b + k]; }
		for (int k = 0; k < 4; k++) t.quat[k] = h->body_quat[4 * b + k];
		{  // Ib = R(iquat) diag(inertia) R(iquat)': xx yy zz xy xz yz
			const double *q = h->body_iquat + 4 * b, *in = h->body_inertia + 3 * b;
			double R[9];
			const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3],
			             q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
			R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
			R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02); R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01); R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
			auto e = [&](int r, int c) { return R[3 * r] * in[0] * R[3 * c] + R[3 * r + 1] * in[1] * R[3 * c + 1] + R[3 * r + 2] * in[2] * R[3 * c + 2]; };
			t.ibody[0] = e(0, 0); t.ibody[1] = e(1, 1); t.ibody[2] = e(2, 2); t.ibody[3] = e(0, 1); t.ibody[4] = e(0, 2); t.ibody[5] = e(1, 2);
		}
		t.mass = h->body_mass[b];
		if (h->body_jntnum[b] == 1) {
			const int j = h->body_jntadr[b];
			for (int k = 0; k < 3; k++) { t.jaxis[k] = h->jnt_axis[3 * j + k]; t.jpos[k] = h->jnt_pos[3 * j + k]; }
			const int qa = h->jnt_qposadr[j], da = h->jnt_dofadr[j];  // (== j for the lane = env topologies; the split step's smooth kernel reads the tape of models with free / ball joints too: their first coordinate)
			t.qpos0 = h->qpos0[qa];
			t.stiffness = h->jnt_stiffness[j];
			t.spring = h->qpos_spring[qa];
			t.damping = h->dof_damping[da];
			t.armature = h->dof_armature[da];
			t.hdamping = h->timestep[0] * h->dof_damping[da];
		}
	}
	for (int i = 0; i < h->nu; i++) {
		LeTapeAct &a = ta[i];
		a.gear = h->actuator_gear[6 * i];
		a.ctrllo = h->actuator_ctrlrange[2 * i];
		a.ctrlhi = h->actuator_ctrlrange[2 * i + 1];
		for (int k = 0; k < 3; k++) { a.gain[k] = h->actuator_gainprm[3 * i + k]; a.bias[k] = h->actuator_biasprm[3 * i + k]; }
		a.forcelo = h->actuator_forcerange[2 * i];
		a.forcehi = h->actuator_forcerange[2 * i + 1];
	}
}

const char *mjb_lane_env_name(int topo)
{
#define MJB_LE_X(id, T) \
	if (topo == id) return T::name;
	MJB_LE_TOPOS(MJB_LE_X)
#undef MJB_LE_X
	return "";
}

// topo >= 0: a compiled-in topology; MJB_LE_TOPO_JIT: the model's own, compiled by hiprtc on first use (h = the model; returns
// MJB_LE_UNAVAILABLE when that is not possible: the caller launches the generic kernel)
int mjb_launch_lane_env(const KernelParams *Pdev, int topo, const mjb_model_desc *h, int nenv_batch, int env_lo, int env_hi, int nsteps, unsigned int step0,
                        void *stream)
{
	const int n = env_hi - env_lo;
	if (n <= 0) return 0;
	// (measurement knob: envs per wavefront < 64 -- the wavefront starts with its upper lanes off)
	static const int wl = [] { const char *v = getenv("MJB_LANE_ENV_WAVE_LANES"); const int k = v ? atoi(v) : 64; return (k == 16 || k == 32) ? k : 64; }();
	const dim3 grid((unsigned int)((n + wl - 1) / wl)), block(wl);
	// LDS budget per wavefront from the CUs the launch leaves idle: one wavefront per CU may take all of its LDS
	// (per DEVICE: a process may drive several GPUs, one batch each -- ADVICE r05)
	int cur_dev = 0;
	(void)hipGetDevice(&cur_dev);
	const int ncu = device_cus(cur_dev);
	static const int forced = [] { const char *v = getenv("MJB_LANE_ENV_LDS_KB"); return v ? atoi(v) : 0; }();  // measurement knob: 40 / 80 / 160
	// (from the BATCH's size, not the launch's env range: the instantiations differ in where data waits, and the compiler contracts a
	//  few multiply-adds differently around that -- results agree to rounding, not bit for bit -- so every launch of one batch, whole or
	//  a prefix / the rest of a split step, runs the same one)
	const int waves = (nenv_batch + 63) / 64;
	int lp = (ncu > 0 && waves <= ncu) ? 160 : ((ncu > 0 && waves <= 2 * ncu) ? 80 : 40);
	if (forced == 40 || forced == 80 || forced == 160) lp = forced;
	// DUO: two wavefronts per 64 envs (mjb_lane_env_kernel.h) while the batch leaves at least every second SIMD idle -- the step's
	// position half and velocity half side by side.  MJB_LANE_ENV_DUO=0 / 1: never / whenever the LDS budget allows (measurement knob)
	static const int duo_env = [] { const char *v = getenv("MJB_LANE_ENV_DUO"); return v ? atoi(v) : -1; }();
	const int form_ov = le_form_override.load();
	const int duo_mode = form_ov >= 0 ? form_ov : duo_env;
	static const int duo_max_waves = [] { const char *v = getenv("MJB_LANE_ENV_DUO_MAX_WAVES"); return v ? atoi(v) : -1; }();
	int duo = (wl == 64 && lp >= 80 && duo_mode != 0 && (duo_mode > 0 || waves <= (duo_max_waves >= 0 ? duo_max_waves : 2 * ncu))) ? 1 : 0;
	// ... pipelined (nothing computed twice) when a workgroup has a CU's LDS to itself.  MJB_LANE_ENV_DUO=1: the two-halves form only
	if (duo && lp == 160 && duo_mode != 1) duo = 2;
	// ... and THREE wavefronts (the pose chain alone on the first) while three SIMDs per block are there: the same LDS layout, so the same bound
	if (duo == 2 && duo_mode != 2 && 3 * waves <= 4 * ncu) duo = 3;
	if (duo_mode > 0 && lp < 80) lp = 80, duo = 1;  // (a forced two-wavefront form on a batch that would run four wavefronts per CU: two per CU)
	if (topo == MJB_LE_TOPO_JIT) {
		if (!h) return (int)hipErrorInvalidValue;
		{  // a larger model than the compiled-in ones: the smallest LDS budget its (qpos, qvel) pairs and body forces fit (fewer wavefronts per CU then)
			int need = h->nv;
			for (int b = 1; b < h->nbody; b++) {
				bool moves = false;
				for (int a = b; a > 0; a = h->body_parentid[a]) moves = moves || h->body_jntnum[a] > 0;
				if (moves) need += 3;
			}
			const int fit = need <= 40 ? 40 : (need <= 80 ? 80 : 160);
			if (need > 160) {
				set_jit_error("the model needs more than 160 KB of LDS per wavefront");
				return MJB_LE_UNAVAILABLE;
			}
			if (fit > lp) lp = fit;
			if (duo == 3 && need + 5 * ((need - h->nv) / 3) + 18 + (h->nv + 1) / 2 + 1 + h->nbody > 160) duo = 2;
			if (duo == 2 && need + 5 * ((need - h->nv) / 3) + 12 + (h->nv + 1) / 2 + 1 > 160) duo = 1;
			if (duo == 1 && need + (h->nv + 1) / 2 + 1 > lp) duo = 0;
		}
		const JitKernel &k = jit_get(*h, lp, duo);
		if (!k.fn) {
			set_jit_error(k.error);
			return MJB_LE_UNAVAILABLE;
		}
		const KernelParams MJB_AS4 *Pd = (const KernelParams MJB_AS4 *)Pdev;
		int a_nsteps = nsteps, a_lo = env_lo, a_hi = env_hi;
		unsigned int a_step0 = step0;
		void *args[] = { (void *)&Pd, (void *)&a_nsteps, (void *)&a_step0, (void *)&a_lo, (void *)&a_hi };
		le_form_last = duo;
		return (int)hipModuleLaunchKernel(k.fn, grid.x, 1, 1, duo == 3 ? 192 : (duo ? 128 : block.x), 1, 1, 0, (hipStream_t)stream, args, nullptr);
	}
#define MJB_LE_GO(T, LPV)                                                                                                                    \
	{                                                                                                                                         \
		auto kern = mjb_lane_env_kernel<T, LPV>;                                                                                              \
		constexpr int bytes = Lds<T, LPV>::bytes();                                                                                           \
		const hipError_t attr = lds_attr_once(reinterpret_cast<const void *>(kern), bytes, cur_dev);                                          \
		if (attr != hipSuccess) return (int)attr;                                                                                             \
		le_form_last = 0;                                                                                                                     \
		hipLaunchKernelGGL(kern, grid, block, bytes, (hipStream_t)stream, (const KernelParams MJB_AS4 *)Pdev, nsteps, step0, env_lo, env_hi); \
		return (int)hipGetLastError();                                                                                                        \
	}
#define MJB_LE_GO_DUO(T, LPV)                                                                                                                \
	{                                                                                                                                         \
		auto kern = mjb_lane_env_duo_kernel<T, LPV>;                                                                                          \
		constexpr int bytes = duo_bytes<T, LPV>();                                                                                            \
		const hipError_t attr = lds_attr_once(reinterpret_cast<const void *>(kern), bytes, cur_dev);                                          \
		if (attr != hipSuccess) return (int)attr;                                                                                             \
		le_form_last = 1;                                                                                                                     \
		hipLaunchKernelGGL(kern, grid, dim3(128), bytes, (hipStream_t)stream, (const KernelParams MJB_AS4 *)Pdev, nsteps, step0, env_lo, env_hi); \
		return (int)hipGetLastError();                                                                                                        \
	}
#define MJB_LE_GO_DUO2(T)                                                                                                                   \
	if constexpr (duo2_bytes<T>() <= 160 * 1024) {                                                                                            \
		auto kern = mjb_lane_env_duo2_kernel<T>;                                                                                              \
		constexpr int bytes = duo2_bytes<T>();                                                                                                \
		const hipError_t attr = lds_attr_once(reinterpret_cast<const void *>(kern), bytes, cur_dev);                                          \
		if (attr != hipSuccess) return (int)attr;                                                                                             \
		le_form_last = 2;                                                                                                                     \
		hipLaunchKernelGGL(kern, grid, dim3(128), bytes, (hipStream_t)stream, (const KernelParams MJB_AS4 *)Pdev, nsteps, step0, env_lo, env_hi); \
		return (int)hipGetLastError();                                                                                                        \
	}
#define MJB_LE_GO_TRIO(T)                                                                                                                   \
	if constexpr (trio_bytes<T>() <= 160 * 1024) {                                                                                            \
		auto kern = mjb_lane_env_trio_kernel<T>;                                                                                              \
		constexpr int bytes = trio_bytes<T>();                                                                                                \
		const hipError_t attr = lds_attr_once(reinterpret_cast<const void *>(kern), bytes, cur_dev);                                          \
		if (attr != hipSuccess) return (int)attr;                                                                                             \
		le_form_last = 3;                                                                                                                     \
		hipLaunchKernelGGL(kern, grid, dim3(192), bytes, (hipStream_t)stream, (const KernelParams MJB_AS4 *)Pdev, nsteps, step0, env_lo, env_hi); \
		return (int)hipGetLastError();                                                                                                        \
	}
#define MJB_LE_X(id, T)                      \
	if (topo == id) {                        \
		if (duo == 3) MJB_LE_GO_TRIO(T)      \
		if (duo >= 2) MJB_LE_GO_DUO2(T)      \
		if (duo && lp == 160) MJB_LE_GO_DUO(T, 160) \
		if (duo && lp == 80) MJB_LE_GO_DUO(T, 80)   \
		if (lp == 160) MJB_LE_GO(T, 160)     \
		if (lp == 80) MJB_LE_GO(T, 80)       \
		MJB_LE_GO(T, 40)                     \
	}
	MJB_LE_TOPOS(MJB_LE_X)
#undef MJB_LE_X
#undef MJB_LE_GO
#undef MJB_LE_GO_DUO
#undef MJB_LE_GO_DUO2
#undef MJB_LE_GO_TRIO
	return (int)hipErrorInvalidValue;
}
