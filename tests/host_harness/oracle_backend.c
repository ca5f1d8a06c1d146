/* oracle_backend.c — TEST HARNESS ONLY.  An mjr_backend (include/mjr_host.h) whose stepper is the CPU oracle
 * (oracle/libmjo.so), so the host runtime's scheduler / plugin / request logic (libmjr_host.so) can be
 * exercised by `-m "not gpu"` tests.  The product never links this: MujocoEnv's default backend factory is
 * mjr_make_mjb_backend (HIP) and it fails loudly without a GPU. */
#include <stdlib.h>
#include <string.h>

#include "../../include/mjr_host.h"
#include "../../oracle/mjo.h"

typedef struct {
	mjb_model_desc desc; /* shallow: the host runtime keeps the arrays alive for the backend's lifetime */
	int nenv;
	mjo_data **d;
	double std, rate;
	uint64_t seed;
	int64_t off;
	unsigned step;
	mjr_backend vt;
} ob;

static int ob_nenv(void *s) { return ((ob *)s)->nenv; }
static int ob_field_size(void *s, int f)
{
	ob *b = (ob *)s;
	int n = 0;
	if (mjo_field(&b->desc, b->d[0], f, &n)) return n;
	if (mjo_field_int(&b->desc, b->d[0], f, &n)) return n;
	return -1;
}
static void noise(ob *b, int e)
{
	mjo_ctrl_noise(&b->desc, b->d[e], b->std, b->rate, b->seed, (uint64_t)(b->off + e), b->step);
}
static int ob_step(void *s, int n)
{
	ob *b = (ob *)s;
	for (int k = 0; k < n; k++) {
		for (int e = 0; e < b->nenv; e++) {
			noise(b, e);
			mjo_step(&b->desc, b->d[e]);
		}
		b->step++;
	}
	return 0;
}
static int ob_step1(void *s)
{
	ob *b = (ob *)s;
	for (int e = 0; e < b->nenv; e++) {
		noise(b, e);
		mjo_step1(&b->desc, b->d[e]);
	}
	return 0;
}
static int ob_step2(void *s)
{
	ob *b = (ob *)s;
	for (int e = 0; e < b->nenv; e++) mjo_step2(&b->desc, b->d[e]);
	b->step++;
	return 0;
}
static int ob_step2_rk(void *s, int ncb, int rk)  /* (no prefix entry points: every env is a callback env) */
{
	ob *b = (ob *)s;
	(void)ncb;
	for (int e = 0; e < b->nenv; e++) mjo_step2_rk(&b->desc, b->d[e], rk);
	if (rk == 3) b->step++;
	return 0;
}
static int ob_forward(void *s)
{
	ob *b = (ob *)s;
	for (int e = 0; e < b->nenv; e++) mjo_forward(&b->desc, b->d[e]);
	return 0;
}
static int ob_register_collision(void *s, int t1, int t2, int func)
{
	ob *b = (ob *)s;
	if (func < 0 || func > 2) return -1;
	for (int e = 0; e < b->nenv; e++) mjo_register_collision(b->d[e], t1, t2, func);
	return 0;
}
/* per-env model parameters: the harness models geom sizes / types (mjo_set_geom_size / _type); the others are accepted and NOT
 * modelled -- the CPU tests that use this backend check the service handlers' logic (lookups, gates, mirrors, messages), the
 * physical effect of every override is checked against the oracle on the GPU (tests/test_gpu_env_params.py, test_host_services.py[hip]) */
static int ob_set_env_param(void *s, int what, int lo, int hi, const void *data)
{
	ob *b = (ob *)s;
	if (lo < 0 || hi > b->nenv || lo > hi || !data) return -1;
	for (int e = lo; e < hi; e++) {
		if (what == MJR_ENV_GEOM_SIZE) mjo_set_geom_size(&b->desc, b->d[e], (const double *)data + (size_t)(e - lo) * 3 * b->desc.ngeom);
		else if (what == MJR_ENV_GEOM_TYPE) {
			const int *t = (const int *)data + (size_t)(e - lo) * b->desc.ngeom;
			int ok = 1;
			for (int g = 0; g < b->desc.ngeom; g++)
				if (t[g] != 0 && t[g] != 2 && t[g] != 3 && t[g] != 6) ok = 0; /* ellipsoid / cylinder: no pair function, not modelled */
			if (ok) mjo_set_geom_type(&b->desc, b->d[e], t);
		}
	}
	return what >= MJR_ENV_GRAVITY && what <= MJR_ENV_BODY_MASS ? 0 : -1;
}
static int ob_reset(void *s, const uint8_t *mask)
{
	ob *b = (ob *)s;
	for (int e = 0; e < b->nenv; e++)
		if (!mask || mask[e]) mjo_reset_data(&b->desc, b->d[e]);
	return 0;
}
static int ob_get(void *s, int f, int lo, int hi, double *h)
{
	ob *b = (ob *)s;
	for (int e = lo; e < hi; e++) {
		int n = 0;
		double *p = mjo_field(&b->desc, b->d[e], f, &n);
		if (!p) return -1;
		memcpy(h + (size_t)(e - lo) * n, p, sizeof(double) * (size_t)n);
	}
	return 0;
}
static int ob_set(void *s, int f, int lo, int hi, const double *h)
{
	ob *b = (ob *)s;
	for (int e = lo; e < hi; e++) {
		int n = 0;
		double *p = mjo_field(&b->desc, b->d[e], f, &n);
		if (!p) return -1;
		memcpy(p, h + (size_t)(e - lo) * n, sizeof(double) * (size_t)n);
	}
	return 0;
}
static int ob_noise(void *s, double std, double rate, uint64_t seed, int64_t off)
{
	ob *b = (ob *)s;
	b->std = std; b->rate = rate; b->seed = seed; b->off = off;
	return 0;
}
static int ob_sync(void *s) { (void)s; return 0; }
static const char *ob_err(void *s) { (void)s; return ""; }
static void ob_destroy(void *s)
{
	ob *b = (ob *)s;
	for (int e = 0; e < b->nenv; e++) mjo_free_data(b->d[e]);
	free(b->d);
	free(b);
}

mjr_backend *oracle_backend_factory(const mjb_model_desc *desc, int nenv, int device, void *user)
{
	(void)device; (void)user;
	ob *b = (ob *)calloc(1, sizeof(ob));
	b->desc = *desc;
	b->nenv = nenv;
	b->d = (mjo_data **)calloc((size_t)nenv, sizeof(mjo_data *));
	for (int e = 0; e < nenv; e++) b->d[e] = mjo_make_data(&b->desc);
	mjr_backend vt = { b, ob_nenv, ob_field_size, ob_step, ob_step1, ob_step2, ob_forward, ob_reset, ob_get, ob_set,
		               ob_noise, ob_sync, ob_err, ob_destroy, NULL, NULL, NULL, NULL, NULL, ob_register_collision, ob_set_env_param, NULL, NULL, NULL, NULL, NULL, NULL, ob_step2_rk };
	b->vt = vt;
	return &b->vt;
}
