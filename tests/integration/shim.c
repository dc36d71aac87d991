/* TEST INFRASTRUCTURE for tests/test_integration_stub.py -- NOT part of the product.
 *
 * The build container has the reference tree but no GPU; the GPU box has a GPU but no reference tree.  To execute
 * INTEGRATION.md's stub INSIDE the real reference tree, this shim exports the C-ABI symbols the stub binds:
 *   - host entry points (qrec_mt_bpr_sample_epoch) are forwarded to the REAL libqrec_hip.so (QREC_REAL_LIB);
 *   - "device memory" is host memory, and the two device entry points the stub calls (qrec_bpr_sgd_ordered,
 *     qrec_sumsq) are answered by the oracle's C restatement (QREC_ORACLE_LIB) -- the same functions the GPU kernels are
 *     held to in tests/test_gpu_bpr.py.
 * What the test therefore proves is the BINDING: argument order and types, ownership, the generator hand-over, the
 * epoch loop against the reference's own base classes, data model and evaluation. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void *real_lib, *oracle_lib;
static char err[256] = "";
static int (*real_sample)(uint32_t *, const int64_t *, const int32_t *, int32_t, int32_t, int32_t *);
static double (*orc_sgd)(double *, double *, int32_t, const int32_t *, const int32_t *, const int32_t *, int64_t, double, double, double);
static double (*orc_sumsq)(const double *, int64_t);

static int load(void) {
    if (real_lib) return 0;
    real_lib = dlopen(getenv("QREC_REAL_LIB"), RTLD_NOW | RTLD_LOCAL);
    oracle_lib = dlopen(getenv("QREC_ORACLE_LIB"), RTLD_NOW | RTLD_LOCAL);
    if (!real_lib || !oracle_lib) { snprintf(err, sizeof err, "shim: %s", dlerror()); return -2; }
    real_sample = dlsym(real_lib, "qrec_mt_bpr_sample_epoch");
    orc_sgd = dlsym(oracle_lib, "orc_bpr_sgd_f64");
    orc_sumsq = dlsym(oracle_lib, "orc_sumsq_f64");
    return (real_sample && orc_sgd && orc_sumsq) ? 0 : -2;
}

const char *qrec_last_error(void) { return err; }
int qrec_init(int device) { (void)device; return load(); }
int qrec_malloc(int64_t bytes, void **p) { *p = malloc(bytes > 0 ? (size_t)bytes : 1); return *p ? 0 : -2; }
int qrec_free(void *p) { free(p); return 0; }
int qrec_memcpy_h2d(void *d, const void *h, int64_t n, void *s) { (void)s; memcpy(d, h, (size_t)n); return 0; }
int qrec_memcpy_d2h(void *h, const void *d, int64_t n, void *s) { (void)s; memcpy(h, d, (size_t)n); return 0; }
int qrec_mt_bpr_sample_epoch(uint32_t *state, const int64_t *indptr, const int32_t *items, int32_t n_users, int32_t n_items, int32_t *j) {
    int rc = load();
    return rc ? rc : real_sample(state, indptr, items, n_users, n_items, j);
}
int qrec_bpr_sgd_ordered(void *P, void *Q, int dtype, int32_t d, int32_t ld, const int32_t *u, const int32_t *i, const int32_t *j,
                         int64_t n, double lr, double regU, double regI, double *loss, void *stream) {
    (void)stream;
    if (dtype != 1 || ld != d) { snprintf(err, sizeof err, "shim: fp64, ld == d only"); return -1; }
    *loss = orc_sgd((double *)P, (double *)Q, d, u, i, j, n, lr, regU, regI);
    return 0;
}
int qrec_sumsq(const void *x, int dtype, int64_t rows, int32_t d, int32_t ld, double *out, void *stream) {
    (void)stream; (void)ld;
    if (dtype != 1) return -1;
    *out = orc_sumsq((const double *)x, rows * (int64_t)d);
    return 0;
}
