/* A plain C99 consumer of include/golemflavor_hip.h: builds the descriptor of the notebook posterior
 * (examples/inference.ipynb:250-260) by hand, evaluates the rows given on stdin (6 doubles per line) through
 * gf_lnprob_batch and prints "lnprob fr0 fr1 fr2 status" per row with 17 significant digits.
 *   gcc -std=c99 -Wall -Wextra -Werror -Iinclude tests/cabi/smoke.c -o smoke -Lgolemflavor_amd -lgolemhip -lm
 * Used by tests/test_gpu_parity.py (run + compare with the ctypes path) and tests/test_host_logic.py
 * (the header must compile as C).  Arguments: bestfit_fr0 bestfit_fr1 bestfit_fr2 log_mass0 log_mass1 log_mass2 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "golemflavor_hip.h"

int main(int argc, char** argv)
{
    if (argc != 7) { fprintf(stderr, "usage: smoke bf0 bf1 bf2 logmass0 logmass1 logmass2 < rows\n"); return 2; }
    if (gf_abi_version() != GF_ABI_VERSION || gf_sizeof_model_desc() != sizeof(gf_model_desc)) {
        fprintf(stderr, "header / library mismatch\n");
        return 2;
    }
    gf_model_desc d;
    memset(&d, 0, sizeof(d));
    d.abi_version = GF_ABI_VERSION;
    d.ndim = 6;
    d.mode = GF_MODE_SM_GAUSS;
    d.texture = GF_TEX_NONE;
    d.dimension = 3;
    /* columns: s_12_2, c_13_4, s_23_2 (LIMITEDGAUSS), dcp (uniform), two source angles (uniform) */
    const double lo[6] = {0, 0, 0, 0, 0, -1}, hi[6] = {1, 1, 1, 2 * M_PI, 1, 1};
    const double loc[3] = {0.307, (1 - 0.02206) * (1 - 0.02206), 0.538}, sig[3] = {0.013, 0.00147, 0.069};
    for (int i = 0; i < 6; ++i) {
        d.lo[i] = lo[i]; d.hi[i] = hi[i];
        d.prior_kind[i] = i < 3 ? GF_PRIOR_LIMITEDGAUSS : GF_PRIOR_UNIFORM;
        d.sigma[i] = 1.0;
        if (i < 3) { d.loc[i] = loc[i]; d.sigma[i] = sig[i]; d.log_mass[i] = atof(argv[4 + i]); }
    }
    for (int k = 0; k < 4; ++k) { d.idx_sm[k] = k; d.idx_mm[k] = -1; }
    d.idx_src[0] = 4; d.idx_src[1] = 5;
    d.idx_mass[0] = d.idx_mass[1] = -1;
    d.idx_scale = d.idx_gamma = d.idx_src_x = -1;
    for (int k = 0; k < 3; ++k) d.bestfit_fr[k] = atof(argv[1 + k]);
    d.smearing = 0.02;
    d.offset = -320.0;
    d.flat_llh = 1.0;

    gf_model* m = NULL;
    int rc = gf_model_create(&d, 0, &m);
    if (rc != GF_OK) { fprintf(stderr, "gf_model_create: %s (%s)\n", gf_strerror(rc), gf_last_hip_error()); return 1; }
    size_t cap = 1024, n = 0;
    double* th = (double*)malloc(cap * 6 * sizeof(double));
    double row[6];
    while (scanf("%lf %lf %lf %lf %lf %lf", &row[0], &row[1], &row[2], &row[3], &row[4], &row[5]) == 6) {
        if (n == cap) { cap *= 2; th = (double*)realloc(th, cap * 6 * sizeof(double)); }
        memcpy(th + 6 * n++, row, sizeof(row));
    }
    double* lp = (double*)malloc(n * sizeof(double));
    double* fr = (double*)malloc(n * 3 * sizeof(double));
    int32_t* st = (int32_t*)malloc(n * sizeof(int32_t));
    rc = gf_lnprob_batch(m, th, (int64_t)n, lp, fr, st);
    if (rc != GF_OK) { fprintf(stderr, "gf_lnprob_batch: %s (%s)\n", gf_strerror(rc), gf_last_hip_error()); return 1; }
    for (size_t i = 0; i < n; ++i)
        printf("%.17g %.17g %.17g %.17g %d\n", lp[i], fr[3 * i], fr[3 * i + 1], fr[3 * i + 2], (int)st[i]);
    gf_model_destroy(m);
    free(th); free(lp); free(fr); free(st);
    return 0;
}
