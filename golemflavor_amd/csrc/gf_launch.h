// Internal launch interface between the C ABI (gf_capi.hip) and the kernels (gf_kernels.hip,
// gf_bsm.hip).  All launches are asynchronous on `s`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf_consts.h"

// `ptab`: device table [GF_MAX_DIM][4] = {lo, hi, loc, 1/sigma} per column (GfModel::d_ptab)
hipError_t gf_launch_lnprob_sm(const GfCommon& c, const double* ptab, const double* theta, int layout, int64_t n,
                               double* lnprob, double* fr, int32_t* status, int cus, hipStream_t s);
hipError_t gf_launch_propagate_sm(const GfCommon& c, const double* theta, int layout, int64_t n, double* fr,
                                  int32_t* status, int cus, hipStream_t s);
hipError_t gf_launch_haar(const GfCommon& c, uint64_t seed, int64_t first, int64_t n, double* angles, double* fr,
                          int cus, hipStream_t s);
// BSM (flux-averaged) path; `with_llh` = 0 -> composition only (propagate), 1 -> lnprob
// `uq`, `uq_cap`: the stream's unitarity-arbitration queue and its capacity in items = walkers (NULL / 0 when status == NULL);
// `wq`, `wq_cap`: its walker queue for the deferred tier 2, NULL = tiers 1-2 inline; `t2sn`: [wq_cap][18]
// doubles, where the evaluation kernel leaves the Hamiltonian terms of the walkers it queues; `seen`: see
// gf_launch_uni_resolve
hipError_t gf_launch_bsm(const GfCommon& c, const GfCommon* d_common, const GfBsm* d_bsm, int nbins, const double* ptab, const double* theta, int layout,
                         int64_t n, int with_llh, double* lnprob, double* fr, int32_t* status, GfArbQueue* uq, int64_t uq_cap, GfUniQueue* wq, int64_t wq_cap,
                         double* t2sn, unsigned int* seen, int cus, hipStream_t s);
// gf_unitarity.hip: settles the walkers queued by the evaluation kernels in emulated x87 arithmetic, bin by bin from the highest
// energy down; a bin the reference would raise on turns status[walker] into NON_UNITARY and lnprob[walker] (if given) into NaN
// and ends that walker.  `max_items` (walkers) bounds the grid; the item count itself is read on the device.  `seen` (pinned host memory, may be NULL): the item count
// of the previous launch, written by the kernel and used to size the next grid.  `wq` (may be NULL): the walker queue of
// the deferred tier 2, emptied here for the next launch.
hipError_t gf_launch_uni_resolve(const GfCommon* d_common, const GfBsm* d_bsm, const double* theta, int layout, int64_t n, int ndim,
                                 double* lnprob, int32_t* status, GfArbQueue* uq, GfUniQueue* wq, int64_t max_items, unsigned int* seen, int cus, hipStream_t s);
hipError_t gf_launch_join_rows(const double* fr, const int32_t* status, const double* theta, int ndim, int64_t n, double* out,
                               int cus, hipStream_t s);
hipError_t gf_launch_flavor_hist(const double* fr, int64_t n, int nb, unsigned long long* counts, int cus, hipStream_t s);
hipError_t gf_launch_cube_to_theta(const GfCommon& c, int nscan, const int32_t* cols, const double* base, const double* cube,
                                   int64_t n, double* theta, int cus, hipStream_t s);
// test hook (gf_unitarity.hip): emulated-x87 residuals of explicit (walker, bin) pairs; which = 0 serial chain, 1 three-lane groups
hipError_t gf_launch_uni_debug(const GfCommon* d_common, const GfBsm* d_bsm, const double* theta, int layout, int64_t n, const int64_t* walkers,
                               const int32_t* bins, int64_t npairs, int which, double* out, hipStream_t s);

// ---- the device sampler's settlement of undecided proposals (gf_sampler.hip -> gf_unitarity.hip) ---------------------------
// Step counters of the device sampler (live on the device so that a captured hipGraph can be replayed with constant kernel
// arguments; see gf_sampler.hip)
struct GfStepState {
    uint64_t iteration_base;  // Philox counter word of step_offset 0
    int64_t run_step_base;    // steps of the current gf_sampler_run call done before step_offset 0
    int64_t store_base;       // chain slot of the run's first stored step
    int32_t store;            // this run stores at all
    int32_t thin;
};
constexpr int GF_PEND_STRIDE = GF_MAX_DIM + 2;     // a deferred proposal: theta [GF_MAX_DIM] | lnprob of the proposal | ln(z^(ndim-1) / u)
// A proposal of the stretch move whose unitarity verdict (fr.py:461-499) the in-kernel tiers cannot settle is NOT decided by the
// half-step kernel: the kernel parks it -- row t = chain * (nwalkers / 2) + k of `pend_rows`, item {t, undecided bins} in `pq` --
// and k_stretch_settle, next in stream order, takes the exact (emulated x87) verdict and completes the walker's update: reject
// and count if the reference would have raised, else the usual accept test; it also writes the walker's stored sample.
struct GfSettleArgs {
    const GfStepState* state;
    GfArbQueue* pq;             // capacity nchains * nwalkers / 2
    const double* pend_rows;    // [nchains * nwalkers / 2][GF_PEND_STRIDE]
    unsigned int* ctl;          // [nchains * nwalkers / 2][2]: parts of the walker that have finished, one of them failed (zero between uses)
    double* pos;                // [nchains][nwalkers][ndim]
    double* lnp;                // [nchains][nwalkers]
    uint32_t* naccept;          // [nchains][nwalkers]
    uint32_t* flags;            // [0]: proposals the reference would have raised on
    double* chain;              // [nchains][nstore_cap][nwalkers][ndim] or null
    double* lnp_chain;
    int64_t nstore_cap;
    int32_t nchains, nwalkers, half, step_offset, ndim;
    const GfCommon* commons;    // [nchains] (multi != 0) or [1]
    const GfBsm* const* tbs;    // [nchains] (multi != 0), else null
    const GfBsm* tb;            // multi == 0
    int32_t multi;
};
hipError_t gf_launch_stretch_settle(const GfSettleArgs& a, int cus, hipStream_t s);
