/*
 * golemflavor_hip.h -- C ABI of libgolemhip.so, the MI355X (gfx950) evaluation engine for
 * GolemFlavor's ensemble log-posterior callback.
 *
 * The reference has no FFI: the hot path sits behind a Python callable `ln_prob(theta) -> float`
 * that emcee calls once per walker per step (golemflavor/mcmc.py:27-41).  Each entry point below
 * names the reference interface it replaces (file:line under the reference tree).  Everything is
 * plain pointers and sizes; no torch, no C++ types.  The Python side binds this header with ctypes
 * (golemflavor_amd/_lib.py); INTEGRATION.md shows the stub a GolemFlavor maintainer would add.
 *
 * Conventions
 *   - every function returns a gf_error (0 = GF_OK) and never throws; gf_strerror() explains it;
 *   - arrays are caller-owned, fp64, row-major; theta is [n][ndim] exactly as emcee hands it over;
 *   - per-walker outcomes travel in `status` (gf_status), not in the return code;
 *   - a gf_model is bound to one device and one HIP stream.  Thread-safe per handle: the entry points that use the
 *     model's staging buffers or its unitarity queue (gf_lnprob_batch, gf_propagate_batch, gf_lnprob_cube_batch, the
 *     *_device launches) take a per-model lock, so several host threads may share one model (their calls run one
 *     after the other); different models are independent and may be driven concurrently;
 *   - there is NO CPU fallback: without a gfx950 device gf_model_create returns GF_ERR_NO_DEVICE.
 */
#ifndef GOLEMFLAVOR_HIP_H
#define GOLEMFLAVOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_ABI_VERSION 5
#define GF_MAX_DIM 16
#define GF_MAX_BINS 64
/* CP phases (dcp; the NP matrix's for texture NONE) must stay within +-GF_PHASE_MAX: range (sampled) or value
 * (fixed).  The reference's paramsets box them into [0, 2 pi] (scripts/fr.py:41, scripts/mc_unitary.py:39);
 * gf_model_create returns GF_ERR_UNSUPPORTED otherwise. */
#define GF_PHASE_MAX 1.0e6

typedef enum gf_error {
    GF_OK = 0,
    GF_ERR_INVALID_ARG = 1,   /* NULL pointer, ndim out of range, inconsistent descriptor      */
    GF_ERR_NO_DEVICE = 2,     /* no HIP device / not gfx950                                     */
    GF_ERR_HIP = 3,           /* a HIP runtime call failed; gf_last_hip_error() has the string  */
    GF_ERR_ALLOC = 4,
    GF_ERR_COMM = 5,          /* RCCL failure                                                   */
    GF_ERR_UNSUPPORTED = 6,
    GF_ERR_QUEUE_OVERFLOW = 7 /* ABI 3: a unitarity work queue was found full on the device and (walker, bin) pairs were dropped:
                                 the status array of that batch is incomplete.  The host sizes batches to fit, so this is a
                                 library bug surfacing loudly instead of as a silently missing verdict                     */
} gf_error;

/* per-walker outcome.  Reference behaviour: OUT_OF_PRIOR -> ln_prob returns -inf (llh.py:74-78,
 * ipynb:360-361); NON_UNITARY -> AssertionError raised inside params_to_BSMu (fr.py:398-399,
 * 493-498); the Python wrapper re-raises it unless told to map it to -inf. */
typedef enum gf_status {
    GF_ST_OK = 0,
    GF_ST_OUT_OF_PRIOR = 1,
    GF_ST_NON_UNITARY = 2,
    GF_ST_NAN = 3
} gf_status;

/* golemflavor/enums.py:39-42 (same integer values) */
typedef enum gf_prior_kind { GF_PRIOR_UNIFORM = 1, GF_PRIOR_GAUSSIAN = 2, GF_PRIOR_LIMITEDGAUSS = 3 } gf_prior_kind;
/* golemflavor/enums.py:59-63 (same integer values) */
typedef enum gf_texture { GF_TEX_OEU = 1, GF_TEX_OET = 2, GF_TEX_OUT = 3, GF_TEX_NONE = 4 } gf_texture;

typedef enum gf_mode {
    GF_MODE_PRIOR_ONLY = 0,  /* lnprior + flat llh: scripts/mc_unitary.py:121-143, mc_texture.py:148-170 */
    GF_MODE_SM_GAUSS = 1,    /* notebook posterior: examples/inference.ipynb:307-338,356-364            */
    GF_MODE_BSM_GAUSS = 2    /* golemflavor/llh.py:94-130 with the Gaussian substitute for gf.get_llh
                                (README.md:70-74): flux_averaged_BSMu (fr.py:403-458) -> multi_gaussian  */
} gf_mode;

/* Layout of a device-resident theta block handed to the *_device entry points. */
typedef enum gf_layout {
    GF_LAYOUT_AOS = 0,       /* [n][ndim] row-major: the emcee layout                           */
    GF_LAYOUT_SOA = 1        /* [ndim][n]: one contiguous column per parameter                  */
} gf_layout;

/*
 * Flat model descriptor: what the reference carries in (ParamSet, args, asimov ParamSet), compiled
 * once per run by golemflavor_amd/descriptor.py.  Replaces the functools.partial-bound state of
 * `ln_prob` (examples/inference.ipynb:366-371, scripts/fr.py:182-187).
 */
typedef struct gf_model_desc {
    int32_t abi_version;               /* GF_ABI_VERSION                                        */
    int32_t ndim;                      /* len(llh_paramset), 1..GF_MAX_DIM                      */
    int32_t mode;                      /* gf_mode                                               */
    int32_t texture;                   /* gf_texture (BSM only), fr.py:370-378                  */
    int32_t dimension;                 /* BSM operator dimension d (E^(d-3)), fr.py:394         */
    int32_t nbins;                     /* energy bins of the flux average, fr.py:413-414        */
    /* column of theta that holds each named parameter, or -1 = take the *_fixed value          */
    int32_t idx_sm[4];                 /* s_12_2, c_13_4, s_23_2, dcp (fr.py:116-162)           */
    int32_t idx_mass[2];               /* m21_2, m3x_2 (fr.py:422-435)                          */
    int32_t idx_src[2];                /* source flavor angles sin^4(phi), cos(2psi) (fr.py:82) */
    int32_t idx_scale;                 /* logLam (fr.py:380)                                    */
    int32_t idx_mm[4];                 /* NP mixing angles when texture == NONE (fr.py:378)     */
    int32_t idx_gamma;                 /* astroDeltaGamma (llh.py:105); cancels in fr, kept for fidelity */
    int32_t idx_src_x;                 /* astroX: the source is normalize_fr((x, 1 - x, 0)) with x = this column
                                          (scripts/mc_x.py:43,186-190); -1 = not used; excludes idx_src      */
    int32_t reserved0;                 /* 0                                                     */
    int32_t prior_kind[GF_MAX_DIM];    /* gf_prior_kind per column (llh.py:81-90)               */
    double lo[GF_MAX_DIM];             /* Param.ranges[0]  (closed box, llh.py:74-78)           */
    double hi[GF_MAX_DIM];             /* Param.ranges[1]                                       */
    double loc[GF_MAX_DIM];            /* Param.nominal_value (prior centre)                    */
    double sigma[GF_MAX_DIM];          /* Param.std                                             */
    double log_mass[GF_MAX_DIM];       /* log Gaussian mass of the truncation interval (scipy truncnorm) */
    double sm_fixed[4];                /* defaults: NuFIT point, fr.py:313                      */
    double mass_fixed[2];              /* defaults: MASS_EIGENVALUES, fr.py:42                  */
    double source_ratio[3];            /* args.source_ratio when the source is not sampled      */
    double scale_fixed;
    double mm_fixed[4];
    double gamma_fixed;                /* spectral index when astroDeltaGamma is not sampled    */
    double bestfit_fr[3];              /* injected / best-fit composition (llh.py:32-54)        */
    double smearing;                   /* Gaussian width (llh.py:53)                            */
    double offset;                     /* multi_gaussian offset, default -320 (llh.py:32)       */
    double flat_llh;                   /* value of the flat likelihood, 1.0 (mc_unitary.py:131) */
    double bin_edges[GF_MAX_BINS + 1]; /* args.binning after process_args (scripts/fr.py:122-124) */
} gf_model_desc;

typedef struct gf_model gf_model;      /* opaque: device constants + stream + staging buffers   */

/* ---- library / device ------------------------------------------------------------------- */
int gf_abi_version(void);
const char* gf_strerror(int err);
const char* gf_last_hip_error(void);   /* thread-local text of the last failing HIP/RCCL call    */
size_t gf_sizeof_model_desc(void);     /* sizeof(gf_model_desc) as compiled: lets a binding verify its layout */
/* ABI 3.  "NAME=value ..." of every GF_* environment override this process's library has honoured so far ("" = none).  The ones
 * that can change a result (the unitarity tiers' thresholds, GF_UNI_DUMP) are honoured only under GF_DIAGNOSTICS=1 and listed
 * as "NAME(ignored)" otherwise; bench.py and scan.py print the list in their JSON line. */
int gf_diagnostic_overrides(char* buf, size_t buflen);
int gf_device_count(int* count);
/* ABI 3.  Hand back the device memory the library caches between uses on `device`: the unitarity workspaces of idle pooled
 * streams (up to 8 GiB) and pooled constant blocks -- *released_bytes (may be NULL) counts these -- and the 128 MB of pinned
 * host staging slots of gf_memcpy_d2h. */
int gf_device_trim(int device, size_t* released_bytes);
int gf_device_name(int device, char* buf, size_t buflen);   /* gcnArchName, e.g. "gfx950:..."   */

/* ---- model ------------------------------------------------------------------------------ */
/* Validates `desc`, derives the per-run constants (prior normalisations, Gaussian constants,
 * texture matrix, bin tables) and uploads them.  Replaces building the partial-bound ln_prob. */
int gf_model_create(const gf_model_desc* desc, int device, gf_model** out);
void gf_model_destroy(gf_model* m);
int gf_model_ndim(const gf_model* m);

/* ---- the hot path, host buffers --------------------------------------------------------- */
/* lnprob[i] = ln_prob(theta[i]) for i < n.  Replaces the per-walker Python callback
 * (examples/inference.ipynb:356-364; golemflavor/llh.py:121-130; scripts/mc_unitary.py:134-143).
 * `fr` ([n][3], measured composition the likelihood saw) and `status` ([n]) may be NULL.
 * Synchronous: H2D, one kernel launch, D2H on the model's stream, then a stream sync. */
int gf_lnprob_batch(gf_model* m, const double* theta, int64_t n,
                    double* lnprob, double* fr, int32_t* status);

/* MultiNest-style batch, golemflavor/mn.py:26-45 lnProb(cube, ndim, n_params, ...): cube [n][nscan] lies in the unit
 * cube; column cols[k] of theta is lo + (hi - lo) * cube[.][k] (the model's own box of that column, mn.py:35-36), every
 * other column is base[col] (the paramset's current value, mn.py:37-39); then ln_prob.  The map runs on the device. */
int gf_lnprob_cube_batch(gf_model* m, const double* cube, int64_t n, int nscan, const int32_t* cols, const double* base,
                         double* lnprob, double* fr, int32_t* status);

/* fr[i] = measured flavor composition for theta[i]: chain post-processing of
 * scripts/mc_unitary.py:189-193 (u_to_fr(source_ratio, angles_to_u(x))) and
 * scripts/mc_texture.py:216-221 (flux_averaged_BSMu).  No priors, no likelihood. */
int gf_propagate_batch(gf_model* m, const double* theta, int64_t n, double* fr, int32_t* status);

/* n Haar-distributed mixing matrices: angles ~ U([0,1]^3 x [0,2pi]) (the flat-prior posterior that
 * scripts/mc_unitary.py samples by MCMC), propagated with source_ratio.  Counter-based Philox4x32-10
 * keyed by (seed, draw index): reproducible and independent of the launch geometry.
 * `angles` ([n][4]) may be NULL. */
int gf_haar_draw(gf_model* m, uint64_t seed, int64_t first_draw, int64_t n, double* angles, double* fr);

/* ---- device-resident variants (no PCIe in the timed region) ------------------------------ */
int gf_device_alloc(gf_model* m, size_t bytes, void** dptr);
int gf_device_free(gf_model* m, void* dptr);
int gf_memcpy_h2d(gf_model* m, void* dst_dev, const void* src_host, size_t bytes);  /* async + sync */
/* Synchronous.  From 16 MB on the copy brings its own staging: eight pinned 16 MB slots per device (allocated on first
 * use), filled by the DMA engine while host threads empty the earlier ones into dst_host -- so the rate (43-49 GB/s
 * measured) does not depend on whether dst_host is pinned, touched or fresh memory. */
int gf_memcpy_d2h(gf_model* m, void* dst_host, const void* src_dev, size_t bytes);
/* asynchronous on the model's stream; `layout` is a gf_layout */
int gf_lnprob_batch_device(gf_model* m, const double* d_theta, int layout, int64_t n,
                           double* d_lnprob, double* d_fr, int32_t* d_status);
int gf_propagate_batch_device(gf_model* m, const double* d_theta, int layout, int64_t n,
                              double* d_fr, int32_t* d_status);
int gf_haar_draw_device(gf_model* m, uint64_t seed, int64_t first_draw, int64_t n,
                        double* d_angles, double* d_fr);
/* Flavor-triangle histogram (golemflavor/plot.py:365-370, np.histogramdd on [0,1]^3 with `nbins` bins per
 * axis, last bin closed): counts[nbins][nbins][nbins] += ... ; the device variant is asynchronous and
 * accumulates, the host variant zeroes first. */
int gf_flavor_histogram_device(gf_model* m, const double* d_fr, int64_t n, int nbins, uint64_t* d_counts);
int gf_flavor_histogram(gf_model* m, const double* fr, int64_t n, int nbins, uint64_t* counts);
int gf_model_sync(gf_model* m);
/* Touch the pages of a freshly allocated host buffer from several threads (the content is preserved, and a copy may be filling
 * the buffer at the same time: madvise(MADV_POPULATE_WRITE), or a locked OR of zero per page where the kernel lacks it), so that a following
 * large device-to-host copy (gf_sampler_get_chain: sampler.chain of golemflavor/mcmc.py:43) runs at PCIe speed instead of
 * page-fault speed.  (gf_memcpy_d2h and gf_sampler_postprocess_rows do not need it: their staging threads map the pages.) */
int gf_host_prepare(void* buf, size_t bytes);
/* ABI 3.  The same with the number of threads chosen by the caller (<= 0: the default, up to 16): few threads when the mapping is to
 * run beside the caller's own launches and allocations, which many page-faulting threads hold up. */
int gf_host_prepare_n(void* buf, size_t bytes, int threads);
/* ABI 5.  A result ARENA: host memory of the caller's (a numpy array, a mapping of a shared segment) registered with the HIP runtime, so
 * that the DMA engines write the large read-backs -- sampler.chain (golemflavor/mcmc.py:43), the scans' rows -- STRAIGHT into it at the
 * speed of the PCIe link: no pinned staging ring, no host thread copying (57 GB/s against 28-47 through the ring on the boxes of the
 * pool; profiles/r04/host_register.txt).  Registering maps and pins every page (untouched 2 MiB-page memory: ~25 GB/s, touched: at
 * once; 4 KiB shared-memory pages ~11 GB/s), which is worth it for memory that receives MANY results: a process-lifetime arena
 * (golemflavor_amd.scan.ResultArena), each rank's region of the host segment of a multi-rank job.  Every read-back entry point finds out
 * by itself whether its destination is registered.  gf_host_unregister before the memory is freed or unmapped. */
int gf_host_register(void* buf, size_t bytes);
int gf_host_unregister(void* buf);

/* HIP events on the model's stream (what bench.py times the kernel with) */
int gf_event_create(void** ev);
int gf_event_destroy(void* ev);
int gf_event_record(gf_model* m, void* ev);
int gf_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);  /* synchronises on ev_stop */

/* ---- device-resident ensemble sampler ------------------------------------------------------ */
/* The emcee step either side of the path (golemflavor/mcmc.py:29-49: EnsembleSampler.sample / reset /
 * chain / acceptance_fraction), run entirely on the device: `nchains` independent ensembles of `nwalkers`
 * walkers of the model's posterior, affine-invariant stretch move (scale `a`, emcee's default 2.0),
 * Philox4x32-10 keyed by `seed`; the host sees chains at the end.  Launch shape is chosen per run and does not
 * change the chain: one launch per half-ensemble update (hipGraph replay), or -- small PRIOR_ONLY / SM_GAUSS
 * ensembles -- one workgroup per ensemble with the walkers in LDS and the whole run in one launch; small BSM
 * ensembles split a walker's energy bins over 4 or 16 lanes. */
typedef struct gf_sampler gf_sampler;
int gf_sampler_create(gf_model* m, int nchains, int nwalkers, uint64_t seed, double a, gf_sampler** out);
/* One ensemble per model (grid scans: submitter/sens_dag.py:75-95 and mc_texture_dag.py:57-71 run one job per
 * grid point; here all of a GPU's grid points advance in one launch per half-step).  Chain ch samples the
 * posterior of models[ch]; the models share device, ndim and mode and must outlive the sampler. */
int gf_sampler_create_multi(gf_model* const* models, int nchains, int nwalkers, uint64_t seed, double a, gf_sampler** out);
/* Random stream of each chain, ids [nchains] (default: the chain's index in this sampler).  A grid scan passes the
 * global grid index, so that a grid point's chain is the same whichever rank runs it and whatever else shares its
 * sampler (the reference's jobs are seeded per job, scripts/mc_texture.py:137-138).  Before the first run. */
int gf_sampler_set_stream_ids(gf_sampler* s, const uint64_t* ids);
void gf_sampler_destroy(gf_sampler* s);
/* p0 [nchains][nwalkers][ndim]; evaluates its lnprob on the device (mcmc.py:34 sampler.sample(p0, ...)) */
int gf_sampler_set_state(gf_sampler* s, const double* pos);
/* nsteps stretch-move steps, asynchronous; store != 0 appends every thin-th step to the device chain */
int gf_sampler_run(gf_sampler* s, int64_t nsteps, int thin, int store);
int gf_sampler_sync(gf_sampler* s);
/* ABI 3.  gf_sampler_run(s, nsteps, thin, 1) with the read-back overlapped: while the GPU works through the run, every finished
 * block of 16 steps of all chains is copied to the host on a second stream (pinned staging, as gf_memcpy_d2h).  Synchronous: on
 * return the run is complete and chain [nchains][nstored][nwalkers][ndim], lnprob_chain [nchains][nstored][nwalkers] (may be
 * NULL) hold the whole stored chain, nstored = gf_sampler_nstored(s) after the call -- size them for
 * nstored_before + ceil(nsteps / thin).  Replaces run + sync + gf_sampler_get_chain where `sampler.chain` is read after
 * `sampler.run_mcmc` (golemflavor/mcmc.py:41-43).  *readback_tail_s (may be NULL): seconds from the end of the run on the GPU to the
 * end of the last copy, i.e. the part of the read-back that was not hidden behind the run. */
int gf_sampler_run_to_host(gf_sampler* s, int64_t nsteps, int thin, double* chain, double* lnprob_chain, double* readback_tail_s);
/* clears the stored chain and the acceptance counters, keeps the walkers (mcmc.py:36 sampler.reset()) */
int gf_sampler_reset(gf_sampler* s);
int64_t gf_sampler_nstored(const gf_sampler* s);
int64_t gf_sampler_iterations(const gf_sampler* s);
int gf_sampler_get_state(gf_sampler* s, double* pos, double* lnprob);
/* chain [nchains][nstored][nwalkers][ndim], lnprob_chain [nchains][nstored][nwalkers],
 * naccepted [nchains][nwalkers], nonunitary[1] = proposals the reference would have raised on (fr.py:493-498; they were
 * rejected: every proposal's unitarity verdict is settled on the device before its accept step, ABI 3); NULL = skip */
int gf_sampler_get_chain(gf_sampler* s, double* chain, double* lnprob_chain, uint32_t* naccepted, uint32_t* nonunitary);
/* the stored chain packed into caller-owned DEVICE buffers (what gf_comm_allgather sends); NULL = skip; synchronous */
int gf_sampler_get_chain_device(gf_sampler* s, double* d_chain, double* d_lnprob_chain);
/* mean [nchains][nstored][ndim]: ensemble mean of every stored step, the series behind sampler.acor
 * (golemflavor/mcmc.py:45-51), reduced on the device */
int gf_sampler_walker_mean(gf_sampler* s, double* mean);
/* Chain post-processing on the device (scripts/mc_unitary.py:189-193, scripts/mc_texture.py:216-221,
 * golemflavor/plot.py:365-370): composition of every stored sample and/or its [nbins]^3 histogram per
 * chain.  fr [nchains][nstored][nwalkers][3], status [nchains][nstored][nwalkers],
 * counts [nchains][nbins]^3; NULL = skip. */
int gf_sampler_postprocess(gf_sampler* s, double* fr, int32_t* status, int nbins, uint64_t* counts);
/* same, chain ch propagated with models[ch] (NULL: the sampling models): scripts/mc_texture.py samples the
 * priors (:148-170) and pushes every sample through flux_averaged_BSMu of the grid point (:216-221) */
int gf_sampler_postprocess_with(gf_sampler* s, gf_model* const* models, double* fr, int32_t* status, int nbins,
                                uint64_t* counts);

/* same with DEVICE destinations d_fr [nchains][nstored][nwalkers][3], d_status (NULL = skip); synchronous */
int gf_sampler_postprocess_device(gf_sampler* s, gf_model* const* models, double* d_fr, int32_t* d_status);
/* the rows a scan saves, assembled on the device: d_rows [nchains][nstored][nwalkers][3 + ndim] = composition (NaN where the
 * reference would have raised) then the sample (scripts/mc_texture.py:216-223); synchronous */
int gf_sampler_postprocess_rows_device(gf_sampler* s, gf_model* const* models, double* d_rows);
/* the same rows in HOST memory, rows [nchains][nstored][nwalkers][3 + ndim]: the chains are post-processed one after the
 * other and every finished group of chains crosses PCIe while the next ones are still being evaluated (what a one-rank scan
 * hands to scripts/mc_texture.py's np.save); synchronous */
int gf_sampler_postprocess_rows(gf_sampler* s, gf_model* const* models, double* rows);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI -------------------------------------- */
/* Independent chains (grid points) shard across ranks with no data-path collective; the only
 * exchanges are the broadcast of the packed descriptors at start and the gather of the chain blocks
 * at the end -- the role HTCondor + a shared filesystem play in the reference
 * (submitter/mc_texture_dag.py:57-71, submitter/sens_dag.py:75-95). */
typedef struct gf_comm gf_comm;
#define GF_COMM_ID_BYTES 128
int gf_comm_unique_id(uint8_t id[GF_COMM_ID_BYTES]);                 /* rank 0; ship to the others out of band */
int gf_comm_create(const uint8_t id[GF_COMM_ID_BYTES], int rank, int nranks, int device, gf_comm** out);
void gf_comm_destroy(gf_comm* c);
int gf_comm_broadcast(gf_comm* c, void* host_buf, size_t bytes, int root);            /* host in/out   */
int gf_comm_allgather(gf_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank); /* device  */
/* ABI 3.  Gather to one rank: rank r's block lands at d_recv_on_root + r * bytes_per_rank on `root` (NULL elsewhere); only the
 * root holds nranks x the block.  Stands in for the reference's N jobs saving N chain files to one place
 * (golemflavor/mcmc.py:108-126, submitter/mc_texture_dag.py:57-71): one ncclGroup of point-to-point transfers into the root. */
int gf_comm_gather(gf_comm* c, const void* d_send, void* d_recv_on_root, size_t bytes_per_rank, int root);
/* ABI 3.  The same gather without a communicator, for the ranks of ONE node: gf_ipc_export turns a rank's block (the base of a
 * gf_device_alloc'd buffer) into a 64-byte handle that travels over the caller's control plane; on the root gf_ipc_gather opens the
 * nranks handles (its own slot is ignored: d_own is copied), copies every block to d_recv + r * bytes_per_rank device to device --
 * over xGMI between GPUs -- and closes them.  The senders must keep their blocks until the root has returned (a barrier of the
 * control plane).  Fallback where RCCL cannot be set up; also the one inter-process device path a one-GPU box can exercise. */
#define GF_IPC_HANDLE_BYTES 64
int gf_ipc_export(const void* d_ptr, unsigned char* handle64);
int gf_ipc_gather(int device, const unsigned char* handles, int nranks, int self_rank, const void* d_own, void* d_recv,
                  size_t bytes_per_rank);
int gf_comm_barrier(gf_comm* c);
/* ABI 4.  What the COMMUNICATOR itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) -- the number a bench line
 * quotes as `rccl_nranks`: how many ranks RCCL saw, not how many the caller asked for.  Any pointer may be NULL. */
int gf_comm_info(gf_comm* c, int* nranks, int* rank, int* device);
/* ABI 4.  Device memory without a model handle (the hipIpc probe of golemflavor_amd.dist allocates its 16 bytes before any
 * posterior exists): hipMalloc / hipFree on `device`. */
int gf_device_malloc(int device, size_t bytes, void** dptr);
int gf_device_release(int device, void* dptr);
const char* gf_comm_last_error(void);                 /* thread-local text of the last failing gf_comm_* call */
int gf_comm_library_info(char* buf, size_t buflen);   /* "<ncclGetVersion code> <path of the loaded librccl>" */

#ifdef __cplusplus
}
#endif
#endif /* GOLEMFLAVOR_HIP_H */
