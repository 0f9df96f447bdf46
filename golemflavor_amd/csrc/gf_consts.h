// Per-run constants derived from gf_model_desc by gf_model_create (host) and consumed by the
// kernels.  Small ones travel by value as kernel arguments (scalar loads -> SGPRs); the BSM bin
// tables live in one device buffer per model.
#pragma once
#include <stdint.h>

#define GF_MAX_DIM 16
#define GF_MAX_BINS 64

// Priors + named-column map + Gaussian likelihood constants: every mode needs these.
struct GfCommon {
    int32_t ndim;
    int32_t mode;
    int32_t idx_sm[4];
    int32_t idx_mass[2];
    int32_t idx_src[2];
    int32_t idx_scale;
    int32_t idx_mm[4];
    int32_t idx_gamma;
    int32_t idx_src_x;              // astroX column: source = normalize_fr((x, 1 - x, 0)), scripts/mc_x.py:186-190; -1 = none
    double lo[GF_MAX_DIM];
    double hi[GF_MAX_DIM];
    double loc[GF_MAX_DIM];
    double inv_sigma[GF_MAX_DIM];   // 0 for UNIFORM columns: their z is forced to 0
    double prior_const;             // sum over non-uniform columns of -log sqrt(2pi) - log_mass - log sigma
    double sm_fixed[4];
    double mass_fixed[2];
    double src_fixed[3];
    double src_fixed_sum;           // fp64 left-to-right sum of src_fixed (np.sum, fr.py:535)
    double scale_fixed;
    double mm_fixed[4];
    double bf[3];
    double inv_smear;               // sqrt(1 / smearing^2)      (scipy _PSD: U = u * sqrt(1/s))
    double gauss_c0;                // 3 log(2 pi) + log_pdet
    double gauss_mh;                // -0.5 * inv_smear^2
    double gauss_k;                 // -0.5 * gauss_c0
    double offset;
    double flat_llh;
    // minimax coefficients of fast_cos_phase, passed through the kernel arguments so that they sit in
    // scalar registers (literals get materialised in VGPRs and cost a v_mov per Horner step)
    double cosc[8];
};

// BSM tables (device memory, read with uniform indices -> scalar loads).
struct GfBsm {
    int32_t texture;
    int32_t dimension;
    int32_t nbins;
    int32_t uni_own_bins_only;     // diagnostics (GF_UNI_OWN_BINS_ONLY): the arbitration gets the undecided bins alone, not the bins above them
    // Fixed textures (fr.py:370-378): U~ diag(0, sc1, sc2) U~^dagger = sc1 T1 + sc2 T2 with the rank-1
    // projectors T1 = u~_1 u~_1^dagger, T2 = u~_2 u~_2^dagger (columns 1 and 2 of U~), precomputed in
    // long double on the host.  For texture NONE they are rebuilt per walker from the sampled angles.
    double t1_re[9], t1_im[9];
    double t2_re[9], t2_im[9];
    double inv2e[GF_MAX_BINS];      // 1 / (2 E_k)                fr.py:386
    double epow[GF_MAX_BINS];       // E_k ** (d - 3)             fr.py:394
    double weight[GF_MAX_BINS];     // |b_{k+1} - b_k|            fr.py:414 (the 1/(b_N - b_0) factor cancels in fr.py:457)
    double centre[GF_MAX_BINS];     // sqrt(b_k b_{k+1})          fr.py:413
    // Unitarity arbitration (gf_unitarity.hip, gf_x87.hpp): per-model mixing matrices in the reference's own
    // arithmetic -- angles_to_u (fr.py:116-162) evaluated in long double by gf_model_create, each entry split into
    // (hi, lo) doubles; row-major (re, im) pairs.  npu: the NP matrix of a fixed texture or of fixed NP angles;
    // smu: the SM matrix when its angles are not columns of theta (NuFIT default, fr.py:313,435).
    double npu_hi[18], npu_lo[18];
    double smu_hi[18], smu_lo[18];
    // unitarity tiers (gf_bsm_device.hpp): SM weight at or above which a bin is unitary outright; SM weight at or above
    // which the fp64 estimate may condemn a bin; the estimate (2^11 x the x87 residual) below which a bin is unitary and
    // at or above which it is not
    double uni_a_ok, uni_a_lin;
    double uni_lo, uni_hi;          // a >= uni_a_lin
    double uni_lo_nl;               // a <  uni_a_lin: the estimate acquits only far below the threshold and never condemns
    double rho_max;                 // max_k epow[k] / inv2e[k]: the smallest SM weight over the bins is 1 / (1 + rho_max trN / trS)
    double rho[GF_MAX_BINS];        // epow[k] / inv2e[k]: the bin's SM weight is a = 1 / (1 + rho[k] trN / trS)
    double wsum;                    // sum_k weight[k] (fp64, in index order)
};

// Work queue of the unitarity arbitration (gf_unitarity.hip): the WALKERS whose verdict the fp64 estimate of the reference's
// unitarity residual cannot settle, each with the set of its undecided energy bins.  The arbitration kernel takes a walker's
// bins from the highest energy down and stops at the first that fails (a walker is non-unitary as soon as ONE bin is,
// fr.py:398-399 raises on the first): walkers that end up failing cost one or two bins instead of all of theirs (11 on average
// where a posterior crosses the failing region), and the per-walker part of the chain -- mixing matrices, Hamiltonian terms -- is
// evaluated once per walker, not once per bin.
struct GfArbItem {
    unsigned long long walker;      // index within the launch's piece of the batch
    unsigned long long mask;        // bit k = energy bin k is undecided
};
struct GfArbQueue {
    unsigned int count;             // items pushed by the evaluation kernels of the current launch
    unsigned int done;              // blocks of the resolve kernel that have finished (the last one re-arms the queue)
    unsigned int cap;               // capacity of items[]
    unsigned int overflow;          // set by a producer that found the queue full (its item was DROPPED); k_uni_resolve reports it to the host and clears it
    unsigned int head;              // next item to hand out (k_uni_resolve's lanes fetch their walkers dynamically)
    unsigned int pad_[3];
    GfArbItem items[1];             // [cap]
};

// Queue of walkers for the deferred tier 2 (k_bsm_tier2).  item = walker index.
struct GfUniQueue {
    unsigned int count;             // items pushed by the evaluation kernel of the current launch
    unsigned int done;              // blocks of the resolve kernel that have finished (the last one resets both)
    unsigned int cap;               // capacity of items[]
    unsigned int overflow;          // set by a producer that found the queue full (its item was DROPPED); k_uni_resolve reports it to the host and clears it
    unsigned long long items[1];    // [cap]
};
