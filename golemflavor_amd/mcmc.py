"""emcee-driven MCMC with the reference's call surface (golemflavor/mcmc.py:27-126).

`mcmc(p0, ln_prob, ndim, nwalkers, burnin, nsteps, threads=1)` has the reference's signature,
prints and return value.  emcee itself is an un-pinned third-party dependency of the reference
(requirements.txt:5) and is not installed in this image, so the sampler is the package's own
implementation of the published affine-invariant stretch move (Goodman & Weare 2010; the move
emcee's `EnsembleSampler` runs by default), with emcee-2's attribute surface (`sample`,
`reset`, `chain`, `lnprobability`, `acceptance_fraction`, `acor`) because that is what the
reference's driver touches (mcmc.py:29-49).

Difference that matters for speed: emcee calls `ln_prob` once per walker; this sampler calls a
callable that advertises `vectorized = True` (golemflavor_amd.llh.LnProb) ONCE per half-ensemble
with a `(nwalkers/2, ndim)` block -> one HIP kernel launch.  Any plain Python callable still
works (it is then called per walker, exactly as emcee would).

Sampler parity: the reference never seeds emcee's RNG and holds no test of sampler output, so
chains cannot be compared sample by sample ("sampler parity unpinned"); tests
check the move's invariants and the reference notebook's acceptance fraction / autocorrelation.
"""
import math
import os
import sys

import numpy as np


try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, total=None):
        return it


class AutocorrError(Exception):
    """The chain is too short for a reliable autocorrelation estimate."""


def _autocorr_1d(x):
    n = len(x)
    size = 1 << (2 * n - 1).bit_length()
    f = np.fft.rfft(x - np.mean(x), n=size)
    acf = np.fft.irfft(f * np.conjugate(f))[:n]
    return acf / acf[0]


def integrated_time(x, c=5, tol=50):
    """Integrated autocorrelation time of a (nsteps, ndim) series with Sokal's automatic
    window (smallest M with M >= c * tau(M)); raises AutocorrError if nsteps < tol * tau."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64).T).T
    n, ndim = x.shape
    tau = np.empty(ndim)
    for d in range(ndim):
        rho = _autocorr_1d(x[:, d])
        taus = 2.0 * np.cumsum(rho) - 1.0
        m = np.arange(len(taus)) < c * taus
        window = int(np.argmin(m)) if not m.all() else len(taus) - 1
        tau[d] = taus[window]
    if np.any(tol * tau > n) or not np.all(np.isfinite(tau)):
        raise AutocorrError("chain of %d steps is shorter than %d x tau (%s)" % (n, tol, tau))
    return tau


class EnsembleSampler:
    """Affine-invariant ensemble sampler (stretch move, a = 2), emcee-2 flavoured API."""

    def __init__(self, nwalkers, dim, lnpostfn, a=2.0, args=(), kwargs=None, threads=1, seed=None):
        if nwalkers % 2:
            raise AssertionError("The number of walkers must be even.")
        if nwalkers < 2 * dim:
            raise AssertionError("The number of walkers needs to be more than twice the dimension "
                                 "of your parameter space.")
        self.k, self.dim, self.a = int(nwalkers), int(dim), float(a)
        self.lnprobfn = lnpostfn
        self.args, self.kwargs = tuple(args), dict(kwargs or {})
        self.threads = threads        # accepted for signature compatibility; batching replaces the pool
        self._random = np.random.RandomState(seed)
        self.vectorized = bool(getattr(lnpostfn, "vectorized", False))
        self.reset()

    # -- state -----------------------------------------------------------------------
    def reset(self):
        self.iterations = 0           # every iteration since the reset, stored or not (acceptance_fraction divides by it)
        self._nstored = 0             # steps kept in the chain (every `thin`-th)
        self.naccepted = np.zeros(self.k)
        self._chain = np.empty((self.k, 0, self.dim))
        self._lnprob = np.empty((self.k, 0))

    @property
    def random_state(self):
        return self._random.get_state()

    @property
    def chain(self):
        """(nwalkers, nsteps, ndim), as emcee-2 (mcmc.py:43 reshapes it to (-1, ndim))."""
        return self._chain[:, :self._nstored]

    @property
    def flatchain(self):
        return self.chain.reshape(-1, self.dim)

    @property
    def lnprobability(self):
        return self._lnprob[:, :self._nstored]

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iterations, 1)

    @property
    def acor(self):
        return self.get_autocorr_time()

    def get_autocorr_time(self, c=5, tol=50):
        return integrated_time(np.mean(self.chain, axis=0), c=c, tol=tol)

    # -- evaluation ---------------------------------------------------------------------
    def _get_lnprob(self, pos):
        if np.any(np.isinf(pos)):
            raise ValueError("At least one parameter value was infinite.")
        if np.any(np.isnan(pos)):
            raise ValueError("At least one parameter value was NaN.")
        if self.vectorized:
            lp = np.asarray(self.lnprobfn(pos, *self.args, **self.kwargs), dtype=np.float64)
        else:
            lp = np.array([float(self.lnprobfn(p, *self.args, **self.kwargs)) for p in pos])
        if lp.shape != (pos.shape[0],):
            raise ValueError("lnpostfn returned shape %s for %d walkers" % (lp.shape, pos.shape[0]))
        if np.any(np.isnan(lp)):
            raise ValueError("lnprob returned NaN.")
        return lp

    def _propose_stretch(self, p0, p1, lnprob0):
        """Move the walkers `p0` using the complementary ensemble `p1`."""
        s, c = np.atleast_2d(p0), np.atleast_2d(p1)
        ns, nc = len(s), len(c)
        zz = ((self.a - 1.0) * self._random.rand(ns) + 1.0) ** 2 / self.a
        rint = self._random.randint(nc, size=(ns,))
        q = c[rint] - zz[:, None] * (c[rint] - s)
        newlnprob = self._get_lnprob(q)
        lnpdiff = (self.dim - 1.0) * np.log(zz) + newlnprob - lnprob0
        accept = lnpdiff > np.log(self._random.rand(ns))
        return q, newlnprob, accept

    def sample(self, p0, lnprob0=None, rstate0=None, iterations=1, thin=1, storechain=True):
        """Generator over iterations yielding (pos, lnprob, random_state) like emcee-2."""
        if rstate0 is not None:
            self._random.set_state(rstate0)
        p = np.array(p0, dtype=np.float64)
        if p.shape != (self.k, self.dim):
            raise ValueError("p0 must have shape (%d, %d), got %s" % (self.k, self.dim, p.shape))
        halfk = self.k // 2
        lnprob = np.array(lnprob0, dtype=np.float64) if lnprob0 is not None else self._get_lnprob(p)
        if np.any(np.isnan(lnprob)):
            raise ValueError("The initial lnprob was NaN.")
        thin = int(thin)
        if storechain:
            n_new = (int(iterations) + thin - 1) // thin          # steps 0, thin, 2 thin, ... of this call
            self._chain = np.concatenate((self._chain[:, :self._nstored],
                                          np.zeros((self.k, n_new, self.dim))), axis=1)
            self._lnprob = np.concatenate((self._lnprob[:, :self._nstored],
                                           np.zeros((self.k, n_new))), axis=1)
        s0 = self._nstored
        first, second = slice(halfk), slice(halfk, self.k)
        for i in range(int(iterations)):
            for S0, S1 in ((first, second), (second, first)):
                q, newlnp, acc = self._propose_stretch(p[S0], p[S1], lnprob[S0])
                if np.any(acc):
                    idx = np.arange(self.k)[S0][acc]
                    lnprob[idx] = newlnp[acc]
                    p[idx] = q[acc]
                    self.naccepted[idx] += 1
            if storechain and i % thin == 0:
                ind = s0 + i // thin
                self._chain[:, ind, :] = p
                self._lnprob[:, ind] = lnprob
                self._nstored = ind + 1
            self.iterations += 1
            yield p, lnprob, self.random_state

    def run_mcmc(self, pos0, N, **kwargs):
        results = None
        for results in self.sample(pos0, iterations=N, **kwargs):
            pass
        return results


class DeviceEnsembleSampler:
    """The same sampler, resident on the GPU (`gf_sampler_*` in include/golemflavor_hip.h).

    Proposal, lnprob and accept/reject of a half-ensemble run in ONE kernel launch and the walkers
    never leave HBM; `nchains` independent ensembles of one posterior are advanced together, or -- when
    `lnpostfn` is a list -- one ensemble per posterior (the grid points of a scan; the posteriors share
    ndim and mode, everything else may differ).  The
    attribute surface is emcee-2's (`sample`, `run_mcmc`, `reset`, `chain`, `lnprobability`,
    `acceptance_fraction`, `acor`); with `nchains > 1` the arrays gain a leading chain axis.
    Random numbers come from Philox4x32-10 keyed by `seed` (reproducible, independent of launch
    geometry), not from numpy's global state.
    """

    def __init__(self, nwalkers, dim, lnpostfn, a=2.0, nchains=1, seed=0, threads=1, stream_ids=None):
        import ctypes as C
        from . import _lib
        multi = isinstance(lnpostfn, (list, tuple))
        fns = list(lnpostfn) if multi else [lnpostfn]
        models = [getattr(f, "model", f) for f in fns]
        if multi:
            if not models:
                raise ValueError("empty list of posteriors")
            if nchains not in (1, len(models)):
                raise ValueError("nchains=%d but %d posteriors given" % (nchains, len(models)))
            nchains = len(models)
        for model in models:
            if not hasattr(model, "_h"):
                raise TypeError("DeviceEnsembleSampler needs a golemflavor_amd LnProb / Model")
            if model.ndim != dim:
                raise AssertionError("dim %d does not match the model's %d parameters" % (dim, model.ndim))
        model, lnpostfn = models[0], fns[0]
        if nwalkers % 2:
            raise AssertionError("The number of walkers must be even.")
        if nwalkers < 2 * dim:
            raise AssertionError("The number of walkers needs to be more than twice the dimension "
                                 "of your parameter space.")
        self._C, self._lib, self._L = C, _lib, _lib.lib()
        self.model, self.lnprobfn = model, lnpostfn
        self.k, self.dim, self.a, self.nchains = int(nwalkers), int(dim), float(a), int(nchains)
        self.models = models if multi else None               # one posterior per chain (kept alive here)
        h = C.c_void_p()
        if multi:
            handles = (C.c_void_p * self.nchains)(*[m._h.value if hasattr(m._h, "value") else m._h for m in models])
            _lib.check(self._L.gf_sampler_create_multi(handles, self.nchains, self.k, int(seed), self.a, C.byref(h)),
                       "gf_sampler_create_multi")
        else:
            _lib.check(self._L.gf_sampler_create(model._h, self.nchains, self.k, int(seed), self.a, C.byref(h)),
                       "gf_sampler_create")
        self._h = h
        self._have_state = False
        self.on_nonunitary = getattr(lnpostfn, "on_nonunitary", "raise")
        if stream_ids is not None:
            # one random stream per chain, named by the caller (a scan: the global grid index) instead of by the
            # chain's position in this sampler: the chain of a grid point is then the same on any rank / in any stack
            ids = np.ascontiguousarray(stream_ids, dtype=np.uint64)
            if ids.shape != (self.nchains,):
                raise ValueError("stream_ids must hold one id per chain (%d), got %s" % (self.nchains, ids.shape))
            _lib.check(self._L.gf_sampler_set_stream_ids(self._h, ids.ctypes.data_as(C.POINTER(C.c_uint64))),
                       "gf_sampler_set_stream_ids")

    # -- control ------------------------------------------------------------------------
    def _set_state(self, p0):
        p = np.ascontiguousarray(p0, dtype=np.float64)
        if p.shape == (self.k, self.dim) and self.nchains == 1:
            p = p[None]
        if p.shape != (self.nchains, self.k, self.dim):
            raise ValueError("p0 must have shape (%d, %d, %d), got %s" % (self.nchains, self.k, self.dim, p.shape))
        if not np.all(np.isfinite(p)):
            raise ValueError("At least one parameter value was infinite or NaN.")
        self._lib.check(self._L.gf_sampler_set_state(self._h, p.ctypes.data_as(self._lib._dp)), "gf_sampler_set_state")
        self._have_state = True

    def reset(self):
        self._lib.check(self._L.gf_sampler_reset(self._h), "gf_sampler_reset")

    def run_mcmc(self, pos0, N, thin=1, storechain=True):
        """Advance N steps (asynchronous launches, then one sync); returns (pos, lnprob, None)."""
        if pos0 is not None:
            self._set_state(pos0)
        if not self._have_state:
            raise ValueError("no starting position")
        self._lib.check(self._L.gf_sampler_run(self._h, int(N), int(thin), 1 if storechain else 0), "gf_sampler_run")
        self._lib.check(self._L.gf_sampler_sync(self._h), "gf_sampler_sync")
        self._check_flags()
        pos, lnp = self.state
        return pos, lnp, None

    def run_mcmc_to_host(self, pos0, N, thin=1, lnprob=False, out=None):
        """run_mcmc(pos0, N, thin) with the chain's read-back overlapped with the run (gf_sampler_run_to_host): every finished
        block of steps crosses PCIe while the GPU computes the next ones.  Returns the stored chain in the device's order,
        (nchains, nstored, nwalkers, ndim) [and the lnprob chain, (nchains, nstored, nwalkers)] -- `flat_steps()` /
        `chain` of the same run, already on the host."""
        if pos0 is not None:
            self._set_state(pos0)
        if not self._have_state:
            raise ValueError("no starting position")
        thin = int(thin)
        ns = int(self._L.gf_sampler_nstored(self._h)) + (int(N) + thin - 1) // thin
        c = np.empty((self.nchains, ns, self.k, self.dim)) if out is None else out
        if c.shape != (self.nchains, ns, self.k, self.dim) or c.dtype != np.float64 or not c.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape %r" % ((self.nchains, ns, self.k, self.dim),))
        lnp = np.empty((self.nchains, ns, self.k)) if lnprob else None
        tail = self._C.c_double(0.0)
        self._lib.check(self._L.gf_sampler_run_to_host(self._h, int(N), thin, c.ctypes.data_as(self._lib._dp),
                                                       lnp.ctypes.data_as(self._lib._dp) if lnprob else None,
                                                       self._C.byref(tail)),
                        "gf_sampler_run_to_host")
        self.readback_tail_s = float(tail.value)          # what of the read-back was not hidden behind the run
        self._check_flags()
        return (c, lnp) if lnprob else c

    def run_async(self, pos0, N, thin=1, storechain=True):
        """Enqueue N steps on the model's stream and return immediately; pair with `wait()`.  Samplers of
        different models live on different streams, so several small ensembles overlap on the GPU."""
        if pos0 is not None:
            self._set_state(pos0)
        if not self._have_state:
            raise ValueError("no starting position")
        self._lib.check(self._L.gf_sampler_run(self._h, int(N), int(thin), 1 if storechain else 0), "gf_sampler_run")

    def wait(self):
        self._lib.check(self._L.gf_sampler_sync(self._h), "gf_sampler_sync")
        self._check_flags()

    def sample(self, p0, iterations=1, thin=1, storechain=True, chunk=None):
        """Generator with emcee-2's shape: yields (pos, lnprob, state) after every `chunk` steps
        (default: ~100 yields per call) so progress bars keep working without a sync per step."""
        if p0 is not None:
            self._set_state(p0)
        chunk = int(chunk or max(1, iterations // 100))
        thin = int(thin)
        chunk = max(thin, (chunk // thin) * thin)       # every gf_sampler_run call starts on a stored step
        done = 0
        while done < iterations:
            m = min(chunk, iterations - done)
            self._lib.check(self._L.gf_sampler_run(self._h, m, int(thin), 1 if storechain else 0), "gf_sampler_run")
            self._lib.check(self._L.gf_sampler_sync(self._h), "gf_sampler_sync")
            self._check_flags()
            done += m
            pos, lnp = self.state
            for _ in range(m):
                yield pos, lnp, None

    def _count_nonunitary(self):
        """Non-unitary proposals since the last reset (device counter).  Every proposal's verdict is settled on the device
        BEFORE its accept step: almost all by the half-step kernel itself (csrc/gf_bsm_device.hpp, tiers 1-2), the rest by
        k_stretch_settle, which replays the reference's arithmetic in emulated x87 (csrc/gf_unitarity.hip) -- so the count is
        exact and a rejected-as-non-unitary proposal never enters the chain."""
        n = (self._C.c_uint32 * 1)()
        self._lib.check(self._L.gf_sampler_get_chain(self._h, None, None, None, n), "gf_sampler_get_chain")
        return int(n[0])

    @property
    def nonunitary_proposals(self):
        """How many proposals since the last reset a reference run would have died on (fr.py:493-498)."""
        return self._count_nonunitary()

    def chain_stats(self):
        """Diagnostics: the per-chain census of the one-workgroup-per-chain sampler (k_stretch_chain) since the last reset, a dict of
        (nchains,) arrays -- seconds spent proposing / settling proposals the chain waited for / settling in bulk, and the counts of
        such proposals and passes.  None where that kernel has not run."""
        fn = getattr(self._L, "gf_internal_sampler_chain_stats", None)
        if fn is None:
            return None
        out = np.zeros((self.nchains, 8), dtype=np.uint64)
        fn.restype, fn.argtypes = self._C.c_int, [self._C.c_void_p, self._C.POINTER(self._C.c_uint64)]
        if fn(self._h, out.ctypes.data_as(self._C.POINTER(self._C.c_uint64))) != 0:
            return None
        return {"propose_s": out[:, 0] * 1e-9, "settle_s": out[:, 1] * 1e-9, "bulk_s": out[:, 2] * 1e-9, "waited_for": out[:, 3],
                "passes_that_waited": out[:, 4], "settled_in_bulk": out[:, 5], "passes": out[:, 6], "bulk_settlements": out[:, 7]}

    def launch_shape(self):
        """Diagnostics: how a BSM sampler on small ensembles launches its steps -- {"shape": "per chain" | "grid" | "undecided",
        "probe_us_per_16_steps": {"per chain": .., "grid": ..}} -- decided at the start of every run of 128 steps or more by timing a block of
        each on its own chains (gf_sampler_run); GF_SAMPLER_CHAIN=1 / 0 forces one.  The chain is the same bit for bit either way."""
        fn = getattr(self._L, "gf_internal_sampler_shape", None)
        if fn is None:
            return None
        out = (self._C.c_double * 3)()
        fn.restype, fn.argtypes = self._C.c_int, [self._C.c_void_p, self._C.POINTER(self._C.c_double)]
        if fn(self._h, out) != 0:
            return None
        return {"shape": ("undecided", "per chain", "grid")[int(out[0])], "probe_us_per_16_steps": {"per chain": float(out[1]), "grid": float(out[2])}}

    def run_to_host_times(self):
        """Diagnostics: where the host thread's time went in the process's last `run_mcmc_to_host` -- seconds issuing the blocks'
        copies (total, longest call), waiting for blocks to complete (total, longest), the number of blocks, and enqueueing the
        blocks of steps (total, longest launch)."""
        fn = getattr(self._L, "gf_internal_run_to_host_times", None)
        if fn is None:
            return None
        out = (self._C.c_double * 8)()
        fn.restype, fn.argtypes = self._C.c_int, [self._C.POINTER(self._C.c_double)]
        if fn(out) != 0:
            return None
        extra = {}
        fn2 = getattr(self._L, "gf_internal_run_prologue_times", None)
        if fn2 is not None:
            pr = (self._C.c_double * 4)()
            fn2.restype, fn2.argtypes = self._C.c_int, [self._C.POINTER(self._C.c_double)]
            if fn2(pr) == 0:
                extra = {"grow_chain_buffers_s": round(pr[0], 4), "graph_capture_s": round(pr[1], 4)}
        return {**extra, "copy_issue_s": round(out[0], 4), "copy_issue_max_s": round(out[1], 4), "block_wait_s": round(out[2], 4), "block_wait_max_s": round(out[3], 4),
                "blocks": int(out[4]), "launch_s": round(out[5], 4), "launch_max_s": round(out[6], 4), "before_first_block_s": round(out[7], 4)}

    def undecided_census(self):
        st = self.chain_stats()
        if st is None:
            return None
        return {"undecided_waited_for": int(st["waited_for"].sum()), "undecided_settled_in_bulk": int(st["settled_in_bulk"].sum()),
                "chains_that_waited": int(np.count_nonzero(st["waited_for"])),
                "slowest_chain": {k: (float(v[np.argmax(st["propose_s"] + st["settle_s"] + st["bulk_s"])])) for k, v in st.items()}}

    def _check_flags(self):
        # reference behaviour: AssertionError out of test_unitarity kills the run (fr.py:493-498); with
        # on_nonunitary="-inf" the count is only taken when somebody asks for it (`nonunitary_proposals`)
        if self.on_nonunitary == "raise":
            bad = self._count_nonunitary()
            if bad:
                raise AssertionError("Matrix is not unitary! (%d proposals)" % bad)

    # -- results --------------------------------------------------------------------------
    @property
    def iterations(self):
        return int(self._L.gf_sampler_iterations(self._h))

    @property
    def state(self):
        pos = np.empty((self.nchains, self.k, self.dim))
        lnp = np.empty((self.nchains, self.k))
        self._lib.check(self._L.gf_sampler_get_state(self._h, pos.ctypes.data_as(self._lib._dp),
                                                     lnp.ctypes.data_as(self._lib._dp)), "gf_sampler_get_state")
        return (pos[0], lnp[0]) if self.nchains == 1 else (pos, lnp)

    def _fetch(self, chain=False, lnprob=False, naccepted=False):
        """Only what is asked for crosses PCIe (the C entry point skips NULL arrays)."""
        from .model import empty_for_download
        ns = int(self._L.gf_sampler_nstored(self._h))
        c = empty_for_download((self.nchains, ns, self.k, self.dim), copier_maps_pages=True) if chain else None
        lnp = empty_for_download((self.nchains, ns, self.k), copier_maps_pages=True) if lnprob else None
        nacc = np.empty((self.nchains, self.k), dtype=np.uint32) if naccepted else None
        self._lib.check(self._L.gf_sampler_get_chain(
            self._h, c.ctypes.data_as(self._lib._dp) if chain else None, lnp.ctypes.data_as(self._lib._dp) if lnprob else None,
            nacc.ctypes.data_as(self._C.POINTER(self._C.c_uint32)) if naccepted else None, None), "gf_sampler_get_chain")
        return c, lnp, nacc

    def chain_to_host(self, out):
        """The stored chain in the device's order, (nchains, nstored, nwalkers, ndim), into an array of the caller's (a rank's
        region of a shared host segment: scan.SharedHostGather)."""
        ns = int(self._L.gf_sampler_nstored(self._h))
        if out.shape != (self.nchains, ns, self.k, self.dim) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape %r" % ((self.nchains, ns, self.k, self.dim),))
        self._lib.check(self._L.gf_sampler_get_chain(self._h, out.ctypes.data_as(self._lib._dp), None, None, None), "gf_sampler_get_chain")
        return out

    @property
    def chain(self):
        """(nwalkers, nsteps, ndim) like emcee-2 [(nchains, nwalkers, nsteps, ndim) when nchains > 1]."""
        c = np.ascontiguousarray(self._fetch(chain=True)[0].transpose(0, 2, 1, 3))
        return c[0] if self.nchains == 1 else c

    @property
    def flatchain(self):
        c = self.chain
        return c.reshape(-1, self.dim) if self.nchains == 1 else c.reshape(self.nchains, -1, self.dim)

    @property
    def lnprobability(self):
        lp = np.ascontiguousarray(self._fetch(lnprob=True)[1].transpose(0, 2, 1))
        return lp[0] if self.nchains == 1 else lp

    @property
    def acceptance_fraction(self):
        nacc = self._fetch(naccepted=True)[2].astype(np.float64) / max(self.iterations, 1)
        return nacc[0] if self.nchains == 1 else nacc

    def postprocess(self, want_fr=True, want_status=False, nbins=None, models=None, step_major=False):
        """Chain post-processing on the device: the measured composition of every stored sample
        (scripts/mc_unitary.py:189-193, mc_texture.py:216-221) and/or its flavor histogram
        (golemflavor/plot.py:365-370).  Returns a dict with 'fr' (nwalkers, nsteps, 3), 'status',
        'hist' (nbins, nbins, nbins) -- each with a leading chain axis when nchains > 1.

        models: one Model / LnProb per chain to propagate with instead of the sampled posterior
        (mc_texture.py samples the priors and propagates with the grid point's texture model).
        step_major: leave 'fr' / 'status' in the device order (nsteps, nwalkers, ...), as `flat_steps`."""
        C = self._C
        from .model import empty_for_download
        ns = int(self._L.gf_sampler_nstored(self._h))
        fr = empty_for_download((self.nchains, ns, self.k, 3)) if want_fr else None
        st = empty_for_download((self.nchains, ns, self.k), dtype=np.int32) if want_status else None
        hist = np.zeros((self.nchains, nbins, nbins, nbins), dtype=np.uint64) if nbins else None
        handles = None
        if models is not None:
            ms = [getattr(m, "model", m) for m in models]
            if len(ms) != self.nchains:
                raise ValueError("%d post-processing models for %d chains" % (len(ms), self.nchains))
            handles = (C.c_void_p * self.nchains)(*[m._h.value if hasattr(m._h, "value") else m._h for m in ms])
        self._lib.check(self._L.gf_sampler_postprocess_with(
            self._h, handles, fr.ctypes.data_as(self._lib._dp) if want_fr else None,
            st.ctypes.data_as(self._lib._ip) if want_status else None, int(nbins or 0),
            hist.ctypes.data_as(C.POINTER(C.c_uint64)) if nbins else None), "gf_sampler_postprocess_with")
        out = {}
        if want_fr:
            f = fr if step_major else np.ascontiguousarray(fr.transpose(0, 2, 1, 3))
            out["fr"] = f[0] if self.nchains == 1 else f
        if want_status:
            t = st if step_major else np.ascontiguousarray(st.transpose(0, 2, 1))
            out["status"] = t[0] if self.nchains == 1 else t
        if nbins:
            out["hist"] = hist[0] if self.nchains == 1 else hist
        return out

    @property
    def nstored(self):
        return int(self._L.gf_sampler_nstored(self._h))

    def chain_to_device(self, d_chain):
        """The stored chain, packed [nchains][nstored][nwalkers][ndim], into a device buffer of the caller's (what a
        multi-GPU gather sends: `dist.RcclBackend.allgather_device`)."""
        self._lib.check(self._L.gf_sampler_get_chain_device(self._h, d_chain, None), "gf_sampler_get_chain_device")

    def postprocess_to_device(self, d_fr, d_status=None, models=None):
        """`postprocess` with device destinations: d_fr [nchains][nstored][nwalkers][3], d_status optional."""
        C = self._C
        handles = None
        if models is not None:
            ms = [getattr(m, "model", m) for m in models]
            handles = (C.c_void_p * self.nchains)(*[m._h.value if hasattr(m._h, "value") else m._h for m in ms])
        self._lib.check(self._L.gf_sampler_postprocess_device(self._h, handles, d_fr, d_status), "gf_sampler_postprocess_device")

    def postprocess_rows_to_device(self, d_rows, models=None):
        """The rows a scan saves -- composition (NaN where the reference would have raised) then the sample -- assembled
        on the device: d_rows [nchains][nstored][nwalkers][3 + ndim]."""
        C = self._C
        handles = None
        if models is not None:
            ms = [getattr(m, "model", m) for m in models]
            handles = (C.c_void_p * self.nchains)(*[m._h.value if hasattr(m._h, "value") else m._h for m in ms])
        self._lib.check(self._L.gf_sampler_postprocess_rows_device(self._h, handles, d_rows), "gf_sampler_postprocess_rows_device")

    def postprocess_rows(self, models=None, out=None):
        """The same rows on the host, (nchains, nstored * nwalkers, 3 + ndim): the finished chains cross PCIe while the later
        ones are still being post-processed."""
        C = self._C
        handles = None
        if models is not None:
            ms = [getattr(m, "model", m) for m in models]
            handles = (C.c_void_p * self.nchains)(*[m._h.value if hasattr(m._h, "value") else m._h for m in ms])
        ns = int(self._L.gf_sampler_nstored(self._h))
        if out is None:
            out = np.empty((self.nchains, ns * self.k, 3 + self.dim))
        if out.shape != (self.nchains, ns * self.k, 3 + self.dim) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape %r" % ((self.nchains, ns * self.k, 3 + self.dim),))
        self._lib.check(self._L.gf_sampler_postprocess_rows(self._h, handles, out.ctypes.data_as(self._lib._dp)),
                        "gf_sampler_postprocess_rows")
        return out

    def flat_steps(self):
        """The stored samples in the order the device holds them, (nsteps*nwalkers, ndim) [leading chain axis
        when nchains > 1]: `flatchain` without the transposition to emcee's walker-major order -- for
        consumers that treat the chain as a bag of samples (histograms, post-processing)."""
        c = self._fetch(chain=True)[0]
        c = c.reshape(self.nchains, -1, self.dim)
        return c[0] if self.nchains == 1 else c

    @property
    def acor(self):
        return self.get_autocorr_time()

    def walker_mean(self):
        """Ensemble mean of every stored step, (nsteps, ndim) [leading chain axis when nchains > 1],
        reduced on the device: only nsteps x ndim numbers cross PCIe."""
        ns = int(self._L.gf_sampler_nstored(self._h))
        out = np.empty((self.nchains, ns, self.dim))
        self._lib.check(self._L.gf_sampler_walker_mean(self._h, out.ctypes.data_as(self._lib._dp)), "gf_sampler_walker_mean")
        return out[0] if self.nchains == 1 else out

    def get_autocorr_time(self, c=5, tol=50):
        m = self.walker_mean()                                 # emcee-2: acor of the ensemble-averaged chain
        if self.nchains == 1:
            return integrated_time(m, c=c, tol=tol)
        return np.array([integrated_time(x, c=c, tol=tol) for x in m])

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._L.gf_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mcmc(p0, ln_prob, ndim, nwalkers, burnin, nsteps, threads=1, device_resident=None, seed=None):
    """Run the MCMC: burn-in, reset, production; returns samples reshaped to (-1, ndim).

    Same signature, prints and return as golemflavor/mcmc.py:27-53.  When `ln_prob` is a
    golemflavor_amd LnProb the whole chain runs on the GPU (DeviceEnsembleSampler: proposal, lnprob
    and accept in one launch per half-ensemble, walkers resident in HBM); `device_resident=False`
    forces the host-driven sampler (one launch + PCIe round trip per half-ensemble), which is also
    what any plain Python callable gets.  `seed` keys the device sampler's Philox stream (default:
    drawn from numpy's global RNG, so `np.random.seed(args.seed)` makes runs reproducible as in the
    reference's scripts)."""
    if device_resident is None:
        device_resident = hasattr(getattr(ln_prob, "model", None), "_h")
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    if device_resident:
        sampler = DeviceEnsembleSampler(nwalkers, ndim, ln_prob, seed=seed)
    else:
        # (emcee seeds its RandomState from the OS; keyed by the global RNG instead, a host-driven run is as reproducible
        # under np.random.seed as a device-resident one)
        sampler = EnsembleSampler(nwalkers, ndim, ln_prob, threads=threads, seed=seed)

    print("Running burn-in")
    pos = p0
    for result in tqdm(sampler.sample(p0, iterations=burnin), total=burnin):
        pos, prob, state = result
    sampler.reset()
    print("Finished burn-in")

    print("Running")
    for _ in tqdm(sampler.sample(pos, iterations=nsteps), total=nsteps):
        pass
    print("Finished")

    samples = sampler.chain.reshape((-1, ndim))
    print('acceptance fraction', sampler.acceptance_fraction)
    print('sum of acceptance fraction', np.sum(sampler.acceptance_fraction))
    print('np.unique(samples[:,0]).shape', np.unique(samples[:, 0]).shape)
    try:
        print('autocorrelation', sampler.acor)
    except Exception:
        print('WARNING : NEED TO RUN MORE SAMPLES')

    return samples


def _flag(text):
    """'True' / 'False' command-line values (the reference's misc.parse_bool)."""
    t = str(text).strip().lower()
    if t in ("true", "t", "1", "yes", "y"):
        return True
    if t in ("false", "f", "0", "no", "n"):
        return False
    raise ValueError("{0!r} is not a boolean".format(text))


def _seed_type(text):
    from .enums import MCMCSeedType
    try:
        return MCMCSeedType[str(text).upper()]
    except KeyError:
        import argparse
        raise argparse.ArgumentTypeError("--mcmc-seed-type must be one of %s" % [e.name.lower() for e in MCMCSeedType])


def mcmc_argparse(parser):
    """The sampler's command-line group, as the reference's scripts compose it (golemflavor/mcmc.py:56-85: same flags,
    types and defaults): --run-mcmc, --burnin 100, --nwalkers 60, --nsteps 2000, --mcmc-seed-type uniform|gaussian,
    --plot-angles, --plot-elements (the plotting flags are accepted and carried; plotting is out of scope here)."""
    from .enums import MCMCSeedType
    g = parser
    g.add_argument('--run-mcmc', type=_flag, default=True, help='Run the MCMC')
    g.add_argument('--burnin', type=int, default=100, help='Amount to burnin')
    g.add_argument('--nwalkers', type=int, default=60, help='Number of walkers')
    g.add_argument('--nsteps', type=int, default=2000, help='Number of steps to run')
    g.add_argument('--mcmc-seed-type', type=_seed_type, default=MCMCSeedType.UNIFORM, choices=list(MCMCSeedType),
                   help='Type of distrbution to make the initial MCMC seed')
    g.add_argument('--plot-angles', type=_flag, default=False, help='Plot MCMC triangle in the angles space')
    g.add_argument('--plot-elements', type=_flag, default=False, help='Plot MCMC triangle in the mixing elements space')
    return parser


def solve_ratio(fr):
    """'1_2_0' for small-integer ratios, else two-decimal floats (golemflavor/misc.py:34-41; the reference
    reduces with a floating-point gcd, restated here with a tolerance)."""
    fr = [float(x) for x in fr]

    def fgcd(a, b):
        while abs(b) > 1e-9:
            a, b = b, a % b
        return a
    den = 0.0
    for x in fr:
        den = fgcd(den, x) if den else x
    f = [int(round(x / den)) if den else 0 for x in fr]
    if any(v not in (1, 2, 0) for v in f) or any(abs(x / den - v) > 1e-6 for x, v in zip(fr, f)):
        return '{0:.2f}_{1:.2f}_{2:.2f}'.format(*fr)
    return '{0}_{1}_{2}'.format(*f)


def chain_identifier(args):
    """File-name stem of the reference's chain files so that its plotting scripts find ours
    (golemflavor/misc.py:44-51 gen_identifier): `_DIM{d}_sfr_{..}[_mfr_{..}][_{texture}]`; `_mfr_` only for
    injected (Asimov / realisation) data."""
    stem = '_DIM{0}'.format(args.dimension)
    stem += '_sfr_' + solve_ratio(args.source_ratio)
    data = getattr(getattr(args, "data", None), "name", None)
    if data in ("ASIMOV", "REALISATION") or (data is None and getattr(args, "injected_ratio", None) is not None):
        stem += '_mfr_' + solve_ratio(args.injected_ratio)
    tex = getattr(args, "texture", None)
    if tex is not None and getattr(tex, "name", "NONE") != "NONE":
        stem += '_{0}'.format(tex.name)
    return stem


def flat_seed(paramset, nwalkers):
    """p0 ~ U(seed_lo, seed_hi) per parameter, shape (nwalkers, ndim); global np.random like the
    reference (mcmc.py:88-96) so `np.random.seed(args.seed)` reproduces its p0."""
    seeds = np.array(paramset.seeds, dtype=np.float64)
    return np.random.uniform(low=seeds[:, 0], high=seeds[:, 1], size=[nwalkers, len(paramset)])


def gaussian_seed(paramset, nwalkers):
    """p0 ~ N(values, stds) (mcmc.py:99-105)."""
    return np.random.normal(paramset.values, paramset.stds, size=[nwalkers, len(paramset)])


def save_chains(chains, outfile):
    """np.save the chains to `outfile`(.npy), creating the directory (mcmc.py:108-126).

    The reference appends '.npy' even when the name already ends with it; here the suffix is
    added only when missing, so the file lands where the message says."""
    of = outfile if outfile.endswith('.npy') else outfile + '.npy'
    d = os.path.dirname(of)
    if d:
        os.makedirs(d, exist_ok=True)
    print('Saving chains to location {0}'.format(of))
    np.save(of, chains)
    return of
