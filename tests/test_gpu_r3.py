"""GPU, round 3: args.no_bsm, the unitarity queues' overflow report, gf_device_trim."""
import argparse
import os
import subprocess
import sys

import numpy as np
import pytest

from common import BIN_EDGES, uniform_theta
from golemflavor_amd import _lib
from golemflavor_amd import configs as Cf
from golemflavor_amd import fr as fr_utils
from golemflavor_amd import llh as llh_utils
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_bsm_is_the_standard_propagation_on_the_device(oracle):
    """args.no_bsm (golemflavor/fr.py:437-438).  PARITY UNPINNED: the reference's own branch cannot run (it hands u_to_fr a
    2-D source flux), so there is no reference output; the build defines it as u_to_fr(source_ratio, sm_u)
    and this test holds the device to the oracle's u_to_fr / lnprior / multi_gaussian restatements (each pinned on its own
    golden set) composed that way -- and checks that the BSM machinery is really out of the loop."""
    rng = np.random.default_rng(5)
    asimov, ps = Cf.fr_paramsets(6, fr_utils.fr_to_angles((1, 1, 1)))
    src = np.array([0.0, 1.0, 0.0])
    args = argparse.Namespace(source_ratio=src, dimension=6, texture=Texture.OEU, binning=BIN_EDGES, no_bsm=True)
    th = uniform_theta(ps, 5000, rng, seeds=True)
    th[:, 11] = rng.uniform(-56, -30, len(th))                  # logLam over the whole range, failing region included
    th[::97, 0] = 1.5                                           # some rows outside the box
    om = oracle.make_model(ps, "SM_GAUSS", bestfit_fr=(1 / 3,) * 3, smearing=0.02, source_ratio=src)
    assert list(om.idx_sm) == [0, 1, 2, 3] and list(om.idx_src) == [-1, -1]
    want_lp, want_fr = oracle.lnprob_batch(om, th, want_fr=True)
    # the reference-named entry points
    frs = fr_utils.flux_averaged_BSMu(th, args, -2.0, ps)
    ok = np.isfinite(want_lp)
    assert np.abs(frs[ok] - want_fr[ok]).max() < 1e-10
    one = fr_utils.flux_averaged_BSMu(th[1], args, -2.0, ps)
    assert one.shape == (3,) and np.abs(one - want_fr[1]).max() < 1e-10
    f = llh_utils.bsm_ln_prob(args, asimov, ps, smearing=0.02)   # on_nonunitary="raise": nothing may raise, nothing is diagonalised
    lp = f(th)
    assert np.array_equal(np.isneginf(lp), np.isneginf(want_lp)) and np.isneginf(lp[::97]).all()
    assert np.max(np.abs(lp[ok] - want_lp[ok]) / np.abs(want_lp[ok])) < 1e-10
    assert f.model.mode == _lib.GF_MODE_SM_GAUSS
    # independent of the new-physics scale: the same rows with another logLam give the same composition
    th2 = th.copy()
    th2[:, 11] = -50.0
    assert np.array_equal(fr_utils.flux_averaged_BSMu(th2, args, -2.0, ps)[ok], frs[ok])
    # and it is NOT what the BSM path returns at a high scale
    args.no_bsm = False
    args.texture = Texture.OET
    th3 = th[ok][:64].copy()
    th3[:, 11] = -42.0
    assert np.abs(fr_utils.flux_averaged_BSMu(th3, args, -2.0, ps) - want_fr[ok][:64]).max() > 1e-3
    f.close()
    # a paramset without the mass splittings: NuFIT mixing whatever theta holds (fr.py:433-435)
    ps4 = Cf.unitary_paramset()
    args4 = argparse.Namespace(source_ratio=np.array([1.0, 2.0, 0.0]) / 3, dimension=3, texture=Texture.OUT, binning=BIN_EDGES,
                               no_bsm=True)
    got = fr_utils.flux_averaged_BSMu(uniform_theta(ps4, 16, rng), args4, -2.0, ps4)
    nufit = np.asarray(fr_utils.u_to_fr((1, 2, 0), fr_utils.NUFIT_U), dtype=float)
    assert np.abs(got - nufit).max() < 1e-12


_OVERFLOW_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
from common import BIN_EDGES, uniform_theta
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
ps = Cf.texture_paramset(6)
rng = np.random.default_rng(2)
n = 400_000
th = uniform_theta(ps, n, rng, seeds=True)
th[:, -1] = rng.uniform(-56, -30, n)
desc = compile_model(ps, "BSM_GAUSS", texture=Texture.OEU, dimension=6, binning=BIN_EDGES, source_ratio=(0., 1., 0.),
                     bestfit_fr=(1 / 3,) * 3, smearing=0.02)
out = []
with Model(desc) as m:
    try:
        m.lnprob(th)
        out.append("host:ok")
    except _lib.GolemHipError as exc:
        out.append("host:%d" % exc.code)
    d_th = m.alloc(th.nbytes).upload(th); d_out, d_st = m.alloc(8 * n), m.alloc(4 * n)
    m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr)
    try:
        m.sync()
        out.append("device:ok")
    except _lib.GolemHipError as exc:
        out.append("device:%d" % exc.code)
    # the report is consumed: a batch that fits goes through afterwards
    lp, st = m.lnprob(th[:1000])
    out.append("after:%d" % int(np.sum(st == 2) > 0))
print(" ".join(out), "|" + _lib.diagnostic_overrides())
"""


def test_a_full_unitarity_queue_is_reported_not_swallowed(tmp_path):
    """queue_pairs drops a (walker, bin) pair when the arbitration queue is full.  The host cuts batches so that this
    cannot happen; if the invariant is ever broken the call must FAIL (GF_ERR_QUEUE_OVERFLOW), not return statuses with
    verdicts missing.  The invariant is broken on purpose here: GF_DIAG_UQ_OVERCOMMIT (pieces 4x what the queue holds, a
    result-changing override: needs GF_DIAGNOSTICS=1) with EVERY (walker, bin) pair sent to arbitration (tier 1 off, band of 12
    decades) through a 131 072-item queue."""
    script = tmp_path / "child.py"
    script.write_text(_OVERFLOW_CHILD)
    env = dict(os.environ, GF_DIAGNOSTICS="1", GF_DIAG_UQ_OVERCOMMIT="1", GF_UQ_MAX_ITEMS="131072", GF_UNI_BAND_DECADES="12",
               GF_UNI_NO_WEIGHT_GATE="1", PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = res.stdout.strip().splitlines()[-1]
    assert line.startswith("host:%d device:%d after:1 |" % (_lib.GF_ERR_QUEUE_OVERFLOW, _lib.GF_ERR_QUEUE_OVERFLOW)), line
    assert "GF_DIAG_UQ_OVERCOMMIT=1" in line and "GF_UNI_BAND_DECADES=12" in line
    # without the deliberate overcommit the same run is clean
    env.pop("GF_DIAG_UQ_OVERCOMMIT")
    res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    assert res.stdout.strip().splitlines()[-1].startswith("host:ok device:ok after:1 |")


def test_device_trim_releases_idle_workspaces():
    """gf_device_trim hands back what the pool keeps between uses: after a status batch the stream's unitarity workspace
    (queues + side buffer, ~170 B per walker) stays with the pooled stream; trim frees it, and the next model works."""
    ps = Cf.texture_paramset(6)
    rng = np.random.default_rng(3)
    n = 200_000
    th = uniform_theta(ps, n, rng, seeds=True)
    th[:, -1] = rng.uniform(-56, -36, n)
    desc = compile_model(ps, "BSM_GAUSS", texture=Texture.OET, dimension=6, binning=BIN_EDGES, source_ratio=(0., 1., 0.),
                         bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(desc) as m:
        want = m.lnprob(th)
    freed = _lib.device_trim(0)
    assert freed >= n * 20 * 8                                   # at least the arbitration queue of that batch
    assert _lib.device_trim(0) < freed                           # nothing of that size left to free
    with Model(desc) as m:
        got = m.lnprob(th)
    assert np.array_equal(want[0], got[0], equal_nan=True) and np.array_equal(want[1], got[1])
    assert _lib.lib().gf_device_trim(99, None) == _lib.GF_ERR_NO_DEVICE


def test_large_reads_come_back_through_the_pinned_ring_intact():
    """Device-to-host copies of 16 MB and more go through four pinned 32 MB slots and host threads (gf_internal_d2h); smaller
    ones are a plain hipMemcpy.  Sizes either side of the threshold, not multiples of a slot or of a page, come back bit for
    bit, also into a destination that is not page-aligned."""
    rng = np.random.default_rng(11)
    with Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY")) as m:
        for nbytes in (8 * 1000, (16 << 20) - 8, (16 << 20) + 8, (32 << 20) + 8, 5 * (32 << 20) + 8 * 12345):
            src = rng.integers(0, 2 ** 63, nbytes // 8, dtype=np.int64).view(np.float64)
            d = m.alloc(nbytes).upload(src)
            got = d.download((nbytes // 8,))
            assert np.array_equal(got.view(np.int64), src.view(np.int64)), nbytes
            back = np.empty(nbytes // 8 + 3)
            _lib.check(_lib.lib().gf_memcpy_d2h(m._h, back[3:].ctypes.data, d.ptr, nbytes), "d2h")
            assert np.array_equal(back[3:].view(np.int64), src.view(np.int64)), nbytes
            d.free()
            _lib.device_trim(0)                                  # frees the pinned slots too: the next large read allocates them again


@pytest.mark.parametrize("case", ["sm_one_workgroup", "sm_grid", "sm_grid_through_the_ring", "sm_grid_through_the_ring_block_pipelines",
                                  "sm_grid_one_wide_chain", "bsm_stacked"])
def test_run_to_host_returns_the_chain_of_the_plain_run(case, golden, monkeypatch):
    """gf_sampler_run_to_host copies every finished block of steps to the host while the run goes on.  The result must be,
    bit for bit, what run_mcmc followed by the chain's read-back gives for the same seed: one-workgroup sampler (a single
    mark), the grid sampler (graph replays of 16 steps + an eager tail), stacked BSM chains with the settle step; with a
    stored prefix from an earlier run, with thinning, with the lnprob chain."""
    from golemflavor_amd import mcmc as mcmc_utils
    rng = np.random.default_rng(9)
    if case.endswith("_block_pipelines"):           # round 4: one read-back pipe per run is the default; the A/B switch gives each block its own
        monkeypatch.setenv("GF_RUN_TO_HOST_NO_PIPE", "1")
        case = case[:-len("_block_pipelines")]
    if case == "bsm_stacked":
        asimov, ps = Cf.fr_paramsets(6, fr_utils.fr_to_angles((1, 1, 1)))
        fs = []
        for tex, x in ((Texture.OET, 0.2), (Texture.OUT, 0.7), (Texture.OEU, 0.5)):
            args = argparse.Namespace(source_ratio=np.array([x, 1 - x, 0.0]), dimension=6, texture=tex, binning=BIN_EDGES)
            fs.append(llh_utils.bsm_ln_prob(args, asimov, ps, smearing=0.02, on_nonunitary="-inf"))
        nw, nd, nch = 128, 12, 3
        box = np.array(ps.seeds, dtype=float)
        p0 = rng.uniform(box[:, 0], box[:, 1], size=(nch, nw, nd))
        p0[:, :, 11] = rng.uniform(-40, -33, size=(nch, nw))          # up into OEU's failing region: proposals get parked
        make = lambda: mcmc_utils.DeviceEnsembleSampler(nw, nd, fs, seed=4)
    else:
        asimov, ps = Cf.notebook_paramsets(golden["g6_asimov_angles"])
        f = llh_utils.notebook_ln_prob(asimov, ps)
        fs = [f]
        # (blocks of 16 steps below 16 MB are plain strided copies; 32 chains of 4096 walkers make 100 MB blocks: the pinned ring)
        # one chain of 65 536 walkers: a block of 16 steps is ONE row of 50 MB, wider than a 16 MB slot -> the row-by-row fallback
        nw, nd, nch = {"sm_one_workgroup": (100, 6, 5), "sm_grid": (4096, 6, 2), "sm_grid_through_the_ring": (4096, 6, 32),
                       "sm_grid_one_wide_chain": (65536, 6, 1)}[case]
        box = np.array(ps.seeds, dtype=float)
        p0 = rng.uniform(box[:, 0], box[:, 1], size=(nch, nw, nd))
        make = lambda: mcmc_utils.DeviceEnsembleSampler(nw, nd, f, nchains=nch, seed=4)
    for thin, first, second in ((1, 0, 75), (1, 40, 53), (3, 20, 100)):
        a, b = make(), make()
        for smp in (a, b):
            smp.on_nonunitary = "-inf"
            smp.run_mcmc(p0, 30, storechain=False)
            smp.reset()
            if first:
                smp.run_mcmc(None, first, thin=thin)                 # a stored prefix
        a.run_mcmc(None, second, thin=thin)
        want, want_lnp, _ = a._fetch(chain=True, lnprob=True)
        got, got_lnp = b.run_mcmc_to_host(None, second, thin=thin, lnprob=True)
        assert got.shape == want.shape == (nch, (first + thin - 1) // thin + (second + thin - 1) // thin, nw, nd)
        assert np.array_equal(got, want) and np.array_equal(got_lnp, want_lnp), (case, thin, first)
        assert np.array_equal(b.state[0], a.state[0]) and a.nonunitary_proposals == b.nonunitary_proposals
        assert np.array_equal(b.run_mcmc_to_host(None, 5, thin=thin)[:, :got.shape[1]], got)        # and the chain goes on
        assert np.array_equal(b.run_mcmc_to_host(None, 0), b._fetch(chain=True)[0])                 # nothing new: the stored chain
        a.close(); b.close()
    for f in fs:
        f.close()


def test_read_backs_into_a_registered_arena_equal_those_through_the_ring(golden):
    """ABI 5: a destination registered with the runtime (`gf_host_register`, scan.ResultArena) is written by the DMA engines directly --
    no pinned ring, no host copy threads.  Every large read-back route must deliver the same bytes either way: a device buffer's
    download (1-D, above the ring's threshold), the stored chain (`chain_to_host`: pitched, row = chain), the chain read back while
    it is sampled (`run_mcmc_to_host`: one pitched block per 16 steps through the pipe), and post-processed rows (gated pipeline)."""
    from golemflavor_amd import mcmc as mcmc_utils, scan
    asimov, ps = Cf.notebook_paramsets(golden["g6_asimov_angles"])
    f = llh_utils.notebook_ln_prob(asimov, ps)
    nch, nw, nd, nsteps = 32, 4096, 6, 48                                        # 100 MB blocks of 16 steps: the ring's territory
    box = np.array(ps.seeds, dtype=float)
    p0 = np.random.default_rng(5).uniform(box[:, 0], box[:, 1], size=(nch, nw, nd))
    arena = scan.ResultArena(nch * nsteps * nw * (3 + nd) * 8)
    assert arena.registered, arena.register_error
    try:
        # 1-D download
        m = f.model
        src = np.random.default_rng(1).standard_normal(40 << 20 >> 3)
        d = m.alloc(src.nbytes).upload(src)
        got = d.download(src.shape, out=arena.take(src.shape))
        assert np.array_equal(got.view(np.int64), src.view(np.int64)) and np.shares_memory(got, arena.array)
        d.free()
        # stored chain, streamed chain, rows
        a, b = (mcmc_utils.DeviceEnsembleSampler(nw, nd, f, nchains=nch, seed=4) for _ in range(2))
        for smp in (a, b):
            smp.run_mcmc(p0, 20, storechain=False)
            smp.reset()
        want = a.run_mcmc_to_host(None, nsteps)                                     # fresh numpy memory: the ring
        got = b.run_mcmc_to_host(None, nsteps, out=arena.take(want.shape))          # registered: direct
        assert np.shares_memory(got, arena.array) and np.array_equal(got, want)
        keep = got.copy()
        assert np.array_equal(b.chain_to_host(arena.take(want.shape)), keep)
        rows_want = a.postprocess_rows()
        rows_got = b.postprocess_rows(out=arena.take(rows_want.shape))
        assert np.shares_memory(rows_got, arena.array) and np.array_equal(rows_got, rows_want, equal_nan=True)
        a.close(); b.close()
    finally:
        arena.close()
    assert not arena.registered and arena.take((1,)) is None


def test_large_device_buffers_are_cached_not_freed():
    """gf_devcache.h (round 4): a device buffer of 64 MB or more goes to the library's free list when it is freed and is handed out
    again for the next request it fits -- the driver wipes freed memory with the DMA engine, which halved every read-back that
    followed a free.  Small buffers are freed as before; `gf_device_trim` hands the cached ones back."""
    import ctypes as C
    L = _lib.lib()
    L.gf_devcache_stats.restype = None
    L.gf_devcache_stats.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_ulonglong)]

    def stats():
        idle, live, reuses = C.c_size_t(), C.c_size_t(), C.c_ulonglong()
        L.gf_devcache_stats(0, C.byref(idle), C.byref(live), C.byref(reuses))
        return idle.value, live.value, reuses.value

    _lib.device_trim(0)
    idle0, live0, reuse0 = stats()
    assert idle0 == 0
    p = C.c_void_p()
    _lib.check(L.gf_device_malloc(0, 200 << 20, C.byref(p)), "gf_device_malloc")
    first = p.value
    assert stats()[1] == live0 + (200 << 20)
    _lib.check(L.gf_device_release(0, p), "gf_device_release")
    assert stats()[0] == 200 << 20 and stats()[1] == live0                       # idle in the cache, not freed
    q = C.c_void_p()
    _lib.check(L.gf_device_malloc(0, 180 << 20, C.byref(q)), "gf_device_malloc")    # fits (within +25 %): the same block
    assert q.value == first and stats()[2] == reuse0 + 1 and stats()[0] == 0
    r = C.c_void_p()
    _lib.check(L.gf_device_malloc(0, 100 << 20, C.byref(r)), "gf_device_malloc")    # nothing idle: a new block
    assert r.value != first
    _lib.check(L.gf_device_release(0, q), "gf_device_release")
    _lib.check(L.gf_device_release(0, r), "gf_device_release")
    s = C.c_void_p()
    _lib.check(L.gf_device_malloc(0, 110 << 20, C.byref(s)), "gf_device_malloc")    # 200 MB is more than 25 % too large: the 100 MB block is too small -> new
    assert s.value not in (first, r.value)
    _lib.check(L.gf_device_release(0, s), "gf_device_release")
    small = C.c_void_p()
    _lib.check(L.gf_device_malloc(0, 1 << 20, C.byref(small)), "gf_device_malloc")  # below the threshold: not tracked
    live = stats()[1]
    _lib.check(L.gf_device_release(0, small), "gf_device_release")
    assert stats()[1] == live
    assert stats()[0] == (200 << 20) + (100 << 20) + (110 << 20)
    assert _lib.device_trim(0) >= (200 << 20) + (100 << 20) + (110 << 20)
    assert stats()[0] == 0
