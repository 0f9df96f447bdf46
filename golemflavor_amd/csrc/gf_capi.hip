// gf_capi.hip -- host side of the C ABI declared in include/golemflavor_hip.h.
// Owns: descriptor validation, derivation of the per-run constants, device/stream/staging-buffer
// management and the launch calls.  There is deliberately no CPU evaluation path in this library.
#include <hip/hip_runtime.h>
#include <chrono>
#include <pthread.h>
#include <sys/mman.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/golemflavor_hip.h"
#include "gf_consts.h"
#include "gf_launch.h"
#include "gf_devcache.h"                // large device allocations are cached, not handed back to the driver (hipMalloc / hipFree are macros from here on)

static_assert(GF_MAX_DIM == 16 && GF_MAX_BINS == 64, "header / device constant mismatch");

extern "C" const char* gf_internal_env(const char* name, int affects_results);

namespace {

thread_local char g_err[512] = "";

int hip_fail(hipError_t e, const char* what)
{
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return GF_ERR_HIP;
}

#define GF_HIP(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

typedef long double ld;
typedef std::complex<long double> cld;

// golemflavor/fr.py:116-162 in the algebraic form, long double, host side only:
// used once per model for the fixed-texture projectors.
void mixing_matrix_ld(const double ang[4], cld u[3][3])
{
    const ld s12_2 = ang[0], c13_4 = ang[1], s23_2 = ang[2], dcp = ang[3];
    const ld c13_2 = std::sqrt(c13_4);
    const ld s12 = std::sqrt(s12_2), c12 = std::sqrt(1.0L - s12_2);
    const ld c13 = std::sqrt(c13_2), s13 = std::sqrt(1.0L - c13_2);
    const ld s23 = std::sqrt(s23_2), c23 = std::sqrt(1.0L - s23_2);
    const cld ep(std::cos(dcp), std::sin(dcp)), em = std::conj(ep);
    u[0][0] = c12 * c13;                       u[0][1] = s12 * c13;                       u[0][2] = s13 * em;
    u[1][0] = -s12 * c23 - c12 * s23 * s13 * ep; u[1][1] = c12 * c23 - s12 * s23 * s13 * ep; u[1][2] = s23 * c13;
    u[2][0] = s12 * s23 - c12 * c23 * s13 * ep;  u[2][1] = -c12 * s23 - s12 * c23 * s13 * ep; u[2][2] = c23 * c13;
}

// golemflavor/fr.py:138-161 angles_to_u operation by operation in long double (np.float128 on x86-64, the same libm):
// U = np.dot(np.dot(p1, p2), p3), np.dot accumulating from zero in index order.  Host side, once per model, for the
// per-model matrices of the unitarity arbitration (gf_unitarity.hip) -- their entries must be the reference's to the
// last bit, not merely to 1e-19.
void angles_to_u_ref_ld(const double ang[4], cld u[3][3])
{
    const ld s12_2 = ang[0], c13_4 = ang[1], s23_2 = ang[2];
    const ld c13_2 = sqrtl(c13_4);
    const ld t12 = asinl(sqrtl(s12_2)), t13 = acosl(sqrtl(c13_2)), t23 = asinl(sqrtl(s23_2));
    const ld c12 = cosl(t12), s12 = sinl(t12), c13 = cosl(t13), s13 = sinl(t13), c23 = cosl(t23), s23 = sinl(t23);
    const ld dcp = ang[3];
    const cld em(cosl(dcp), -sinl(dcp)), ep(cosl(dcp), sinl(dcp));      // EXP(-+1j * dcp)
    auto mul = [](cld a, cld b) { return cld(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real()); };
    auto dot = [&](const cld a[3][3], const cld b[3][3], cld out[3][3]) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                cld acc(0.0L, 0.0L);
                for (int k = 0; k < 3; ++k) { const cld p = mul(a[i][k], b[k][j]); acc = cld(acc.real() + p.real(), acc.imag() + p.imag()); }
                out[i][j] = acc;
            }
    };
    const cld p1[3][3] = {{1.0L, 0.0L, 0.0L}, {0.0L, c23, s23}, {0.0L, -s23, c23}};
    const cld p2[3][3] = {{c13, 0.0L, cld(s13 * em.real(), s13 * em.imag())}, {0.0L, 1.0L, 0.0L}, {cld(-s13 * ep.real(), -s13 * ep.imag()), 0.0L, c13}};
    const cld p3[3][3] = {{c12, s12, 0.0L}, {-s12, c12, 0.0L}, {0.0L, 0.0L, 1.0L}};
    cld t[3][3];
    dot(p1, p2, t);
    dot(t, p3, u);
}

void split_matrix_ld(const cld u[3][3], double hi[18], double lo[18])
{
    for (int k = 0; k < 9; ++k) {
        const ld v[2] = {u[k / 3][k % 3].real(), u[k / 3][k % 3].imag()};
        for (int q = 0; q < 2; ++q) {
            hi[2 * k + q] = (double)v[q];
            lo[2 * k + q] = (double)(v[q] - (ld)hi[2 * k + q]);
        }
    }
}

// Every environment override the library honours goes through gf_internal_env and is REMEMBERED: gf_diagnostic_overrides()
// lists them, bench.py and scan.py print the list in their JSON line.  Overrides that can change a RESULT (the unitarity
// tiers' thresholds: a verdict; GF_UNI_DUMP: fr[0]) are honoured only when GF_DIAGNOSTICS=1 is set as well, so that a stray
// variable in somebody's shell cannot silently change what a run computes; ignoring one is reported once on stderr.
std::mutex g_env_mu;
char g_env_seen[1024] = "";

bool finite_all(const double* p, int n)
{
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(p[i])) return false;
    return true;
}

}  // namespace

struct gf_model {
    GfCommon c;
    GfBsm hb;
    void* d_block = nullptr;     // the model's constant block (from the per-device pool): d_ptab | d_bsm
    GfBsm* d_bsm = nullptr;
    GfCommon* d_common = nullptr;  // device copy of `c` (same block)
    double* d_ptab = nullptr;    // [GF_MAX_DIM][4] = {lo, hi, loc, 1/sigma}: the kernels' LDS constant table
    hipStream_t stream = nullptr;   // created on first use (ensure_stream)
    std::mutex mu;
    int device = 0;
    int cus = 256;
    // staging for the host-buffer entry points (grown on demand, reused across calls)
    int64_t cap = 0;             // rows the device buffers hold
    double* d_theta = nullptr;
    double* d_out = nullptr;     // lnprob [cap] then fr [3 cap]
    int32_t* d_status = nullptr;
    int64_t hcap = 0;            // rows the pinned mirror holds (large batches stream through it in chunks: run_host)
    void* h_pin = nullptr;       // pinned mirror: theta [hcap][ndim] | lnprob [hcap] | fr [hcap][3] | status [hcap]
    size_t h_pin_bytes = 0;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};   // the chunk pipeline's slot events
    double* d_cube = nullptr;    // gf_lnprob_cube_batch: the unit-cube rows on the device
    size_t cube_cap = 0;
    // unitarity arbitration queue (BSM models, grown on demand when a status array is requested)
    std::mutex call_mu;          // serialises the entry points that use the model's staging buffers / queue
};

namespace {

// Per-device cache of what a model needs from the runtime: the device's identity, a non-blocking stream and
// one small block of device memory for its constant tables.  A grid scan creates and destroys hundreds of
// models (one per grid point); hipStreamCreate / hipMalloc / hipFree / hipGetDeviceProperties cost ~1 ms each
// and hipFree synchronises the device, so destroyed models hand their stream and block back to this pool.
constexpr int64_t GF_ZEROCOPY_MAX_ROWS = 2048;
constexpr int POOL_MAX_DEVICES = 64;
constexpr size_t POOL_MAX_ITEMS = 1024;
constexpr size_t WORK_CACHE_MAX_BYTES = (size_t)8 << 30;     // idle unitarity workspaces kept per device (of 288 GB)
constexpr size_t CONST_PTAB_BYTES = sizeof(double) * GF_MAX_DIM * 4;
constexpr size_t CONST_BSM_OFFSET = (CONST_PTAB_BYTES + 255) / 256 * 256;
constexpr size_t CONST_COMMON_OFFSET = (CONST_BSM_OFFSET + sizeof(GfBsm) + 255) / 256 * 256;
constexpr size_t CONST_BLOCK_BYTES = CONST_COMMON_OFFSET + sizeof(GfCommon);

// What the unitarity verdict of a batch needs besides the caller's arrays (gf_bsm.hip, gf_unitarity.hip): the arbitration
// queue, the walker queue and side buffer of the deferred tier 2, and one pinned word through which the arbitration kernel
// tells the host how long its queue was.  It belongs to the STREAM, not to the model: launches on one stream run in order,
// so every model that launches there can use the same workspace -- the 64 per-grid-point models of a texture scan, which
// propagate their chains one after the other on the sampler's stream, share one instead of allocating (and, worse,
// freeing: hipFree ~0.25 ms and a device synchronisation each) three buffers apiece -- and it stays with the stream when
// the stream goes back to the pool.
struct UniWork {
    std::mutex mu;                 // held from sizing the workspace to the last launch that uses it
    GfArbQueue* d_uq = nullptr;    // [uq_cap] walkers with their undecided bins
    GfUniQueue* d_wq = nullptr;    // [wq_cap] walkers
    double* d_t2sn = nullptr;      // [wq_cap][18]
    unsigned int* h_seen = nullptr;
    int64_t uq_cap = 0, wq_cap = 0;
    size_t bytes() const
    {
        return (d_uq ? sizeof(GfArbItem) * (size_t)uq_cap : 0) + (d_wq ? sizeof(unsigned long long) * (size_t)wq_cap : 0) +
               (d_t2sn ? sizeof(double) * 18 * (size_t)wq_cap : 0);
    }
    void release()
    {
        if (d_uq) (void)hipFree(d_uq);
        if (d_wq) (void)hipFree(d_wq);
        if (d_t2sn) (void)hipFree(d_t2sn);
        d_uq = nullptr; d_wq = nullptr; d_t2sn = nullptr; uq_cap = wq_cap = 0;
    }
};

struct DevicePool {
    int state = 0;                 // 0 unknown, 1 gfx950, -1 something else
    int cus = 256;
    std::vector<hipStream_t> streams;
    std::vector<hipStream_t> copy_streams;               // high-priority streams for the large read-backs (pool_copy_stream)
    std::vector<void*> blocks;
    std::unordered_map<hipStream_t, UniWork*> work;      // never erased while the stream lives
};
std::mutex g_pool_mu;
DevicePool g_pool[POOL_MAX_DEVICES];

// returns GF_OK and fills cus when `device` is a gfx950
int pool_device(int device, int* cus)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count || device >= POOL_MAX_DEVICES) {
        (void)hipGetLastError();
        std::snprintf(g_err, sizeof(g_err), "no HIP device %d (found %d)", device, count);
        return GF_ERR_NO_DEVICE;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    DevicePool& dp = g_pool[device];
    if (dp.state == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); dp.state = -1; }
        else {
            dp.state = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : -1;
            dp.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
    }
    if (dp.state != 1) {
        std::snprintf(g_err, sizeof(g_err), "device %d is not gfx950", device);
        return GF_ERR_NO_DEVICE;
    }
    *cus = dp.cus;
    return GF_OK;
}

hipError_t pool_stream(int device, hipStream_t* stream)
{
    *stream = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        DevicePool& dp = g_pool[device];
        if (!dp.streams.empty()) { *stream = dp.streams.back(); dp.streams.pop_back(); }
    }
    return *stream ? hipSuccess : hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
}

// A stream for a read-back that is to run BESIDE kernels of another stream.  The runtime multiplexes its streams onto a handful of
// hardware queues, round-robin, and a copy stream that shares the compute stream's queue has its barrier packets queued behind the
// kernels enqueued there.  Streams of another PRIORITY get hardware queues of their own, so the copy streams are created with the
// greatest priority and kept in a pool of their own.  (Built while hunting the read-backs that ran at half speed; the cause turned out
// to be the driver wiping freed memory -- gf_devcache.h -- and the priority made no measurable difference: kept, it is the safer
// arrangement.  GF_COPY_STREAM_PLAIN=1: a stream like any other.)
hipError_t pool_copy_stream(int device, hipStream_t* stream)
{
    *stream = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        DevicePool& dp = g_pool[device];
        if (!dp.copy_streams.empty()) { *stream = dp.copy_streams.back(); dp.copy_streams.pop_back(); }
    }
    if (*stream) return hipSuccess;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    static const bool plain = gf_internal_env("GF_COPY_STREAM_PLAIN", 0) != nullptr;            // A/B: a stream like any other
    return (plain || least == greatest) ? hipStreamCreateWithFlags(stream, hipStreamNonBlocking)
                                        : hipStreamCreateWithPriority(stream, hipStreamNonBlocking, greatest);
}

hipError_t pool_block(int device, void** block)
{
    *block = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        DevicePool& dp = g_pool[device];
        if (!dp.blocks.empty()) { *block = dp.blocks.back(); dp.blocks.pop_back(); }
    }
    return *block ? hipSuccess : hipMalloc(block, CONST_BLOCK_BYTES);
}

UniWork* work_for(int device, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    UniWork*& w = g_pool[device].work[stream];
    if (!w) w = new (std::nothrow) UniWork();
    return w;
}

// the stream must be idle (the caller synchronised it)
void pool_release(int device, hipStream_t stream, void* block)
{
    UniWork* drop = nullptr;
    bool trim = false;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        DevicePool& dp = g_pool[device];
        if (stream) {
            // the stream's workspace goes back to the pool with it, unless the idle workspaces of this device already hold
            // WORK_CACHE_MAX_BYTES: then its buffers are released (the small bookkeeping object stays)
            auto it = dp.work.find(stream);
            if (it != dp.work.end() && it->second) {
                size_t total = 0;
                for (auto& kv : dp.work) if (kv.second) total += kv.second->bytes();
                if (dp.streams.size() >= POOL_MAX_ITEMS) { drop = it->second; dp.work.erase(it); }
                else if (total > WORK_CACHE_MAX_BYTES) { drop = it->second; trim = true; }
            }
        }
        if (stream && dp.streams.size() < POOL_MAX_ITEMS) { dp.streams.push_back(stream); stream = nullptr; }
        if (block && dp.blocks.size() < POOL_MAX_ITEMS) { dp.blocks.push_back(block); block = nullptr; }
    }
    if (drop) {
        std::lock_guard<std::mutex> lk(drop->mu);
        drop->release();
        if (!trim) { if (drop->h_seen) (void)hipHostFree(drop->h_seen); }
    }
    if (drop && !trim) delete drop;
    if (stream) (void)hipStreamDestroy(stream);
    if (block) (void)hipFree(block);
}

// device buffers for `n` rows, pinned mirror for `n_pin` rows (n_pin < n: the batch streams through the mirror in chunks)
int ensure_staging(gf_model* m, int64_t n, int64_t n_pin)
{
    const size_t nd = (size_t)m->c.ndim;
    if (n > m->cap) {
        int64_t cap = m->cap ? m->cap : 1024;
        while (cap < n) cap *= 2;
        if (m->d_theta) (void)hipFree(m->d_theta);
        if (m->d_out) (void)hipFree(m->d_out);
        if (m->d_status) (void)hipFree(m->d_status);
        m->d_theta = nullptr; m->d_out = nullptr; m->d_status = nullptr; m->cap = 0;
        GF_HIP(hipMalloc((void**)&m->d_theta, sizeof(double) * nd * cap));
        GF_HIP(hipMalloc((void**)&m->d_out, sizeof(double) * 4 * cap));
        GF_HIP(hipMalloc((void**)&m->d_status, sizeof(int32_t) * cap));
        m->cap = cap;
    }
    if (n_pin > m->hcap) {
        int64_t hcap = m->hcap ? m->hcap : 1024;
        while (hcap < n_pin) hcap *= 2;
        if (m->h_pin) (void)hipHostFree(m->h_pin);
        m->h_pin = nullptr; m->hcap = 0;
        m->h_pin_bytes = sizeof(double) * (nd + 4) * hcap + sizeof(int32_t) * hcap;
        GF_HIP(hipHostMalloc(&m->h_pin, m->h_pin_bytes, hipHostMallocDefault));
        m->hcap = hcap;
    }
    return GF_OK;
}

int check_dev_ptr(const void* p, size_t align)
{
    if (!p) return GF_ERR_INVALID_ARG;
    if (((uintptr_t)p) % align) {
        std::snprintf(g_err, sizeof(g_err), "device pointer %p is not %zu-byte aligned", p, align);
        return GF_ERR_INVALID_ARG;
    }
    return GF_OK;
}

// The arbitration queue must hold every walker of one piece of the batch (gf_launch_bsm cuts AoS batches into pieces of
// uq_cap walkers; SoA batches go in one piece).  Items are walkers since round 3 (16 B each: index + mask of undecided bins),
// not (walker, bin) pairs: a queue for the 8.4 M walkers of a piece is 128 MiB where round 2's was 1 GiB for 6.7 M.
constexpr int64_t UQ_MAX_ITEMS = 1 << 23;
// from this batch size on (one lane per walker in the evaluation kernel) tier 2 runs as its own compact kernel
constexpr int64_t GF_TIER2_SPLIT_MIN = 65536;
constexpr int64_t WQ_MAX_WALKERS = 1 << 23;    // per piece: 8.4 M walkers, 1.2 GB of side buffer

// `items_limit` (out): how many items of the queue a piece of this batch may use (its capacity, or GF_UQ_MAX_ITEMS if smaller)
int ensure_uq(UniWork* w, int nbins, hipStream_t st, int layout, int64_t n, int64_t* items_limit)
{
    (void)nbins;
    const int64_t nb = 1;                     // one item per walker
    int64_t need = n;
    int64_t max_items = UQ_MAX_ITEMS;
    if (const char* e = gf_internal_env("GF_UQ_MAX_ITEMS", 0)) {     // tests: a small queue, so that a modest batch is cut into pieces
        const long long v = std::atoll(e);
        if (v >= 4096 && v < UQ_MAX_ITEMS) max_items = v;
    }
    if (layout == GF_LAYOUT_AOS && need > max_items) need = max_items > nb ? max_items : nb;
    *items_limit = layout == GF_LAYOUT_AOS ? (max_items > nb ? max_items : nb) : (int64_t)0x7fffffffffffLL;
    if (need > 0xffffffffLL) {
        std::snprintf(g_err, sizeof(g_err), "a structure-of-arrays batch of %lld walkers with a status array exceeds the arbitration queue", (long long)n);
        return GF_ERR_UNSUPPORTED;
    }
    if (!w->h_seen) {
        GF_HIP(hipHostMalloc((void**)&w->h_seen, 64, hipHostMallocDefault));
        w->h_seen[0] = 0xffffffffu;              // nothing seen yet: the first launch takes the full grid
        w->h_seen[1] = 0;                        // host-only flag: full grids on request (gf_internal_full_arbitration_grids)
        w->h_seen[2] = 0;                        // written by k_uni_resolve: a queue overflowed (check_queue_overflow)
        w->h_seen[3] = w->h_seen[4] = 0;         // running totals: pairs arbitrated, arbitration launches (gf_internal_uni_stats)
    }
    int64_t cap = w->uq_cap ? w->uq_cap : 4096;
    while (cap < need) cap *= 2;
    // the walker queue (and the side buffer, 144 B per walker) of the deferred tier 2: one piece of the batch -- gf_launch_bsm
    // cuts an AoS batch into pieces that fit both queues
    const bool defer = n >= GF_TIER2_SPLIT_MIN;
    int64_t need_w = defer ? n : 0;
    if (layout == GF_LAYOUT_AOS && need_w > WQ_MAX_WALKERS) need_w = WQ_MAX_WALKERS;
    if (need_w > w->wq_cap) {                 // grow at least geometrically
        const int64_t twice = 2 * w->wq_cap < WQ_MAX_WALKERS ? 2 * w->wq_cap : WQ_MAX_WALKERS;
        if (layout == GF_LAYOUT_AOS && twice > need_w) need_w = twice;
    }
    if (cap == w->uq_cap && need_w <= w->wq_cap) return GF_OK;
    GF_HIP(hipStreamSynchronize(st));          // earlier launches may still use the old buffers
    GfUniQueue hdr = {0, 0, 0, 0, {0}};
    if (cap != w->uq_cap) {
        if (w->d_uq) (void)hipFree(w->d_uq);
        w->d_uq = nullptr; w->uq_cap = 0;
        GF_HIP(hipMalloc((void**)&w->d_uq, sizeof(GfArbQueue) + sizeof(GfArbItem) * (size_t)cap));
        GfArbQueue ah;
        std::memset(&ah, 0, sizeof(ah));
        ah.cap = (unsigned int)cap;
        GF_HIP(hipMemcpyAsync(w->d_uq, &ah, offsetof(GfArbQueue, items), hipMemcpyHostToDevice, st));
        GF_HIP(hipStreamSynchronize(st));
        w->uq_cap = cap;
    }
    if (need_w > w->wq_cap) {
        if (w->d_wq) (void)hipFree(w->d_wq);
        if (w->d_t2sn) (void)hipFree(w->d_t2sn);
        w->d_wq = nullptr; w->d_t2sn = nullptr; w->wq_cap = 0;
        GF_HIP(hipMalloc((void**)&w->d_wq, sizeof(GfUniQueue) + sizeof(unsigned long long) * (size_t)need_w));
        GF_HIP(hipMalloc((void**)&w->d_t2sn, sizeof(double) * 18 * (size_t)need_w));
        hdr.cap = (unsigned int)need_w;
        GF_HIP(hipMemcpyAsync(w->d_wq, &hdr, offsetof(GfUniQueue, items), hipMemcpyHostToDevice, st));
        GF_HIP(hipStreamSynchronize(st));
        w->wq_cap = need_w;
    }
    return GF_OK;
}

// Did an arbitration launch on this stream report a full queue?  Call after a stream synchronise (or ahead of new launches:
// stale reports of earlier asynchronous launches).  The report is consumed.
int check_queue_overflow(int device, hipStream_t st)
{
    UniWork* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool[device].work.find(st);
        if (it != g_pool[device].work.end()) w = it->second;
    }
    if (!w || !w->h_seen) return GF_OK;
    if (__atomic_exchange_n(&w->h_seen[2], 0u, __ATOMIC_RELAXED) == 0u) return GF_OK;
    std::snprintf(g_err, sizeof(g_err), "a unitarity queue overflowed: (walker, bin) pairs were dropped and the status array of that batch is "
                                        "incomplete (the host cuts batches to fit the queues: this is a library bug)");
    return GF_ERR_QUEUE_OVERFLOW;
}

// One BSM launch with or without the verdict's workspace (the stream's, locked for the duration of the launches)
int launch_bsm_on(gf_model* m, hipStream_t st, const double* d_theta, int layout, int64_t n, int with_llh, double* d_lnprob, double* d_fr,
                  int32_t* d_status, const char* what)
{
    hipError_t e;
    if (!d_status) {
        e = gf_launch_bsm(m->c, m->d_common, m->d_bsm, m->hb.nbins, m->d_ptab, d_theta, layout, n, with_llh, d_lnprob, d_fr, nullptr,
                          nullptr, 0, nullptr, 0, nullptr, nullptr, m->cus, st);
    } else {
        UniWork* w = work_for(m->device, st);
        if (!w) return GF_ERR_ALLOC;
        const int ro = check_queue_overflow(m->device, st);        // of an earlier asynchronous launch on this stream
        if (ro != GF_OK) return ro;
        std::lock_guard<std::mutex> lk(w->mu);
        int64_t limit = 0;
        const int rq = ensure_uq(w, m->hb.nbins, st, layout, n, &limit);
        if (rq != GF_OK) return rq;
        e = gf_launch_bsm(m->c, m->d_common, m->d_bsm, m->hb.nbins, m->d_ptab, d_theta, layout, n, with_llh, d_lnprob, d_fr, d_status,
                          w->d_uq, w->uq_cap < limit ? w->uq_cap : limit, n >= GF_TIER2_SPLIT_MIN ? w->d_wq : nullptr, w->wq_cap, w->d_t2sn, w->h_seen, m->cus, st);
    }
    if (e != hipSuccess) return hip_fail(e, what);
    return GF_OK;
}

int launch_lnprob(gf_model* m, hipStream_t st, const double* d_theta, int layout, int64_t n, double* d_lnprob, double* d_fr,
                  int32_t* d_status)
{
    if (n == 0) return GF_OK;
    if (m->c.mode == GF_MODE_BSM_GAUSS) return launch_bsm_on(m, st, d_theta, layout, n, 1, d_lnprob, d_fr, d_status, "lnprob launch");
    const hipError_t e = gf_launch_lnprob_sm(m->c, m->d_ptab, d_theta, layout, n, d_lnprob, d_fr, d_status, m->cus, st);
    if (e != hipSuccess) return hip_fail(e, "lnprob launch");
    return GF_OK;
}

int launch_propagate(gf_model* m, hipStream_t st, const double* d_theta, int layout, int64_t n, double* d_fr, int32_t* d_status)
{
    if (n == 0) return GF_OK;
    if (m->c.mode == GF_MODE_BSM_GAUSS) return launch_bsm_on(m, st, d_theta, layout, n, 0, nullptr, d_fr, d_status, "propagate launch");
    const hipError_t e = gf_launch_propagate_sm(m->c, d_theta, layout, n, d_fr, d_status, m->cus, st);
    if (e != hipSuccess) return hip_fail(e, "propagate launch");
    return GF_OK;
}

// A model's stream is created (or taken from the pool) the first time one of its entry points needs it:
// creating a HIP stream costs milliseconds (tools/rtcost.hip: 3.8 ms), and the models of a stacked grid
// sampler only lend their constants -- their launches go to the sampler's stream.
int ensure_stream(gf_model* m)
{
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->stream) return GF_OK;
    hipError_t e = pool_stream(m->device, &m->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamCreate");
    return GF_OK;
}
#define GF_STREAM(m)                      \
    do {                                  \
        int rs_ = ensure_stream(m);       \
        if (rs_ != GF_OK) return rs_;     \
    } while (0)

}  // namespace

extern "C" {

int gf_abi_version(void) { return GF_ABI_VERSION; }

const char* gf_strerror(int err)
{
    switch (err) {
    case GF_OK: return "ok";
    case GF_ERR_INVALID_ARG: return "invalid argument";
    case GF_ERR_NO_DEVICE: return "no gfx950 HIP device available (this library has no CPU fallback)";
    case GF_ERR_HIP: return "HIP runtime error";
    case GF_ERR_ALLOC: return "allocation failed";
    case GF_ERR_COMM: return "RCCL error";
    case GF_ERR_UNSUPPORTED: return "unsupported configuration";
    case GF_ERR_QUEUE_OVERFLOW: return "unitarity queue overflow";
    default: return "unknown error";
    }
}

const char* gf_last_hip_error(void) { return g_err; }

// internal (gf_sampler.hip, gf_comm.hip): one thread-local error text for the whole library
void gf_internal_set_error(const char* msg) { std::snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

size_t gf_sizeof_model_desc(void) { return sizeof(gf_model_desc); }

// internal: getenv with a record.  `affects_results` != 0: honoured only under GF_DIAGNOSTICS=1.
const char* gf_internal_env(const char* name, int affects_results)
{
    const char* v = std::getenv(name);
    if (!v) return nullptr;
    std::lock_guard<std::mutex> lk(g_env_mu);
    char item[160];
    if (affects_results) {
        const char* d = std::getenv("GF_DIAGNOSTICS");
        if (!d || d[0] != '1') {
            std::snprintf(item, sizeof(item), "%s(ignored)", name);
            if (!std::strstr(g_env_seen, item)) {
                std::fprintf(stderr, "libgolemhip: %s is set but GF_DIAGNOSTICS=1 is not: ignored (it would change results)\n", name);
                if (std::strlen(g_env_seen) + std::strlen(item) + 2 < sizeof(g_env_seen)) { if (g_env_seen[0]) std::strcat(g_env_seen, " "); std::strcat(g_env_seen, item); }
            }
            return nullptr;
        }
    }
    std::snprintf(item, sizeof(item), "%s=%.100s", name, v);
    if (!std::strstr(g_env_seen, item) && std::strlen(g_env_seen) + std::strlen(item) + 2 < sizeof(g_env_seen)) {
        if (g_env_seen[0]) std::strcat(g_env_seen, " ");
        std::strcat(g_env_seen, item);
    }
    return v;
}

int gf_diagnostic_overrides(char* buf, size_t buflen)
{
    if (!buf || buflen == 0) return GF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_env_mu);
    std::snprintf(buf, buflen, "%s", g_env_seen);
    return GF_OK;
}

int gf_device_count(int* count)
{
    if (!count) return GF_ERR_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return GF_OK; }
    *count = n;
    return GF_OK;
}

int gf_device_name(int device, char* buf, size_t buflen)
{
    if (!buf || buflen == 0) return GF_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    GF_HIP(hipGetDeviceProperties(&prop, device));
    std::snprintf(buf, buflen, "%s", prop.gcnArchName);
    return GF_OK;
}

int gf_model_create(const gf_model_desc* d, int device, gf_model** out)
{
    if (!d || !out) return GF_ERR_INVALID_ARG;
    *out = nullptr;
    g_err[0] = 0;
    if (d->abi_version != GF_ABI_VERSION) {
        std::snprintf(g_err, sizeof(g_err), "descriptor abi_version %d != library %d", d->abi_version, GF_ABI_VERSION);
        return GF_ERR_INVALID_ARG;
    }
    if (d->ndim < 1 || d->ndim > GF_MAX_DIM) return GF_ERR_INVALID_ARG;
    if (d->mode < GF_MODE_PRIOR_ONLY || d->mode > GF_MODE_BSM_GAUSS) return GF_ERR_INVALID_ARG;
    auto idx_ok = [&](int i) { return i >= -1 && i < d->ndim; };
    for (int k = 0; k < 4; ++k)
        if (!idx_ok(d->idx_sm[k]) || !idx_ok(d->idx_mm[k])) return GF_ERR_INVALID_ARG;
    for (int k = 0; k < 2; ++k)
        if (!idx_ok(d->idx_mass[k]) || !idx_ok(d->idx_src[k])) return GF_ERR_INVALID_ARG;
    if (!idx_ok(d->idx_scale) || !idx_ok(d->idx_gamma)) return GF_ERR_INVALID_ARG;
    if ((d->idx_src[0] < 0) != (d->idx_src[1] < 0)) return GF_ERR_INVALID_ARG;
    if (!idx_ok(d->idx_src_x) || (d->idx_src_x >= 0 && d->idx_src[0] >= 0)) return GF_ERR_INVALID_ARG;
    if (d->idx_src_x >= 0 && d->mode == GF_MODE_BSM_GAUSS) {
        std::snprintf(g_err, sizeof(g_err), "an astroX source column is not defined for the flux-averaged (BSM) posterior");
        return GF_ERR_UNSUPPORTED;
    }
    // CP phases (dcp, and the NP matrix's for texture NONE): the kernels' sine / cosine reduce |x| < GF_PHASE_MAX
    // only (every paramset of the reference boxes them into [0, 2 pi]: scripts/fr.py:41, mc_unitary.py:39)
    auto phase_ok = [&](int idx, double fixed) {
        if (idx >= 0) return std::fabs(d->lo[idx]) <= GF_PHASE_MAX && std::fabs(d->hi[idx]) <= GF_PHASE_MAX;
        return std::fabs(fixed) <= GF_PHASE_MAX;
    };
    if (!phase_ok(d->idx_sm[3], d->sm_fixed[3]) ||
        (d->mode == GF_MODE_BSM_GAUSS && d->texture == GF_TEX_NONE && !phase_ok(d->idx_mm[3], d->mm_fixed[3]))) {
        std::snprintf(g_err, sizeof(g_err), "CP phase range or value beyond +-%g is not supported", GF_PHASE_MAX);
        return GF_ERR_UNSUPPORTED;
    }

    gf_model* m = new (std::nothrow) gf_model();
    if (!m) return GF_ERR_ALLOC;
    GfCommon& c = m->c;
    std::memset(&c, 0, sizeof(c));
    std::memset(&m->hb, 0, sizeof(m->hb));
    c.ndim = d->ndim;
    c.mode = d->mode;
    for (int k = 0; k < 4; ++k) { c.idx_sm[k] = d->idx_sm[k]; c.idx_mm[k] = d->idx_mm[k]; c.sm_fixed[k] = d->sm_fixed[k]; c.mm_fixed[k] = d->mm_fixed[k]; }
    for (int k = 0; k < 2; ++k) { c.idx_mass[k] = d->idx_mass[k]; c.idx_src[k] = d->idx_src[k]; c.mass_fixed[k] = d->mass_fixed[k]; }
    c.idx_scale = d->idx_scale;
    c.idx_gamma = d->idx_gamma;
    c.idx_src_x = d->idx_src_x;
    c.scale_fixed = d->scale_fixed;

    // priors: llh.py:81-90 + scipy truncnorm.logpdf = ((-z^2/2 - log sqrt(2pi)) - log_mass) - log(sigma)
    const double logC = std::log(std::sqrt(2.0 * M_PI));
    double pc = 0.0;
    for (int i = 0; i < d->ndim; ++i) {
        c.lo[i] = d->lo[i];
        c.hi[i] = d->hi[i];
        const int kind = d->prior_kind[i];
        if (kind == GF_PRIOR_UNIFORM) {
            c.loc[i] = 0.0;
            c.inv_sigma[i] = 0.0;
        } else if (kind == GF_PRIOR_GAUSSIAN || kind == GF_PRIOR_LIMITEDGAUSS) {
            if (!(d->sigma[i] > 0.0) || !std::isfinite(d->loc[i]) || !std::isfinite(d->log_mass[i])) {
                std::snprintf(g_err, sizeof(g_err), "column %d: Gaussian prior needs finite loc/log_mass and sigma > 0", i);
                delete m;
                return GF_ERR_INVALID_ARG;
            }
            c.loc[i] = d->loc[i];
            c.inv_sigma[i] = 1.0 / d->sigma[i];
            pc += ((-logC) - d->log_mass[i]) - std::log(d->sigma[i]);
        } else {
            delete m;
            return GF_ERR_INVALID_ARG;
        }
    }
    c.prior_const = pc;

    for (int k = 0; k < 3; ++k) { c.src_fixed[k] = d->source_ratio[k]; c.bf[k] = d->bestfit_fr[k]; }
    c.src_fixed_sum = (d->source_ratio[0] + d->source_ratio[1]) + d->source_ratio[2];
    // multi_gaussian, llh.py:53-54 + scipy _multivariate.py:514-539: cov = smearing^2 I
    if (d->mode != GF_MODE_PRIOR_ONLY) {
        if (!(d->smearing > 0.0) || !finite_all(d->bestfit_fr, 3)) { delete m; return GF_ERR_INVALID_ARG; }
        const double s = std::pow(d->smearing, 2);
        c.inv_smear = std::sqrt(1.0 / s);
        c.gauss_c0 = 3.0 * std::log(2.0 * M_PI) + ((std::log(s) + std::log(s)) + std::log(s));
        c.gauss_mh = -0.5 * (c.inv_smear * c.inv_smear);
        c.gauss_k = -0.5 * c.gauss_c0;
    }
    c.offset = d->offset;
    c.flat_llh = d->flat_llh;
    {
        static const double cosc[8] = {2.7117413873509064e-15, -7.641995277350052e-13, 1.605889634387573e-10,
                                       -2.505210587009456e-08, 2.75573191979119e-06, -0.00019841269841110079,
                                       0.008333333333332799, -0.16666666666666657};
        for (int k = 0; k < 8; ++k) c.cosc[k] = cosc[k];
    }

    GfBsm& b = m->hb;
    if (d->mode == GF_MODE_BSM_GAUSS) {
        if (d->nbins < 1 || d->nbins > GF_MAX_BINS || d->texture < GF_TEX_OEU || d->texture > GF_TEX_NONE) {
            delete m;
            return GF_ERR_INVALID_ARG;
        }
        if (d->texture == GF_TEX_NONE && (d->idx_mm[0] < 0 && !finite_all(d->mm_fixed, 4))) { delete m; return GF_ERR_INVALID_ARG; }
        b.texture = d->texture;
        b.dimension = d->dimension;
        b.nbins = d->nbins;
        for (int k = 0; k < d->nbins; ++k) {
            const double e = std::sqrt(d->bin_edges[k] * d->bin_edges[k + 1]);     // fr.py:413
            if (!(e > 0.0) || !std::isfinite(e)) { delete m; return GF_ERR_INVALID_ARG; }
            b.centre[k] = e;
            b.weight[k] = std::fabs(d->bin_edges[k + 1] - d->bin_edges[k]);        // fr.py:414
            b.inv2e[k] = 1.0 / (2 * e);                                            // fr.py:386
            b.epow[k] = std::pow(e, (double)(d->dimension - 3));                   // fr.py:394
            const double rho = b.epow[k] / b.inv2e[k];
            if (k == 0 || rho > b.rho_max) b.rho_max = rho;
            b.rho[k] = rho;
            b.wsum += b.weight[k];
        }
        if (d->texture != GF_TEX_NONE) {
            const double z = 0. + 1e-9;                                            // fr.py:370
            double ang[4];
            switch (d->texture) {
            case GF_TEX_OEU: ang[0] = 0.5; ang[1] = 1.0; ang[2] = z; ang[3] = z; break;
            case GF_TEX_OET: ang[0] = z; ang[1] = 0.25; ang[2] = z; ang[3] = z; break;
            default: ang[0] = z; ang[1] = 1.0; ang[2] = 0.5; ang[3] = z; break;
            }
            cld u[3][3];
            mixing_matrix_ld(ang, u);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    const cld t1 = u[i][1] * std::conj(u[j][1]);
                    const cld t2 = u[i][2] * std::conj(u[j][2]);
                    b.t1_re[3 * i + j] = (double)t1.real(); b.t1_im[3 * i + j] = (double)t1.imag();
                    b.t2_re[3 * i + j] = (double)t2.real(); b.t2_im[3 * i + j] = (double)t2.imag();
                }
        }
        // Unitarity tiers (gf_bsm_device.hpp).  Tier 1: SM weight a >= 2e-11 -> unitary (80-bit residual <= 1.3e-19 / a
        // over 30 000 pairs, tools/uni_weight_bound.py: five-fold margin; no walker with a > 1.1e-13 fails).  Tier 2, the
        // fp64 estimate, measured against the 80-bit residual on 180 000 walkers of all (dimension, texture) pairs binned by
        // a (tools/uni_estimate_spread.py, profiles/r02/uni_estimate_spread.txt): log10(estimate / residual) lies in
        // [-2.2, +2.1] wherever fp64 resolves the SM term (a >= 1e-16) and in [-4.0, +4.5] below -- there the estimate
        // acquits only with that margin and never condemns.  (GF_UNI_BAND_DECADES: symmetric override of the resolved
        // regime's band, diagnostics; 0 = estimate only.)
        {
            double lo_dec = 2.7, hi_dec = 2.6, lo_nl_dec = 4.6;
            b.uni_a_ok = 2e-11;
            b.uni_a_lin = 1e-16;
            if (const char* e = gf_internal_env("GF_UNI_BAND_DECADES", 1)) {
                const double v = std::atof(e);
                if (v >= 0.0 && v <= 12.0) { lo_dec = hi_dec = lo_nl_dec = v; if (v == 0.0) b.uni_a_lin = 0.0; }
            }
            b.uni_lo = 1e-7 * 2048.0 * std::pow(10.0, -lo_dec);
            b.uni_hi = 1e-7 * 2048.0 * std::pow(10.0, hi_dec);
            b.uni_lo_nl = 1e-7 * 2048.0 * std::pow(10.0, -lo_nl_dec);
            if (gf_internal_env("GF_UNI_NO_WEIGHT_GATE", 1)) b.uni_a_ok = 2.0;              // diagnostics: tier 1 off
            if (const char* e = gf_internal_env("GF_UNI_A_OK", 1)) b.uni_a_ok = std::atof(e);  // diagnostics: tier 1's threshold
            b.uni_own_bins_only = gf_internal_env("GF_UNI_OWN_BINS_ONLY", 1) ? 1 : 0;             // diagnostics: A/B of uni_arbitration_mask
            if (gf_internal_env("GF_UNI_DUMP", 1)) { b.uni_lo = b.uni_lo_nl = -1.0; b.uni_hi = 1e300; }   // diagnostics: fr[0] <- the estimate
        }
        // per-model matrices of the unitarity arbitration, in the reference's own operation order
        {
            const double z = 0. + 1e-9;                                            // fr.py:370
            double np_ang[4];
            switch (d->texture) {
            case GF_TEX_OEU: np_ang[0] = 0.5; np_ang[1] = 1.0; np_ang[2] = z; np_ang[3] = z; break;
            case GF_TEX_OET: np_ang[0] = z; np_ang[1] = 0.25; np_ang[2] = z; np_ang[3] = z; break;
            case GF_TEX_OUT: np_ang[0] = z; np_ang[1] = 1.0; np_ang[2] = 0.5; np_ang[3] = z; break;
            default: for (int k = 0; k < 4; ++k) np_ang[k] = d->mm_fixed[k]; break;   // used only when idx_mm < 0
            }
            cld u[3][3];
            angles_to_u_ref_ld(np_ang, u);
            split_matrix_ld(u, b.npu_hi, b.npu_lo);
            angles_to_u_ref_ld(d->sm_fixed, u);                                     // fr.py:313, 435
            split_matrix_ld(u, b.smu_hi, b.smu_lo);
        }
    }

    int cus = 256;
    const int drc = pool_device(device, &cus);
    if (drc != GF_OK) { delete m; return drc; }
    m->device = device;
    m->cus = cus;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = pool_block(device, &m->d_block);
    if (e == hipSuccess) {
        // one upload: prior table, then (BSM) the bin / texture tables, into the model's constant block
        alignas(16) unsigned char img[CONST_BLOCK_BYTES];
        double* tab = reinterpret_cast<double*>(img);
        for (int i = 0; i < GF_MAX_DIM; ++i) {
            tab[4 * i] = c.lo[i]; tab[4 * i + 1] = c.hi[i]; tab[4 * i + 2] = c.loc[i]; tab[4 * i + 3] = c.inv_sigma[i];
        }
        const bool bsm = d->mode == GF_MODE_BSM_GAUSS;
        if (bsm) std::memcpy(img + CONST_BSM_OFFSET, &m->hb, sizeof(GfBsm));
        std::memcpy(img + CONST_COMMON_OFFSET, &c, sizeof(GfCommon));          // kernels that take the constants by pointer
        // stream-ordered, never the null stream: a synchronous hipMemcpy issued while another host thread is
        // capturing a sampler graph fails on this runtime and poisons that capture.  The stream comes from the
        // pool and goes straight back (the model gets its own only when an entry point needs one).
        hipStream_t up = nullptr;
        e = pool_stream(device, &up);
        if (e == hipSuccess) e = hipMemcpyAsync(m->d_block, img, CONST_BLOCK_BYTES, hipMemcpyHostToDevice, up);
        if (e == hipSuccess) e = hipStreamSynchronize(up);
        if (up) pool_release(device, up, nullptr);
        m->d_ptab = reinterpret_cast<double*>(m->d_block);
        m->d_bsm = bsm ? reinterpret_cast<GfBsm*>(static_cast<unsigned char*>(m->d_block) + CONST_BSM_OFFSET) : nullptr;
        m->d_common = reinterpret_cast<GfCommon*>(static_cast<unsigned char*>(m->d_block) + CONST_COMMON_OFFSET);
    }
    if (e != hipSuccess) {
        int rc = hip_fail(e, "gf_model_create");
        gf_model_destroy(m);
        return rc;
    }
    *out = m;
    return GF_OK;
}

void gf_model_destroy(gf_model* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    pool_release(m->device, m->stream, m->d_block);
    if (m->d_theta) (void)hipFree(m->d_theta);
    if (m->d_out) (void)hipFree(m->d_out);
    if (m->d_status) (void)hipFree(m->d_status);
    if (m->h_pin) (void)hipHostFree(m->h_pin);
    for (int k = 0; k < 2; ++k) {
        if (m->ev_up[k]) (void)hipEventDestroy(m->ev_up[k]);
        if (m->ev_down[k]) (void)hipEventDestroy(m->ev_down[k]);
    }
    if (m->d_cube) (void)hipFree(m->d_cube);
    delete m;
}

int gf_model_ndim(const gf_model* m) { return m ? m->c.ndim : -1; }

// internal (not in the public header): gf_sampler.hip reaches the model's constants and stream through these
int gf_model_internal(gf_model* m, const GfCommon** c, const GfBsm** d_bsm, const double** d_ptab, void** stream, int* device)
{
    if (!m) return GF_ERR_INVALID_ARG;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    *c = &m->c; *d_bsm = m->d_bsm; *d_ptab = m->d_ptab; *stream = (void*)m->stream; *device = m->device;
    return GF_OK;
}

// constants only: does not give the model a stream
int gf_model_constants(gf_model* m, const GfCommon** c, const GfBsm** d_bsm, const double** d_ptab, int* device, int* cus,
                       int* nbins)
{
    if (!m) return GF_ERR_INVALID_ARG;
    *c = &m->c; *d_bsm = m->d_bsm; *d_ptab = m->d_ptab; *device = m->device; *cus = m->cus;
    *nbins = m->c.mode == GF_MODE_BSM_GAUSS ? m->hb.nbins : 0;
    return GF_OK;
}

// the model's kernels on a stream of the caller's (the sampler's)
int gf_model_lnprob_on(gf_model* m, void* stream, const double* d_theta, int layout, int64_t n, double* d_lnprob,
                       double* d_fr, int32_t* d_status)
{
    if (!m || n < 0) return GF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->call_mu);
    return launch_lnprob(m, (hipStream_t)stream, d_theta, layout, n, d_lnprob, d_fr, d_status);
}

// internal: a second stream from the device's pool (gf_sampler.hip: copies that overlap the sampler stream's kernels)
int gf_internal_borrow_stream(int device, void** stream)
{
    if (device < 0 || device >= POOL_MAX_DEVICES || !stream) return GF_ERR_INVALID_ARG;
    hipStream_t st = nullptr;
    const hipError_t e = pool_stream(device, &st);
    if (e != hipSuccess) return hip_fail(e, "pool_stream");
    *stream = (void*)st;
    return GF_OK;
}
void gf_internal_return_stream(int device, void* stream)          // idle (synchronised) streams only
{
    if (device >= 0 && device < POOL_MAX_DEVICES && stream) pool_release(device, (hipStream_t)stream, nullptr);
}
// internal: a stream for a large read-back that overlaps another stream's kernels (pool_copy_stream: hardware queues of its own)
int gf_internal_borrow_copy_stream(int device, void** stream)
{
    if (device < 0 || device >= POOL_MAX_DEVICES || !stream) return GF_ERR_INVALID_ARG;
    hipStream_t st = nullptr;
    const hipError_t e = pool_copy_stream(device, &st);
    if (e != hipSuccess) return hip_fail(e, "pool_copy_stream");
    *stream = (void*)st;
    return GF_OK;
}
void gf_internal_return_copy_stream(int device, void* stream)     // idle (synchronised) streams only
{
    if (device < 0 || device >= POOL_MAX_DEVICES || !stream) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool[device].copy_streams.push_back((hipStream_t)stream);
}

// internal: while `on`, every arbitration launch on `stream` takes the full grid whatever the previous one found
// (gf_launch_uni_resolve); a workspace is created if the stream has none yet
void gf_internal_full_arbitration_grids(int device, void* stream, int on)
{
    if (device < 0 || device >= POOL_MAX_DEVICES) return;
    UniWork* w = work_for(device, (hipStream_t)stream);
    if (!w) return;
    std::lock_guard<std::mutex> lk(w->mu);
    if (!w->h_seen) {
        if (hipHostMalloc((void**)&w->h_seen, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); w->h_seen = nullptr; return; }
        w->h_seen[0] = 0xffffffffu;
        w->h_seen[2] = w->h_seen[3] = w->h_seen[4] = 0;
    }
    w->h_seen[1] = on ? 1u : 0u;
}

int gf_model_propagate_on(gf_model* m, void* stream, const double* d_theta, int layout, int64_t n, double* d_fr,
                          int32_t* d_status)
{
    if (!m || n < 0) return GF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->call_mu);
    return launch_propagate(m, (hipStream_t)stream, d_theta, layout, n, d_fr, d_status);
}

// ---- host-buffer entry points ------------------------------------------------------------
// Rows per chunk of the large-batch pipeline, and the batch size from which it is used.  A large batch streams through two
// pinned slots: the host copies chunk c + 1 into its slot while chunk c crosses PCIe (and the other way round for the
// results).  Copying the whole batch into a pinned mirror first and transferring it then -- the path below this size, where
// it is one memcpy and one transfer -- runs at 1 / (1/33 + 1/57) = 21 GB/s on the MI355X box, the pipeline at the slower of
// the two (tools/h2d_probe.hip); the mirror of a 4 M-row batch also took 0.1 s to allocate (hipHostMalloc: 4.7 GB/s).
constexpr int64_t PIPE_CHUNK_ROWS = 65536;
constexpr int64_t PIPE_MIN_ROWS = 4 * PIPE_CHUNK_ROWS;

namespace { void parallel_memcpy(char* dst, const char* src, size_t len); }      // the host copy pool, below

static int run_host_pipelined(gf_model* m, const double* theta, int64_t n, double* lnprob, double* fr, int32_t* status, bool with_llh)
{
    // From 2 M rows on the chunks are four times as long and the copies into and out of the slots are shared out over the
    // host copy pool (copy_rows: 1 MB pieces): one thread copies at 33 GB/s, PCIe takes 57 (round 3).
    const int64_t CH = n >= 32 * PIPE_CHUNK_ROWS ? 4 * PIPE_CHUNK_ROWS : PIPE_CHUNK_ROWS;
    int rc = ensure_staging(m, n, 2 * CH);
    if (rc != GF_OK) return rc;
    for (int k = 0; k < 2; ++k) {
        if (!m->ev_up[k]) GF_HIP(hipEventCreateWithFlags(&m->ev_up[k], hipEventDisableTiming));
        if (!m->ev_down[k]) GF_HIP(hipEventCreateWithFlags(&m->ev_down[k], hipEventDisableTiming));
    }
    const size_t nd = (size_t)m->c.ndim;
    double* h_theta = (double*)m->h_pin;
    double* h_out = h_theta + nd * m->hcap;
    double* h_fr = h_out + m->hcap;
    int32_t* h_st = (int32_t*)(h_fr + 3 * m->hcap);
    const int64_t nchunks = (n + CH - 1) / CH;
    for (int64_t c = 0; c < nchunks; ++c) {
        const int slot = (int)(c & 1);
        const int64_t off = c * CH, len = n - off < CH ? n - off : CH;
        if (c >= 2) GF_HIP(hipEventSynchronize(m->ev_up[slot]));        // the slot's previous chunk has left for the device
        parallel_memcpy(reinterpret_cast<char*>(h_theta + nd * CH * slot), reinterpret_cast<const char*>(theta + nd * off), sizeof(double) * nd * len);
        GF_HIP(hipMemcpyAsync(m->d_theta + nd * off, h_theta + nd * CH * slot, sizeof(double) * nd * len, hipMemcpyHostToDevice, m->stream));
        GF_HIP(hipEventRecord(m->ev_up[slot], m->stream));
    }
    double* d_ln = m->d_out;
    double* d_fr = m->d_out + m->cap;
    if (with_llh)
        rc = launch_lnprob(m, m->stream, m->d_theta, GF_LAYOUT_AOS, n, d_ln, fr ? d_fr : nullptr, status ? m->d_status : nullptr);
    else
        rc = launch_propagate(m, m->stream, m->d_theta, GF_LAYOUT_AOS, n, d_fr, status ? m->d_status : nullptr);
    if (rc != GF_OK) return rc;
    // results: chunk c comes down into its slot while the host copies chunk c - 1 out of the other (the transfers follow the
    // kernel, and with it every upload that read these slots, in stream order)
    auto copy_out = [&](int64_t c) {
        const int slot = (int)(c & 1);
        const int64_t off = c * CH, len = n - off < CH ? n - off : CH;
        if (with_llh) parallel_memcpy(reinterpret_cast<char*>(lnprob + off), reinterpret_cast<const char*>(h_out + CH * slot), sizeof(double) * len);
        if (fr) parallel_memcpy(reinterpret_cast<char*>(fr + 3 * off), reinterpret_cast<const char*>(h_fr + 3 * CH * slot), sizeof(double) * 3 * len);
        if (status) parallel_memcpy(reinterpret_cast<char*>(status + off), reinterpret_cast<const char*>(h_st + CH * slot), sizeof(int32_t) * len);
    };
    for (int64_t c = 0; c < nchunks; ++c) {
        const int slot = (int)(c & 1);
        const int64_t off = c * CH, len = n - off < CH ? n - off : CH;
        if (with_llh) GF_HIP(hipMemcpyAsync(h_out + CH * slot, d_ln + off, sizeof(double) * len, hipMemcpyDeviceToHost, m->stream));
        if (fr) GF_HIP(hipMemcpyAsync(h_fr + 3 * CH * slot, d_fr + 3 * off, sizeof(double) * 3 * len, hipMemcpyDeviceToHost, m->stream));
        if (status) GF_HIP(hipMemcpyAsync(h_st + CH * slot, m->d_status + off, sizeof(int32_t) * len, hipMemcpyDeviceToHost, m->stream));
        GF_HIP(hipEventRecord(m->ev_down[slot], m->stream));
        if (c >= 1) {
            GF_HIP(hipEventSynchronize(m->ev_down[slot ^ 1]));
            copy_out(c - 1);
        }
    }
    GF_HIP(hipEventSynchronize(m->ev_down[(nchunks - 1) & 1]));
    copy_out(nchunks - 1);
    return status ? check_queue_overflow(m->device, m->stream) : GF_OK;
}

static int run_host(gf_model* m, const double* theta, int64_t n, double* lnprob, double* fr, int32_t* status,
                    bool with_llh)
{
    if (!m || n < 0 || (n > 0 && (!theta || (with_llh && !lnprob) || (!with_llh && !fr)))) return GF_ERR_INVALID_ARG;
    if (n == 0) return GF_OK;
    // one caller at a time per model: the staging buffers (and the arbitration queue) are the model's
    std::lock_guard<std::mutex> lk(m->call_mu);
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    static const bool pipe_off = gf_internal_env("GF_NO_HOST_PIPELINE", 0) != nullptr;  // diagnostics / A-B
    if (n >= PIPE_MIN_ROWS && !pipe_off) return run_host_pipelined(m, theta, n, lnprob, fr, status, with_llh);
    int rc = ensure_staging(m, n, n);
    if (rc != GF_OK) return rc;
    const size_t nd = (size_t)m->c.ndim;
    double* h_theta = (double*)m->h_pin;
    double* h_out = h_theta + nd * m->hcap;
    double* h_fr = h_out + m->hcap;
    int32_t* h_st = (int32_t*)(h_fr + 3 * m->hcap);
    std::memcpy(h_theta, theta, sizeof(double) * nd * n);
    // Small batches (emcee's half-ensemble of a 100-walker chain is 50 rows): the kernel reads theta from and
    // writes its results to the pinned, device-mapped staging buffer directly -- one launch and one stream
    // sync instead of launch + two DMA transfers, each of which costs more than the kilobytes they move.
    static const bool zc_off = gf_internal_env("GF_NO_ZEROCOPY", 0) != nullptr;       // diagnostics / A-B
    if (n <= GF_ZEROCOPY_MAX_ROWS && !zc_off) {
        if (with_llh)
            rc = launch_lnprob(m, m->stream, h_theta, GF_LAYOUT_AOS, n, h_out, fr ? h_fr : nullptr, status ? h_st : nullptr);
        else
            rc = launch_propagate(m, m->stream, h_theta, GF_LAYOUT_AOS, n, h_fr, status ? h_st : nullptr);
        if (rc != GF_OK) return rc;
        GF_HIP(hipStreamSynchronize(m->stream));
        if (with_llh) std::memcpy(lnprob, h_out, sizeof(double) * n);
        if (fr) std::memcpy(fr, h_fr, sizeof(double) * 3 * n);
        if (status) std::memcpy(status, h_st, sizeof(int32_t) * n);
        return status ? check_queue_overflow(m->device, m->stream) : GF_OK;
    }
    GF_HIP(hipMemcpyAsync(m->d_theta, h_theta, sizeof(double) * nd * n, hipMemcpyHostToDevice, m->stream));
    double* d_ln = m->d_out;
    double* d_fr = m->d_out + m->cap;
    if (with_llh)
        rc = launch_lnprob(m, m->stream, m->d_theta, GF_LAYOUT_AOS, n, d_ln, fr ? d_fr : nullptr, status ? m->d_status : nullptr);
    else
        rc = launch_propagate(m, m->stream, m->d_theta, GF_LAYOUT_AOS, n, d_fr, status ? m->d_status : nullptr);
    if (rc != GF_OK) return rc;
    if (with_llh) GF_HIP(hipMemcpyAsync(h_out, d_ln, sizeof(double) * n, hipMemcpyDeviceToHost, m->stream));
    if (fr) GF_HIP(hipMemcpyAsync(h_fr, d_fr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, m->stream));
    if (status) GF_HIP(hipMemcpyAsync(h_st, m->d_status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, m->stream));
    GF_HIP(hipStreamSynchronize(m->stream));
    if (with_llh) std::memcpy(lnprob, h_out, sizeof(double) * n);
    if (fr) std::memcpy(fr, h_fr, sizeof(double) * 3 * n);
    if (status) std::memcpy(status, h_st, sizeof(int32_t) * n);
    return status ? check_queue_overflow(m->device, m->stream) : GF_OK;
}

int gf_lnprob_batch(gf_model* m, const double* theta, int64_t n, double* lnprob, double* fr, int32_t* status)
{
    return run_host(m, theta, n, lnprob, fr, status, true);
}

// MultiNest-style batch (golemflavor/mn.py:26-45 lnProb): cube [n][nscan] in the unit cube; column cols[k] of theta is
// lo + (hi - lo) * cube[.][k] with the model's own box for that column, every other column is base[col].  The map runs
// on the device; only the cube crosses PCIe.
int gf_lnprob_cube_batch(gf_model* m, const double* cube, int64_t n, int nscan, const int32_t* cols, const double* base,
                         double* lnprob, double* fr, int32_t* status)
{
    if (!m || n < 0 || nscan < 1 || nscan > m->c.ndim || !cols || !base || (n > 0 && (!cube || !lnprob))) return GF_ERR_INVALID_ARG;
    bool seen[GF_MAX_DIM] = {false};
    for (int k = 0; k < nscan; ++k) {
        if (cols[k] < 0 || cols[k] >= m->c.ndim || seen[cols[k]]) return GF_ERR_INVALID_ARG;
        seen[cols[k]] = true;
    }
    if (n == 0) return GF_OK;
    std::lock_guard<std::mutex> lk(m->call_mu);
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    int rc = ensure_staging(m, n, n);
    if (rc != GF_OK) return rc;
    const size_t nd = (size_t)m->c.ndim;
    double* h_cube = (double*)m->h_pin;                      // the theta slot of the pinned mirror holds the (smaller) cube
    double* h_out = h_cube + nd * m->hcap;
    double* h_fr = h_out + m->hcap;
    int32_t* h_st = (int32_t*)(h_fr + 3 * m->hcap);
    double* d_ln = m->d_out;
    double* d_fr = m->d_out + m->cap;
    // device side: the cube rows get their own buffer, grown on demand (d_theta receives the expanded rows)
    if ((int64_t)m->cube_cap < n * nscan) {
        if (m->d_cube) (void)hipFree(m->d_cube);
        m->d_cube = nullptr; m->cube_cap = 0;
        GF_HIP(hipMalloc((void**)&m->d_cube, sizeof(double) * (size_t)n * nscan));
        m->cube_cap = (size_t)n * nscan;
    }
    std::memcpy(h_cube, cube, sizeof(double) * (size_t)n * nscan);
    GF_HIP(hipMemcpyAsync(m->d_cube, h_cube, sizeof(double) * (size_t)n * nscan, hipMemcpyHostToDevice, m->stream));
    hipError_t e = gf_launch_cube_to_theta(m->c, nscan, cols, base, m->d_cube, n, m->d_theta, m->cus, m->stream);
    if (e != hipSuccess) return hip_fail(e, "cube map launch");
    rc = launch_lnprob(m, m->stream, m->d_theta, GF_LAYOUT_AOS, n, d_ln, fr ? d_fr : nullptr, status ? m->d_status : nullptr);
    if (rc != GF_OK) return rc;
    GF_HIP(hipMemcpyAsync(h_out, d_ln, sizeof(double) * n, hipMemcpyDeviceToHost, m->stream));
    if (fr) GF_HIP(hipMemcpyAsync(h_fr, d_fr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, m->stream));
    if (status) GF_HIP(hipMemcpyAsync(h_st, m->d_status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, m->stream));
    GF_HIP(hipStreamSynchronize(m->stream));
    std::memcpy(lnprob, h_out, sizeof(double) * n);
    if (fr) std::memcpy(fr, h_fr, sizeof(double) * 3 * n);
    if (status) std::memcpy(status, h_st, sizeof(int32_t) * n);
    return status ? check_queue_overflow(m->device, m->stream) : GF_OK;
}

int gf_propagate_batch(gf_model* m, const double* theta, int64_t n, double* fr, int32_t* status)
{
    return run_host(m, theta, n, nullptr, fr, status, false);
}

int gf_haar_draw(gf_model* m, uint64_t seed, int64_t first_draw, int64_t n, double* angles, double* fr)
{
    if (!m || n < 0 || (n > 0 && !fr)) return GF_ERR_INVALID_ARG;
    if (n == 0) return GF_OK;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    double *d_fr = nullptr, *d_ang = nullptr;
    GF_HIP(hipMalloc((void**)&d_fr, sizeof(double) * 3 * n));
    if (angles) {
        hipError_t e = hipMalloc((void**)&d_ang, sizeof(double) * 4 * n);
        if (e != hipSuccess) { (void)hipFree(d_fr); return hip_fail(e, "hipMalloc(angles)"); }
    }
    int rc = gf_haar_draw_device(m, seed, first_draw, n, d_ang, d_fr);
    hipError_t e = hipSuccess;
    if (rc == GF_OK) e = hipMemcpyAsync(fr, d_fr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, m->stream);
    if (rc == GF_OK && e == hipSuccess && angles)
        e = hipMemcpyAsync(angles, d_ang, sizeof(double) * 4 * n, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    (void)hipFree(d_fr);
    if (d_ang) (void)hipFree(d_ang);
    if (rc != GF_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "gf_haar_draw");
    return GF_OK;
}

// ---- device-resident entry points ----------------------------------------------------------
int gf_device_alloc(gf_model* m, size_t bytes, void** dptr)
{
    if (!m || !dptr) return GF_ERR_INVALID_ARG;
    GF_HIP(hipSetDevice(m->device));
    GF_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return GF_OK;
}

int gf_device_free(gf_model* m, void* dptr)
{
    if (!m) return GF_ERR_INVALID_ARG;
    if (!dptr) return GF_OK;
    GF_HIP(hipSetDevice(m->device));
    if (m->stream) GF_HIP(hipStreamSynchronize(m->stream));
    GF_HIP(hipFree(dptr));
    return GF_OK;
}

int gf_memcpy_h2d(gf_model* m, void* dst_dev, const void* src_host, size_t bytes)
{
    if (!m || !dst_dev || !src_host) return GF_ERR_INVALID_ARG;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    GF_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, m->stream));
    GF_HIP(hipStreamSynchronize(m->stream));
    return GF_OK;
}

// ---- large device-to-host copies: a ring of pinned slots + host threads ----------------------------------------------------
// hipMemcpy into pageable memory is at the mercy of the runtime's own choice between pinning the destination in place and
// staging: measured on one box, in one process, five 1.9 GB read-backs into fresh arrays ran at 42, 52, 10, 51 and 15 GB/s
// (tools/numa_probe.py; not a NUMA effect: the same with the threads bound to either node).  So reads of 16 MB and more bring
// their own staging: eight pinned 16 MB slots per device (allocated once), the DMA engine fills slot c while host threads copy
// slot c - 1 (and c - 2 ...) into the destination -- which also touches the destination's pages for the first time, from several
// threads, so no separate page-mapping pass is needed.  PCIe runs at its pinned-memory rate whatever the destination is.
namespace {
constexpr size_t D2H_SLOT = (size_t)16 << 20;
constexpr int D2H_SLOTS = 8;
constexpr size_t D2H_RING_MIN = (size_t)16 << 20;
struct D2HRing {
    std::mutex mu;                                   // one large read-back at a time per device
    void* slot[D2H_SLOTS] = {};
    hipEvent_t ev[D2H_SLOTS] = {};
};
D2HRing g_d2h[POOL_MAX_DEVICES];

// The host threads that empty the slots: a pool created on first use and kept (starting seven threads per 16 MB slot cost
// 0.15-0.2 ms against the 0.33 ms the copy itself takes).  A job is a packed source of nrows x width bytes going to rows
// `dpitch` apart; it is cut into 1 MB pieces that the workers and the calling thread take from a shared counter, so the
// pieces balance themselves whatever the row length.  Leaked on purpose at exit (the workers sleep on the condition
// variable); a forked child starts without a pool.
struct CopyPool {
    std::mutex mu;
    std::condition_variable wake, done;
    std::vector<std::thread> workers;
    char* dst = nullptr; const char* src = nullptr;
    size_t dpitch = 0, width = 0, total = 0, piece = 0, ntasks = 0;
    std::atomic<size_t> next{0};
    unsigned generation = 0, busy = 0;

    static void piece_copy(char* dst, size_t dpitch, const char* src, size_t width, size_t lo, size_t hi)
    {
        while (lo < hi) {                                          // [lo, hi) of the packed source, row by row
            const size_t r = lo / width, off = lo - r * width;
            const size_t n = (width - off) < (hi - lo) ? (width - off) : (hi - lo);
            std::memcpy(dst + r * dpitch + off, src + lo, n);
            lo += n;
        }
    }
    void take()
    {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= ntasks) return;
            const size_t lo = i * piece, hi = lo + piece < total ? lo + piece : total;
            piece_copy(dst, dpitch, src, width, lo, hi);
        }
    }
    void worker()
    {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                wake.wait(lk, [&] { return generation != seen; });
                seen = generation;
            }
            take();
            std::lock_guard<std::mutex> lk(mu);
            if (--busy == 0) done.notify_one();
        }
    }
    void run(char* d, size_t dp, const char* s, size_t w, size_t nrows)
    {
        const size_t bytes = w * nrows;
        if (workers.empty() || bytes < ((size_t)2 << 20)) { piece_copy(d, dp, s, w, 0, bytes); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            dst = d; dpitch = dp; src = s; width = w; total = bytes;
            piece = (size_t)1 << 20;
            ntasks = (bytes + piece - 1) / piece;
            next.store(0, std::memory_order_relaxed);
            busy = (unsigned)workers.size();
            ++generation;
        }
        wake.notify_all();
        take();
        std::unique_lock<std::mutex> lk(mu);
        done.wait(lk, [&] { return busy == 0; });
    }
};
CopyPool* g_copy_pool = nullptr;
std::mutex g_copy_pool_mu;                        // one job at a time (the rings of two devices may drain concurrently)

void copy_rows(char* dst, size_t dpitch, const char* src, size_t width, size_t nrows)
{
    std::lock_guard<std::mutex> lk(g_copy_pool_mu);
    if (!g_copy_pool) {
        size_t nt = 0;
        if (const char* v = gf_internal_env("GF_D2H_THREADS", 0)) { const long k = std::atol(v); if (k >= 1 && k <= 64) nt = (size_t)k; }
        if (!nt) { const unsigned hw = std::thread::hardware_concurrency(); nt = hw >= 32 ? 8 : (hw >= 8 ? 4 : 1); }
        g_copy_pool = new CopyPool();
        for (size_t k = 1; k < nt; ++k) {
            g_copy_pool->workers.emplace_back([p = g_copy_pool] { p->worker(); });
            g_copy_pool->workers.back().detach();
        }
        static bool hooked = false;
        if (!hooked) { hooked = true; pthread_atfork(nullptr, nullptr, [] { g_copy_pool = nullptr; new (&g_copy_pool_mu) std::mutex(); }); }
    }
    g_copy_pool->run(dst, dpitch, src, width, nrows);
}
void parallel_memcpy(char* dst, const char* src, size_t len) { copy_rows(dst, len, src, len, 1); }

// Is the host range [p, p + span) REGISTERED memory (gf_host_register) or hipHostMalloc'ed -- memory the device can write itself?
// Asked once per read-back call (two attribute queries: the first byte and the last); GF_NO_DIRECT_D2H=1: always "no" (A/B).
// *alias: the device's address of p.
bool host_range_is_pinned(const void* p, size_t span, void** alias = nullptr)
{
    if (!p || span == 0) return false;
    static const bool off = gf_internal_env("GF_NO_DIRECT_D2H", 0) != nullptr;
    if (off) return false;
    const char* ends[2] = {static_cast<const char*>(p), static_cast<const char*>(p) + span - 1};
    for (const char* q : ends) {
        hipPointerAttribute_t a;
        std::memset(&a, 0, sizeof(a));
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain pageable memory: an error on some runtimes
        if (a.type != hipMemoryTypeHost) return false;                                                 // ... hipMemoryTypeUnregistered on others
    }
    if (alias) {
        *alias = nullptr;
        if (hipHostGetDevicePointer(alias, const_cast<void*>(p), 0) != hipSuccess || !*alias) { (void)hipGetLastError(); return false; }
    }
    return true;
}

// Rows of device memory into registered host memory by a KERNEL that stores through the host memory's device alias: 55-56 GB/s from 32
// workgroups (tools/experiments/d2h_kernel_probe.hip), and unaffected by what halves the runtime's DMA for ~0.6 s after a large hipFree
// (the driver wiping the freed memory with the DMA engine: tools/vram_realloc_probe4.py, profiles/r04/host_register.txt).  NOT the
// default (enqueue_d2h_rows): beside other kernels it costs them a dispatch each, and the device cache keeps the wipe from happening.
// A tile is 256 lanes x 16 elements of one row; few workgroups on purpose: the link is the bound.
extern "C++" {
template <typename T>
__global__ __launch_bounds__(256) void k_d2h_rows(T* __restrict__ dst, size_t dpitch_e, const T* __restrict__ src, size_t spitch_e,
                                                  size_t width_e, size_t height)
{
    constexpr size_t TILE = 256 * 16;
    const size_t tiles_per_row = (width_e + TILE - 1) / TILE, ntiles = tiles_per_row * height;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t r = t / tiles_per_row, c0 = (t - r * tiles_per_row) * TILE;
        const T* s = src + r * spitch_e + c0;
        T* d = dst + r * dpitch_e + c0;
        const size_t n = width_e - c0 < TILE ? width_e - c0 : TILE;
#pragma unroll 4
        for (size_t i = threadIdx.x; i < n; i += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
    }
}
}  // extern "C++"
typedef double gf_v2d __attribute__((ext_vector_type(2)));

// enqueue on `st`: `height` rows of `width` bytes, `spitch` apart on the device, to the registered host rows `dpitch` apart behind
// `dst_alias` (their device address).
hipError_t enqueue_d2h_rows(hipStream_t st, void* dst_host, void* dst_alias, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height)
{
    // The runtime's DMA by default: it needs no compute unit, so it runs beside the sampler's and the post-processing kernels without
    // costing them a dispatch (the copy kernel beside them: C5 89 -> 177 us per half-step, C4's rows 33 GB/s).  The kernel is the
    // faster of the two only while the driver wipes freed memory with the DMA engine -- which the device cache (gf_devcache.h) now
    // keeps from happening.  GF_D2H_KERNEL=1: the kernel (A/B).
    static const bool dma = gf_internal_env("GF_D2H_KERNEL", 0) == nullptr;
    const bool words = ((width | dpitch | spitch | (size_t)(uintptr_t)dst_alias | (size_t)(uintptr_t)src) & 7u) == 0;
    if (dma || !words || !dst_alias)
        return height == 1 ? hipMemcpyAsync(dst_host, src, width, hipMemcpyDeviceToHost, st)
                           : hipMemcpy2DAsync(dst_host, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, st);
    static const int blocks = [] { const char* v = gf_internal_env("GF_D2H_KERNEL_BLOCKS", 0); const int k = v ? std::atoi(v) : 0; return k >= 1 && k <= 4096 ? k : 32; }();
    const bool wide = ((width | dpitch | spitch | (size_t)(uintptr_t)dst_alias | (size_t)(uintptr_t)src) & 15u) == 0;
    if (wide)
        hipLaunchKernelGGL(k_d2h_rows<gf_v2d>, dim3(blocks), dim3(256), 0, st, static_cast<gf_v2d*>(dst_alias), dpitch / 16,
                           static_cast<const gf_v2d*>(src), spitch / 16, width / 16, height);
    else
        hipLaunchKernelGGL(k_d2h_rows<double>, dim3(blocks), dim3(256), 0, st, static_cast<double*>(dst_alias), dpitch / 8,
                           static_cast<const double*>(src), spitch / 8, width / 8, height);
    return hipGetLastError();
}
}  // namespace

// internal (also gf_sampler.hip): synchronous copy of `bytes` from device memory to any host memory, through the ring,
// in order on `stream`.  `gate` (may be NULL): called before a chunk is issued with the end offset of that chunk; returns
// once the source bytes [0, upto) are final (gf_sampler_postprocess_rows: the event of the group of chains they belong to),
// non-zero to abandon the copy -- so ONE pipeline runs over a source that is still being produced.
int gf_internal_d2h_gated(int device, void* stream, void* dst_host, const void* src_dev, size_t bytes,
                          int (*gate)(void* ctx, size_t upto), void* gate_ctx)
{
    if (device < 0 || device >= POOL_MAX_DEVICES || !dst_host || !src_dev) return GF_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    static const bool ring_off = gf_internal_env("GF_NO_D2H_PIPELINE", 0) != nullptr;            // diagnostics / A-B
    void* alias = nullptr;
    if (bytes >= D2H_RING_MIN && host_range_is_pinned(dst_host, bytes, &alias)) {
        // a registered destination (gf_host_register): written directly (enqueue_d2h_rows), piece by piece as the source is completed
        const size_t piece = (size_t)64 << 20;
        for (size_t off = 0; off < bytes; off += piece) {
            const size_t len = bytes - off < piece ? bytes - off : piece;
            if (gate && gate(gate_ctx, off + len) != 0) {
                (void)hipStreamSynchronize(st);
                std::snprintf(g_err, sizeof(g_err), "gf_internal_d2h: the source was not completed");
                return GF_ERR_HIP;
            }
            GF_HIP(enqueue_d2h_rows(st, static_cast<char*>(dst_host) + off, static_cast<char*>(alias) + off, len, static_cast<const char*>(src_dev) + off, len, len, 1));
        }
        GF_HIP(hipStreamSynchronize(st));
        return GF_OK;
    }
    if (bytes < D2H_RING_MIN || ring_off) {
        if (gate && gate(gate_ctx, bytes) != 0) return GF_ERR_HIP;
        GF_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        GF_HIP(hipStreamSynchronize(st));
        return GF_OK;
    }
    D2HRing& R = g_d2h[device];
    std::lock_guard<std::mutex> lk(R.mu);
    for (int k = 0; k < D2H_SLOTS; ++k) {
        if (!R.slot[k]) GF_HIP(hipHostMalloc(&R.slot[k], D2H_SLOT, hipHostMallocDefault));
        if (!R.ev[k]) GF_HIP(hipEventCreateWithFlags(&R.ev[k], hipEventDisableTiming));
    }
    const size_t nchunks = (bytes + D2H_SLOT - 1) / D2H_SLOT;
    std::atomic<size_t> issued{0}, drained{0};
    std::atomic<int> failed{0};
    char* dst = static_cast<char*>(dst_host);
    std::thread consumer([&]() {
        for (size_t c = 0; c < nchunks; ++c) {
            while (issued.load(std::memory_order_acquire) <= c && !failed.load()) std::this_thread::yield();
            if (failed.load()) return;
            if (hipEventSynchronize(R.ev[c % D2H_SLOTS]) != hipSuccess) { failed.store(1); return; }
            const size_t off = c * D2H_SLOT, len = bytes - off < D2H_SLOT ? bytes - off : D2H_SLOT;
            parallel_memcpy(dst + off, static_cast<const char*>(R.slot[c % D2H_SLOTS]), len);
            drained.store(c + 1, std::memory_order_release);
        }
    });
    hipError_t e = hipSuccess;
    for (size_t c = 0; c < nchunks && e == hipSuccess && !failed.load(); ++c) {
        while (c >= drained.load(std::memory_order_acquire) + D2H_SLOTS && !failed.load()) std::this_thread::yield();   // the slot is free
        const size_t off = c * D2H_SLOT, len = bytes - off < D2H_SLOT ? bytes - off : D2H_SLOT;
        if (gate && gate(gate_ctx, off + len) != 0) { failed.store(2); break; }
        e = hipMemcpyAsync(R.slot[c % D2H_SLOTS], static_cast<const char*>(src_dev) + off, len, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(R.ev[c % D2H_SLOTS], st);
        if (e == hipSuccess) issued.store(c + 1, std::memory_order_release);
    }
    if (e != hipSuccess) failed.store(1);
    consumer.join();
    if (e != hipSuccess) return hip_fail(e, "gf_internal_d2h");
    if (failed.load()) {
        std::snprintf(g_err, sizeof(g_err), failed.load() == 2 ? "gf_internal_d2h: the source was not completed" : "gf_internal_d2h: event wait failed");
        return GF_ERR_HIP;
    }
    return GF_OK;
}
int gf_internal_d2h(int device, void* stream, void* dst_host, const void* src_dev, size_t bytes)
{
    return gf_internal_d2h_gated(device, stream, dst_host, src_dev, bytes, nullptr, nullptr);
}

// internal (gf_sampler.hip): the same pipeline for a pitched block -- `height` rows of `width` bytes, `spitch` apart on the
// device and `dpitch` apart on the host (the stored prefix of every chain of a sampler: row = chain).  A slot takes as many
// whole rows as fit (one hipMemcpy2DAsync); a row wider than a slot goes through the 1-D pipeline on its own.
int gf_internal_d2h_2d(int device, void* stream, void* dst_host, size_t dpitch, const void* src_dev, size_t spitch, size_t width,
                       size_t height)
{
    if (device < 0 || device >= POOL_MAX_DEVICES || !dst_host || !src_dev) return GF_ERR_INVALID_ARG;
    if (width == 0 || height == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    char* dst = static_cast<char*>(dst_host);
    const char* src = static_cast<const char*>(src_dev);
    if (width == dpitch && width == spitch) return gf_internal_d2h_gated(device, stream, dst_host, src_dev, width * height, nullptr, nullptr);
    static const bool ring_off = gf_internal_env("GF_NO_D2H_PIPELINE", 0) != nullptr;
    void* alias = nullptr;
    if (width * height >= D2H_RING_MIN && host_range_is_pinned(dst, (height - 1) * dpitch + width, &alias)) {
        GF_HIP(enqueue_d2h_rows(st, dst, alias, dpitch, src, spitch, width, height));
        GF_HIP(hipStreamSynchronize(st));
        return GF_OK;
    }
    if (width * height < D2H_RING_MIN || ring_off) {
        GF_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, st));
        GF_HIP(hipStreamSynchronize(st));
        return GF_OK;
    }
    if (width > D2H_SLOT) {
        for (size_t r = 0; r < height; ++r) {
            const int rc = gf_internal_d2h_gated(device, stream, dst + r * dpitch, src + r * spitch, width, nullptr, nullptr);
            if (rc != GF_OK) return rc;
        }
        return GF_OK;
    }
    D2HRing& R = g_d2h[device];
    std::lock_guard<std::mutex> lk(R.mu);
    for (int k = 0; k < D2H_SLOTS; ++k) {
        if (!R.slot[k]) GF_HIP(hipHostMalloc(&R.slot[k], D2H_SLOT, hipHostMallocDefault));
        if (!R.ev[k]) GF_HIP(hipEventCreateWithFlags(&R.ev[k], hipEventDisableTiming));
    }
    const size_t rps = D2H_SLOT / width;                          // rows per slot (>= 1)
    const size_t nchunks = (height + rps - 1) / rps;
    std::atomic<size_t> issued{0}, drained{0};
    std::atomic<int> failed{0};
    std::thread consumer([&]() {
        for (size_t c = 0; c < nchunks; ++c) {
            while (issued.load(std::memory_order_acquire) <= c && !failed.load()) std::this_thread::yield();
            if (failed.load()) return;
            if (hipEventSynchronize(R.ev[c % D2H_SLOTS]) != hipSuccess) { failed.store(1); return; }
            const size_t r0 = c * rps, nr = height - r0 < rps ? height - r0 : rps;
            const char* slot = static_cast<const char*>(R.slot[c % D2H_SLOTS]);
            copy_rows(dst + r0 * dpitch, dpitch, slot, width, nr);
            drained.store(c + 1, std::memory_order_release);
        }
    });
    hipError_t e = hipSuccess;
    for (size_t c = 0; c < nchunks && e == hipSuccess && !failed.load(); ++c) {
        while (c >= drained.load(std::memory_order_acquire) + D2H_SLOTS && !failed.load()) std::this_thread::yield();
        const size_t r0 = c * rps, nr = height - r0 < rps ? height - r0 : rps;
        e = hipMemcpy2DAsync(R.slot[c % D2H_SLOTS], width, src + r0 * spitch, spitch, width, nr, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(R.ev[c % D2H_SLOTS], st);
        if (e == hipSuccess) issued.store(c + 1, std::memory_order_release);
    }
    if (e != hipSuccess) failed.store(1);
    consumer.join();
    if (e != hipSuccess) return hip_fail(e, "gf_internal_d2h_2d");
    if (failed.load()) { std::snprintf(g_err, sizeof(g_err), "gf_internal_d2h_2d: event wait failed"); return GF_ERR_HIP; }
    return GF_OK;
}

// ---- one read-back pipeline over MANY pitched blocks (gf_sampler_run_to_host) ---------------------------------------------------------
// gf_internal_d2h_2d is a pipeline per call: thread start, ring fill and ring drain every time.  A chain that leaves the device block of
// steps by block of steps while it is sampled (62 blocks of 201 MB for the C5 scan at the reference's length) paid that 62 times and
// reached 31 GB/s where the link gives 48 (profiles/r04/readback_overlap.txt).  A pipe keeps ONE consumer thread and the ring's state
// across blocks: gf_internal_d2h_pipe_rows returns as soon as the block's chunks have been ISSUED (the DMA of chunk c runs while the host
// threads empty chunk c - 1 into the destination), the next block's chunks follow without a gap, and only _close drains.
struct gf_d2h_pipe {
    int device = 0;
    hipStream_t st = nullptr;
    D2HRing* ring = nullptr;
    struct Chunk { char* dst; size_t dpitch, width, nrows; } desc[D2H_SLOTS] = {};
    std::atomic<size_t> issued{0}, drained{0};
    std::atomic<int> failed{0}, closing{0};
    std::thread consumer;
    bool direct_pending = false;            // blocks went straight into a registered destination: _close waits for the stream
};

// The ring (eight pinned 16 MB slots, allocated once per device) and the consumer thread are set up when the first chunk needs them:
// a pipe whose blocks all go straight into a registered destination never touches either (and a run does not start with 128 MB of
// hipHostMalloc, 30 ms when it is the process's first and now and then several times that).
static int pipe_start(gf_d2h_pipe* p)
{
    if (p->ring) return GF_OK;
    D2HRing* ring = &g_d2h[p->device];
    ring->mu.lock();                                      // one large read-back at a time per device; released by _close
    for (int k = 0; k < D2H_SLOTS; ++k) {
        hipError_t e = hipSuccess;
        if (!ring->slot[k]) e = hipHostMalloc(&ring->slot[k], D2H_SLOT, hipHostMallocDefault);
        if (e == hipSuccess && !ring->ev[k]) e = hipEventCreateWithFlags(&ring->ev[k], hipEventDisableTiming);
        if (e != hipSuccess) { ring->mu.unlock(); return hip_fail(e, "gf_internal_d2h_pipe: ring"); }
    }
    p->ring = ring;
    p->consumer = std::thread([p]() {
        for (size_t c = 0;; ++c) {
            while (p->issued.load(std::memory_order_acquire) <= c) {
                if (p->failed.load() || p->closing.load()) {
                    if (p->issued.load(std::memory_order_acquire) <= c) return;       // closing and nothing more to empty
                    break;
                }
                std::this_thread::yield();
            }
            if (p->failed.load()) return;
            const int k = (int)(c % D2H_SLOTS);
            if (hipEventSynchronize(p->ring->ev[k]) != hipSuccess) { p->failed.store(1); return; }
            const gf_d2h_pipe::Chunk& d = p->desc[k];
            copy_rows(d.dst, d.dpitch, static_cast<const char*>(p->ring->slot[k]), d.width, d.nrows);
            p->drained.store(c + 1, std::memory_order_release);
        }
    });
    return GF_OK;
}

int gf_internal_d2h_pipe_open(int device, void* stream, gf_d2h_pipe** out)
{
    if (device < 0 || device >= POOL_MAX_DEVICES || !out) return GF_ERR_INVALID_ARG;
    *out = nullptr;
    gf_d2h_pipe* p = new (std::nothrow) gf_d2h_pipe();
    if (!p) return GF_ERR_ALLOC;
    p->device = device; p->st = (hipStream_t)stream; p->ring = nullptr;
    *out = p;
    return GF_OK;
}

// `height` rows of `width` bytes, `spitch` apart on the device, to rows `dpitch` apart on the host; returns once every chunk is issued
int gf_internal_d2h_pipe_rows(gf_d2h_pipe* p, void* dst_host, size_t dpitch, const void* src_dev, size_t spitch, size_t width, size_t height)
{
    if (!p || !dst_host || !src_dev) return GF_ERR_INVALID_ARG;
    if (width == 0 || height == 0) return GF_OK;
    char* dst = static_cast<char*>(dst_host);
    const char* src = static_cast<const char*>(src_dev);
    hipError_t e = hipSuccess;
    void* alias = nullptr;
    if (host_range_is_pinned(dst, (height - 1) * dpitch + width, &alias)) {
        // a registered destination: the block goes straight to where its rows belong (enqueue_d2h_rows) -- no slot, no host thread
        e = enqueue_d2h_rows(p->st, dst, alias, dpitch, src, spitch, width, height);
        if (e != hipSuccess) { p->failed.store(1); return hip_fail(e, "gf_internal_d2h_pipe_rows"); }
        p->direct_pending = true;
        return GF_OK;
    }
    {
        const int rc = pipe_start(p);
        if (rc != GF_OK) { p->failed.store(1); return rc; }
    }
    auto issue = [&](char* d, size_t dp, const char* sp, size_t spi, size_t w, size_t nr) {
        const size_t c = p->issued.load(std::memory_order_relaxed);
        while (c >= p->drained.load(std::memory_order_acquire) + D2H_SLOTS && !p->failed.load()) std::this_thread::yield();   // the slot is free
        if (p->failed.load()) return;
        const int k = (int)(c % D2H_SLOTS);
        p->desc[k] = {d, dp, w, nr};
        e = nr == 1 ? hipMemcpyAsync(p->ring->slot[k], sp, w, hipMemcpyDeviceToHost, p->st)
                    : hipMemcpy2DAsync(p->ring->slot[k], w, sp, spi, w, nr, hipMemcpyDeviceToHost, p->st);
        if (e == hipSuccess) e = hipEventRecord(p->ring->ev[k], p->st);
        if (e == hipSuccess) p->issued.store(c + 1, std::memory_order_release);
    };
    if (width > D2H_SLOT) {                                 // a row wider than a slot: in pieces
        for (size_t r = 0; r < height && e == hipSuccess && !p->failed.load(); ++r)
            for (size_t off = 0; off < width && e == hipSuccess && !p->failed.load(); off += D2H_SLOT)
                issue(dst + r * dpitch + off, 0, src + r * spitch + off, 0, width - off < D2H_SLOT ? width - off : D2H_SLOT, 1);
    } else {
        const size_t rps = D2H_SLOT / width;                // whole rows per slot
        for (size_t r0 = 0; r0 < height && e == hipSuccess && !p->failed.load(); r0 += rps)
            issue(dst + r0 * dpitch, dpitch, src + r0 * spitch, spitch, width, height - r0 < rps ? height - r0 : rps);
    }
    if (e != hipSuccess) { p->failed.store(1); return hip_fail(e, "gf_internal_d2h_pipe_rows"); }
    return p->failed.load() ? GF_ERR_HIP : GF_OK;
}

int gf_internal_d2h_pipe_close(gf_d2h_pipe* p)
{
    if (!p) return GF_OK;
    p->closing.store(1);
    if (p->consumer.joinable()) p->consumer.join();
    if (p->direct_pending && hipStreamSynchronize(p->st) != hipSuccess) p->failed.store(1);
    const int bad = p->failed.load();
    if (p->ring) p->ring->mu.unlock();
    delete p;
    if (bad) { std::snprintf(g_err, sizeof(g_err), "gf_internal_d2h_pipe: a copy or an event wait failed"); return GF_ERR_HIP; }
    return GF_OK;
}

// diagnostics (tools/readback_ab.py, not part of the ABI): what the link delivers in this process -- `bytes` of device memory copied into
// PINNED host memory in 64 MiB pieces, no host copy behind them; GB/s
int gf_internal_pinned_d2h_rate(int device, size_t bytes, double* gbps)
{
    if (!gbps || bytes < ((size_t)64 << 20)) return GF_ERR_INVALID_ARG;
    GF_HIP(hipSetDevice(device));
    const size_t piece = (size_t)64 << 20;
    void *h = nullptr, *d = nullptr;
    hipStream_t st = nullptr;
    hipError_t e = hipHostMalloc(&h, 2 * piece, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&d, piece);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, piece, hipMemcpyDeviceToHost, st);          // warm-up
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    size_t done = 0;
    for (int k = 0; e == hipSuccess && done < bytes; ++k, done += piece)
        e = hipMemcpyAsync(static_cast<char*>(h) + (k & 1) * piece, d, piece, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (st) (void)hipStreamDestroy(st);
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    if (e != hipSuccess) return hip_fail(e, "gf_internal_pinned_d2h_rate");
    *gbps = (double)done / dt / 1e9;
    return GF_OK;
}

int gf_memcpy_d2h(gf_model* m, void* dst_host, const void* src_dev, size_t bytes)
{
    if (!m || !dst_host || !src_dev) return GF_ERR_INVALID_ARG;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    return gf_internal_d2h(m->device, (void*)m->stream, dst_host, src_dev, bytes);
}

int gf_lnprob_batch_device(gf_model* m, const double* d_theta, int layout, int64_t n, double* d_lnprob, double* d_fr,
                           int32_t* d_status)
{
    if (!m || n < 0 || (layout != GF_LAYOUT_AOS && layout != GF_LAYOUT_SOA)) return GF_ERR_INVALID_ARG;
    if (n == 0) return GF_OK;
    int rc = check_dev_ptr(d_theta, 16);
    if (rc == GF_OK) rc = check_dev_ptr(d_lnprob, 8);
    if (rc != GF_OK) return rc;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    std::lock_guard<std::mutex> lk(m->call_mu);
    return launch_lnprob(m, m->stream, d_theta, layout, n, d_lnprob, d_fr, d_status);
}

int gf_propagate_batch_device(gf_model* m, const double* d_theta, int layout, int64_t n, double* d_fr,
                              int32_t* d_status)
{
    if (!m || n < 0 || (layout != GF_LAYOUT_AOS && layout != GF_LAYOUT_SOA)) return GF_ERR_INVALID_ARG;
    if (n == 0) return GF_OK;
    int rc = check_dev_ptr(d_theta, 16);
    if (rc == GF_OK) rc = check_dev_ptr(d_fr, 8);
    if (rc != GF_OK) return rc;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    std::lock_guard<std::mutex> lk(m->call_mu);
    return launch_propagate(m, m->stream, d_theta, layout, n, d_fr, d_status);
}

int gf_haar_draw_device(gf_model* m, uint64_t seed, int64_t first_draw, int64_t n, double* d_angles, double* d_fr)
{
    if (!m || n < 0 || first_draw < 0) return GF_ERR_INVALID_ARG;
    if (n == 0) return GF_OK;
    int rc = check_dev_ptr(d_fr, 8);
    if (rc == GF_OK && d_angles) rc = check_dev_ptr(d_angles, 32);
    if (rc != GF_OK) return rc;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    hipError_t e = gf_launch_haar(m->c, seed, first_draw, n, d_angles, d_fr, m->cus, m->stream);
    if (e != hipSuccess) return hip_fail(e, "haar launch");
    return GF_OK;
}

// counts[nb][nb][nb] += histogram of n compositions resident on the device (asynchronous)
int gf_flavor_histogram_device(gf_model* m, const double* d_fr, int64_t n, int nbins, uint64_t* d_counts)
{
    if (!m || n < 0 || nbins < 1 || nbins > 1024) return GF_ERR_INVALID_ARG;
    if (n == 0) return GF_OK;
    int rc = check_dev_ptr(d_fr, 8);
    if (rc == GF_OK) rc = check_dev_ptr(d_counts, 8);
    if (rc != GF_OK) return rc;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    hipError_t e = gf_launch_flavor_hist(d_fr, n, nbins, (unsigned long long*)d_counts, m->cus, m->stream);
    if (e != hipSuccess) return hip_fail(e, "histogram launch");
    return GF_OK;
}

// host convenience: fr [n][3] -> counts [nbins]^3 (zeroed first)
int gf_flavor_histogram(gf_model* m, const double* fr, int64_t n, int nbins, uint64_t* counts)
{
    if (!m || n < 0 || (n > 0 && !fr) || !counts || nbins < 1 || nbins > 1024) return GF_ERR_INVALID_ARG;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    const size_t nbin3 = (size_t)nbins * nbins * nbins;
    double* d_fr = nullptr;
    uint64_t* d_c = nullptr;
    GF_HIP(hipMalloc((void**)&d_c, sizeof(uint64_t) * nbin3));
    hipError_t e = hipMemsetAsync(d_c, 0, sizeof(uint64_t) * nbin3, m->stream);
    if (e == hipSuccess && n > 0) e = hipMalloc((void**)&d_fr, sizeof(double) * 3 * n);
    if (e == hipSuccess && n > 0) e = hipMemcpyAsync(d_fr, fr, sizeof(double) * 3 * n, hipMemcpyHostToDevice, m->stream);
    int rc = GF_OK;
    if (e == hipSuccess && n > 0) rc = gf_flavor_histogram_device(m, d_fr, n, nbins, d_c);
    if (e == hipSuccess && rc == GF_OK) e = hipMemcpyAsync(counts, d_c, sizeof(uint64_t) * nbin3, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (d_fr) (void)hipFree(d_fr);
    (void)hipFree(d_c);
    if (rc != GF_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "gf_flavor_histogram");
    return GF_OK;
}

// Make a freshly allocated host buffer ready to receive a large device-to-host copy at PCIe speed: its pages are
// touched (one byte per page is WRITTEN: the content is clobbered) from several threads.  Measured on the MI355X box
// (tools/pcie_probe.*): a D2H into untouched malloc / np.empty memory runs at 11-20 GB/s (page faults inside the copy),
// into touched pageable memory at 48-56 GB/s -- the rate of pinned memory, whose allocation itself costs 4.7 GB/s;
// touching 2 GiB takes 17 ms with 8 threads.
int gf_host_prepare_n(void* buf, size_t bytes, int threads);
int gf_host_prepare(void* buf, size_t bytes) { return gf_host_prepare_n(buf, bytes, 0); }

// threads <= 0: up to 16 (as many as finish 2 GiB in ~15 ms); a caller that lets the mapping run BESIDE its own work asks for
// few: sixteen threads taking page faults hold up the main thread's launches and allocations (they share the address space's
// lock), two do not and still stay ahead of a PCIe copy (profiles/r03/readback.txt)
int gf_host_prepare_n(void* buf, size_t bytes, int threads)
{
    if (!buf && bytes) return GF_ERR_INVALID_ARG;
    const size_t page = 4096;
    const size_t min_per_thread = 16u << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? hw / 2 : 4;
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    if (threads > 0) nt = (size_t)(threads > 64 ? 64 : threads);
    if (const char* e = gf_internal_env("GF_PREPARE_THREADS", 0)) { const long k = std::atol(e); if (k >= 1 && k <= 64) nt = (size_t)k; }   // A/B
    if (bytes / min_per_thread < nt) nt = bytes / min_per_thread ? bytes / min_per_thread : 1;
    // Mapping WITHOUT touching the content, so that a buffer may be prepared while a copy fills it: madvise(MADV_POPULATE_WRITE)
    // (Linux 5.14+: the kernel faults the range in writable, in one call per thread's share), and where that is refused a locked
    // OR of zero into one byte per page -- written in assembly: the compiler turns an __atomic_fetch_or(p, 0) into a LOAD, and a
    // load of an untouched anonymous page maps the shared zero page, i.e. nothing (measured: "12.6 GB in 1 ms").
    auto touch = [=](size_t lo, size_t hi) {
        char* p = static_cast<char*>(buf);
        const uintptr_t a = reinterpret_cast<uintptr_t>(p + lo), b = reinterpret_cast<uintptr_t>(p + hi);
        const uintptr_t pa = (a + page - 1) & ~(uintptr_t)(page - 1), pb = b & ~(uintptr_t)(page - 1);
        bool populated = false;
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23          /* linux/mman.h, Linux 5.14+; older headers lack the name, older kernels answer EINVAL */
#endif
        static const bool force_poke = gf_internal_env("GF_PREPARE_FORCE_POKE", 0) != nullptr;       // tests: the path taken where madvise is refused
        if (pb > pa && !force_poke) populated = madvise(reinterpret_cast<void*>(pa), pb - pa, MADV_POPULATE_WRITE) == 0;
        // (x86-64: a locked OR written in assembly -- the compiler turns __atomic_fetch_or(p, 0) into a LOAD there; elsewhere a
        // compare-and-swap of the byte with itself, which no compiler may drop: it is a write whenever it succeeds)
#if defined(__x86_64__)
        auto poke = [](char* q) { __asm__ __volatile__("lock; orb $0, (%0)" : : "r"(q) : "memory", "cc"); };
#else
        auto poke = [](char* q) {
            char v = __atomic_load_n(q, __ATOMIC_RELAXED);
            while (!__atomic_compare_exchange_n(q, &v, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
        };
#endif
        if (!populated)
            for (size_t o = lo; o < hi; o += page) poke(p + o);
        if (hi > lo) { poke(p + lo); poke(p + hi - 1); }                 // the partial pages at either end
    };
    if (nt == 1) { touch(0, bytes); return GF_OK; }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + page - 1) / page * page;
    for (size_t k = 0; k < nt; ++k) {
        const size_t lo = k * per, hi = (k + 1 == nt || (k + 1) * per > bytes) ? bytes : (k + 1) * per;
        if (lo >= bytes) break;
        th.emplace_back(touch, lo, hi);
    }
    for (auto& t : th) t.join();
    return GF_OK;
}

// ABI 5: a result arena -- host memory registered with the runtime, so that the read-backs' DMA writes it directly (host_range_is_pinned)
int gf_host_register(void* buf, size_t bytes)
{
    if (!buf || bytes == 0) return GF_ERR_INVALID_ARG;
    GF_HIP(hipHostRegister(buf, bytes, hipHostRegisterPortable));
    return GF_OK;
}
int gf_host_unregister(void* buf)
{
    if (!buf) return GF_ERR_INVALID_ARG;
    GF_HIP(hipHostUnregister(buf));
    return GF_OK;
}

int gf_model_sync(gf_model* m)
{
    if (!m) return GF_ERR_INVALID_ARG;
    if (!m->stream) return GF_OK;                               // no stream yet: nothing was ever enqueued
    GF_HIP(hipStreamSynchronize(m->stream));
    return check_queue_overflow(m->device, m->stream);          // of the *_device launches this call waited for
}

// internal, diagnostics (tools/): {pairs in the last arbitration launch, pairs arbitrated so far, launches so far} of the
// model's stream (wrapping 32-bit counters); synchronise first
int gf_internal_uni_stats(gf_model* m, unsigned int out[3])
{
    if (!m || !out) return GF_ERR_INVALID_ARG;
    out[0] = out[1] = out[2] = 0;
    if (!m->stream) return GF_OK;
    UniWork* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool[m->device].work.find(m->stream);
        if (it != g_pool[m->device].work.end()) w = it->second;
    }
    if (w && w->h_seen) { out[0] = w->h_seen[0]; out[1] = w->h_seen[3]; out[2] = w->h_seen[4]; }
    return GF_OK;
}

// internal, diagnostics (tools/arb_probe.py): the items of the last arbitration launch on the model's stream, still in the
// queue's memory after the kernel re-armed it: items_out[min(count, max)][2] = (walker, mask of undecided bins)
int gf_internal_uni_dump(gf_model* m, unsigned long long* items_out, unsigned int max, unsigned int* count)
{
    if (!m || !items_out || !count || !m->stream) return GF_ERR_INVALID_ARG;
    UniWork* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool[m->device].work.find(m->stream);
        if (it != g_pool[m->device].work.end()) w = it->second;
    }
    if (!w || !w->h_seen || !w->d_uq) return GF_ERR_INVALID_ARG;
    GF_HIP(hipStreamSynchronize(m->stream));
    unsigned int n = w->h_seen[0];
    *count = n;
    if (n > max) n = max;
    if (n > (unsigned int)w->uq_cap) n = (unsigned int)w->uq_cap;
    if (n) {
        GF_HIP(hipMemcpyAsync(items_out, w->d_uq->items, sizeof(GfArbItem) * n, hipMemcpyDeviceToHost, m->stream));
        GF_HIP(hipStreamSynchronize(m->stream));
    }
    return GF_OK;
}

// internal, test hook (tests/test_gpu_unitarity_r3.py): the emulated-x87 unitarity residual (fr.py:489-494) of explicit
// (walker, bin) pairs of a device-resident theta block; which = 0: the serial chain of gf_x87.hpp (one lane per pair), 1: its
// three-lane distribution (what k_uni_resolve runs).  All pointers are device pointers; synchronous.
int gf_internal_uni_residuals(gf_model* m, const double* d_theta, int layout, int64_t n, const int64_t* d_walkers, const int32_t* d_bins,
                              int64_t npairs, int which, double* d_out)
{
    if (!m || !d_theta || !d_walkers || !d_bins || !d_out || m->c.mode != GF_MODE_BSM_GAUSS) return GF_ERR_INVALID_ARG;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    const hipError_t e = gf_launch_uni_debug(m->d_common, m->d_bsm, d_theta, layout, n, d_walkers, d_bins, npairs, which, d_out, m->stream);
    if (e != hipSuccess) return hip_fail(e, "uni debug launch");
    GF_HIP(hipStreamSynchronize(m->stream));
    return GF_OK;
}

// internal (gf_sampler.hip): the same report for launches the sampler put on `stream` and has just synchronised
int gf_internal_check_overflow(int device, void* stream)
{
    if (device < 0 || device >= POOL_MAX_DEVICES) return GF_ERR_INVALID_ARG;
    return check_queue_overflow(device, (hipStream_t)stream);
}

// Release what the library keeps cached on `device` between uses: the unitarity workspaces (arbitration queue, walker queue,
// side buffer: up to 8 GiB in all) of pooled, idle streams, and the pooled constant blocks.  Streams in use keep theirs.
// *released_bytes (may be NULL): device memory handed back.  For long-lived processes that ran one large scan and go on
// with small work.
int gf_device_trim(int device, size_t* released_bytes)
{
    int cus = 0;
    const int rc = pool_device(device, &cus);
    if (rc != GF_OK) return rc;
    GF_HIP(hipSetDevice(device));
    // the idle streams leave the pool while their workspaces are released (nobody can pick one up half-way) and return after
    std::vector<hipStream_t> streams;
    std::vector<UniWork*> idle;
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        DevicePool& dp = g_pool[device];
        streams.swap(dp.streams);
        for (hipStream_t st : streams) {
            auto it = dp.work.find(st);
            if (it != dp.work.end() && it->second) idle.push_back(it->second);
        }
        blocks.swap(dp.blocks);
    }
    size_t total = gf_devcache_trim(device);            // the cached large buffers (gf_devcache.h) go back to the driver too
    for (UniWork* w : idle) {
        std::lock_guard<std::mutex> lk(w->mu);
        total += w->bytes();
        w->release();
    }
    for (void* b : blocks) { (void)hipFree(b); total += CONST_BLOCK_BYTES; }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        DevicePool& dp = g_pool[device];
        for (hipStream_t st : streams) dp.streams.push_back(st);
    }
    // the pinned staging slots of large device-to-host reads (host memory: not part of released_bytes, which counts device
    // memory); a read in progress keeps them
    {
        D2HRing& R = g_d2h[device];
        std::unique_lock<std::mutex> lk(R.mu, std::try_to_lock);
        if (lk.owns_lock())
            for (int k = 0; k < D2H_SLOTS; ++k) {
                if (R.slot[k]) { (void)hipHostFree(R.slot[k]); R.slot[k] = nullptr; }
                if (R.ev[k]) { (void)hipEventDestroy(R.ev[k]); R.ev[k] = nullptr; }
            }
    }
    if (released_bytes) *released_bytes = total;
    return GF_OK;
}

int gf_event_create(void** ev)
{
    if (!ev) return GF_ERR_INVALID_ARG;
    hipEvent_t e;
    GF_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return GF_OK;
}

int gf_event_destroy(void* ev)
{
    if (!ev) return GF_OK;
    GF_HIP(hipEventDestroy((hipEvent_t)ev));
    return GF_OK;
}

int gf_event_record(gf_model* m, void* ev)
{
    if (!m || !ev) return GF_ERR_INVALID_ARG;
    GF_HIP(hipSetDevice(m->device));
    GF_STREAM(m);
    GF_HIP(hipEventRecord((hipEvent_t)ev, m->stream));
    return GF_OK;
}

int gf_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms)
{
    if (!ev_start || !ev_stop || !ms) return GF_ERR_INVALID_ARG;
    GF_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    GF_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return GF_OK;
}

}  // extern "C"
