// x87_int_probe -- would an INTEGER representation of the emulated x87 format (64-bit significand in a u64, exponent and sign in two
// 32-bit words) beat the double-double one of gf_x87.hpp (every + - * in ~25 fp64 instructions, each waiting for the one before)?
// The arbitration kernels are bound by exactly that chain: one (walker, bin) is ~10 000 dependent instructions on nine lanes.
// This probe implements x * y and x + y in integers, checks them bit for bit against gf_x87.hpp's on random operands (ties,
// cancellations and far-apart exponents included), and times a dependent chain x <- x * a + b in both forms: cycles per
// (mul + add) on ONE wave (latency) and on a full GPU (throughput).
//   hipcc -O3 --offload-arch=gfx950 -I golemflavor_amd/csrc tools/experiments/x87_int_probe.hip -o tools/x87_int_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#define GFX87_INLINE_ALL
#include "gf_x87.hpp"
using namespace gfx87;

struct xi { uint64_t m; int32_t e; uint32_t s; };        // value = (-1)^s m 2^(e - 63), bit 63 of m set; m == 0: zero

__host__ __device__ inline xi xi_round128(uint64_t hi, uint64_t lo, int e, uint32_t s)   // hi has bit 63 set (or hi == lo == 0)
{
    const uint64_t guard = lo >> 63;
    const uint64_t sticky = (lo << 1) != 0 ? 1ull : 0ull;
    hi += guard & (sticky | (hi & 1ull));
    const bool carry = hi == 0 && (guard != 0);            // 0xffff... + 1
    xi r;
    r.m = carry ? (1ull << 63) : hi;
    r.e = e + (carry ? 1 : 0);
    r.s = s;
    return r;
}
__device__ inline xi xi_mul(xi a, xi b)
{
    uint64_t hi = __umul64hi(a.m, b.m), lo = a.m * b.m;
    int e = a.e + b.e + 1;
    const bool top = (hi >> 63) != 0;
    const uint64_t hi2 = (hi << 1) | (lo >> 63), lo2 = lo << 1;
    hi = top ? hi : hi2; lo = top ? lo : lo2; e -= top ? 0 : 1;
    xi r = xi_round128(hi, lo, e, a.s ^ b.s);
    const bool zero = a.m == 0 || b.m == 0;
    r.m = zero ? 0 : r.m; r.e = zero ? 0 : r.e;
    return r;
}
__device__ inline xi xi_add(xi a, xi b)
{
    // |a| >= |b|
    const bool swap = (b.m != 0) && (a.m == 0 || b.e > a.e || (b.e == a.e && b.m > a.m));
    const xi x = swap ? b : a, y = swap ? a : b;
    const unsigned d = (unsigned)(x.e - y.e);
    // y's significand shifted right by d into a 128-bit window below x's, what falls off collected in the lowest bit
    uint64_t yh, yl;
    if (d == 0) { yh = y.m; yl = 0; }
    else if (d < 64) { yh = y.m >> d; yl = y.m << (64 - d); }
    else if (d == 64) { yh = 0; yl = y.m; }
    else if (d < 128) { yh = 0; yl = (y.m >> (d - 64)) | ((y.m << (128 - d)) != 0 ? 1ull : 0ull); }
    else { yh = 0; yl = y.m != 0 ? 1ull : 0ull; }
    uint64_t hi, lo; int e = x.e;
    if (x.s == y.s) {
        lo = yl; hi = x.m + yh;
        const bool carry = hi < x.m;
        if (carry) { lo = (lo >> 1) | (lo & 1ull) | (hi << 63); hi = (hi >> 1) | (1ull << 63); e += 1; }
    } else {
        lo = 0 - yl; hi = x.m - yh - (yl != 0 ? 1ull : 0ull);
        // normalise
        if (hi == 0 && lo == 0) { xi z = {0, 0, 0}; return z; }
        int lz = hi != 0 ? __clzll((long long)hi) : 64 + __clzll((long long)lo);
        if (lz >= 64) { hi = lo << (lz - 64); lo = 0; }
        else if (lz > 0) { hi = (hi << lz) | (lo >> (64 - lz)); lo <<= lz; }
        e -= lz;
    }
    xi r = xi_round128(hi, lo, e, x.s);
    if (y.m == 0) r = x;
    return r;
}

// ---- host conversions ---------------------------------------------------------------------------------------
static xi to_xi(long double v)
{
    xi r = {0, 0, 0};
    if (v == 0) return r;
    r.s = std::signbit(v) ? 1u : 0u;
    int ex; long double f = frexpl(fabsl(v), &ex);          // f in [0.5, 1)
    r.m = (uint64_t)ldexpl(f, 64);
    r.e = ex - 1;
    return r;
}
static long double from_xi(xi a) { long double v = ldexpl((long double)a.m, a.e - 63); return a.s ? -v : v; }
static x87 to_dd(long double v) { x87 r; r.hi = (double)v; r.lo = (double)(v - (long double)r.hi); return r; }
static long double from_dd(x87 a) { return (long double)a.hi + (long double)a.lo; }

__global__ void k_check(const xi* a, const xi* b, const x87* da, const x87* db, xi* pm, xi* ps, x87* dm, x87* ds, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pm[i] = xi_mul(a[i], b[i]); ps[i] = xi_add(a[i], b[i]);
    dm[i] = x_mul(da[i], db[i]); ds[i] = x_add(da[i], db[i]);
}
template <int WHICH>
__global__ void k_chain(const xi* a, const x87* da, int iters, long long* cycles, double* sink)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const long long t0 = clock64();
    double out;
    if (WHICH == 0) {
        xi x = a[i], p = a[i + 1], q = a[i + 2];
        for (int k = 0; k < iters; ++k) x = xi_add(xi_mul(x, p), q);
        out = (double)x.m + x.e;
    } else {
        x87 x = da[i], p = da[i + 1], q = da[i + 2];
        for (int k = 0; k < iters; ++k) x = x_add(x_mul(x, p), q);
        out = x.hi + x.lo;
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[i] = out;
}

int main()
{
    const int n = 1 << 20;
    std::mt19937_64 rng(7);
    std::vector<xi> a(n + 4), b(n + 4); std::vector<x87> da(n + 4), db(n + 4);
    for (int i = 0; i < n + 4; ++i) {
        auto draw = [&](int kind) -> long double {
            uint64_t m = rng() | (1ull << 63);
            if (kind == 1) m &= ~0x7ffull;                   // exactly a double: ties in the product / sum
            if (kind == 2) m = (m & ~0xffffffffull);          // short significand: exact results
            int e = (int)(rng() % 41) - 20;
            long double v = ldexpl((long double)m, e - 63);
            return (rng() & 1) ? -v : v;
        };
        long double x = draw(i % 4), y = draw((i / 4) % 4);
        if (i % 7 == 0) y = -x * (1.0L + ldexpl((long double)(int)(rng() % 5) - 2, -62));     // cancellation to the last bits
        if (i % 11 == 0) y = ldexpl(y, -70 - (int)(rng() % 80));                                  // far-apart exponents
        if (i % 13 == 0) y = ldexpl(1.0L, (int)(rng() % 30) - 70) * (x > 0 ? 1 : -1) * ldexpl(fabsl(x), 0) * 0 + ldexpl(x, -64) * ((rng() & 1) ? 1 : -1);   // exactly half an ulp: ties
        a[i] = to_xi(x); b[i] = to_xi(y); da[i] = to_dd(x); db[i] = to_dd(y);
    }
    xi *ga, *gb, *gpm, *gps; x87 *gda, *gdb, *gdm, *gds;
    hipMalloc(&ga, sizeof(xi) * (n + 4)); hipMalloc(&gb, sizeof(xi) * (n + 4)); hipMalloc(&gpm, sizeof(xi) * n); hipMalloc(&gps, sizeof(xi) * n);
    hipMalloc(&gda, sizeof(x87) * (n + 4)); hipMalloc(&gdb, sizeof(x87) * (n + 4)); hipMalloc(&gdm, sizeof(x87) * n); hipMalloc(&gds, sizeof(x87) * n);
    hipMemcpy(ga, a.data(), sizeof(xi) * (n + 4), hipMemcpyHostToDevice); hipMemcpy(gb, b.data(), sizeof(xi) * (n + 4), hipMemcpyHostToDevice);
    hipMemcpy(gda, da.data(), sizeof(x87) * (n + 4), hipMemcpyHostToDevice); hipMemcpy(gdb, db.data(), sizeof(x87) * (n + 4), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, ga, gb, gda, gdb, gpm, gps, gdm, gds, n);
    std::vector<xi> pm(n), ps(n); std::vector<x87> dm(n), ds(n);
    hipMemcpy(pm.data(), gpm, sizeof(xi) * n, hipMemcpyDeviceToHost); hipMemcpy(ps.data(), gps, sizeof(xi) * n, hipMemcpyDeviceToHost);
    hipMemcpy(dm.data(), gdm, sizeof(x87) * n, hipMemcpyDeviceToHost); hipMemcpy(ds.data(), gds, sizeof(x87) * n, hipMemcpyDeviceToHost);
    long bad_m = 0, bad_s = 0, bad_ref_m = 0, bad_ref_s = 0;
    for (int i = 0; i < n; ++i) {
        const long double x = from_xi(a[i]), y = from_xi(b[i]);
        volatile long double pr = x * y, sr = x + y;          // the CPU's x87 unit (64-bit significand, nearest even)
        if (from_xi(pm[i]) != from_dd(dm[i])) ++bad_m;
        if (from_xi(ps[i]) != from_dd(ds[i])) ++bad_s;
        if (from_xi(pm[i]) != pr) ++bad_ref_m;
        if (from_xi(ps[i]) != sr) ++bad_ref_s;
    }
    std::printf("operand pairs %d: integer vs double-double  mul %ld  add %ld differ;  integer vs the CPU's x87 unit  mul %ld  add %ld differ\n", n, bad_m, bad_s, bad_ref_m, bad_ref_s);
    // timing
    long long* gc; double* gsink; hipMalloc(&gc, sizeof(long long) * 8192); hipMalloc(&gsink, sizeof(double) * (8192 * 256));
    for (int which = 0; which < 2; ++which)
        for (int blocks : {1, 2048, 8192}) {
            const int iters = 2000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (which == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(64), 0, 0, ga, gda, iters, gc, gsink);
                else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(64), 0, 0, ga, gda, iters, gc, gsink);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long c0 = 0; hipMemcpy(&c0, gc, sizeof(c0), hipMemcpyDeviceToHost);
            std::printf("%-14s %5d wave(s): %7.1f shader-clock cycles per (mul + add) on wave 0, %8.3f ms per launch = %.3g (mul + add) per second\n",
                        which == 0 ? "integer" : "double-double", blocks, (double)c0 / iters, ms, (double)blocks * 64 * iters / (ms * 1e-3));
        }
    return (bad_m || bad_s || bad_ref_m || bad_ref_s) ? 1 : 0;
}
