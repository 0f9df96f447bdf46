// gf_devcache.hip -- see gf_devcache.h
#define GF_DEVCACHE_IMPL
#include "gf_devcache.h"

#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

extern "C" const char* gf_internal_env(const char* name, int affects_results);

namespace {
constexpr size_t GF_DEVCACHE_MIN = (size_t)64 << 20;        // smaller buffers are freed as before (their wipe takes microseconds)
struct Block { void* ptr; size_t bytes; int device; unsigned long long age; };
std::mutex g_mu;
std::vector<Block> g_idle;                                  // freed by the library, still allocated
std::unordered_map<void*, Block> g_live;                    // large blocks handed out (so that the free knows their size)
unsigned long long g_clock = 0, g_reuses = 0;

size_t cap_bytes()
{
    static const size_t cap = [] {
        const char* v = gf_internal_env("GF_DEVICE_CACHE_GB", 0);
        const double gb = v ? std::atof(v) : 96.0;
        return gb <= 0.0 ? (size_t)0 : (size_t)(gb * 1073741824.0);
    }();
    return cap;
}
}  // namespace

extern "C" hipError_t gf_cached_malloc(void** ptr, size_t bytes)
{
    if (!ptr) return hipErrorInvalidValue;
    if (bytes < GF_DEVCACHE_MIN || cap_bytes() == 0) return hipMalloc(ptr, bytes);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = 0; }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        size_t best = g_idle.size();
        for (size_t i = 0; i < g_idle.size(); ++i) {
            const Block& b = g_idle[i];
            if (b.device != device || b.bytes < bytes || b.bytes > bytes + bytes / 4) continue;
            if (best == g_idle.size() || b.bytes < g_idle[best].bytes) best = i;
        }
        if (best != g_idle.size()) {
            Block b = g_idle[best];
            g_idle.erase(g_idle.begin() + (long)best);
            *ptr = b.ptr;
            g_live[b.ptr] = b;
            ++g_reuses;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) {
        // out of memory with idle blocks in the cache: hand them back and try once more
        (void)hipGetLastError();
        if (gf_devcache_trim(device) == 0) return e;
        e = hipMalloc(ptr, bytes);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[*ptr] = Block{*ptr, bytes, device, 0};
    return hipSuccess;
}

extern "C" hipError_t gf_cached_free(void* ptr)
{
    if (!ptr) return hipSuccess;
    Block b{};
    bool large = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(ptr);
        if (it != g_live.end()) { b = it->second; g_live.erase(it); large = true; }
    }
    if (!large) return hipFree(ptr);
    // what hipFree guarantees its caller: nothing on the device still uses the block when the call returns
    int cur = 0;
    const bool other = hipGetDevice(&cur) == hipSuccess && cur != b.device;
    if (other) (void)hipSetDevice(b.device);
    const hipError_t es = hipDeviceSynchronize();
    if (other) (void)hipSetDevice(cur);
    if (es != hipSuccess) { (void)hipFree(ptr); return es; }
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        b.age = ++g_clock;
        g_idle.push_back(b);
        size_t total = 0;
        for (const Block& x : g_idle) total += x.bytes;
        while (total > cap_bytes() && !g_idle.empty()) {             // over the cap: the block that has been idle longest goes
            size_t old = 0;
            for (size_t i = 1; i < g_idle.size(); ++i) if (g_idle[i].age < g_idle[old].age) old = i;
            total -= g_idle[old].bytes;
            drop.push_back(g_idle[old].ptr);
            g_idle.erase(g_idle.begin() + (long)old);
        }
    }
    for (void* p : drop) (void)hipFree(p);
    return hipSuccess;
}

extern "C" size_t gf_devcache_trim(int device)
{
    std::vector<Block> drop;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_idle.size();) {
            if (device < 0 || g_idle[i].device == device) { drop.push_back(g_idle[i]); g_idle.erase(g_idle.begin() + (long)i); }
            else ++i;
        }
    }
    size_t total = 0;
    for (const Block& b : drop) { (void)hipFree(b.ptr); total += b.bytes; }
    return total;
}

extern "C" void gf_devcache_stats(int device, size_t* idle_bytes, size_t* live_bytes, unsigned long long* reuses)
{
    std::lock_guard<std::mutex> lk(g_mu);
    size_t idle = 0, live = 0;
    for (const Block& b : g_idle) if (device < 0 || b.device == device) idle += b.bytes;
    for (const auto& kv : g_live) if (device < 0 || kv.second.device == device) live += kv.second.bytes;
    if (idle_bytes) *idle_bytes = idle;
    if (live_bytes) *live_bytes = live;
    if (reuses) *reuses = g_reuses;
}
