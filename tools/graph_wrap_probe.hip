// graph_wrap_probe -- does `rocprofv3 --kernel-trace` survive hipGraph replays once the stream's AQL queue has wrapped?
//
// Round 3 saw `rocprofv3 --kernel-trace -- python3 bench.py` die with SIGSEGV inside hipGraphLaunch (frames: libamdhip64 ->
// HSA queue intercept -> rocprofiler-sdk, a copy running off the end of a mapping at a 1 MiB-aligned address) about 35 graph
// replays into the C5 scan, only with graphs, only in the full command (tens of thousands of packets), never in shorter runs.
// 16 384 packets x 64 B = 1 MiB = one HSA ring of HIP's default size: the hypothesis is that a BATCH of graph packets which
// straddles the ring's wrap-around is handed to the profiler's interceptor as (pointer, count) and read linearly.
// This program replays a graph of K trivial kernels R times on one stream and logs its progress, so that
//   K = 65, R = 200  (13 000 packets: no wrap)            must pass,
//   K = 65, R = 300  (19 500 packets: one wrap)           is predicted to die near replay 16384 / (packets per replay),
//   the same with a stream synchronise after every replay separates "queue full" from "queue wrapped",
//   the same with ROC_AQL_QUEUE_SIZE=65536               is predicted to pass (no wrap within the run),
//   the same with "bound 8" (at most 8 replays in flight) is what gf_sampler_run does since round 4.
// Nothing of libgolemhip is involved: if this dies under the profiler and passes without it, the fault is not the product's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void k_touch(unsigned int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1u); }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv)
{
    const int K = argc > 1 ? std::atoi(argv[1]) : 65;
    const int R = argc > 2 ? std::atoi(argv[2]) : 300;
    const bool sync_each = argc > 3 && std::strcmp(argv[3], "sync") == 0;
    // "bound N": replay k is not enqueued before replay k - N has completed (an event per replay, a ring of N) -- what
    // gf_sampler_run does since round 4 (N = 8): the host never runs more than N replays ahead of the GPU
    const int bound = (argc > 4 && std::strcmp(argv[3], "bound") == 0) ? std::atoi(argv[4]) : 0;
    hipEvent_t ring[64] = {};
    // "event": an event is recorded behind every replay and never waited for; "copy N": bound N, and between two replays a
    // second stream carries a 1 MiB device-to-host copy and an event (what gf_sampler_run_to_host's read-back does)
    const bool event_only = argc > 3 && std::strcmp(argv[3], "event") == 0;
    const int copy_bound = (argc > 4 && std::strcmp(argv[3], "copy") == 0) ? std::atoi(argv[4]) : 0;
    hipStream_t st2 = nullptr;
    hipEvent_t ev2 = nullptr, ev1 = nullptr;
    void *pinned = nullptr, *dbig = nullptr;
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned int* d = nullptr;
    CHECK(hipMalloc((void**)&d, sizeof(unsigned int)));
    CHECK(hipMemsetAsync(d, 0, sizeof(unsigned int), st));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, d);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const int nb = bound > 0 ? bound : copy_bound;
    if (event_only) CHECK(hipEventCreateWithFlags(&ev1, hipEventDisableTiming));
    if (copy_bound > 0) {
        CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        CHECK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
        CHECK(hipHostMalloc(&pinned, 1 << 20, hipHostMallocDefault));
        CHECK(hipMalloc(&dbig, 1 << 20));
    }
    std::fprintf(stderr, "graph of %d kernels, %d replays%s\n", K, R, sync_each ? ", synchronised after each" : "");
    for (int r = 0; r < R; ++r) {
        std::fprintf(stderr, "replay %d (kernels enqueued so far %lld)\n", r, (long long)r * K);
        std::fflush(stderr);
        if (nb > 0 && nb <= 64) {
            hipEvent_t& e = ring[r % nb];
            if (e) CHECK(hipEventSynchronize(e));                    // replay r - bound has completed
            else CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        CHECK(hipGraphLaunch(ge, st));
        if (nb > 0 && nb <= 64) CHECK(hipEventRecord(ring[r % nb], st));
        if (event_only) CHECK(hipEventRecord(ev1, st));
        if (copy_bound > 0) {
            CHECK(hipMemcpy2DAsync(pinned, 4096, dbig, 8192, 4096, 128, hipMemcpyDeviceToHost, st2));
            CHECK(hipEventRecord(ev2, st2));
            if (r % 3 == 2) CHECK(hipEventSynchronize(ev2));
        }
        if (sync_each) CHECK(hipStreamSynchronize(st));
    }
    CHECK(hipStreamSynchronize(st));
    unsigned int h = 0;
    CHECK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("done: %u kernel executions (expected %lld)\n", h, (long long)R * K);
    return h == (unsigned int)((long long)R * K) ? 0 : 1;
}
