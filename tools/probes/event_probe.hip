// event_probe.hip -- what do hipEventQuery / hipEventElapsedTime return, and what do they leave in hipGetLastError,
// in the pattern the filter's statistics ring uses?  (development probe for the round-2 'invalid resource handle')
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <thread>
#include <atomic>
#include <vector>
__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
struct Slot { hipEvent_t ev[2], copied; unsigned* host; };
static std::map<int, long> tally_q, tally_e, tally_last;
int main(int argc, char** argv) {
    const int use_null = argc > 1 ? atoi(argv[1]) : 0;
    const unsigned evflags = argc > 2 ? (unsigned)strtoul(argv[2], 0, 0) : hipEventDisableSystemFence;
    const int with_thread = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t s = nullptr, side = nullptr;
    if (!use_null) hipStreamCreate(&s);
    hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    float *d, *d2; hipMalloc(&d, 4096); hipMalloc(&d2, 4096); hipMemset(d, 0, 4096); hipMemset(d2, 0, 4096);
    unsigned* dm; hipMalloc(&dm, 512);
    unsigned* hm; hipHostMalloc((void**)&hm, 65 * 512, hipHostMallocDefault);
    // semantics first
    {
        hipEvent_t a, b, c; float ms;
        hipEventCreateWithFlags(&a, evflags); hipEventCreateWithFlags(&b, evflags); hipEventCreateWithFlags(&c, hipEventDisableTiming);
        hipError_t e = hipEventElapsedTime(&ms, a, b);
        printf("elapsed(never recorded) = %d (%s); sticky after = %d\n", (int)e, hipGetErrorString(e), (int)hipGetLastError());
        hipEventRecord(a, s); spin<<<1, 64, 0, s>>>(d, 1 << 22); hipEventRecord(b, s);
        e = hipEventElapsedTime(&ms, a, b);
        printf("elapsed(in flight) = %d (%s); sticky after = %d\n", (int)e, hipGetErrorString(e), (int)hipGetLastError());
        e = hipEventQuery(b);
        printf("query(in flight) = %d (%s); sticky after = %d\n", (int)e, hipGetErrorString(e), (int)hipGetLastError());
        hipStreamSynchronize(s);
        e = hipEventElapsedTime(&ms, a, b);
        printf("elapsed(done) = %d ms=%f\n", (int)e, ms);
        hipEventRecord(c, s);
        e = hipEventElapsedTime(&ms, a, c);
        printf("elapsed(a, disable-timing) = %d (%s); sticky after = %d\n", (int)e, hipGetErrorString(e), (int)hipGetLastError());
        // a successful call in between: does it clear the sticky error?
        (void)hipEventElapsedTime(&ms, a, c);
        hipDeviceSynchronize();
        printf("sticky after a later successful call = %d\n", (int)hipGetLastError());
    }
    Slot ring[65];
    for (int i = 0; i < 65; i++) {
        hipEventCreateWithFlags(&ring[i].copied, hipEventDisableTiming);
        hipEventCreateWithFlags(&ring[i].ev[0], evflags);
        hipEventCreateWithFlags(&ring[i].ev[1], evflags);
        ring[i].host = hm + i * 128;
    }
    std::atomic<int> stop{0};
    std::thread th;
    if (with_thread) th = std::thread([&] {   // another host thread busy with its own stream and events
        hipStream_t t; hipStreamCreateWithFlags(&t, hipStreamNonBlocking);
        hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
        while (!stop.load()) { spin<<<1, 64, 0, t>>>(d2, 1000); hipEventRecord(e, t); hipEventSynchronize(e); }
    });
    int head = 0, count = 0; long bad = 0;
    (void)hipGetLastError();
    for (int it = 0; it < 20000; it++) {
        // resolve (poll)
        while (count > 0) {
            Slot& fs = ring[head];
            hipError_t q = hipEventQuery(fs.copied);
            tally_q[(int)q]++;
            if (q == hipErrorNotReady) break;
            float ms = 0;
            hipError_t e = hipEventElapsedTime(&ms, fs.ev[0], fs.ev[1]);
            tally_e[(int)e]++;
            if (e != hipSuccess && bad++ < 10) printf("it %d slot %d: elapsed -> %d (%s)\n", it, head, (int)e, hipGetErrorString(e));
            head = (head + 1) % 64; count--;
        }
        if (count == 64) { hipEventSynchronize(ring[head].copied); continue; }
        Slot& fs = ring[(head + count) % 64];
        hipMemsetAsync(dm, 0, 512, s);
        hipEventRecord(fs.ev[0], s);
        spin<<<64, 64, 0, s>>>(d, 200 + (it % 7) * 300);
        hipEventRecord(fs.ev[1], s);
        // something beside on another stream, joined by events (as the accumulation's side stream is)
        hipEventRecord(ring[64].copied, s); hipStreamWaitEvent(side, ring[64].copied, 0);
        spin<<<1, 64, 0, side>>>(d2, 500);
        spin<<<1, 64, 0, s>>>(d, 100);
        hipMemcpyAsync(fs.host, dm, 512, hipMemcpyDeviceToHost, s);
        hipEventRecord(fs.copied, s);
        count++;
        hipError_t le = hipGetLastError();
        tally_last[(int)le]++;
    }
    hipDeviceSynchronize();
    stop = 1; if (with_thread) th.join();
    for (auto& kv : tally_q) printf("query      code %d: %ld\n", kv.first, kv.second);
    for (auto& kv : tally_e) printf("elapsed    code %d: %ld\n", kv.first, kv.second);
    for (auto& kv : tally_last) printf("last-error code %d: %ld\n", kv.first, kv.second);
    return 0;
}
