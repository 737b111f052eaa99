// Which HIP streams of one process share a hardware queue: work on two streams that share one runs in order, never side by side.
// Creates streams the way the library does (one, then an allocation and a memset, then five more), occupies stream i with a spinning
// kernel or a large copy from page-locked memory, and times a one-lane kernel put on stream j right after it.
// Build: hipcc --offload-arch=gfx950 -O2 -o stream_queues stream_queues.hip ; run: ./stream_queues [nstreams]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(long long cycles, int* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) *out = 1;
}
__global__ void tiny(int* out) { *out = 2; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 6;
    std::vector<hipStream_t> s(n);
    int* d = nullptr;
    (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
    (void)hipMalloc(&d, 64);
    (void)hipMemset(d, 0, 64);
    for (int i = 1; i < n; i++) (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    const size_t bytes = 128u << 20;
    void *h = nullptr, *dbuf = nullptr;
    (void)hipHostMalloc(&h, bytes, hipHostMallocDefault);
    (void)hipMalloc(&dbuf, bytes);
    hipEvent_t ev;
    (void)hipEventCreate(&ev);
    const long long spin_cycles = 200000;   // wall_clock64 counts at 100 MHz: 2 ms
    for (int mode = 0; mode < 2; mode++) {
        std::printf("%s on stream i, then a one-lane kernel on stream j: ms until j's kernel is done (rows i, columns j)\n",
                    mode == 0 ? "2 ms kernel" : "128 MB copy to the device");
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) {
                (void)hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                if (mode == 0) spin<<<1, 64, 0, s[i]>>>(spin_cycles, d);
                else (void)hipMemcpyAsync(dbuf, h, bytes, hipMemcpyHostToDevice, s[i]);
                tiny<<<1, 64, 0, s[j]>>>(d + 1);
                (void)hipEventRecord(ev, s[j]);
                (void)hipEventSynchronize(ev);
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                std::printf(" %6.2f", ms);
            }
            std::printf("\n");
        }
    }
    // two copies at once: on streams i and j, total time
    std::printf("two 128 MB copies to the device, one on stream 0 and one on stream j: ms for both\n");
    void* dbuf2 = nullptr; void* h2 = nullptr;
    (void)hipMalloc(&dbuf2, bytes);
    (void)hipHostMalloc(&h2, bytes, hipHostMallocDefault);
    for (int j = 0; j < n; j++) {
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        (void)hipMemcpyAsync(dbuf, h, bytes, hipMemcpyHostToDevice, s[0]);
        (void)hipMemcpyAsync(dbuf2, h2, bytes, hipMemcpyHostToDevice, s[j]);
        (void)hipDeviceSynchronize();
        std::printf(" %6.2f", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::printf("\n");
    return 0;
}
