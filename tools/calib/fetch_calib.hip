// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for the access widths the
// library's kernels use: streaming reads of 1 GiB with 1, 4 and 16 bytes per lane and a 4-byte
// per-lane streaming write. Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
// Run under: rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__device__ __forceinline__ void k_read(const T* __restrict__ p, size_t n, unsigned* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) {
        const T v = p[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
        for (unsigned k = 0; k < sizeof(T); k += (sizeof(T) >= 4 ? 4 : 1)) acc += b[k];
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read_u8(const uint8_t* __restrict__ p, size_t n, unsigned* __restrict__ sink) { k_read<uint8_t>(p, n, sink); }
__global__ void k_read_u32(const uint32_t* __restrict__ p, size_t n, unsigned* __restrict__ sink) { k_read<uint32_t>(p, n, sink); }
__global__ void k_read_u128(const uint4* __restrict__ p, size_t n, unsigned* __restrict__ sink) { k_read<uint4>(p, n, sink); }
__global__ void k_write_u32(uint32_t* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (uint32_t)i;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    uint8_t* buf = nullptr;
    unsigned* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    const int blocks = 256 * 16;
    hipLaunchKernelGGL(k_read_u8, dim3(blocks), dim3(256), 0, 0, buf, bytes, sink);
    hipLaunchKernelGGL(k_read_u32, dim3(blocks), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_read_u128, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(k_write_u32, dim3(blocks), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::printf("calibration kernels ran over %zu bytes each\n", bytes);
    return 0;
}
