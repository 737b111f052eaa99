// Checks v_pk_minimum3_f16 / v_pk_maximum3_f16 on pixel values held as 16-bit integers (denormal half floats) against integer
// minimum / maximum: 65536 random triples of packed pairs. Build: hipcc --offload-arch=gfx950 -O2 -o pk_min3_check pk_min3_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t pkmin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pkmax3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__global__ void k(const uint32_t* in, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
    out[2 * i] = pkmin3(a, b, c);
    out[2 * i + 1] = pkmax3(a, b, c);
}
int main() {
    const int n = 1 << 16;
    uint32_t* h = new uint32_t[3 * n]; uint32_t* o = new uint32_t[2 * n];
    uint32_t s = 12345;
    for (int i = 0; i < 3 * n; i++) { s = s * 1664525u + 1013904223u; uint32_t lo = (s >> 8) & 0x1FF, hi = (s >> 20) & 0x1FF; if (i % 7 == 0) lo = 0; if (i % 11 == 0) hi = 255; h[i] = lo | (hi << 16); }
    uint32_t *di, *dout; hipMalloc(&di, 12 * n); hipMalloc(&dout, 8 * n);
    hipMemcpy(di, h, 12 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, di, dout, n);
    hipMemcpy(o, dout, 8 * n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        uint32_t a = h[3 * i], b = h[3 * i + 1], c = h[3 * i + 2];
        auto mn = [](uint32_t x, uint32_t y) { return x < y ? x : y; }; auto mx = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
        uint32_t emin = mn(mn(a & 0xFFFF, b & 0xFFFF), c & 0xFFFF) | (mn(mn(a >> 16, b >> 16), c >> 16) << 16);
        uint32_t emax = mx(mx(a & 0xFFFF, b & 0xFFFF), c & 0xFFFF) | (mx(mx(a >> 16, b >> 16), c >> 16) << 16);
        if (o[2 * i] != emin || o[2 * i + 1] != emax) { if (bad < 5) printf("i %d: %08x %08x %08x -> %08x %08x want %08x %08x\n", i, a, b, c, o[2*i], o[2*i+1], emin, emax); bad++; }
    }
    printf("pk min3/max3 f16 on integer patterns 0..511: %d mismatches of %d\n", bad, n);
    return bad != 0;
}
