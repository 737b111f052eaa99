// Calibration of the vector-ALU issue rate of gfx950 for the instruction kinds the library's kernels are
// made of (and the ones they could be made of): how many cycles a SIMD needs per wave64 instruction when
// 1, 2, 4 or 8 waves per SIMD issue long runs of independent instructions. Each wave runs ITERS trips of
// 16 instructions on 16 different accumulators (an accumulator is touched again 16 instructions later:
// no dependency stall) and stamps the shader clock (s_memtime) and the 100 MHz constant clock
// (s_memrealtime) before and after. A workgroup is 4 waves = one per SIMD; its LDS allocation is sized
// so that exactly W workgroups fit a CU and the grid is 256 * W workgroups, so every SIMD of the chip
// holds W waves for the whole run (a run lasts milliseconds: the waves of a launch overlap almost fully).
//
// Output: one line per (instruction, W): cycles per wave-instruction per SIMD = median over waves of
// (shader clock delta) / (W * ITERS * 16), the same from the wall time of the launch and the clock the
// waves saw (shader clock delta / constant clock delta).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_calib valu_calib.hip      Run: ./valu_calib
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 8192
#define NOPS 71
static const char* const kOpNames[NOPS] = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32 const", "v_lshrrev_b32 vgpr", "v_lshlrev_b32 const", "v_min_i32", "v_max_u32", "v_min_u16", "v_max_i16", "v_sub_u16", "v_add_u16", "v_cndmask_b32", "v_mul_u32_u24", "v_mul_lo_u32", "v_mad_u32_u24", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_add_u32 dpp row_shr:1", "v_mov_b32 dpp row_shr:1", "v_mov_b32 dpp wave_shr:1", "v_add_u32 sdwa WORD_1", "v_sub_u16 sdwa hi", "v_pk_min_i16", "v_pk_max_i16", "v_pk_sub_i16", "v_pk_add_u16", "v_pk_lshrrev_b16", "v_perm_b32", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_lshl_or_b32", "v_lshl_add_u32", "v_bfe_u32", "v_bfi_b32", "v_alignbit_b32", "v_alignbyte_b32", "v_min3_i32", "v_med3_i32", "v_sad_u8", "v_sad_u16", "v_msad_u8", "v_dot4_u32_u8", "v_dot2_u32_u16", "v_bcnt_u32_b32", "v_mbcnt_lo_u32_b32", "v_cmp_gt_i32 (vcc)", "v_cvt_f32_i32", "v_rndne_f32", "v_ffbl_b32", "v_add_u32 (2 fresh srcs)", "v_sub_u32 (2 fresh srcs)", "v_or_b32 (2 fresh srcs)", "v_and_b32 (2 fresh srcs)", "v_add_u32 (src = sgpr)", "v_add_u32 (src = literal)", "v_lshrrev_b32 (fresh dst)", "v_mov_b32 (fresh)", "v_pk_min_i16 (2 fresh srcs)",
    "v_pk_minimum3_f16 (denormal operands)", "v_pk_maximum3_f16 (denormal operands)", "v_pk_min_f16 (denormal operands)", "v_pk_max_f16 (denormal operands)",
    "v_min3_i16", "v_max3_u16", "v_pk_minimum3_f16 (normal operands)"};

template <int OP>
__device__ __forceinline__ void one(uint32_t& a, uint32_t b, uint32_t c, uint32_t d, uint32_t sg) {
    if (OP < 0) {}
    else if (OP == 55) asm volatile("v_add_u32 %0, %3, %1" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 56) asm volatile("v_sub_u32 %0, %3, %1" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 57) asm volatile("v_or_b32 %0, %3, %1" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 58) asm volatile("v_and_b32 %0, %3, %1" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 59) asm volatile("v_add_u32 %0, %4, %0" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 60) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 61) asm volatile("v_lshrrev_b32 %0, 1, %3" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 62) asm volatile("v_mov_b32 %0, %3" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 63) asm volatile("v_pk_min_i16 %0, %3, %1" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 64 || OP == 70) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %3" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 65) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %3" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 66) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 67) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 68) asm volatile("v_min3_i16 %0, %0, %1, %3" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 69) asm volatile("v_max3_u16 %0, %0, %1, %3" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(sg) : "vcc");
    else if (OP == 0) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 2) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 4) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 6) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 7) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 8) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 9) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 10) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 11) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 12) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 13) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 14) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 16) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 17) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 18) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 19) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 20) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 21) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 22) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 23) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 24) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 25) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 26) asm volatile("v_sub_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 27) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 28) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 29) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 30) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 31) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 33) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 34) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 35) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 36) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 37) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 38) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 39) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 40) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 41) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 42) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 43) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 44) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 45) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 46) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 47) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 48) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 49) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 50) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 51) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 52) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 53) asm volatile("v_rndne_f32 %0, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    else if (OP == 54) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
}

template <int OP>
__global__ __launch_bounds__(256) void k_calib(uint64_t* __restrict__ delta, uint32_t* __restrict__ sink, uint32_t seed) {
    extern __shared__ uint32_t lds_hold[];                     // only its size matters: it limits the workgroups per CU
    const uint32_t b = seed * 2654435761u + threadIdx.x, c = 0x0C010C00u ^ seed;
    uint64_t t0, t1, r0, r1;
    uint32_t a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        a[k] = b + k;
        if (OP >= 64 && OP <= 69) a[k] = (a[k] * 0x00010001u) & 0x00FF00FFu;            // pixel values in 16-bit halves: denormal half floats
        if (OP == 70) a[k] = ((a[k] * 0x00010001u) & 0x03FF03FFu) | 0x3C003C00u;      // half floats in [1, 2)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) one<OP>(a[k], a[(k + 5) & 15], c, a[(k + 9) & 15], seed);
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) x ^= a[k];
    if (x == 0x12345678u && seed == 77) *sink = x;
    if ((threadIdx.x & 63) == 0) {
        delta[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
        delta[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
    }
}

template <int OP>
static void run(int ncu, uint64_t* d_delta, uint32_t* d_sink) {
    for (int W : {1, 2, 4, 8}) {
        const int grid = ncu * W;
        const size_t lds = ((size_t)(160 * 1024 / W) - 1024) & ~(size_t)255;   // exactly W workgroups fit a CU's 160 KB
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_calib<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            std::printf("cannot set the LDS size\n"); return;
        }
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_calib<OP>, dim3(grid), dim3(256), lds, 0, d_delta, d_sink, 1u);   // warm-up
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_calib<OP>, dim3(grid), dim3(256), lds, 0, d_delta, d_sink, 2u);
        (void)hipEventRecord(e1, 0);
        if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h((size_t)grid * 8);
        (void)hipMemcpy(h.data(), d_delta, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (size_t i = 0; i < (size_t)grid * 4; i++) { cyc.push_back((double)h[2 * i]); ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double med = cyc[cyc.size() / 2], clk = ghz[ghz.size() / 2];
        const double n = (double)ITERS * 16;
        std::printf("%-26s W=%d  cycles/instr/SIMD %.2f (from wall time: %.2f)  wall %.3f ms  clock %.2f GHz\n",
                    kOpNames[OP], W, med / (W * n), ms * 1e6 * clk / (W * n), ms, clk);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
}

static bool g_only[NOPS];
static bool g_filter = false;
template <int OP>
static void run_if(int ncu, uint64_t* d_delta, uint32_t* d_sink) {
    if (!g_filter || g_only[OP]) run<OP>(ncu, d_delta, d_sink);
}

int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) { const int k = std::atoi(argv[i]); if (k >= 0 && k < NOPS) { g_only[k] = true; g_filter = true; } }   // run only the listed cases
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int ncu = prop.multiProcessorCount;
    std::printf("device %s, %d CUs, max clock %d MHz\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
    uint64_t* d_delta = nullptr; uint32_t* d_sink = nullptr;
    if (hipMalloc(&d_delta, (size_t)ncu * 8 * 4 * 16) != hipSuccess || hipMalloc(&d_sink, 4) != hipSuccess) return 1;
    run_if<0>(ncu, d_delta, d_sink);
    run_if<1>(ncu, d_delta, d_sink);
    run_if<2>(ncu, d_delta, d_sink);
    run_if<3>(ncu, d_delta, d_sink);
    run_if<4>(ncu, d_delta, d_sink);
    run_if<5>(ncu, d_delta, d_sink);
    run_if<6>(ncu, d_delta, d_sink);
    run_if<7>(ncu, d_delta, d_sink);
    run_if<8>(ncu, d_delta, d_sink);
    run_if<9>(ncu, d_delta, d_sink);
    run_if<10>(ncu, d_delta, d_sink);
    run_if<11>(ncu, d_delta, d_sink);
    run_if<12>(ncu, d_delta, d_sink);
    run_if<13>(ncu, d_delta, d_sink);
    run_if<14>(ncu, d_delta, d_sink);
    run_if<15>(ncu, d_delta, d_sink);
    run_if<16>(ncu, d_delta, d_sink);
    run_if<17>(ncu, d_delta, d_sink);
    run_if<18>(ncu, d_delta, d_sink);
    run_if<19>(ncu, d_delta, d_sink);
    run_if<20>(ncu, d_delta, d_sink);
    run_if<21>(ncu, d_delta, d_sink);
    run_if<22>(ncu, d_delta, d_sink);
    run_if<23>(ncu, d_delta, d_sink);
    run_if<24>(ncu, d_delta, d_sink);
    run_if<25>(ncu, d_delta, d_sink);
    run_if<26>(ncu, d_delta, d_sink);
    run_if<27>(ncu, d_delta, d_sink);
    run_if<28>(ncu, d_delta, d_sink);
    run_if<29>(ncu, d_delta, d_sink);
    run_if<30>(ncu, d_delta, d_sink);
    run_if<31>(ncu, d_delta, d_sink);
    run_if<32>(ncu, d_delta, d_sink);
    run_if<33>(ncu, d_delta, d_sink);
    run_if<34>(ncu, d_delta, d_sink);
    run_if<35>(ncu, d_delta, d_sink);
    run_if<36>(ncu, d_delta, d_sink);
    run_if<37>(ncu, d_delta, d_sink);
    run_if<38>(ncu, d_delta, d_sink);
    run_if<39>(ncu, d_delta, d_sink);
    run_if<40>(ncu, d_delta, d_sink);
    run_if<41>(ncu, d_delta, d_sink);
    run_if<42>(ncu, d_delta, d_sink);
    run_if<43>(ncu, d_delta, d_sink);
    run_if<44>(ncu, d_delta, d_sink);
    run_if<45>(ncu, d_delta, d_sink);
    run_if<46>(ncu, d_delta, d_sink);
    run_if<47>(ncu, d_delta, d_sink);
    run_if<48>(ncu, d_delta, d_sink);
    run_if<49>(ncu, d_delta, d_sink);
    run_if<50>(ncu, d_delta, d_sink);
    run_if<51>(ncu, d_delta, d_sink);
    run_if<52>(ncu, d_delta, d_sink);
    run_if<53>(ncu, d_delta, d_sink);
    run_if<54>(ncu, d_delta, d_sink);
    run_if<55>(ncu, d_delta, d_sink);
    run_if<56>(ncu, d_delta, d_sink);
    run_if<57>(ncu, d_delta, d_sink);
    run_if<58>(ncu, d_delta, d_sink);
    run_if<59>(ncu, d_delta, d_sink);
    run_if<60>(ncu, d_delta, d_sink);
    run_if<61>(ncu, d_delta, d_sink);
    run_if<62>(ncu, d_delta, d_sink);
    run_if<63>(ncu, d_delta, d_sink);
    run_if<64>(ncu, d_delta, d_sink);
    run_if<65>(ncu, d_delta, d_sink);
    run_if<66>(ncu, d_delta, d_sink);
    run_if<67>(ncu, d_delta, d_sink);
    run_if<68>(ncu, d_delta, d_sink);
    run_if<69>(ncu, d_delta, d_sink);
    run_if<70>(ncu, d_delta, d_sink);
    return 0;
}
