// mfma_probe5.hip -- what does a non-MFMA instruction cost beside back-to-back v_mfma_f32_32x32x2_f32 (64 cycles each, one wave per SIMD,
// 16 independent accumulators = all 256 AGPRs, operands in registers)?  Each variant adds 14-16 fillers per 16 MFMAs in a different
// shape; the cycle efficiency (SQ_INSTS_MFMA * 64 / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)) is read from a --pmc pass (tools/r02_run7.sh),
// the printed figure assumes 2.4 GHz.
//   0  no filler                                   5  14 s_nop 0, 2 per gap in 7 gaps
//   1  14 v_add_f32, 2 per gap in 7 gaps           6  14 v_add_f32 in ONE gap (behind MFMA 15)
//   2  14 v_add_f32, 1 per gap in 14 gaps          7  14 v_add_f32 in ONE gap (behind MFMA 7)
//   3  as 1, on registers no MFMA reads            8  16 v_mov_b32 (1 source), 1 per gap
//   4  14 s_add_u32, 2 per gap in 7 gaps           9  as 1 with s_setprio 3 around the kernel body
// hipcc --offload-arch=gfx950 -O3 tools/mfma_probe5.hip -o tools/mfma_probe5.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe(float* out, int groups) {
    constexpr int U = 6;
    const int lane = threadIdx.x & 63;
    f32x16 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 X[U], Y[U];
    for (int u = 0; u < U; ++u) { X[u] = f32x4{1.f + lane, 2.f, 3.f, 4.f}; Y[u] = f32x4{0.5f, 1.5f, 2.5f, 3.5f + u}; }
    float d0 = lane, d1 = 1.f, d2 = 2.f;
    int s0 = groups;
    if (VAR == 9) __builtin_amdgcn_s_setprio(3);
    for (int g = 0; g < groups; ++g) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc[t >> 2][t & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[u][t >> 2], Y[u][t & 3], acc[t >> 2][t & 3], 0, 0, 0);
                const bool odd7 = (t & 1) && t < 14;
                if ((VAR == 1 || VAR == 9) && odd7) asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(X[u][1]));
                if (VAR == 2 && t < 14) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d0) : "v"(X[u][1]));
                if (VAR == 3 && odd7) asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(d2));
                if (VAR == 4 && odd7) asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5" : "+s"(s0));
                if (VAR == 5 && odd7) asm volatile("s_nop 0\n\ts_nop 0");
                if ((VAR == 6 && t == 15) || (VAR == 7 && t == 7))
                    asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\t"
                                 "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\t"
                                 "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(d2));
                if (VAR == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(d0) : "v"(d2));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = d0 + d1 + (float)s0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR>
void run(float* out, int grid, int groups) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe<VAR>, dim3(grid), dim3(256), 0, 0, out, groups);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)groups * 6 * 16, cyc = ms * 1e-3 * 2.4e9;
    printf("variant %d: %.3f ms  %.1f cycles per MFMA at 2.4 GHz (64 = peak)\n", VAR, ms, cyc / mfma);
}

int main() {
    const int grid = 256, groups = 258;
    float* out; (void)hipMalloc(&out, grid * 256 * 4);
    run<0>(out, grid, groups); run<1>(out, grid, groups); run<2>(out, grid, groups); run<3>(out, grid, groups); run<4>(out, grid, groups);
    run<5>(out, grid, groups); run<6>(out, grid, groups); run<7>(out, grid, groups); run<8>(out, grid, groups); run<9>(out, grid, groups);
    run<0>(out, grid, groups);
    return 0;
}
