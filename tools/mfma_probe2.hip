// mfma_probe2.hip -- does a second workgroup per CU hide one workgroup's VALU phases behind the other's MFMAs?
// v_mfma_f32_16x16x4_f32, 64 accumulator registers per wave, <= 256 registers so two 256-thread workgroups fit a CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VALU_BURST, int PRIO>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x4 acc[16];
    f32x4 a[16];
    float b[8], side[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 16; ++t) { acc[t] = f32x4{0, 0, 0, 0}; a[t] = f32x4{1.f + lane, 2.f, 3.f, 4.f + t}; }
    for (int t = 0; t < 8; ++t) b[t] = 0.5f * t + lane;
    for (int i = threadIdx.x; i < 2048; i += 256) ((f32x4*)smem)[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) {          // 16 k-quads x 16 tiles x 4 = 1024 MFMAs
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j], b[(kq + j) & 7], acc[t], 0, 0, 0);
                a[t] = *(const f32x4*)(smem + ((kq * 16 + t) & 31) * 1024 + lane * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((kq & 3) == 3) __syncthreads();
        }
        // layer-boundary style VALU burst
        if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int v = 0; v < VALU_BURST; ++v) {
            float x = side[v & 7] + b[v & 7];
            asm volatile("v_max_i32 %0, 0, %1" : "=v"(x) : "v"(x));
            side[(v + 1) & 7] = x;
        }
    }
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int x = 0; x < 8; ++x) s += side[x];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VB, int PRIO>
void run(int grid) {
    const int iters = 200;
    float* out;
    hipMalloc(&out, grid * 256 * 4);
    auto kern = probe<VB, PRIO>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 65536, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 65536, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 1024 * (grid / 256.0);
    printf("prio %d VALU burst %4d  grid %3d (%d WG/CU): %.3f ms  -> %.2f cycles per MFMA per SIMD at 2.4 GHz (ideal 32)\n", PRIO, VB, grid, grid / 256, ms,
           ms * 1e-3 * 2.4e9 / mfma_per_simd);
    hipFree(out);
}

int main() {
    for (int grid : {256, 512}) { run<0, 0>(grid); run<400, 0>(grid); run<1200, 0>(grid); run<400, 1>(grid); run<1200, 1>(grid); }
    return 0;
}
