// mfma_probe3.hip -- why does wgrad_big_kernel's loop sustain only ~84 % of the fp32 MFMA rate?
// 16 independent v_mfma_f32_32x32x2_f32 accumulators (all 256 AGPRs), k-major issue order as in the kernel, with
//   MODE 0: operands in registers only                       (pure MFMA issue)
//   MODE 1: + two 16-byte global loads per 16 MFMAs from a 64 KB buffer (L2/L1 resident), three register sets rotated by name
//   MODE 2: same loads from a streaming buffer (HBM)
//   MODE 3: MODE 0 with tile-major order (4 dependent MFMAs per accumulator, as the forward kernel issues them)
// hipcc --offload-arch=gfx950 -O3 tools/mfma_probe3.hip -o tools/mfma_probe3.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe(const float* __restrict__ A, const float* __restrict__ B, long long rows, float* out, int groups) {
    constexpr int U = 6;
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    f32x16 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 ca[U], cb[U], na[U], nb[U], fa[U], fb[U];
    for (int u = 0; u < U; ++u) { ca[u] = f32x4{1.f + lane, 2.f, 3.f, 4.f}; cb[u] = f32x4{0.5f, 1.5f, 2.5f, 3.5f + u}; na[u] = ca[u]; nb[u] = cb[u]; fa[u] = ca[u]; fb[u] = cb[u]; }
    const long long slices = gridDim.x, gstride = slices * 12;
    long long row = (long long)blockIdx.x * 12 + kh;
    int ru = 0;
    const float* abase = A + 4 * i + ((threadIdx.x >> 6) >> 1) * 128;
    const float* bbase = B + 4 * i + ((threadIdx.x >> 6) & 1) * 128;
    auto request = [&](f32x4& X, f32x4& Y) __attribute__((always_inline)) {
        if (MODE == 1 || MODE == 2) {
            const long long r = (MODE == 1) ? (row & 63) : (row < rows ? row : rows - 1);
            X = *(const f32x4*)(abase + r * 256);
            Y = *(const f32x4*)(bbase + r * 256);
            row += 2;
            if (++ru == U) { ru = 0; row += gstride - 12; }
        }
    };
    auto step = [&](f32x4 (&X)[U], f32x4 (&Y)[U], f32x4 (&FX)[U], f32x4 (&FY)[U]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 3) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[u][tm], Y[u][tn], acc[tm][tn], 0, 0, 0);
            } else {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[u][tm], Y[u][tn], acc[tm][tn], 0, 0, 0);
            }
            request(FX[u], FY[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int g = 0; g < groups; g += 3) {
        step(ca, cb, fa, fb);
        step(na, nb, ca, cb);
        step(fa, fb, na, nb);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void fill(float* p, long long n, unsigned seed, float zero_frac) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = (float)(h & 0xFFFFFF) / 8388608.0f - 1.0f;
        p[i] = ((h >> 24) / 256.0f < zero_frac) ? 0.0f : v;
    }
}

template <int MODE>
void run(const float* A, const float* B, long long rows, float* out, int grid, int groups) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, A, B, rows, out, groups);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, A, B, rows, out, groups);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)groups * 6 * 16;              // per wave
    const double cyc = ms * 1e-3 * 2.4e9;                     // nominal clock
    printf("MODE %d: %.3f ms  %.1f cycles per MFMA at 2.4 GHz (64 = peak)  -> %.1f %% of peak\n", MODE, ms, cyc / mfma, 6400.0 / (cyc / mfma));
}

int main() {
    const int grid = 256, groups = 258;                        // 256 WGs x 258 groups x 12 rows = the fine net's 786 432 points (+pad)
    const long long rows = (long long)grid * groups * 12;
    float *A, *B, *out;
    hipMalloc(&A, (rows + 64) * 1024); hipMalloc(&B, (rows + 64) * 1024); hipMalloc(&out, grid * 256 * 4);
    hipMemset(A, 0, (rows + 64) * 1024); hipMemset(B, 0, (rows + 64) * 1024);
    run<0>(A, B, rows, out, grid, groups);
    run<3>(A, B, rows, out, grid, groups);
    run<1>(A, B, rows, out, grid, groups);
    printf("streaming operands, all zeros:\n");
    run<2>(A, B, rows, out, grid, groups);
    for (float zf : {0.5f, 0.0f}) {
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, (rows + 64) * 256, 1u, zf);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, B, (rows + 64) * 256, 2u, zf);
        printf("streaming operands, uniform(-1,1) with %.0f %% zeros:\n", zf * 100);
        run<2>(A, B, rows, out, grid, groups);
        run<2>(A, B, rows, out, grid, groups);
    }
    return 0;
}
