// mfma_probe3.hip -- why does wgrad_big_kernel's loop sustain only ~84 % of the fp32 MFMA rate?
// 16 independent v_mfma_f32_32x32x2_f32 accumulators (all 256 AGPRs), k-major issue order as in the kernel, with
//   MODE 0: operands in registers only                       (pure MFMA issue)
//   MODE 1: + two 16-byte global loads per 16 MFMAs from a 64 KB buffer (L2/L1 resident), three register sets rotated by name
//   MODE 2: same loads from a streaming buffer (HBM)
//   MODE 3: MODE 0 with tile-major order (4 dependent MFMAs per accumulator, as the forward kernel issues them)
//   MODE 5: MODE 2 with the two loads and their address arithmetic spread behind MFMAs 1, 5 and 9 of the k-step instead of
//           sitting in one clump behind the 16th (a wave issues in order: whatever follows the last MFMA of a k-step delays the first
//           MFMA of the next one by its issue time minus the 64 cycles the pipe is still busy)
//   MODE 6: MODE 0 + 14 independent v_add_f32 per k-step spread over the MFMA gaps (does plain VALU beside 64-cycle MFMAs cost MFMA time?)
//   MODE 7: MODE 2 with wave-uniform row addressing kept in SGPRs (global_load saddr form: no vector address arithmetic per k-step)
//   MODE 4: operands staged ONCE per workgroup through LDS (global_load_lds_dwordx4, one 1 KiB row per DMA, one DMA per wave per
//           k-step, NSLOT slots of 12 + 12 rows, a barrier per slot), fragments read back with ds_read_b128 one k-step ahead
// hipcc --offload-arch=gfx950 -O3 tools/mfma_probe3.hip -o tools/mfma_probe3.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
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
    long long rclamped = 0;
    auto step5 = [&](f32x4 (&X)[U], f32x4 (&Y)[U], f32x4 (&FX)[U], f32x4 (&FY)[U]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc[t >> 2][t & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[u][t >> 2], Y[u][t & 3], acc[t >> 2][t & 3], 0, 0, 0);
                if (t == 1) { rclamped = row < rows ? row : rows - 1; __builtin_amdgcn_sched_barrier(0); }
                if (t == 4) { FX[u] = *(const f32x4*)(abase + rclamped * 256); __builtin_amdgcn_sched_barrier(0); }
                if (t == 7) { FY[u] = *(const f32x4*)(bbase + rclamped * 256); __builtin_amdgcn_sched_barrier(0); }
                if (t == 10) { row += 2; if (++ru == U) { ru = 0; row += gstride - 12; } __builtin_amdgcn_sched_barrier(0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    float d0 = lane, d1 = 1.f, d2 = 2.f, d3 = 3.f, d4 = 4.f, d5 = 5.f, d6 = 6.f;
    auto step6 = [&](f32x4 (&X)[U], f32x4 (&Y)[U]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc[t >> 2][t & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[u][t >> 2], Y[u][t & 3], acc[t >> 2][t & 3], 0, 0, 0);
                if (t == 1 || t == 3 || t == 5 || t == 7 || t == 9 || t == 11 || t == 13) {
                    asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(d2));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // wave-uniform part of the address in SGPRs: row r of this k-step for lane half kh is base + (r0 + kh) * 1024, r0 uniform
    const unsigned voffA = (unsigned)(kh * 1024 + (4 * i + ((threadIdx.x >> 6) >> 1) * 128) * 4);
    const unsigned voffB = (unsigned)(kh * 1024 + (4 * i + ((threadIdx.x >> 6) & 1) * 128) * 4);
    long long srow = (long long)blockIdx.x * 12;                       // uniform
    int sru = 0;
    auto step7 = [&](f32x4 (&X)[U], f32x4 (&Y)[U], f32x4 (&FX)[U], f32x4 (&FY)[U]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                acc[t >> 2][t & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[u][t >> 2], Y[u][t & 3], acc[t >> 2][t & 3], 0, 0, 0);
            const long long r = srow + 1 < rows ? srow : rows - 2;
            const char* sa = (const char*)A + r * 1024;
            const char* sb = (const char*)B + r * 1024;
            FX[u] = *(const f32x4*)(sa + voffA);
            FY[u] = *(const f32x4*)(sb + voffB);
            srow += 2;
            if (++sru == U) { sru = 0; srow += gstride - 12; }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (MODE == 5) {
        for (int g = 0; g < groups; g += 3) {
            step5(ca, cb, fa, fb);
            step5(na, nb, ca, cb);
            step5(fa, fb, na, nb);
        }
    } else if (MODE == 6) {
        for (int g = 0; g < groups; g += 3) {
            step6(ca, cb);
            step6(na, nb);
            step6(fa, fb);
        }
        acc[0][0][0] += d0 + d1 + d3 + d4 + d5 + d6;
    } else if (MODE == 7) {
        for (int g = 0; g < groups; g += 3) {
            step7(ca, cb, fa, fb);
            step7(na, nb, ca, cb);
            step7(fa, fb, na, nb);
        }
    } else
    for (int g = 0; g < groups; g += 3) {
        step(ca, cb, fa, fb);
        step(na, nb, ca, cb);
        step(fa, fb, na, nb);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int AHEAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe_lds(const float* __restrict__ A, const float* __restrict__ B, long long rows, float* out, int groups) {
    constexpr int U = 6, NSLOT = AHEAD + 2, SLOT = 24 * 1024;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const long long slices = gridDim.x;
    const unsigned lds0 = (unsigned)(size_t)lds;
    // DMA j (0..5) of a slot, this wave: row r = wave + 4 j of the slot's 24 (12 of A, then 12 of B)
    long long dma_group = 0;                  // group whose rows the next DMA fetches
    int dma_j = 0;
    auto dma = [&]() __attribute__((always_inline)) {
        const int r = wave + 4 * dma_j;
        const bool isb = r >= 12;
        const int rr = isb ? r - 12 : r;
        long long row = (dma_group * slices + blockIdx.x) * 12 + rr;
        row = row < rows ? row : rows - 1;
        const float* g = (isb ? B : A) + row * 256 + 4 * lane;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(dma_group % NSLOT) * SLOT + (unsigned)r * 1024u));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(g) : "memory");
        if (++dma_j == U) { dma_j = 0; ++dma_group; }
    };
    const unsigned aoff = (unsigned)(kh * 1024 + ((wave >> 1) * 128 + 4 * i) * 4);
    const unsigned boff = (unsigned)(12 * 1024 + kh * 1024 + ((wave & 1) * 128 + 4 * i) * 4);
    auto frag = [&](long long g, int u, f32x4& X, f32x4& Y) __attribute__((always_inline)) {
        const char* sl = lds + (unsigned)(g % NSLOT) * SLOT + u * 2048;
        X = *(const f32x4*)(sl + aoff);
        Y = *(const f32x4*)(sl + boff);
    };
    for (int k = 0; k < (AHEAD + 1) * U; ++k) dma();
    if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (AHEAD == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if (AHEAD == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    __syncthreads();
    f32x4 x0, y0, x1, y1;
    frag(0, 0, x0, y0);
    auto mm = [&](const f32x4& X, const f32x4& Y) __attribute__((always_inline)) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[tm], Y[tn], acc[tm][tn], 0, 0, 0);
    };
    // k-step ks of group g: fetch the fragments of the next k-step, multiply the current ones, issue one DMA.  After the MFMAs of
    // k-step 4 the group g+1 has landed (this wave's share: vmcnt; everybody's: barrier), and every wave is done with group g-1,
    // whose slot the DMAs from here on overwrite (group g+1+AHEAD).
    auto kstep = [&](long long g, auto KS, f32x4& cx, f32x4& cy, f32x4& nx, f32x4& ny) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        if (ks < U - 1) frag(g, ks + 1, nx, ny); else frag(g + 1, 0, nx, ny);
        mm(cx, cy);
        if (ks == U - 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (AHEAD == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if (AHEAD == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __syncthreads();
        }
        dma();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (long long g = 0; g < groups; ++g) {
        kstep(g, std::integral_constant<int, 0>{}, x0, y0, x1, y1);
        kstep(g, std::integral_constant<int, 1>{}, x1, y1, x0, y0);
        kstep(g, std::integral_constant<int, 2>{}, x0, y0, x1, y1);
        kstep(g, std::integral_constant<int, 3>{}, x1, y1, x0, y0);
        kstep(g, std::integral_constant<int, 4>{}, x0, y0, x1, y1);
        kstep(g, std::integral_constant<int, 5>{}, x1, y1, x0, y0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int AHEAD>
void run_lds(const float* A, const float* B, long long rows, float* out, int grid, int groups) {
    const int bytes = (AHEAD + 2) * 24 * 1024;
    hipFuncSetAttribute((const void*)probe_lds<AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_lds<AHEAD>, dim3(grid), dim3(256), bytes, 0, A, B, rows, out, groups);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe_lds<AHEAD>, dim3(grid), dim3(256), bytes, 0, A, B, rows, out, groups);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)groups * 6 * 16, cyc = ms * 1e-3 * 2.4e9;
    static float host[256 * 256];
    hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost);
    double cs = 0; for (int k = 0; k < grid * 256; ++k) cs += (double)host[k] * (1 + k % 7);
    printf("MODE 4 (LDS-DMA staged, %d groups ahead): %.3f ms  %.1f cycles per MFMA at 2.4 GHz -> %.1f %% of peak  [%s] checksum %.9e\n", AHEAD, ms, cyc / mfma,
           6400.0 / (cyc / mfma), hipGetErrorString(hipGetLastError()), cs);
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
    run<6>(A, B, rows, out, grid, groups);
    printf("streaming operands, all zeros:\n");
    run<2>(A, B, rows, out, grid, groups);
    run<5>(A, B, rows, out, grid, groups);
    run<7>(A, B, rows, out, grid, groups);
    run_lds<2>(A, B, rows, out, grid, groups);
    for (float zf : {0.5f, 0.0f}) {
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, (rows + 64) * 256, 1u, zf);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, B, (rows + 64) * 256, 2u, zf);
        printf("streaming operands, uniform(-1,1) with %.0f %% zeros:\n", zf * 100);
        run<2>(A, B, rows, out, grid, groups);
        run<2>(A, B, rows, out, grid, groups);
        run<5>(A, B, rows, out, grid, groups);
        run<7>(A, B, rows, out, grid, groups);
        run<7>(A, B, rows, out, grid, groups);
        run_lds<2>(A, B, rows, out, grid, groups);
    }
    return 0;
}
