// mfma_probe.hip -- calibrates v_mfma_f32_32x32x2_f32 issue patterns on gfx950 (diagnostic tool, not product).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// VARIANT 0: 8 accumulators round-robin (independent consecutive MFMAs), operands in registers
// VARIANT 1: t-major, 4 dependent MFMAs per accumulator
// VARIANT 2: VARIANT 1 + one ds_read_b128 per group feeding the A operand pipeline
// VARIANT 3: VARIANT 2 + __syncthreads() every 32 groups
// VARIANT 4: VARIANT 0 with 2 dependent (pairs)
// VARIANT 5: t-major; between the MFMAs of tile t, 4 x (v_accvgpr_read + v_add + v_max_i32) on tile t-2's accumulator
// VARIANT 6: like 5 but the VALU work reads plain VGPRs (no accumulator-file access)
// VARIANT 7: like 5 with 8 VALU ops per gap and no accumulator reads
// VARIANT 8: 4 x (add, max, v_accvgpr_write) per gap
// VARIANT 9: 4 x v_accvgpr_read only per gap; VARIANT 10: 8 x v_accvgpr_read per gap
template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    f32x4 a[8];
    float b[8];
    float side[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 8; ++t) {
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        a[t] = f32x4{1.f + lane, 2.f, 3.f, 4.f + t};
        b[t] = 0.5f * t + lane;
    }
    for (int i = threadIdx.x; i < 8192; i += 256) ((f32x4*)smem)[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    const unsigned long long t0 = stamp();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) {
            if (VARIANT == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[(kq + j) & 7], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (VARIANT == 4) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][2 * jj], b[(kq + jj) & 7], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][2 * jj + 1], b[(kq + jj + 1) & 7], acc[t], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            } else if (VARIANT == 15 || VARIANT == 16) {
                // build with -mllvm -amdgpu-mfma-vgpr-form: accumulators in VGPRs.  In-place relu(acc[e]+b) with explicit VALU
                // instructions dealt 4 per MFMA gap; VARIANT 16 also copies each result into an AGPR (next layer's B operand).
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int e = (t + 6) & 7;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[(kq + j) & 7], acc[t], 0, 0, 0);
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            float v;
                            asm volatile("v_add_f32 %0, %1, %2" : "=v"(v) : "v"(acc[e][4 * j + x]), "v"(b[x]));
                            asm volatile("v_max_i32 %0, 0, %1" : "=v"(v) : "v"(v));
                            acc[e][4 * j + x] = v;
                            if (VARIANT == 16) { float aw; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(aw) : "v"(v)); asm volatile("" :: "a"(aw)); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if (VARIANT == 13 || VARIANT == 14) {
                // accumulator tile e leaves the AGPR file through LDS: ds_write_b128 straight from AGPRs, read back to VGPRs
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int e = (t + 6) & 7;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[(kq + j) & 7], acc[t], 0, 0, 0);
                        f32x4 q = {acc[e][4 * j], acc[e][4 * j + 1], acc[e][4 * j + 2], acc[e][4 * j + 3]};
                        const unsigned addr = 65536 + (threadIdx.x * 16) + ((t * 4 + j) & 7) * 4096;
                        asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "a"(q) : "memory");
                        if (VARIANT == 14) {
                            f32x4 r;
                            const unsigned raddr = 65536 + (threadIdx.x * 16) + ((t * 4 + j + 4) & 7) * 4096;
                            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(raddr) : "memory");
#pragma unroll
                            for (int x = 0; x < 4; ++x) {
                                float v = r[x] + b[x];
                                asm volatile("v_max_i32 %0, 0, %1" : "=v"(v) : "v"(v));
                                side[(t + x) & 7] = v;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if (VARIANT >= 5) {   // 5..12
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int e = (t + 6) & 7;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[(kq + j) & 7], acc[t], 0, 0, 0);
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            float v;
                            if (VARIANT == 11 || VARIANT == 12) v = acc[e][4 * j + x];          // plain read: VGPR when built with -amdgpu-mfma-vgpr-form
                            else if (VARIANT == 5 || VARIANT >= 9) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[e][4 * j + x]));
                            else v = side[(4 * j + x) & 7];
                            if (VARIANT == 12) {        // in-place epilogue: write back into the (idle) accumulator tile
                                v = v + b[x];
                                const int bi2 = __builtin_bit_cast(int, v);
                                acc[e][4 * j + x] = __builtin_bit_cast(float, bi2 > 0 ? bi2 : 0);
                                continue;
                            }
                            if (VARIANT >= 9 && VARIANT <= 10) {
                                if (VARIANT == 10) { float v2; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v2) : "a"(acc[(e + 1) & 7][4 * j + x])); asm volatile("" :: "v"(v2)); }
                                asm volatile("" :: "v"(v));
                                continue;
                            }
                            v = v + b[x];
                            const int bi = __builtin_bit_cast(int, v);
                            side[(t + x) & 7] = __builtin_bit_cast(float, bi > 0 ? bi : 0);
                            if (VARIANT == 8) { float aw; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(aw) : "v"(side[(t + x) & 7])); asm volatile("" :: "a"(aw)); }
                            if (VARIANT == 7) { side[(t + x + 1) & 7] += v * 0.5f; side[(t + x + 2) & 7] += v; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[(kq + j) & 7], acc[t], 0, 0, 0);
                    if (VARIANT >= 2) {
                        if (VARIANT == 3 && t == 0 && (kq & 3) == 0) __syncthreads();
                        a[t] = *(const f32x4*)(smem + ((kq * 8 + t) & 31) * 1024 + lane * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    const unsigned long long t1 = stamp();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int x = 0; x < 8; ++x) s += side[x];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
void run(const char* name, int grid) {
    const int iters = 64;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 4 * 8);
    hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 131072, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (auto v : h) { sum += v; if (v > mx) mx = v; }
    const double n = (double)iters * 16 * 32;
    printf("%-44s grid %3d: %.2f cycles/MFMA mean, %.2f max-wave\n", name, grid, sum / h.size() / n, mx / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int grid : {1, 256}) {
        run<0>("k-major, 8 independent accumulators", grid);
        run<4>("pairs of dependent MFMAs", grid);
        run<1>("t-major, 4 dependent MFMAs per group", grid);
        run<2>("t-major + ds_read_b128 per group", grid);
        run<3>("t-major + ds_read + barrier per 32 groups", grid);
        run<5>("t-major + 4x(accvgpr_read,add,max) per gap", grid);
        run<6>("t-major + 4x(add,max) per gap, VGPR only", grid);
        run<7>("t-major + 4x(add,max,fma,add) per gap, VGPR", grid);
        run<8>("t-major + 4x(add,max,accvgpr_write) per gap", grid);
        run<9>("t-major + 4x accvgpr_read per gap", grid);
        run<10>("t-major + 8x accvgpr_read per gap", grid);
        run<15>("vgpr-form: in-place asm relu(acc[e]+b) 4/gap", grid);
        run<16>("vgpr-form: same + v_accvgpr_write 4/gap", grid);
        run<13>("t-major + ds_write_b128 from AGPR per gap", grid);
        run<14>("t-major + ds_write(AGPR), ds_read, 4x(add,max)", grid);
        run<11>("t-major + 4x(read acc[e],add,max) per gap plain C", grid);
        run<12>("t-major + in-place relu(acc[e]+b) 4 per gap", grid);
    }
    return 0;
}
