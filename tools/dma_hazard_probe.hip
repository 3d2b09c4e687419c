// dma_hazard_probe: does a global_load_lds_dwordx4 read its address VGPRs when it ISSUES, or later (when the CU's L1 -> LDS path takes it)?
//
// An ablation build of the f16s kernel (packing removed: the compiler then reuses a DMA's address registers as the destination of the very next
// ds_read_b128) faulted on addresses whose high dword was a valid pointer's; the shipped builds never do that reuse.  This probe forces the
// sequence with explicit registers:
//        global_load_lds_dwordx4 v[10:11], off        ; address = A + 16 lane
//        ds_read_b128            v[10:13], v14        ; returns {lo32, hi32} of  B + 16 lane  into v[10:11]
// with the L1 -> LDS path kept busy by PRE DMAs from all four waves of the workgroup just before.  Both A and B are valid, so a late read
// shows as B's bytes in the destination instead of A's (no fault either way).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_hazard_probe.hip -o tools/dma_hazard_probe.bin && tools/dma_hazard_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KIB = 1024;
// LDS map: [0, 1 KiB) the address table the ds_read returns; [4 KiB, 8 KiB) this wave's destination under test; [16 KiB, ..) scratch destinations of the PRE DMAs
// HOW: 0 nothing follows the DMA (control); 1 ds_read_b128 into the address registers; 2 a VALU write of B's address into them;
//      3 like 1 after 16 wait states; 4 a global_load_dwordx2 (VMEM return) of B's address into them
template <int PRE, int HOW>
__global__ __launch_bounds__(256) void probe(const char* A, const char* B, const char* junk, const unsigned long long* btab, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned long long b = (unsigned long long)(uintptr_t)(B + lane * 16);
    if (wave == 0) ((u32x4*)smem)[lane] = u32x4{(unsigned)b, (unsigned)(b >> 32), 0u, 0u};
    for (int i = threadIdx.x; i < 4 * KIB / 4; i += 256) ((unsigned*)(smem + 4 * KIB))[i] = 0xDEADBEEFu;
    __syncthreads();
    const char* ga = A + lane * 16;
    const char* gj = junk + (wave * 16 + 0) * KIB + lane * 16;
    const unsigned tab = lds0 + lane * 16;
    const unsigned long long* gb = btab + lane;                 // btab[lane] = B + 16 lane (written by the host)
    const unsigned m_pre = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + 16 * KIB + wave * 16 * KIB));
    const unsigned m_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + 4 * KIB));
    // every wave loads the path with PRE DMAs; wave 0 then issues the DMA under test followed at once by the overwriting instruction
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m_pre) : "memory");
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
        const char* g = gj + i * KIB;
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(g) : "memory");
    }
    if (wave == 0) {
#define PROLOG "s_mov_b32 m0, %2\n\ts_nop 0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v14, %3\n\tv_mov_b32 v16, %4\n\tv_mov_b32 v17, %5\n\tv_mov_b32 v18, %6\n\tv_mov_b32 v19, %7\n\ts_nop 4\n\tglobal_load_lds_dwordx4 v[10:11], off\n\t"
#define EPILOG "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define OPS : : "v"((unsigned)(uintptr_t)ga), "v"((unsigned)((uintptr_t)ga >> 32)), "s"(m_dst), "v"(tab), "v"((unsigned)(uintptr_t)gb), "v"((unsigned)((uintptr_t)gb >> 32)), "v"((unsigned)b), "v"((unsigned)(b >> 32)) : "memory", "v10", "v11", "v12", "v13", "v14", "v16", "v17", "v18", "v19"
        if constexpr (HOW == 0) asm volatile(PROLOG EPILOG OPS);
        if constexpr (HOW == 1) asm volatile(PROLOG "ds_read_b128 v[10:13], v14\n\t" EPILOG OPS);
        if constexpr (HOW == 2) asm volatile(PROLOG "v_mov_b32 v10, v18\n\tv_mov_b32 v11, v19\n\t" EPILOG OPS);
        if constexpr (HOW == 3) asm volatile(PROLOG "s_nop 15\n\tds_read_b128 v[10:13], v14\n\t" EPILOG OPS);
        if constexpr (HOW == 4) asm volatile(PROLOG "global_load_dwordx2 v[10:11], v[16:17], off\n\t" EPILOG OPS);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < KIB / 4; i += 256) out[blockIdx.x * (KIB / 4) + i] = ((const unsigned*)(smem + 4 * KIB))[i];
}

template <int PRE, int HOW>
static void run(const char* name, int grid, const char* A, const char* B, const char* junk, const unsigned long long* btab, unsigned* dout,
                const std::vector<unsigned>& ha, const std::vector<unsigned>& hb) {
    const int lds = 96 * KIB;
    hipFuncSetAttribute((const void*)(probe<PRE, HOW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    std::vector<unsigned> h((size_t)grid * KIB / 4);
    size_t from_a = 0, from_b = 0, other = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL((probe<PRE, HOW>), dim3(grid), dim3(256), lds, 0, A, B, junk, btab, dout);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
        hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
        for (int g = 0; g < grid; ++g)
            for (int l = 0; l < 64; ++l) {                                 // one verdict per lane (16 bytes)
                const unsigned* p = &h[(size_t)g * (KIB / 4) + l * 4];
                if (p[0] == ha[l * 4] && p[3] == ha[l * 4 + 3]) ++from_a;
                else if (p[0] == hb[l * 4] && p[3] == hb[l * 4 + 3]) ++from_b;
                else ++other;
            }
    }
    printf("%-44s PRE %2d: lanes that read A (the address at issue) %8zu   B (the overwritten registers) %8zu   other %zu\n", name, PRE, from_a, from_b, other);
}

int main() {
    const int grid = 256;
    std::vector<unsigned> ha(KIB / 4), hb(KIB / 4);
    for (int i = 0; i < KIB / 4; ++i) { ha[i] = 0xA0000000u | i; hb[i] = 0xB0000000u | i; }
    char *A, *B, *junk; unsigned long long* btab; unsigned* dout;
    hipMalloc(&A, KIB); hipMalloc(&B, KIB); hipMalloc(&junk, 64 * KIB); hipMalloc(&btab, 64 * 8); hipMalloc(&dout, (size_t)grid * KIB);
    hipMemcpy(A, ha.data(), KIB, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), KIB, hipMemcpyHostToDevice); hipMemset(junk, 0, 64 * KIB);
    std::vector<unsigned long long> hbt(64);
    for (int l = 0; l < 64; ++l) hbt[l] = (unsigned long long)(uintptr_t)(B + l * 16);
    hipMemcpy(btab, hbt.data(), 64 * 8, hipMemcpyHostToDevice);
#define RUN(P, H, N) run<P, H>(N, grid, A, B, junk, btab, dout, ha, hb)
    RUN(0, 0, "control (nothing follows)");  RUN(8, 0, "control (nothing follows)");
    RUN(0, 1, "ds_read_b128 into the address registers");  RUN(4, 1, "ds_read_b128 into the address registers");  RUN(8, 1, "ds_read_b128 into the address registers");
    RUN(0, 2, "VALU write of the address registers");  RUN(8, 2, "VALU write of the address registers");
    RUN(8, 3, "ds_read_b128 after 16 wait states");
    RUN(0, 4, "global_load_dwordx2 into the address regs");  RUN(8, 4, "global_load_dwordx2 into the address regs");
    return 0;
}
