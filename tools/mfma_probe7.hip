// mfma_probe7: what one LDS-DMA of the bf16 / f16s weight ring costs the wave that issues it, by instruction form.
//
// The ablation builds (profiles/r03_bf16_shape_ablation.txt, r03_f16s_*.txt) say the DMAs' ISSUE costs the computing waves 12 % (bf16) to
// 22 % (f16s) of the kernel: with one wave per SIMD nothing else issues while a VMEM instruction leaves the wave.  Same ring, same
// geometry as mfma_probe6's "ring" (256 workgroups x 4 waves, 3 x 32 KiB slots, 8 DMAs per wave and slot, 4 MFMAs per 1 KiB quad),
// the DMA written five ways:
//   vaddr    global_load_lds_dwordx4 v[a:a+1], off          (the kernels' form: 64-bit per-lane address)
//   saddr    global_load_lds_dwordx4 v_off, s[b:b+1]        (scalar base + 32-bit per-lane offset)
//   offen    buffer_load_dwordx4 v_off, s[d:d+3], s_o offen lds
//   addtid   buffer_load_dwordx4 off, s[d:d+3], s_o lds     (descriptor with ADD_TID_ENABLE, stride 16: no vector operand at all)
//   none     no DMA (the ring's slots keep their first contents): the bound
// Every variant must produce the same sums as "vaddr" (same bytes in the same LDS places) except "none".
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe7.hip -o tools/mfma_probe7.bin && tools/mfma_probe7.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int QUAD = 1024, SLOTQ = 32, SLOT = SLOTQ * QUAD, NSLOT = 3, NQ = 1184, RD = 3;
enum { VADDR = 0, SADDR = 1, OFFEN = 2, ADDTID = 3, NONE = 4, DWORD = 5, DWORDI = 6 };     // DWORD: four global_load_lds_dword per KiB at one position; DWORDI: one behind each of the quad's four MFMAs
// STAG: the four waves issue at different quad positions (wave w at 1 + 4 j + w) instead of all four at positions 1..8: the CU's one
// L1 -> LDS path takes a 1 KiB DMA in 16 cycles, and a wave that finds it busy waits at issue

__device__ __forceinline__ void set_m0(unsigned v) {
    const unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(s) : "memory");
}
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void sfor(F&& f) { if constexpr (B < E) { f(IC<B>{}); sfor<B + 1, E>(f); } }

struct Ring {
    const char* sbase;          // this wave's share of slot 0 in the stream
    u32x4 desc;                 // buffer descriptor over the same bytes
    unsigned voff;              // lane * 16
    unsigned fetch_off, fetch_lds, lds_lo, lds_hi, read_slot, bytes;
};

template <int DV, int IMM>
__device__ __forceinline__ void dma16(const Ring& r, unsigned half) {          // half: 0 or 4096, uniform
    if constexpr (DV == VADDR) {
        const char* g = r.sbase + r.fetch_off + r.voff + half;
        asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(g), "i"(IMM) : "memory");
    } else if constexpr (DV == SADDR) {
        const char* sb = r.sbase + r.fetch_off + half;                         // uniform: lives in SGPRs
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(r.voff), "s"(sb), "i"(IMM) : "memory");
    } else if constexpr (DV == OFFEN) {
        const unsigned so = r.fetch_off + half;
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(r.voff), "s"(r.desc), "s"(so), "i"(IMM) : "memory");
    } else if constexpr (DV == ADDTID) {
        const unsigned so = r.fetch_off + half;
        asm volatile("buffer_load_dwordx4 off, %0, %1 offset:%2 lds" ::"s"(r.desc), "s"(so), "i"(IMM) : "memory");
    }
}
// one 256-byte piece (j = 0..3) of DMA i: a straight copy, LDS byte = stream byte
template <int IMM>
__device__ __forceinline__ void dma4(const Ring& r, unsigned half) {
    const char* g = r.sbase + r.fetch_off + (r.voff >> 2) + half;
    asm volatile("global_load_lds_dword %0, off offset:%1" ::"v"(g), "i"(IMM) : "memory");
}
__device__ __forceinline__ void ring_dma_piece(const Ring& r, int i, int j) {
    const unsigned half = i >= 4 ? 4096u : 0u;
    if (i == 0 && j == 0) set_m0(r.fetch_lds);
    if (i == 4 && j == 0) set_m0(r.fetch_lds + 4096);
    const int imm = (i & 3) * 1024 + j * 256;
    switch (imm) {
#define C(k) case k * 256: dma4<k * 256>(r, half); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
    }
}
template <int DV>
__device__ __forceinline__ void ring_dma(const Ring& r, int i) {
    if constexpr (DV == DWORD || DV == DWORDI) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ring_dma_piece(r, i, j);
        return;
    }               // DMA i (0..7) of this wave's 8 KiB share of the slot being fetched
    if constexpr (DV == NONE) return;
    const unsigned half = i >= 4 ? 4096u : 0u;
    if (i == 0) set_m0(r.fetch_lds);
    if (i == 4) set_m0(r.fetch_lds + 4096);
    if ((i & 3) == 0) dma16<DV, 0>(r, half); else if ((i & 3) == 1) dma16<DV, 1024>(r, half); else if ((i & 3) == 2) dma16<DV, 2048>(r, half); else dma16<DV, 3072>(r, half);
}
__device__ __forceinline__ void ring_next(Ring& r) {
    r.fetch_off += SLOT; if (r.fetch_off >= r.bytes) r.fetch_off = 0;
    r.fetch_lds += SLOT; if (r.fetch_lds >= r.lds_hi) r.fetch_lds = r.lds_lo;
}
template <int DV, int SKEW = 0>
__device__ __forceinline__ void ring_advance(Ring& r, int wave = 0) {
    if constexpr (DV == DWORD || DV == DWORDI) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if constexpr (DV != NONE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (SKEW > 0)                                                    // wave w leaves the barrier w * SKEW * 16 cycles late: the waves' DMAs no longer meet at the L1 -> LDS path
        for (int i = 0; i < wave * SKEW; ++i) asm volatile("s_nop 15" ::: "memory");
    ring_next(r);
    r.read_slot = r.read_slot + 1 == NSLOT ? 0 : r.read_slot + 1;
}

template <int DV, bool STAG = false, int SKEW = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe(const char* __restrict__ stream, const u32x4* __restrict__ bin, float* out, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4 b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) b[i] = bin[i * 64 + lane];
    f32x4 c[4] = {};
    Ring r;
    r.sbase = stream + wave * 8 * QUAD; r.bytes = NQ * QUAD; r.fetch_off = 0; r.voff = lane * 16;
    {
        const unsigned long long base = (unsigned long long)(uintptr_t)r.sbase;
        // GFX9 buffer resource: base[47:0] | stride[13:0] << 48; num_records; word 3 = dst_sel xyzw, num_format, data_format (32: raw dwords),
        // ADD_TID_ENABLE (bit 23; the data_format field then holds stride[17:14] = 0)
        const unsigned stride = DV == ADDTID ? 16u : 0u;
        r.desc[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
        r.desc[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xFFFFu)) | (stride << 16);
        r.desc[2] = DV == ADDTID ? 0x10000000u : 0xFFFFFFFFu;                   // records of `stride` bytes / bytes
        r.desc[3] = DV == ADDTID ? (1u << 23) : 0x00020000u;
    }
    if (DV == NONE)                                                            // fill the ring once so the reads see data
        for (int i = threadIdx.x; i < NSLOT * SLOT / 16; i += 256) ((u32x4*)smem)[i] = ((const u32x4*)stream)[i];
    r.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8 * QUAD;
    r.lds_hi = r.lds_lo + NSLOT * SLOT; r.fetch_lds = r.lds_lo; r.read_slot = NSLOT - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) ring_dma<DV>(r, i);
    ring_next(r);
#pragma unroll
    for (int i = 0; i < 8; ++i) ring_dma<DV>(r, i);
    ring_advance<DV>(r);
    u32x4 a[32];
    sfor<0, RD>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        a[q] = *(const u32x4*)(smem + r.read_slot * SLOT + lane * 16 + q * QUAD);
    });
    const int n_slots = passes * (NQ / SLOTQ);
    for (int s = 0; s < n_slots; ++s) {
        sfor<0, SLOTQ>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            constexpr int qr = (q + RD) % SLOTQ;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                c[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[q]), __builtin_bit_cast(bf16x8, b[(q & 7) * 4 + p]), c[p], 0, 0, 0);
                if constexpr (DV == DWORDI && qr >= 2 && qr <= 9) { ring_dma_piece(r, qr - 2, p); __builtin_amdgcn_sched_barrier(0); }   // one position later: behind the advance
            }
            if constexpr (q + RD == SLOTQ) ring_advance<DV, SKEW>(r, wave);
            if constexpr (DV == DWORDI) {
            } else if constexpr (!STAG) {
                if constexpr (qr >= 1 && qr <= 8) ring_dma<DV>(r, qr - 1);
            } else if constexpr (qr >= 1) {
                constexpr int k = qr - 1;                                       // 0..30: wave k % 4 issues its DMA k / 4; the 32nd (wave 3, DMA 7) shares position 31
                if (wave == k % 4) ring_dma<DV>(r, k / 4);
                if constexpr (qr == 31) if (wave == 3) ring_dma<DV>(r, 7);
            }
            a[qr] = *(const u32x4*)(smem + r.read_slot * SLOT + lane * 16 + qr * QUAD);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    float sum = 0;
    for (int p = 0; p < 4; ++p) for (int e = 0; e < 4; ++e) sum += c[p][e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

// one slot filled by the four waves' DMAs, copied out: the host compares it with the stream's second slot (fetch_off = SLOT exercises the scalar offset)
template <int DV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void fill_check(const char* __restrict__ stream, u32x4* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < NSLOT * SLOT / 16; i += 256) ((u32x4*)smem)[i] = u32x4{0xDEADu, 0xDEADu, 0xDEADu, 0xDEADu};
    __syncthreads();
    Ring r;
    r.sbase = stream + wave * 8 * QUAD; r.bytes = NQ * QUAD; r.fetch_off = SLOT; r.voff = lane * 16;
    const unsigned long long base = (unsigned long long)(uintptr_t)r.sbase;
    const unsigned stride = DV == ADDTID ? 16u : 0u;
    r.desc[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
    r.desc[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xFFFFu)) | (stride << 16);
    r.desc[2] = DV == ADDTID ? 0x10000000u : 0xFFFFFFFFu;
    r.desc[3] = DV == ADDTID ? (1u << 23) : 0x00020000u;
    r.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8 * QUAD;
    r.lds_hi = r.lds_lo + NSLOT * SLOT; r.fetch_lds = r.lds_lo + SLOT; r.read_slot = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) ring_dma<DV>(r, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < SLOT / 16; i += 256) out[i] = ((const u32x4*)(smem + SLOT))[i];
}

static unsigned rnd_bf16() { union { float f; unsigned u; } v; v.f = (rand() / (float)RAND_MAX) * 2 - 1; return v.u >> 16; }

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    const size_t sb = (size_t)NQ * QUAD, nb = 32 * 64;
    std::vector<unsigned> hs(sb / 4 + 3 * SLOT / 4), hb(nb * 4);
    srand(1);
    for (auto& v : hs) v = rnd_bf16() | (rnd_bf16() << 16);
    for (auto& v : hb) v = rnd_bf16() | (rnd_bf16() << 16);
    char* ds; u32x4* db; float* dout;
    const size_t on = (size_t)grid * 256;
    hipMalloc(&ds, hs.size() * 4); hipMalloc(&db, nb * 16); hipMalloc(&dout, on * 4);
    hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), nb * 16, hipMemcpyHostToDevice);
    const int lds = NSLOT * SLOT;
#define OPT(k) hipFuncSetAttribute((const void*)(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds)
    OPT((probe<VADDR>)); OPT((probe<SADDR>)); OPT((probe<OFFEN>)); OPT((probe<ADDTID>)); OPT((probe<NONE>)); OPT((probe<DWORD>)); OPT((probe<DWORDI>)); OPT((probe<VADDR, true>)); OPT((probe<ADDTID, true>)); OPT((probe<VADDR, false, 1>)); OPT((probe<VADDR, false, 2>)); OPT((probe<ADDTID, false, 1>)); OPT((probe<NONE, false, 1>));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int passes = 12;
    const double flop = (double)grid * 4 * passes * NQ * 4 * (2.0 * 16 * 16 * 32);
    const char* names[] = {"vaddr", "saddr", "offen", "addtid", "none", "vaddr staggered", "addtid staggered", "vaddr skew 16", "vaddr skew 32", "addtid skew 16", "none skew 16", "4 x dword burst", "4 x dword behind MFMAs"};
    std::vector<float> ref(on), got(on);
    {
        u32x4* dchk; hipMalloc(&dchk, SLOT);
        std::vector<unsigned> hc(SLOT / 4);
        for (int v = 0; v < 5; ++v) {
            switch (v) {
                case 4: OPT((fill_check<DWORD>)); hipLaunchKernelGGL((fill_check<DWORD>), dim3(1), dim3(256), lds, 0, ds, dchk); break;
                case 0: OPT((fill_check<VADDR>)); hipLaunchKernelGGL((fill_check<VADDR>), dim3(1), dim3(256), lds, 0, ds, dchk); break;
                case 1: OPT((fill_check<SADDR>)); hipLaunchKernelGGL((fill_check<SADDR>), dim3(1), dim3(256), lds, 0, ds, dchk); break;
                case 2: OPT((fill_check<OFFEN>)); hipLaunchKernelGGL((fill_check<OFFEN>), dim3(1), dim3(256), lds, 0, ds, dchk); break;
                case 3: OPT((fill_check<ADDTID>)); hipLaunchKernelGGL((fill_check<ADDTID>), dim3(1), dim3(256), lds, 0, ds, dchk); break;
            }
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: fill_check failed\n", names[v]); return 1; }
            hipMemcpy(hc.data(), dchk, SLOT, hipMemcpyDeviceToHost);
            size_t bad = 0, first = (size_t)-1;
            for (size_t i = 0; i < SLOT / 4; ++i) if (hc[i] != hs[SLOT / 4 + i]) { if (!bad) first = i; ++bad; }
            printf("%-7s slot image: %zu of %d dwords wrong", v == 4 ? "4xdword" : names[v], bad, SLOT / 4);
            if (bad) printf(" (first at dword %zu: got %08x want %08x)", first, hc[first], hs[SLOT / 4 + first]);
            printf("\n");
        }
    }
    // then the whole probe, one launch each, the cheap way to find a wrong descriptor before timing anything
    for (int v = 0; v < 5; ++v) {
        hipMemset(dout, 0, on * 4);
        switch (v) {
            case 4:
            case 0: hipLaunchKernelGGL((probe<VADDR>), dim3(grid), dim3(256), lds, 0, ds, db, dout, 1); break;
            case 1: hipLaunchKernelGGL((probe<SADDR>), dim3(grid), dim3(256), lds, 0, ds, db, dout, 1); break;
            case 2: hipLaunchKernelGGL((probe<OFFEN>), dim3(grid), dim3(256), lds, 0, ds, db, dout, 1); break;
            case 3: hipLaunchKernelGGL((probe<ADDTID>), dim3(grid), dim3(256), lds, 0, ds, db, dout, 1); break;
        }
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", names[v]); return 1; }
        hipMemcpy(v == 0 ? ref.data() : got.data(), dout, on * 4, hipMemcpyDeviceToHost);
        if (v > 0) {
            size_t bad = 0;
            for (size_t i = 0; i < on; ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
            printf("%-7s sums %s vaddr's (%zu of %zu differ)\n", v == 4 ? "vaddr again" : names[v], bad ? "DIFFER from" : "equal", bad, on);
        }
    }
    for (int round = 0; round < 5; ++round)
        for (int v = 0; v < 13; ++v) {
            float ms = 0;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                switch (v) {
                    case 0: hipLaunchKernelGGL((probe<VADDR>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 1: hipLaunchKernelGGL((probe<SADDR>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 2: hipLaunchKernelGGL((probe<OFFEN>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 3: hipLaunchKernelGGL((probe<ADDTID>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 4: hipLaunchKernelGGL((probe<NONE>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 5: hipLaunchKernelGGL((probe<VADDR, true>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 6: hipLaunchKernelGGL((probe<ADDTID, true>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 7: hipLaunchKernelGGL((probe<VADDR, false, 1>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 8: hipLaunchKernelGGL((probe<VADDR, false, 2>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 9: hipLaunchKernelGGL((probe<ADDTID, false, 1>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 10: hipLaunchKernelGGL((probe<NONE, false, 1>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 11: hipLaunchKernelGGL((probe<DWORD>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 12: hipLaunchKernelGGL((probe<DWORDI>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1); if (rep >= 2) ms += t / 4;
            }
            if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", names[v]); return 1; }
            printf("round %d %-17s %8.4f ms  %6.0f TFLOP/s (%.3f of 2.5 PF)\n", round, names[v], ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15);
        }
    return 0;
}
