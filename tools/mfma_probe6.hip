// mfma_probe6: the operand paths of the bf16 fused-MLP kernel's A fragments (the weight stream), stand-alone, on random bf16 data.
//
// The kernel (csrc/mlp_bf16.hip) moves every weight quad (1 KiB = the A fragment of four v_mfma_f32_16x16x32_bf16) L2 -> LDS by
// LDS-DMA once per workgroup and pass, and every wave reads it back with one ds_read_b128.  Round 2 read the kernel's 6.5 TB/s of
// L2 -> LDS traffic as "the chip's LDS-DMA fill rate" (MI355X_MICROARCH.md quotes 6.4 TB/s for ONE loader wave per CU).  This probe
// measures, with the kernel's geometry (256 workgroups x 4 waves, one wave per SIMD, 3 x 32 KiB ring, 8 DMAs per wave and slot, a
// barrier per slot, 1 184-quad = 1.21 MB stream that stays in L2, B fragments in registers, 4 MFMAs per quad):
//   dma       the fill path alone: every wave issues its DMAs back to back (no MFMA, no LDS reads)      -> chip-wide L2 -> LDS TB/s
//   ring      the kernel's scheme: all quads through the ring
//   hyb1 / 2  1 / 2 of every 4 quads by global_load_dwordx4 straight into VGPRs (every wave its own copy; the CU's L1 serves three
//             of the four waves), the rest through the ring; direct loads issued PD quads ahead
//   direct    all quads straight into VGPRs, no LDS at all
//   lds       no refill at all: a static 96 KiB LDS image (the MFMA + ds_read_b128 ceiling of this operand pattern)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe6.hip -o tools/mfma_probe6.bin && tools/mfma_probe6.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int QUAD = 1024, SLOTQ = 32, SLOT = SLOTQ * QUAD, NSLOT = 3, NQ = 1184, RD = 3;

template <int IMM>
__device__ __forceinline__ void dma16(const char* g) { asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(g), "i"(IMM) : "memory"); }
__device__ __forceinline__ void set_m0(unsigned v) {
    const unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(s) : "memory");
}
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void sfor(F&& f) { if constexpr (B < E) { f(IC<B>{}); sfor<B + 1, E>(f); } }

struct Ring { const char* sbase; unsigned fetch_off, fetch_lds, lds_lo, lds_hi, read_slot, read_off, bytes; };
template <int NDIR>
__device__ __forceinline__ void ring_dma(const Ring& r, int i, int lane) {      // DMA i (0..7) of this wave's 8 KiB share of the slot being fetched
    if ((i & 3) < NDIR) return;                                                 // that quad travels by direct loads
    const char* g = r.sbase + r.fetch_off + lane * 16 + (i >= 4 ? 4096 : 0);
    if (i == 0 || (i == NDIR && NDIR > 0)) set_m0(r.fetch_lds);
    if (i == 4 || (i == 4 + NDIR && NDIR > 0)) set_m0(r.fetch_lds + 4096);
    if ((i & 3) == 0) dma16<0>(g); else if ((i & 3) == 1) dma16<1024>(g); else if ((i & 3) == 2) dma16<2048>(g); else dma16<3072>(g);
}
__device__ __forceinline__ void ring_next(Ring& r) {
    r.fetch_off += SLOT; if (r.fetch_off >= r.bytes) r.fetch_off = 0;
    r.fetch_lds += SLOT; if (r.fetch_lds >= r.lds_hi) r.fetch_lds = r.lds_lo;
}
template <int NDIR>
__device__ __forceinline__ void ring_advance(Ring& r) {
    // everything but this phase's DMAs has landed (direct loads are younger or already consumed: a stricter wait at worst)
    if constexpr (NDIR == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (NDIR == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ring_next(r);
    r.read_slot = r.read_slot + 1 == NSLOT ? 0 : r.read_slot + 1;
    r.read_off += SLOT; if (r.read_off >= r.bytes) r.read_off = 0;
}

// NDIR of every 4 quads direct (0: the kernel's scheme, 4: no LDS at all); PD: direct loads in flight (quads ahead)
template <int NDIR, int PD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe(const char* __restrict__ stream, const u32x4* __restrict__ bin, float* out, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4 b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) b[i] = bin[i * 64 + lane];
    f32x4 c[4] = {};
    Ring r;
    r.sbase = stream + wave * 8 * QUAD; r.bytes = NQ * QUAD; r.fetch_off = 0;
    r.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8 * QUAD;
    r.lds_hi = r.lds_lo + NSLOT * SLOT; r.fetch_lds = r.lds_lo; r.read_slot = NSLOT - 1; r.read_off = r.bytes - SLOT;
    if constexpr (NDIR < 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ring_dma<NDIR>(r, i, lane);
        ring_next(r);
#pragma unroll
        for (int i = 0; i < 8; ++i) ring_dma<NDIR>(r, i, lane);
        ring_advance<NDIR>(r);
    } else { r.read_off = 0; }
    constexpr int NB = PD > RD ? PD : RD;                     // register slots of the A pipeline (position q lives in a[q % 32 % ... ])
    u32x4 a[32];                                              // indexed statically by slot position; only NB are live at a time
    const char* gl = stream + lane * 16;                      // direct loads: the wave's own copy of the quad
    auto direct_load = [&](int pos_abs_off, int q) __attribute__((always_inline)) -> u32x4 {
        unsigned off = r.read_off + pos_abs_off;              // pos_abs_off may run into the next slot(s)
        if (off >= r.bytes) off -= r.bytes;
        return *(const u32x4*)(gl + off);
    };
    // prologue: positions 0 .. lookahead-1 of the first slot
    sfor<0, SLOTQ>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        constexpr bool dir = (q & 3) < NDIR;
        if constexpr (dir && q < PD) a[q] = direct_load(q * QUAD, q);
        if constexpr (!dir && q < RD) a[q] = *(const u32x4*)(smem + r.read_slot * SLOT + lane * 16 + q * QUAD);
    });
    const int n_slots = passes * (NQ / SLOTQ);
    for (int s = 0; s < n_slots; ++s) {
        sfor<0, SLOTQ>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                c[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[q]), __builtin_bit_cast(bf16x8, b[(q & 7) * 4 + p]), c[p], 0, 0, 0);
            // ring lookahead: position q + RD (the advance rides on the first read of the next slot, as in the kernel)
            constexpr int qr = (q + RD) % SLOTQ;
            if constexpr (NDIR < 4) {
                if constexpr (q + RD == SLOTQ) ring_advance<NDIR>(r);
                if constexpr (qr >= 1 && qr <= 8) ring_dma<NDIR>(r, qr - 1, lane);
                if constexpr ((qr & 3) >= NDIR) a[qr] = *(const u32x4*)(smem + r.read_slot * SLOT + lane * 16 + qr * QUAD);
            } else if constexpr (q + RD == SLOTQ) {
                r.read_off += SLOT; if (r.read_off >= r.bytes) r.read_off = 0;
            }
            // direct lookahead: position q + PD, relative to the slot read_off points at (which moved if the advance already happened)
            constexpr int qd = (q + PD) % SLOTQ;
            if constexpr ((qd & 3) < NDIR) {
                constexpr int slots_ahead = (q + PD) / SLOTQ - ((q + RD >= SLOTQ) ? 1 : 0);
                a[qd] = direct_load(slots_ahead * SLOT + qd * QUAD, qd);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    float sum = 0;
    for (int p = 0; p < 4; ++p) for (int e = 0; e < 4; ++e) sum += c[p][e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

// the fill path alone
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dma_only(const char* __restrict__ stream, float* out, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Ring r;
    r.sbase = stream + wave * 8 * QUAD; r.bytes = NQ * QUAD; r.fetch_off = 0;
    r.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8 * QUAD;
    r.lds_hi = r.lds_lo + NSLOT * SLOT; r.fetch_lds = r.lds_lo;
    const int n_slots = passes * (NQ / SLOTQ);
    for (int s = 0; s < n_slots; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ring_dma<0>(r, i, lane);
        ring_next(r);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // two slots of this wave's DMAs in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = ((float*)smem)[threadIdx.x];
}

// static LDS image, no refill
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void lds_only(const char* __restrict__ stream, const u32x4* __restrict__ bin, float* out, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < NSLOT * SLOT / 16; i += 256) ((u32x4*)smem)[i] = ((const u32x4*)stream)[i];
    __syncthreads();
    u32x4 b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) b[i] = bin[i * 64 + lane];
    f32x4 c[4] = {};
    const char* base = smem + lane * 16;
    const int n_slots = passes * (NQ / SLOTQ);
    for (int s = 0; s < n_slots; ++s) {
#pragma unroll
        for (int q = 0; q < SLOTQ; ++q) {
            const u32x4 a = *(const u32x4*)(base + ((s % NSLOT) * SLOTQ + q) * QUAD);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                c[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[(q & 7) * 4 + p]), c[p], 0, 0, 0);
        }
    }
    float sum = 0;
    for (int p = 0; p < 4; ++p) for (int e = 0; e < 4; ++e) sum += c[p][e];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
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
    hipMalloc(&ds, hs.size() * 4); hipMalloc(&db, nb * 16); hipMalloc(&dout, 256 * 256 * 4);
    hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), nb * 16, hipMemcpyHostToDevice);
    const int lds = NSLOT * SLOT;
#define OPT(k) hipFuncSetAttribute((const void*)(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds)
    OPT((probe<0, 3>)); OPT((probe<1, 12>)); OPT((probe<2, 12>)); OPT((probe<4, 12>)); OPT((probe<1, 6>)); OPT((probe<2, 6>)); OPT(dma_only); OPT(lds_only);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int passes = 12;                                    // a fine launch of the 4096-ray batch: 12 passes per workgroup
    const double flop = (double)grid * 4 * passes * NQ * 4 * (2.0 * 16 * 16 * 32), stream_gb = (double)grid * passes * sb / 1e9;
    struct V { const char* name; int id; double dma_frac, direct_frac; };
    const V vs[] = {{"lds", 0, 0, 0}, {"dma", 1, 1, 0}, {"ring", 2, 1, 0}, {"hyb1 PD12", 3, .75, .25}, {"hyb2 PD12", 4, .5, .5}, {"hyb1 PD6", 5, .75, .25},
                    {"hyb2 PD6", 6, .5, .5}, {"direct PD12", 7, 0, 1}};
    for (int round = 0; round < 3; ++round)
        for (const V& v : vs) {
            float ms = 0;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                switch (v.id) {
                    case 0: hipLaunchKernelGGL(lds_only, dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 1: hipLaunchKernelGGL(dma_only, dim3(grid), dim3(256), lds, 0, ds, dout, passes); break;
                    case 2: hipLaunchKernelGGL((probe<0, 3>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 3: hipLaunchKernelGGL((probe<1, 12>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 4: hipLaunchKernelGGL((probe<2, 12>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 5: hipLaunchKernelGGL((probe<1, 6>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 6: hipLaunchKernelGGL((probe<2, 6>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                    case 7: hipLaunchKernelGGL((probe<4, 12>), dim3(grid), dim3(256), lds, 0, ds, db, dout, passes); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1); if (rep >= 2) ms += t / 4;
            }
            if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", v.name); return 1; }
            printf("round %d %-12s %8.4f ms", round, v.name, ms);
            if (v.id != 1) printf("  %6.0f TFLOP/s (%.3f of 2.5 PF)", flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15);
            if (v.dma_frac > 0) printf("  L2->LDS DMA %5.2f TB/s", v.dma_frac * stream_gb / ms);
            if (v.direct_frac > 0) printf("  direct loads (4 waves each) %5.2f TB/s", 4 * v.direct_frac * stream_gb / ms);
            printf("\n");
        }
    return 0;
}
