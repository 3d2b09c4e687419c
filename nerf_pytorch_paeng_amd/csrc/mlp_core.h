// mlp_core.h -- device-side building blocks shared by the fp32 MLP kernels (forward: mlp_fp32.hip,
// backward-data: mlp_train.hip): the LDS weight ring fed by LDS-DMA, the register-resident GEMM part,
// accumulator <-> B-operand moves and the small VALU helpers.  Device code only (.hip translation units).
#pragma once
#include "common.h"
#include "layout.h"

namespace minerf {

constexpr int NSLOT = 4;
constexpr int RING_BYTES = NSLOT * SLOT_BYTES;

// ---------------------------------------------------------------------------------------------
// weight ring: NSLOT x 16 KiB in LDS, filled by LDS-DMA three slots ahead of the consumer.
//
// The DMA is issued from inline asm so that hipcc neither counts it (a counted DMA makes every
// __syncthreads() drain vmcnt(0), i.e. wait for the prefetch just issued) nor moves it; we count it
// ourselves: each ring_advance issues exactly 4 global_load_lds_dwordx4 per wave and waits vmcnt(4)
// before the barrier, i.e. for everything except the 4 issued at the previous advance.
//   advance to slot p: [vmcnt(4)] -> slots <= p+1 of this wave's share have landed; barrier -> of every
//   wave's share, and every wave has completed its LDS reads of slot p-1 (lgkmcnt(0) is part of the
//   barrier); then fetch slot p+3 into ring[(p+3)&3] == ring[(p-1)&3].
// Compiler-counted loads/stores elsewhere in the kernel only ever over-wait because of the uncounted DMA
// (vmcnt retires in order), never under-wait.
// ---------------------------------------------------------------------------------------------
struct WRing {
    const char* sbase;      // wave-uniform global pointer: stream + wave*4 KiB (the wave's quarter of every slot)
    unsigned voff;          // per-lane byte offset inside a quad: lane*16
    unsigned fetch_off;     // stream byte offset of the slot being fetched
    unsigned stream_bytes;
    unsigned fetch_lds;     // LDS byte address (wave-uniform) this wave's share of the next fetch lands at
    unsigned lds_lo, lds_hi;  // this wave's share of ring slot 0 / one past the last slot
    unsigned read_slot;     // ring slot being consumed
#ifdef MN_DIAG
    unsigned long long* dlog;   // fine-grained stamp log (one tile of one wave)
    unsigned dcnt;
#endif
};

#ifdef MN_DIAG
// diagnostic build only (never shipped, never timed): s_memtime stamps around the kernel's segments.  A stamp drains
// the LDS queue, so it perturbs what it measures (~100+ cycles each): read SHARES and DIFFERENCES, not totals.
__device__ __forceinline__ unsigned long long mn_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MN_STAMP(i) do { const unsigned long long t_ = mn_stamp(); seg[i] += t_ - tprev; tprev = t_; } while (0)
#define MN_KQ_STAMP(ring) do { if ((ring).dlog && (ring).dcnt < 500) { (ring).dlog[(ring).dcnt] = mn_stamp(); (ring).dcnt++; } } while (0)
#else
#define MN_STAMP(i) do {} while (0)
#define MN_KQ_STAMP(ring) do {} while (0)
#endif

// one global_load_lds_dwordx4: 64 lanes x 16 B = one 1 KiB quad.  SGPR-base form: global address = sbase + voff + IMM,
// LDS destination = M0 + IMM + lane*16 (the instruction offset applies to BOTH sides).  M0 is compiler-reserved:
// saved and restored inside the statement.
template <int IMM>
__device__ __forceinline__ void dma16(const char* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr), "i"(IMM)
        : "memory");
}

// DMA number I (0..3) of the slot being fetched
template <int I>
__device__ __forceinline__ void ring_dma(const WRing& r) {
    dma16<I * QUAD_BYTES>(r.sbase + r.fetch_off, r.voff, r.fetch_lds);
}

__device__ __forceinline__ void ring_next_fetch(WRing& r) {
    r.fetch_off += SLOT_BYTES;
    if (r.fetch_off >= r.stream_bytes) r.fetch_off = 0;
    r.fetch_lds += SLOT_BYTES;
    if (r.fetch_lds >= r.lds_hi) r.fetch_lds = r.lds_lo;
}

// Start consuming the next slot.  The four DMAs of a slot are NOT issued here in a burst (every extra issue slot
// between two MFMAs beyond ~8 delays the matrix pipe, tools/mfma_probe.hip): ring_read() issues DMA k together
// with the read of the slot's quad k+1, i.e. spread over the four groups that follow the barrier.
//
// ALLOW = operations that may still be in flight.  The slot being entered (p) is followed in the queue by the DMAs of
// slots p+1 and p+2 (4 each), so any ALLOW <= 8 is correct; 4 keeps one slot of slack (inference kernels).  The
// training kernels interleave one row store per k-quad with the DMAs (stores share vmcnt and retire in order): with
// ALLOW = 8 a store has one to two slot phases (4-8k cycles) to reach memory before an advance waits for it.
template <int ALLOW = 4>
__device__ __forceinline__ void ring_advance(WRing& r) {
    static_assert(ALLOW >= 0 && ALLOW <= 8, "more than 8 outstanding operations could include the entered slot's DMAs");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ALLOW) : "memory");
    __syncthreads();
    ring_next_fetch(r);
    r.read_slot = (r.read_slot + 1) & (NSLOT - 1);
}

// qs is a compile-time constant at every call site once the GEMM loops are unrolled: the chain below folds away
__device__ __forceinline__ f32x4 ring_read(const char* smem, const WRing& r, int lane, int qs) {
    if (qs == 1) ring_dma<0>(r);
    else if (qs == 2) ring_dma<1>(r);
    else if (qs == 3) ring_dma<2>(r);
    else if (qs == 4) ring_dma<3>(r);
    return *(const f32x4*)(smem + r.read_slot * SLOT_BYTES + lane * 16 + qs * QUAD_BYTES);
}

// ---------------------------------------------------------------------------------------------
// one GEMM part: acc[0..NT) += A(stream) x B, B = KS per-lane registers b[0..KS).
// `a` is the A-operand pipeline: on entry it holds this part's quads (kq=0, t<NT), already read from
// LDS; after the 4 MFMAs that consume a[t] the quad NT positions further down the stream is read into
// it, so every LDS read has a full k-quad (NT*4 MFMAs) of lead time.  On exit `a` holds the first
// NT_NEXT quads of the NEXT part (parts are slot aligned, the stream is consumed strictly in order).
// MFMA order is t-major: 4 back-to-back dependent MFMAs per tile (dependent issue = 64 cycles = the
// issue interval of v_mfma_f32_32x32x2_f32, so the chain costs nothing).
// ---------------------------------------------------------------------------------------------
struct NoHook { __device__ __forceinline__ void operator()(int, int) const {} };

// `hook(kq, t)` runs inside the pinned group (kq, t) after its MFMAs and LDS read, with compile-time-constant
// arguments once the loops are unrolled: the training kernels use it to spread their row stores (and the ReLU-mask
// packing) over the GEMM that consumes the same registers, instead of a burst at the layer boundary that would block
// the texture-address path (shared by the 4 waves and by the weight DMAs) for ~9k cycles per layer.
template <int NT, int KS, int NT_NEXT, int ALLOW = 4, typename Hook = NoHook, int NB = 0>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[8], const float (&b)[NB], f32x4 (&a)[8], const char* smem,
                                          WRing& ring, int lane, Hook hook = Hook()) {
    static_assert(KS % 4 == 0 && KS <= NB, "k-steps come in quads");
    constexpr int KQ = KS / 4;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[4 * kq + j], acc[t], 0, 0, 0);
            if (kq + 1 < KQ) {
                const int q = (kq + 1) * NT + t;                       // quad index inside this part
                if (q % SLOT_QUADS == 0) ring_advance<ALLOW>(ring);
                a[t] = ring_read(smem, ring, lane, q % SLOT_QUADS);
            } else if (t < NT_NEXT) {
                if (t == 0) ring_advance<ALLOW>(ring);                 // next part starts a fresh slot
                a[t] = ring_read(smem, ring, lane, t);
            }
            hook(kq, t);
            // pin the (4 MFMA, 1 LDS read) group order: left alone, hipcc sinks each read to just before
            // its first use and exposes the LDS latency on every tile
            __builtin_amdgcn_sched_barrier(0);
        }
        MN_KQ_STAMP(ring);
    }
#pragma unroll
    for (int t = NT; t < NT_NEXT; ++t) a[t] = ring_read(smem, ring, lane, t);
}

// acc tile t <- 32 floats of a natural-order vector in LDS (bias): register r of lane half hh holds
// feature 32t + (r&3) + 8*(r>>2) + 4*hh  -> four 16-byte reads at 32t + 8g + 4hh.
template <int NT>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[8], const float* vec_lds, int hh) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(vec_lds + 32 * t + 8 * g + 4 * hh);
            acc[t][4 * g + 0] = v[0]; acc[t][4 * g + 1] = v[1]; acc[t][4 * g + 2] = v[2]; acc[t][4 * g + 3] = v[3];
        }
}

// ReLU as ONE integer instruction, pinned where it is written (volatile): max_i32(bits, 0) keeps every non-negative
// float (and +NaN) and maps negatives and -0.0 to +0.0.  A plain fmaxf() costs two instructions (hipcc canonicalises
// the MFMA output first) and hipcc defers them to directly in front of the MFMA that consumes the value, which then
// waits out the VALU latency plus hazard nops: ~20 cycles per k-step in the trunk layers (MN_DIAG stamps).
__device__ __forceinline__ float relu_pinned(float x) {
    float r;
    asm volatile("v_max_i32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int NT, bool RELU, int NB>
__device__ __forceinline__ void acc_to_b(const f32x16 (&acc)[8], float (&h)[NB]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[16 * t + r] = RELU ? relu_pinned(acc[t][r]) : acc[t][r];
}

// sum_i h[i] * w[feat(i, hh)] over this lane's half; w natural order in LDS
template <int N, int NB>
__device__ __forceinline__ float dot_half(const float (&h)[NB], const float* w_lds, int hh) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {        // q = 4t + g : features 32t + 8g + 4hh + {0..3}
        const f32x4 w = *(const f32x4*)(w_lds + 8 * q + 4 * hh);
        s0 = __builtin_fmaf(h[4 * q + 0], w[0], s0);
        s1 = __builtin_fmaf(h[4 * q + 1], w[1], s1);
        s2 = __builtin_fmaf(h[4 * q + 2], w[2], s2);
        s3 = __builtin_fmaf(h[4 * q + 3], w[3], s3);
    }
    return (s0 + s1) + (s2 + s3);
}

// row-major copy of a B-operand register set: registers 16t+4g..+3 of lane half hh are features 32t+8g+4hh+{0..3}
// of this lane's point -> one 16-byte store each.  `row` already includes the + 4*hh.
template <int NT, int NB>
__device__ __forceinline__ void store_rows(const float (&h)[NB], float* row, bool valid) {
    if (valid) {
#pragma unroll
        for (int q = 0; q < 4 * NT; ++q) {
            f32x4 v; v[0] = h[4 * q + 0]; v[1] = h[4 * q + 1]; v[2] = h[4 * q + 2]; v[3] = h[4 * q + 3];
            *(f32x4*)(row + 8 * q) = v;
        }
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// chunk q (registers 4q..4q+3 = features 8q + 4hh + {0..3}) of a B-operand register set -> its row (one 16-byte store).
// Unconditional: an exec-masked store costs a saveexec/branch/restore triple inside the MFMA stream.  Callers make the
// padding lanes (sample >= S, inactive tail waves) recompute a VALID point, so their stores write the same bytes again.
template <int NB>
__device__ __forceinline__ void store_chunk(const float (&h)[NB], int q, float* row) {
    f32x4 v; v[0] = h[4 * q + 0]; v[1] = h[4 * q + 1]; v[2] = h[4 * q + 2]; v[3] = h[4 * q + 3];
    *(f32x4*)(row + 8 * q) = v;
}

// ReLU' bit masks.  Registers are packed in order, each word shifted left by one per value: register 4q+e lands in bit
// 31 - (4*(q&7)+e) of word q>>3 (words always fill: 8 chunks each).  The registers are post-ReLU (+0.0 or positive,
// relu_pinned), so "non-zero bit pattern" == "> 0": min(bits, 1) shifted in, 2 VALU per value.  Pinned with asm volatile:
// written in C++, hipcc defers all 128 packs of a layer to the end of the GEMM (~2200 exposed cycles per layer).
template <int NB>
__device__ __forceinline__ void mask_bits_chunk(const float (&h)[NB], int q, unsigned (&bits)[4]) {   // 4 independent VALU
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("v_min_u32 %0, 1, %1" : "=v"(bits[e]) : "v"(h[4 * q + e]));
}
template <int NW>
__device__ __forceinline__ void mask_merge_chunk(const unsigned (&bits)[4], int q, unsigned (&mw)[NW]) {   // depth-3 tree, 4 VALU
    unsigned t01, t23;
    asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(t01) : "v"(bits[0]), "v"(bits[1]));
    asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(t23) : "v"(bits[2]), "v"(bits[3]));
    asm volatile("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(t01) : "v"(t01), "v"(t23));
    asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(mw[q >> 3]) : "v"(t01));
}
template <int NB, int NW>
__device__ __forceinline__ void mask_pack_chunk(const float (&h)[NB], int q, unsigned (&mw)[NW]) {
    unsigned bits[4];
    mask_bits_chunk(h, q, bits);
    mask_merge_chunk(bits, q, mw);
}
// register 4q+e masked by its bit: extract the bit sign-extended (0 / ~0) and AND the float's bits -- two VALU per value, pinned:
// from the C++ shifts hipcc builds and + compare + select (three; VALU work beside fp32 MFMAs is paid in MFMA cycles).
template <int NW>
__device__ __forceinline__ float mask_apply(float v, int q, int e, const unsigned (&mw)[NW]) {
    int m;
    float r;
    asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(mw[q >> 3]), "s"(31 - (4 * (q & 7) + e)));
    asm volatile("v_and_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
    return r;
}

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

}  // namespace minerf
