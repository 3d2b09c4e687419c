// mlp_f16s_core.h -- the split-precision machinery shared by the translation units of that variant (mlp_f16s.hip: host packers + the
// inference launch; mlp_f16s_stash.hip: the training forward; dgrad_f16s.hip: the backward-data chain): constants, blob layout, the weight
// ring, the fragment file, MFMA wrappers, packing schedule, job(), and the forward kernel template.  See mlp_f16s.hip for the design.
#pragma once
#include <string.h>
#include <type_traits>
#include <vector>
#include "common.h"
#include "layout.h"

// where the next unit's inputs are requested in the view-direction layer (job, k-step): behind a ring advance (pair 161 of the tail = slot 10, quad 2)
namespace minerf {
namespace f16s {
constexpr int PF_T = 3, PF_KS = 1;

typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x2b __attribute__((ext_vector_type(2)));

constexpr int NP = 2;                                      // point tiles (16 points each) per wave
constexpr int MT = 16, KF = 32;                            // output features per job, k per MFMA
constexpr int NBUF = 4, LA = NBUF - 1;                     // A-operand pipeline: quad PAIRS in flight (a body's pair count and the tail's padding jump are multiples of 4)
constexpr int SLOT_QUADS_S = 32, SLOT_BYTES_S = SLOT_QUADS_S * QUAD_BYTES, NSLOT_S = 3, RING_BYTES_S = NSLOT_S * SLOT_BYTES_S;
constexpr int DMA_PER_WAVE = SLOT_QUADS_S / 4;
constexpr int KPE = 2, KH = 8, NT = 16;                    // k-steps over gamma(x) (63 -> 64 channels), over a 256-wide activation; output tiles of a 256-wide layer
constexpr int TAIL_USED_P = 128 + 8 + 64 + 4;              // quad PAIRS of the tail body that carry weights
constexpr int TAIL_PAIRS = 208;                            // ... padded to whole slots (416 quads = 13 slots)
constexpr float SC_DN = 1.0f / 2048.0f, SC_UP = 2048.0f;   // 2^-11, 2^11

struct BlobLayoutS {
    uint32_t stream_off, stream_bytes, side_off, side_floats;
    uint32_t bias_trunk, bias_feat, bias_d, head_b, wdir_t, total_bytes;
};
static BlobLayoutS make_layout(int D, int W, int skip) {
    BlobLayoutS b{};
    const int in_d = 3 + 6 * KERNEL_LD;
    uint32_t pairs = KPE * NT;
    for (int l = 1; l < D; ++l) pairs += KH * NT + ((skip >= 0 && l == skip + 1) ? KPE * NT : 0);
    pairs += TAIL_PAIRS;
    b.stream_off = HEADER_BYTES;
    b.stream_bytes = pairs * 2 * QUAD_BYTES;
    b.side_off = b.stream_off + b.stream_bytes;
    uint32_t f = 0;
    b.bias_trunk = f; f += (uint32_t)D * W;
    b.bias_feat = f;  f += W;
    b.bias_d = f;     f += W / 2;
    b.head_b = f;     f += 4;                       // colour bias (3), density bias
    b.wdir_t = f;     f += (uint32_t)in_d * (W / 2);
    b.side_floats = round_up_u32(f, 4);
    b.total_bytes = b.side_off + b.side_floats * 4;
    return b;
}

// shapes this variant is built for
static int check_net(const mi_nerf_net* net) {
    MN_CHECK_ARG(net != nullptr, "net is NULL");
    MN_CHECK_ARG(net->W == 256, "the f16-split variant is built for W=256 only (got %d)", net->W);
    MN_CHECK_ARG(net->D >= 2 && net->D <= 16 && net->L_x >= 0 && net->L_x <= KERNEL_LX && net->L_d >= 0 && net->L_d <= KERNEL_LD && net->skip >= -1,
                 "unsupported network for the f16-split variant (D=%d L_x=%d L_d=%d skip=%d)", net->D, net->L_x, net->L_d, net->skip);
    return MI_NERF_OK;
}
static uint32_t bwd_stream_bytes_s(int D) { return (uint32_t)(NT * (KH / 2) + NT * KH * D) * 2u * QUAD_BYTES; }


// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
struct Args {
    const char* stream;
    const float* side;
    const float* rays;
    const float* z;
    float* out;
    unsigned n_wtiles;          // 32-point tiles (n_rays * tpr); a wave's unit of work is ONE tile = 2 point tiles
    unsigned n_iter;            // units per wave (the same for every wave: the ring barriers are workgroup-wide)
    unsigned ppr;               // ray-major walk: units per ray (tpr); 0: flat walk
    int S, tpr, D, skip_layer;
    unsigned stream_bytes, side_floats;
    unsigned o_bias_trunk, o_bias_feat, o_bias_d, o_head_b, o_wdir_t;
    // STASH instantiation (training forward): what mlp_fp32_kernel<..., STASH = true> leaves for the backward pass, in ITS layouts
    float* stash_h;             // [D][P][W]   post-ReLU output of trunk layer l
    float* stash_f;             // [P][W]      linear_feat output
    float* stash_g;             // [P][W/2]    post-ReLU linear_d output
    unsigned* mask_h;           // [D][n_wtiles][64][4]  ReLU' bits of stash_h in the fp32 kernel's lane / register order (mlp_core.h mask_pack_chunk)
    unsigned* mask_g;           // [n_wtiles][64][2]     ... of stash_g
    long long stash_rows;       // P
};

struct Ring {
    const char* sbase;      // stream + wave's 8 KiB share
    unsigned voff;          // lane*16
    unsigned fetch_off, stream_bytes;
    unsigned fetch_lds, lds_lo, lds_hi;
    unsigned read_slot;
};
// LDS-DMA of the weight stream (see mlp_bf16.hip): M0 is ours alone in this kernel
__device__ __forceinline__ void set_m0(unsigned lds_in) {
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_addr) : "memory");
}
template <int IMM>
__device__ __forceinline__ void dma16(const char* gaddr_lane) {
    asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gaddr_lane), "i"(IMM) : "memory");
}
__device__ __forceinline__ void ring_dma(const Ring& r, int i) {
    const char* g = r.sbase + r.fetch_off + r.voff + (i >= 4 ? 4096 : 0);
    if (i == 0) set_m0(r.fetch_lds);
    if (i == 4) set_m0(r.fetch_lds + 4096);
    if ((i & 3) == 0) dma16<0>(g);
    else if ((i & 3) == 1) dma16<1024>(g);
    else if ((i & 3) == 2) dma16<2048>(g);
    else dma16<3072>(g);
}
__device__ __forceinline__ void ring_next_fetch(Ring& r) {
    r.fetch_off += SLOT_BYTES_S;
    if (r.fetch_off >= r.stream_bytes) r.fetch_off = 0;
    r.fetch_lds += SLOT_BYTES_S;
    if (r.fetch_lds >= r.lds_hi) r.fetch_lds = r.lds_lo;
}
__device__ __forceinline__ void ring_advance(Ring& r) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    ring_next_fetch(r);
    r.read_slot = (r.read_slot + 1 == NSLOT_S) ? 0 : r.read_slot + 1;
}
// quad at slot position qs; positions 1..8 also issue one of the slot's DMAs (never a burst)
__device__ __forceinline__ u32x4b ring_read(const char* smem, const Ring& r, int lane, int qs) {
    if (qs >= 1 && qs <= DMA_PER_WAVE) ring_dma(r, qs - 1);
    return *(const u32x4b*)(smem + r.read_slot * SLOT_BYTES_S + lane * 16 + qs * QUAD_BYTES);
}

template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}
// THE FRAGMENT FILE (all 256 AGPRs, hand-numbered; see mlp_bf16.hip): set, point tile, fragment, part (0 hi, 1 lo)
__host__ __device__ constexpr int frag_reg(int set, int p, int f, int part) { return ((((set * NP + p) * 8 + f) * 2) + part) * 4; }
__host__ __device__ constexpr int tile_reg(int set, int p, int t, int part) { return frag_reg(set, p, t >> 1, part) + 2 * (t & 1); }

// MFMAs as asm statements (hipcc allocates only the VGPR side).  B operand: a fragment-file register (IC<R>) or a VGPR fragment.
template <int R>
__device__ __forceinline__ void mfma_c(f32x4& acc, const u32x4b& afrag, IC<R>, const f32x4& c) {          // acc = A B + c
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, a[%3:%4], %2" : "=&v"(acc) : "v"(afrag), "v"(c), "n"(R), "n"(R + 3));
}
__device__ __forceinline__ void mfma_c(f32x4& acc, const u32x4b& afrag, const u32x4b& bfrag, const f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(afrag), "v"(bfrag), "v"(c));
}
template <int R>
__device__ __forceinline__ void mfma_z(f32x4& acc, const u32x4b& afrag, IC<R>) {                          // acc = A B
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, a[%2:%3], 0" : "=&v"(acc) : "v"(afrag), "n"(R), "n"(R + 3));
}
__device__ __forceinline__ void mfma_z(f32x4& acc, const u32x4b& afrag, const u32x4b& bfrag) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(afrag), "v"(bfrag));
}
template <int R>
__device__ __forceinline__ void mfma_a(f32x4& acc, const u32x4b& afrag, IC<R>) {                          // acc += A B
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(afrag), "n"(R), "n"(R + 3));
}
__device__ __forceinline__ void mfma_a(f32x4& acc, const u32x4b& afrag, const u32x4b& bfrag) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(afrag), "v"(bfrag));
}

// ---- packing: a pair of accumulator elements (registers 2e, 2e+1 of a finished tile) -> one dword of the hi fragment, one of the lo
// fragment.  Six stages of two instructions (one stage per MFMA gap), or as one block where a job has more work than gaps.
// RANGE CONTRACT (round 4): an activation y >= 65520 converts to hi = +inf, lo = -inf; every unit of the next layer then accumulates
// inf - inf = NaN -- and the ReLU here is gfx950's NaN-PROPAGATING maximum (v_pk_maximum3_f16 / v_maximum3_f32, IEEE 754-2019), not
// v_pk_max_f16 / v_max_f32, which return the other operand for a NaN and turned the whole layer into plausible zeros.  So an activation
// beyond the f16 range yields NaN in every output that depends on it (all four for a trunk unit; the three colours for a feature / direction
// unit), never a finite value; y <= -65520 in front of a ReLU is exact (hi = lo = 0, what fp32 gives).  tests/test_gpu_f16s.py
// test_f16s_forward_out_of_range_is_never_a_finite_colour; same instruction count as before.
struct PairTmp { float y0, y1, Y0, Y1; unsigned hi, lo; };
template <bool RELU, int RH, int RL, int STAGE>
__device__ __forceinline__ void pack_stage(const float h0, const float h1, const float l0, const float l1, PairTmp& t, float dn, float up, float nup) {
    if constexpr (STAGE == 0) asm volatile("v_fma_f32 %0, %2, %6, %3\n\tv_fma_f32 %1, %4, %6, %5" : "=&v"(t.y0), "=&v"(t.y1) : "v"(l0), "v"(h0), "v"(l1), "v"(h1), "s"(dn));
    else if constexpr (STAGE == 1) asm volatile("v_fma_f32 %0, %2, %6, %3\n\tv_fma_f32 %1, %4, %6, %5" : "=&v"(t.Y0), "=&v"(t.Y1) : "v"(h0), "v"(l0), "v"(h1), "v"(l1), "s"(up));
    else if constexpr (STAGE == 2) {
        if (RELU) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_maximum3_f16 %0, %0, 0, 0" : "=&v"(t.hi) : "v"(t.y0), "v"(t.y1));
        else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=&v"(t.hi) : "v"(t.y0), "v"(t.y1));
    } else if constexpr (STAGE == 3) {
        if (RELU) asm volatile("v_maximum3_f32 %0, %0, 0, 0\n\tv_maximum3_f32 %1, %1, 0, 0" : "+v"(t.Y0), "+v"(t.Y1));
    } else if constexpr (STAGE == 4) {
        // lo halves: f16((y - hi) * 2^11) = f16(fma(hi, -2^11, Y)): the residual is exact, rounded once.  mixlo writes bits 15:0 of the
        // destination, mixhi bits 31:16 (each keeps the other half): the pair lands packed.
        asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(t.lo) : "v"(t.hi), "s"(nup), "v"(t.Y0), "v"(t.Y1));
    } else asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1" ::"v"(t.hi), "v"(t.lo), "n"(RH), "n"(RL));
}
template <bool RELU, int RH, int RL>
__device__ __forceinline__ void pack_block(const float h0, const float h1, const float l0, const float l1, PairTmp& t, float dn, float up, float nup) {
    pack_stage<RELU, RH, RL, 0>(h0, h1, l0, l1, t, dn, up, nup); pack_stage<RELU, RH, RL, 1>(h0, h1, l0, l1, t, dn, up, nup);
    pack_stage<RELU, RH, RL, 2>(h0, h1, l0, l1, t, dn, up, nup); pack_stage<RELU, RH, RL, 3>(h0, h1, l0, l1, t, dn, up, nup);
    pack_stage<RELU, RH, RL, 4>(h0, h1, l0, l1, t, dn, up, nup); pack_stage<RELU, RH, RL, 5>(h0, h1, l0, l1, t, dn, up, nup);
}
// The standard schedule (jobs of >= 8 k-steps): the four packing gaps of groups 1..6 (sub-steps 2..5) carry the 24 stages of the
// previous job's two tiles x two pairs: slot n = 4 (KS - 1) + (SUB - 2) -> pair n / 6 (tile p = pair >> 1, element pair e = pair & 1),
// stage n % 6.  Done by group 6; the half fragment packed across a layer boundary feeds k-step 7.  A tile's first stage is >= 8 MFMA
// issues behind the MFMA that finished it.
template <bool RELU, int SET, int T, int KS, int SUB>
__device__ __forceinline__ void pack_sched(const f32x4 (&ph)[NP], const f32x4 (&pl)[NP], PairTmp (&t)[4], float dn, float up, float nup) {
    if constexpr (KS >= 1 && KS <= 6 && SUB >= 2) {
        constexpr int n = 4 * (KS - 1) + (SUB - 2), pair = n / 6, stage = n % 6, p = pair >> 1, e = pair & 1;
        pack_stage<RELU, tile_reg(SET, p, T, 0) + e, tile_reg(SET, p, T, 1) + e, stage>(ph[p][2 * e], ph[p][2 * e + 1], pl[p][2 * e], pl[p][2 * e + 1], t[pair], dn, up, nup);
    }
}
// ... and the block form for short jobs: pair n (0..3) whole, in one gap
template <bool RELU, int SET, int T, int PAIR>
__device__ __forceinline__ void pack_pair_block(const f32x4 (&ph)[NP], const f32x4 (&pl)[NP], PairTmp (&t)[4], float dn, float up, float nup) {
    constexpr int p = PAIR >> 1, e = PAIR & 1;
    pack_block<RELU, tile_reg(SET, p, T, 0) + e, tile_reg(SET, p, T, 1) + e>(ph[p][2 * e], ph[p][2 * e + 1], pl[p][2 * e], pl[p][2 * e + 1], t[PAIR], dn, up, nup);
}

// ---- STASH (training forward): a packed tile also leaves its four values per lane (features 16 T + 4 q4 + {0..3} of one point) as a
// 16-byte piece of the point's row, and -- ReLU layers -- their ReLU' bits in the layout mlp_dgrad_kernel reads.  That kernel's lane
// (j, hh) holds, for point j of the 32-point tile, the features f with (f >> 2) & 1 == hh, bit 31 - (4 ((f >> 3) & 7) + (f & 3)) of word
// f >> 6 (mlp_core.h mask_pack_chunk).  Here lane (q4, col) of point tile p holds f = 16 T + 4 q4 + i: hh = q4 & 1, word T >> 2, nibble
// (2 T + (q4 >> 1)) & 7 counted from the top.  Lanes q4 and q4 ^ 2 (lane ^ 32) fill alternate nibbles of the same words: the layer's
// words are OR-ed across that pair once, at the end (finish_masks).
#define F16S_STASH_STORE_FLAGS "nt sc1"
template <bool RELU, bool MASK, int T>
__device__ __forceinline__ void stash_tile(const PairTmp& e0, const PairTmp& e1, float* rowp, unsigned (&mw)[4], unsigned nib_sh) {
    f32x4 v;
    if constexpr (RELU) {              // operand order pinned: max(+0, -0) must come out +0 (the bits double as the mask)
        asm volatile("v_max_f32 %0, 0, %1" : "=v"(v[0]) : "v"(e0.y0));
        asm volatile("v_max_f32 %0, 0, %1" : "=v"(v[1]) : "v"(e0.y1));
        asm volatile("v_max_f32 %0, 0, %1" : "=v"(v[2]) : "v"(e1.y0));
        asm volatile("v_max_f32 %0, 0, %1" : "=v"(v[3]) : "v"(e1.y1));
    } else { v[0] = e0.y0; v[1] = e0.y1; v[2] = e1.y0; v[3] = e1.y1; }
    // streaming stores: the rows are read next by another kernel, and as ordinary stores they cost this kernel 1.4 ms of 3.6 (the L2
    // allocates a line per 64-byte piece); A/B on one box: default 3.62 ms, nt 2.98, nt sc1 2.84, sc1 4.02, sc0 sc1 4.13, no stores 2.21
    // (an asm statement: hipcc does not see a store here, so the wait states it would put between a 128-bit store and a VALU write of
    // the data registers are ours to add -- without them the next instruction's result went to memory instead of the activation)
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2 " F16S_STASH_STORE_FLAGS "\n\ts_nop 1" ::"v"(rowp), "v"(v), "n"(MT * T * 4) : "memory");
    if constexpr (MASK) {
        unsigned b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_min_u32 %0, 1, %1" : "=v"(b[i]) : "v"(v[i]));     // post-ReLU: non-zero bits == positive
        const unsigned nib = (((b[0] << 1) | b[1]) << 2) | ((b[2] << 1) | b[3]);
        mw[T >> 2] |= (nib << nib_sh) << (24 - 8 * (T & 3));
    }
}
// the layer's mask words of both point tiles, completed across the lane pair and written in the backward kernel's order; words cleared
template <int NWORD>
__device__ __forceinline__ void finish_masks(unsigned (&mw)[NP][4], unsigned* dst_tile, int col, int q4, bool active) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        unsigned w[NWORD];
#pragma unroll
        for (int k = 0; k < NWORD; ++k) { w[k] = mw[p][k] | (unsigned)__shfl_xor((int)mw[p][k], 32, 64); mw[p][k] = 0u; }
        if (active && q4 < 2) {
            unsigned* d = dst_tile + (size_t)(col + 16 * p + 32 * (q4 & 1)) * NWORD;
            if constexpr (NWORD == 4) { u32x4b m; m[0] = w[0]; m[1] = w[1]; m[2] = w[2]; m[3] = w[3]; *(u32x4b*)d = m; }
            else { d[0] = w[0]; d[1] = w[1]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One job: output tile of 16 features x NP point tiles over KS k-steps = stream quad PAIRS Q0..Q0+KS-1 of the current body.
// csel(p): C operand of the hi.hi chain (the bias).  bh / bl (p_c, ks_c): B operand hi / lo -- IC<register> or a VGPR fragment.
// Group ks: [hi.hi p0][hi.hi p1][hi.lo p0][hi.lo p1][lo.hi p0][lo.hi p1]; after the first two the ring work (advance, one A read each, the
// slot's DMAs), after every one hook(ks_c, sub_c).  QEND / QPAD: pair positions >= QEND skip QPAD pairs (the tail's padding).
// ---------------------------------------------------------------------------------------------
template <int Q0, int KS, int QEND, int QPAD, typename CSel, typename BH, typename BL, typename Hook>
__device__ __forceinline__ void job(f32x4 (&ah)[NP], f32x4 (&al)[NP], CSel csel, BH bh, BL bl, u32x4b (&a)[NBUF][2], const char* smem, Ring& ring, int lane, Hook hook) {
    static_for<0, KS>([&](auto ks_c) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_c)::value;
        constexpr int cur = (Q0 + ks) % NBUF;
        constexpr int q0 = Q0 + ks + LA;                                  // pair being fetched into a[q0 % NBUF]
        constexpr int qn = (q0 >= QEND) ? q0 + QPAD : q0;
        static_for<0, 6>([&](auto s_c) __attribute__((always_inline)) {
            constexpr int sub = decltype(s_c)::value, p = sub & 1;
            if constexpr (sub < 2) {
                if constexpr (ks == 0) mfma_c(ah[p], a[cur][0], bh(IC<p>{}, ks_c), csel(p));
                else mfma_a(ah[p], a[cur][0], bh(IC<p>{}, ks_c));
                if constexpr (sub == 0 && (2 * qn) % SLOT_QUADS_S == 0) ring_advance(ring);
                a[q0 % NBUF][sub] = ring_read(smem, ring, lane, (2 * qn + sub) % SLOT_QUADS_S);
            } else if constexpr (sub < 4) {
                if constexpr (ks == 0) mfma_z(al[p], a[cur][0], bl(IC<p>{}, ks_c));
                else mfma_a(al[p], a[cur][0], bl(IC<p>{}, ks_c));
            } else {
                mfma_a(al[p], a[cur][1], bh(IC<p>{}, ks_c));
            }
            hook(ks_c, s_c);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        });
        // the C operands stay allocated until the next group (the matrix pipe reads them after issue)
        if constexpr (ks == 0) asm volatile("" ::"v"(csel(0)), "v"(csel(1)));
    });
}

template <bool STASH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_f16s_kernel(const Args a) {
    constexpr int W = 256, LX = KERNEL_LX, LD = KERNEL_LD, IN_X = 3 + 6 * LX, IN_D = 3 + 6 * LD;
    constexpr int BIG = 1 << 30;
    asm volatile("" ::: "a255");                             // reserve the whole accumulation-register file
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* side = (float*)(smem + RING_BYTES_S);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, q4 = lane >> 4, pq = q4 & 1;  // lane quarters 2, 3 duplicate the encoding work of 0, 1
    const float dn = SC_DN, up = SC_UP, nup = -SC_UP;
    for (unsigned i = tid * 4; i < a.side_floats; i += 256 * 4) *(f32x4*)(side + i) = *(const f32x4*)(a.side + i);
    float* scratch = side + a.side_floats + wave * (W / 2);                       // per wave: hoisted direction bias of its tile's ray
    char* pe_wave = (char*)(side + a.side_floats + 4 * (W / 2)) + wave * (2 * NP * KPE * QUAD_BYTES);   // parked gamma(x) fragments [part][p][ks]
    char* pe_lds = pe_wave + lane * 16;

    Ring ring;
    ring.sbase = a.stream + wave * (DMA_PER_WAVE * QUAD_BYTES);
    ring.voff = lane * 16;
    ring.fetch_off = 0;
    ring.stream_bytes = a.stream_bytes;
    ring.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (DMA_PER_WAVE * QUAD_BYTES);
    ring.lds_hi = ring.lds_lo + RING_BYTES_S;
    ring.fetch_lds = ring.lds_lo;
    ring.read_slot = NSLOT_S - 1;
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) ring_dma(ring, i);       // slot 0
    ring_next_fetch(ring);
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) ring_dma(ring, i);       // slot 1; slot p+2 streams in while slot p is consumed
    u32x4b aq[NBUF][2];
    ring_advance(ring);                                             // also publishes the side tables (barrier)
#pragma unroll
    for (int i = 0; i < LA; ++i) { aq[i][0] = ring_read(smem, ring, lane, 2 * i); aq[i][1] = ring_read(smem, ring, lane, 2 * i + 1); }

    // ---- tile walk: a wave takes one 32-sample tile (two 16-point MFMA tiles) per unit; ray-major when every wave gets whole rays
    const unsigned NW = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    auto unit_of = [&](unsigned it) -> unsigned {
        if (a.ppr) { const unsigned blk = it / a.ppr; return (blk * NW + wid) * a.ppr + (it - blk * a.ppr); }
        return it * NW + wid;
    };
    unsigned n_tile; float nx_r[6], nx_z;
    auto load_inputs = [&](unsigned it) __attribute__((always_inline)) {
        unsigned t = unit_of(it);
        n_tile = t;
        if (t >= a.n_wtiles) t = a.n_wtiles - 1;                    // inactive: recompute the last tile, store nothing
        const unsigned ray = t / (unsigned)a.tpr, chunk = t - ray * (unsigned)a.tpr;
        const float* rp = a.rays + (size_t)ray * 6;
#pragma unroll
        for (int e = 0; e < 6; ++e) nx_r[e] = rp[e];
        const int sample = (int)chunk * 32 + 16 * pq + col;
        nx_z = a.z[(size_t)ray * a.S + (sample < a.S ? sample : a.S - 1)];
    };
    load_inputs(0);
    unsigned bias_ray = ~0u;

    f32x4 ah[NP], al[NP], ph[NP], pl[NP];                    // accumulators of the running job (hi, lo) and of the finished one
    f32x4 cin, cnext;                                        // bias of the current / next job (shared by the point tiles)
    auto csel1 = [&](int) __attribute__((always_inline)) -> const f32x4& { return cin; };
    u32x4b peh[NP][KPE], pel[NP][KPE];
    // STASH: row pointers of the tensor being written (this lane's 16-byte column of its two points), the running layer's ReLU' words
    float* rowp[NP] = {nullptr, nullptr};
    unsigned mw[NP][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    const unsigned nib_sh = (q4 >> 1) ? 0u : 4u;
    unsigned tile_cur = 0; bool tile_active = false; int lyr = 0;
    auto end_trunk_layer = [&]() __attribute__((always_inline)) {          // after the job that packed (and stashed) a trunk layer's last tile
        if constexpr (STASH) {
            finish_masks<4>(mw, a.mask_h + ((size_t)lyr * a.n_wtiles + tile_cur) * 256, col, q4, tile_active);
            ++lyr;
#pragma unroll
            for (int p = 0; p < NP; ++p) rowp[p] += a.stash_rows * W;
        }
    };

    // A trunk layer reads fragment set SIN (the second half of fragment 7 is still being packed from the previous tiles when it
    // starts), writes set 1 - SIN, leaves its last tile in ph / pl; the bias of the NEXT job is read while a job's last groups compute.
    auto trunk_layer = [&](auto skip_c, auto sin_c, const float* bias, const float* next_bias) __attribute__((always_inline)) {
        constexpr bool SKIP = decltype(skip_c)::value;
        constexpr int SIN = decltype(sin_c)::value, SOUT = 1 - SIN;
        constexpr int KS = SKIP ? KH + KPE : KH;
        u32x4b perh[NP][KPE], perl[NP][KPE];                 // skip layer: the parked gamma(x) fragments, re-read just in time
        auto bh = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> decltype(auto) {
            constexpr int p = decltype(p_c)::value, ks = decltype(ks_c)::value;
            if constexpr (ks >= KH) return (const u32x4b&)perh[p][ks - KH];
            else return IC<frag_reg(SIN, p, ks, 0)>{};
        };
        auto bl = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> decltype(auto) {
            constexpr int p = decltype(p_c)::value, ks = decltype(ks_c)::value;
            if constexpr (ks >= KH) return (const u32x4b&)perl[p][ks - KH];
            else return IC<frag_reg(SIN, p, ks, 1)>{};
        };
        static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
            constexpr int t = decltype(t_c)::value;
            PairTmp pt[4];
            auto hook = [&](auto ks_c, auto s_c) __attribute__((always_inline)) {
                constexpr int ks = decltype(ks_c)::value, sub = decltype(s_c)::value;
                if constexpr (t == 0) pack_sched<true, SIN, NT - 1, ks, sub>(ph, pl, pt, dn, up, nup);
                else pack_sched<true, SOUT, t - 1, ks, sub>(ph, pl, pt, dn, up, nup);
                if constexpr (STASH && ks == 7 && (sub == 3 || sub == 4))      // the tile just packed: its rows and ReLU' bits
                    stash_tile<true, true, (t == 0 ? NT - 1 : t - 1)>(pt[2 * (sub - 3)], pt[2 * (sub - 3) + 1], rowp[sub - 3], mw[sub - 3], nib_sh);
                if constexpr (ks == 7 && sub == 2) {              // bias of the next job (C operand of its first MFMAs)
                    const float* v = (t + 1 < NT) ? bias + MT * (t + 1) + 4 * q4 : next_bias + 4 * q4;
                    cnext = *(const f32x4*)v;
                }
                if constexpr (SKIP && ks == KH - 1 && sub >= 2) {  // gamma(x) fragments of k-steps 8, 9: one 16-byte read per gap
                    constexpr int idx = sub - 2, p = idx & 1, part = idx >> 1;
                    if constexpr (part == 0) { perh[p][0] = *(const u32x4b*)(pe_lds + ((0 * NP + p) * KPE + 0) * QUAD_BYTES); perh[p][1] = *(const u32x4b*)(pe_lds + ((0 * NP + p) * KPE + 1) * QUAD_BYTES); }
                    else { perl[p][0] = *(const u32x4b*)(pe_lds + ((1 * NP + p) * KPE + 0) * QUAD_BYTES); perl[p][1] = *(const u32x4b*)(pe_lds + ((1 * NP + p) * KPE + 1) * QUAD_BYTES); }
                }
            };
            job<t * KS, KS, BIG, 0>(ah, al, csel1, bh, bl, aq, smem, ring, lane, hook);
#pragma unroll
            for (int p = 0; p < NP; ++p) { ph[p] = ah[p]; pl[p] = al[p]; }
            cin = cnext;
            if constexpr (t == 0) end_trunk_layer();             // job 0 packed the PREVIOUS layer's last tile
        });
    };

    for (unsigned it = 0; it < a.n_iter; ++it) {
        // ---- prologue: this unit's points, gamma(x) fragments (hi, lo), hoisted view-direction bias -----------------------------
        const bool active = n_tile < a.n_wtiles;
        const unsigned tcur = active ? n_tile : a.n_wtiles - 1;
        const unsigned tray = tcur / (unsigned)a.tpr, chunk = tcur - tray * (unsigned)a.tpr;
        bool valid[NP]; size_t out_idx[NP];
#pragma unroll
        for (int h = 0; h < NP; ++h) {                           // results of point tile h land on lane quarter 0, lane = point
            const int sample = (int)chunk * 32 + 16 * h + col;
            valid[h] = active && sample < a.S;
            out_idx[h] = (size_t)tray * a.S + (sample < a.S ? sample : a.S - 1);
        }
        if constexpr (STASH) {
            tile_cur = tcur; tile_active = active; lyr = 0;
#pragma unroll
            for (int h = 0; h < NP; ++h) rowp[h] = a.stash_h + out_idx[h] * W + 4 * q4;
        }
        const float in_o[3] = {nx_r[0], nx_r[1], nx_r[2]}, in_d[3] = {nx_r[3], nx_r[4], nx_r[5]};
        const float in_z = nx_z;
        {
            // pts = rays_o + rays_d * z (nerf_process.py:69-70); gamma(x) per channel with the fp32 kernel's accurate sin / cos
            const float pt[3] = {in_o[0] + in_d[0] * in_z, in_o[1] + in_d[1] * in_z, in_o[2] + in_d[2] * in_z};
            const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(pt[0]), __builtin_fabsf(pt[1])), __builtin_fabsf(pt[2])) * (float)(1 << (LX - 1));
            const bool fast = amax < SINCOS_FAST_LIMIT;
            float sn[LX][3], cs[LX][3];
#pragma unroll
            for (int k = 0; k < LX; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float y = pt[c] * (float)(1 << k);
                    sn[k][c] = fast ? sin_cos_fast(y, 0) : sin_cos_slow(y, 0);
                    cs[k][c] = fast ? sin_cos_fast(y, 1) : sin_cos_slow(y, 1);
                }
            auto chan = [&](int u) __attribute__((always_inline)) -> float {     // channel u of gamma(x); u is a constant at every use
                if (u >= IN_X) return 0.0f;
                if (u < 3) return pt[u];
                const int k = (u - 3) / 6, r = (u - 3) % 6;
                return r < 3 ? sn[k][r] : cs[k][r - 3];
            };
            // fragment (part, point tile pq, k-step ks): lane quarter qq reads channels 32 ks + 8 qq + j of point `col` at
            // [fragment][(qq * 16 + col) * 16 bytes]: this lane writes those 16 bytes for every qq, hi and lo
            char* wr_h = pe_wave + ((0 * NP + pq) * KPE) * QUAD_BYTES + col * 16;
            char* wr_l = pe_wave + ((1 * NP + pq) * KPE) * QUAD_BYTES + col * 16;
#pragma unroll
            for (int ks = 0; ks < KPE; ++ks)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    u32x4b vh, vl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float y0 = chan(KF * ks + 8 * qq + 2 * i), y1 = chan(KF * ks + 8 * qq + 2 * i + 1);
                        unsigned hi, lo;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(y0), "v"(y1));
                        const float Y0 = y0 * SC_UP, Y1 = y1 * SC_UP;
                        asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                     : "=&v"(lo) : "v"(hi), "s"(nup), "v"(Y0), "v"(Y1));
                        vh[i] = hi; vl[i] = lo;
                    }
                    *(u32x4b*)(wr_h + ks * QUAD_BYTES + qq * 256) = vh;
                    *(u32x4b*)(wr_l + ks * QUAD_BYTES + qq * 256) = vl;
                }
        }
        // LDS operations of one wave execute in order: the fragments written above are complete when these reads return
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ks = 0; ks < KPE; ++ks) {
                peh[p][ks] = *(const u32x4b*)(pe_lds + ((0 * NP + p) * KPE + ks) * QUAD_BYTES);
                pel[p][ks] = *(const u32x4b*)(pe_lds + ((1 * NP + p) * KPE + ks) * QUAD_BYTES);
            }
        // hoisted view-direction term of linear_d (fp32): scratch[n] = b_d[n] + sum_f Wd[n][W+f] * gamma(d/|d|)[f], once per ray
        if (tray != bias_ray) {
            bias_ray = tray;
            const float dx = in_d[0], dy = in_d[1], dz = in_d[2];
            const float nrm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
            const float vdir[3] = {dx / nrm, dy / nrm, dz / nrm};
            float g[IN_D];
            g[0] = vdir[0]; g[1] = vdir[1]; g[2] = vdir[2];
#pragma unroll
            for (int k = 0; k < LD; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float y = vdir[c] * (float)(1 << k);
                    g[3 + 6 * k + c] = sin_cos_fast(y, 0);
                    g[3 + 6 * k + 3 + c] = sin_cos_fast(y, 1);
                }
            const float* wdt = side + a.o_wdir_t;
            const float* bd = side + a.o_bias_d;
#pragma unroll
            for (int n0 = 0; n0 < W / 2; n0 += 64) {
                const int n = n0 + lane;
                float s = bd[n];
#pragma unroll
                for (int f = 0; f < IN_D; ++f) s = __builtin_fmaf(wdt[f * (W / 2) + n], g[f], s);
                scratch[n] = s;
            }
        }
        // ---- layer 0: 16 jobs of 2 k-steps over gamma(x) (VGPR fragments), output into set 0 -------------------------------------------
        if constexpr (STASH) {                                   // (re-stated here so that the eight words are not live across the prologue, the kernel's register peak)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) mw[p][k] = 0u;
        }
        {
            const float* b0 = side + a.o_bias_trunk + 4 * q4;
            cin = *(const f32x4*)b0;
            auto bh = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> const u32x4b& { return peh[decltype(p_c)::value][decltype(ks_c)::value]; };
            auto bl = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> const u32x4b& { return pel[decltype(p_c)::value][decltype(ks_c)::value]; };
            static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                PairTmp pt[4];
                auto hook = [&](auto ks_c, auto s_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, sub = decltype(s_c)::value;
                    // 12 MFMAs per job and four pairs to pack: a whole pair per gap (these jobs run VALU bound; 3 % of the MFMAs)
                    if constexpr (t > 0 && ks == 0 && sub >= 2) pack_pair_block<true, 0, t - 1, sub - 2>(ph, pl, pt, dn, up, nup);
                    if constexpr (STASH && t > 0 && ks == 1 && (sub == 3 || sub == 4))
                        stash_tile<true, true, t - 1>(pt[2 * (sub - 3)], pt[2 * (sub - 3) + 1], rowp[sub - 3], mw[sub - 3], nib_sh);
                    if constexpr (ks == 1 && sub == 2) {
                        const float* v = (t + 1 < NT) ? b0 + MT * (t + 1) : side + a.o_bias_trunk + W + 4 * q4;
                        cnext = *(const f32x4*)v;
                    }
                };
                job<t * KPE, KPE, BIG, 0>(ah, al, csel1, bh, bl, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) { ph[p] = ah[p]; pl[p] = al[p]; }
                cin = cnext;
            });
        }
        // ---- trunk layers 1..D-1 ping-pong between the two fragment sets with a static polarity (pairs 0->1, 1->0) --------------------
        auto layer_01 = [&](int l) __attribute__((always_inline)) {
            const float* bias = side + a.o_bias_trunk + l * W;
            const float* nb = (l + 1 < a.D) ? bias + W : side + a.o_bias_feat;
            if (l == a.skip_layer) trunk_layer(std::true_type{}, IC<0>{}, bias, nb);
            else trunk_layer(std::false_type{}, IC<0>{}, bias, nb);
        };
        auto layer_10 = [&](int l) __attribute__((always_inline)) {
            const float* bias = side + a.o_bias_trunk + l * W;
            const float* nb = (l + 1 < a.D) ? bias + W : side + a.o_bias_feat;
            if (l == a.skip_layer) trunk_layer(std::true_type{}, IC<1>{}, bias, nb);
            else trunk_layer(std::false_type{}, IC<1>{}, bias, nb);
        };
        int l = 1;                                               // layer 0 wrote set 0 (its last tile is still in ph / pl)
#pragma unroll 1
        for (; l + 1 < a.D; l += 2) { layer_01(l); layer_10(l + 1); }
        // ---- tail: feature layer, density tile, view-direction layer, colour tile, store ---------------------------------------------
        auto tail = [&](auto sin_c) __attribute__((always_inline)) {
            constexpr int SIN = decltype(sin_c)::value, SOUT = 1 - SIN;
            f32x4 hdh[NP], hdl[NP], hch[NP], hcl[NP], cind[NP], cnextd[NP], cinh;      // density / colour tiles; per-point-tile direction bias
            float dens[NP];
            auto cseld = [&](int p) __attribute__((always_inline)) -> const f32x4& { return cind[p]; };
            auto cselh = [&](int) __attribute__((always_inline)) -> const f32x4& { return cinh; };
            const float* bf = side + a.o_bias_feat + 4 * q4;
            auto bh_in = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(SIN, decltype(p_c)::value, decltype(ks_c)::value, 0)>{}; };
            auto bl_in = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(SIN, decltype(p_c)::value, decltype(ks_c)::value, 1)>{}; };
            auto bh_out = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(SOUT, decltype(p_c)::value, decltype(ks_c)::value, 0)>{}; };
            auto bl_out = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(SOUT, decltype(p_c)::value, decltype(ks_c)::value, 1)>{}; };
            // feature layer: no activation on its outputs; its first job still packs the trunk's last tile (ReLU)
            static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                PairTmp pt[4];
                auto hook = [&](auto ks_c, auto s_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, sub = decltype(s_c)::value;
                    if constexpr (t == 0) pack_sched<true, SIN, NT - 1, ks, sub>(ph, pl, pt, dn, up, nup);
                    else pack_sched<false, SOUT, t - 1, ks, sub>(ph, pl, pt, dn, up, nup);
                    if constexpr (STASH && ks == 7 && (sub == 3 || sub == 4)) {
                        if constexpr (t == 0) stash_tile<true, true, NT - 1>(pt[2 * (sub - 3)], pt[2 * (sub - 3) + 1], rowp[sub - 3], mw[sub - 3], nib_sh);
                        else stash_tile<false, false, t - 1>(pt[2 * (sub - 3)], pt[2 * (sub - 3) + 1], rowp[sub - 3], mw[sub - 3], nib_sh);
                    }
                    if constexpr (ks == 7 && sub == 2) {
                        if constexpr (t + 1 < NT) cnext = *(const f32x4*)(bf + MT * (t + 1));
                        else {                                          // density tile: row 3 = density bias (lane quarter 0 only)
                            const float db = side[a.o_head_b + 3];
                            cnext[0] = 0.0f; cnext[1] = 0.0f; cnext[2] = 0.0f; cnext[3] = q4 == 0 ? db : 0.0f;
                        }
                    }
                };
                job<t * KH, KH, BIG, 0>(ah, al, csel1, bh_in, bl_in, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) { ph[p] = ah[p]; pl[p] = al[p]; }
                cin = cnext;
                if constexpr (STASH && t == 0) {                 // the trunk's last tile is out: from here on the rows are linear_feat's (no activation)
                    end_trunk_layer();
#pragma unroll
                    for (int p = 0; p < NP; ++p) rowp[p] = a.stash_f + out_idx[p] * W + 4 * q4;
                }
            });
            // density tile over the trunk output (row 3); packs the feature layer's last tile; reads the direction bias of tile 0
            {
                PairTmp pt[4];
                auto hook = [&](auto ks_c, auto s_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, sub = decltype(s_c)::value;
                    pack_sched<false, SOUT, NT - 1, ks, sub>(ph, pl, pt, dn, up, nup);
                    if constexpr (ks == 7 && (sub == 2 || sub == 3)) cnextd[sub - 2] = *(const f32x4*)(scratch + 4 * q4);
                    if constexpr (STASH && ks == 7 && (sub == 4 || sub == 5))
                        stash_tile<false, false, NT - 1>(pt[2 * (sub - 4)], pt[2 * (sub - 4) + 1], rowp[sub - 4], mw[sub - 4], nib_sh);
                };
                job<128, KH, BIG, 0>(hdh, hdl, csel1, bh_in, bl_in, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) cind[p] = cnextd[p];
                if constexpr (STASH) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) rowp[p] = a.stash_g + out_idx[p] * (W / 2) + 4 * q4;
                }
            }
            // view-direction layer: 8 jobs over the feature layer's output; ReLU'd tiles go into fragments 0..3 of set SIN (the
            // trunk output is dead once the density tile has run)
            static_for<0, NT / 2>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                PairTmp pt[4];
                auto hook = [&](auto ks_c, auto s_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, sub = decltype(s_c)::value;
                    if constexpr (t > 0) pack_sched<true, SIN, t - 1, ks, sub>(ph, pl, pt, dn, up, nup);
                    if constexpr (STASH && t > 0 && ks == 6 && (sub == 4 || sub == 5))      // (group 7's gaps carry this layer's own loads)
                        stash_tile<true, true, t - 1>(pt[2 * (sub - 4)], pt[2 * (sub - 4) + 1], rowp[sub - 4], mw[sub - 4], nib_sh);
                    if constexpr (t == 0 && ks == 0 && sub == 3) {          // the density tile's last MFMA is 4 issues (16 wait states) back: its tuples may go (keep_tuple, common.h)
                        keep_tuple(hdh[0]); keep_tuple(hdh[1]); keep_tuple(hdl[0]); keep_tuple(hdl[1]);
                    }
                    if constexpr (t == 0 && ks == 7 && sub >= 4)            // the density tile finished >= 40 MFMAs ago: keep its one useful value
                        asm volatile("v_fma_f32 %0, %1, %3, %2" : "=v"(dens[sub - 4]) : "v"(hdl[sub - 4][3]), "v"(hdh[sub - 4][3]), "s"(dn));
                    if constexpr (t == PF_T && ks == PF_KS && sub == 3) load_inputs(it + 1 < a.n_iter ? it + 1 : it);      // next unit's ray and depths
                    if constexpr (ks == 7 && sub == 2) {
                        if constexpr (t + 1 < NT / 2) { cnextd[0] = *(const f32x4*)(scratch + MT * (t + 1) + 4 * q4); cnextd[1] = cnextd[0]; }
                        else {                                          // colour tile: rows 0..2 = colour bias (lane quarter 0 only)
                            const f32x4 hb4 = *(const f32x4*)(side + a.o_head_b);
                            cinh[0] = q4 == 0 ? hb4[0] : 0.0f; cinh[1] = q4 == 0 ? hb4[1] : 0.0f; cinh[2] = q4 == 0 ? hb4[2] : 0.0f; cinh[3] = 0.0f;
                        }
                    }
                };
                job<136 + t * KH, KH, BIG, 0>(ah, al, cseld, bh_out, bl_out, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) { ph[p] = ah[p]; pl[p] = al[p]; cind[p] = cnextd[p]; }
            });
            // colour tile over the view-direction output (rows 0..2): 4 k-steps; the last direction tile (second half of fragment 3) is
            // packed in its first group, a whole pair per gap (it feeds k-step 3)
            {
                PairTmp pt[4];
                auto hook = [&](auto ks_c, auto s_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, sub = decltype(s_c)::value;
                    if constexpr (ks == 0 && sub >= 2) pack_pair_block<true, SIN, NT / 2 - 1, sub - 2>(ph, pl, pt, dn, up, nup);
                    if constexpr (STASH && ks == 1 && (sub == 2 || sub == 3))
                        stash_tile<true, true, NT / 2 - 1>(pt[2 * (sub - 2)], pt[2 * (sub - 2) + 1], rowp[sub - 2], mw[sub - 2], nib_sh);
                };
                job<200, KH / 2, TAIL_USED_P, TAIL_PAIRS - TAIL_USED_P>(hch, hcl, cselh, bh_in, bl_in, aq, smem, ring, lane, hook);
            }
            // the MFMAs are asm statements: hipcc does not know that the colour tile is still in flight (XDL write -> VALU read), nor that
            // the tiles' fourth registers, which nothing reads, are still to be WRITTEN: the whole tuples pass through the wait statement
            // (keep_tuple, common.h), and nothing else sits between the job and it
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(hcl[0]), "+v"(hcl[1]), "+v"(hch[0]), "+v"(hch[1]) : : "memory");
            // (behind the colour tile's window, not inside it: until round 4 a shuffle result landed in a tile's unread fourth register)
            if constexpr (STASH) finish_masks<2>(mw, a.mask_g + (size_t)tile_cur * 128, col, q4, tile_active);
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (valid[p] && q4 == 0) {                      // cat([rgb, density]) NeRF.py:51
                    f32x4 o;
                    o[0] = __builtin_fmaf(hcl[p][0], SC_DN, hch[p][0]); o[1] = __builtin_fmaf(hcl[p][1], SC_DN, hch[p][1]);
                    o[2] = __builtin_fmaf(hcl[p][2], SC_DN, hch[p][2]); o[3] = dens[p];
                    *(f32x4*)(a.out + out_idx[p] * 4) = o;
                }
        };
        if (l < a.D) { layer_01(l); tail(IC<1>{}); }
        else tail(IC<0>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}


}  // namespace f16s

// host side of a forward launch (inference: mlp_f16s.hip; training forward with stash: mlp_f16s_stash.hip)
struct StashF16s { float* h; float* f; float* g; unsigned* mask_h; unsigned* mask_g; };
template <bool STASH>
static int launch_f16s(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev, int64_t n_rays, int S,
                       float* raw_dev, const StashF16s* stash, hipStream_t st) {
    using namespace f16s;
    if (int rc = check_net(net)) return rc;
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    if (n_rays == 0) return MI_NERF_OK;
    MN_CHECK_ARG(packed_dev && rays_dev && z_dev && raw_dev, "NULL device pointer");
    const BlobLayoutS L = make_layout(net->D, net->W, net->skip);
    Args a{};
    a.stream = (const char*)packed_dev + L.stream_off;
    a.side = (const float*)((const char*)packed_dev + L.side_off);
    a.rays = rays_dev; a.z = z_dev; a.out = raw_dev;
    a.S = S; a.tpr = (S + 31) / 32;
    const long long n_wtiles = (long long)n_rays * a.tpr;
    MN_CHECK_ARG(n_wtiles < (1LL << 30), "too many points for one launch: %lld rays x %d samples", (long long)n_rays, S);
    a.n_wtiles = (unsigned)n_wtiles;
    a.D = net->D;
    a.skip_layer = (net->skip >= 0 && net->skip + 1 < net->D) ? net->skip + 1 : -1;
    a.stream_bytes = L.stream_bytes; a.side_floats = L.side_floats;
    a.o_bias_trunk = L.bias_trunk; a.o_bias_feat = L.bias_feat; a.o_bias_d = L.bias_d; a.o_head_b = L.head_b; a.o_wdir_t = L.wdir_t;
    const size_t lds = RING_BYTES_S + (size_t)a.side_floats * 4 + 4 * (256 / 2) * 4 + (size_t)4 * 2 * NP * KPE * QUAD_BYTES;
    MN_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded: %zu bytes", lds);
    static LdsOptIn opt_in = {};
    if (int rc = ensure_lds_opt_in(opt_in, (const void*)mlp_f16s_kernel<STASH>)) return rc;
    const int n_cus = device_cus();
    const long long n_wg = (n_wtiles + 3) / 4;
    const int grid = (int)(n_wg < n_cus ? n_wg : n_cus);
    const long long NW = (long long)grid * 4;
    const long long it_flat = (n_wtiles + NW - 1) / NW, it_ray = (((long long)n_rays + NW - 1) / NW) * a.tpr;
    if ((long long)n_rays >= NW && it_ray <= it_flat) { a.ppr = (unsigned)a.tpr; a.n_iter = (unsigned)it_ray; }      // ray-major: whole rays per wave
    else { a.ppr = 0; a.n_iter = (unsigned)it_flat; }
    if constexpr (STASH) {
        a.stash_h = stash->h; a.stash_f = stash->f; a.stash_g = stash->g; a.mask_h = stash->mask_h; a.mask_g = stash->mask_g;
        a.stash_rows = (long long)n_rays * S;
    }
    hipLaunchKernelGGL(mlp_f16s_kernel<STASH>, dim3(grid), dim3(256), lds, st, a);
    MN_LAUNCH_CHECK("mlp_f16s_kernel");
    return MI_NERF_OK;
}

}  // namespace minerf
