// dgrad_f16s.hip -- the backward-data chain of the training step in split precision (the design: mlp_f16s.hip; the fp32 kernel it replaces:
// mlp_train.hip mlp_dgrad_kernel).
#include "mlp_f16s_core.h"

namespace minerf {
namespace f16s {

// ---------------------------------------------------------------------------------------------
// BACKWARD DATA in split precision: mlp_dgrad_kernel's chain (mlp_train.hip) on this file's machinery.  A unit is one 32-point tile;
// the gradient w.r.t. a layer's output sits in the accumulators (hi, lo), is masked with the forward's ReLU' bits while it is packed and is
// the B fragment of the next transposed GEMM; every pre-activation gradient ("delta") row is written as fp32, true scale, for the
// weight-gradient products.  Gradients are tiny: the whole chain runs on d_raw * s, s the power of two that puts max|d_raw| near 2^8
// (absmax_bits, the same device word wgrad_f16s_kernel reads); rows are multiplied back by 1 / s as they are stored.
//   G0   d feature   = Wd[:, :W]^T (mask_g . colour head^T d rgb)        16 jobs x 4 k-steps, B fragments built in the prologue (VGPRs)
//   G1   d trunk out = W_feat^T d feature + dens_w d sigma; x mask_h[D-1] 16 jobs x 8 k-steps
//   G(l) d h_{l-1}   = W_l[:, h block]^T delta_l; x mask_h[l-1]           l = D-1 .. 1
// Packing slots of a pair (the forward's 24-slot schedule): [y = lo 2^-11 + hi (+ dens_w d sigma)] [ReLU' bit -> and] [hi = cvt; Y = 2^11 y]
// [lo = f16(Y - 2^11 hi)] [two fragment-file writes] [second pair of a tile: its 16-byte row piece].
// ---------------------------------------------------------------------------------------------
struct DArgs {
    const char* stream; unsigned stream_bytes;
    const float* color_w; const float* dens_w;      // fp32 forward blob's side tables: [3][W/2], [W]
    const float* d_raw; const unsigned* mask_h; const unsigned* mask_g;
    float* delta_h; float* delta_f; float* delta_d;
    long long P, n_valid;                           // row pitch of the delta tensors; rows that exist
    unsigned n_wtiles, n_iter, ppr;
    int S, tpr, D;
    const unsigned* absmax_bits;
    unsigned* delta_absmax_bits;                    // out: max |delta * s| over every stored row piece (bits of a non-negative float): how much of the
};                                                  // f16 range the scaled chain used (>= 65504: a conversion saturated)
struct DPair { float y0, y1; unsigned hi, lo; };
// VAR: 0 plain, 1 ReLU' mask, 2 rank-1 term + mask.  POS: bit position of element 2e of the tile in the pre-shifted mask word.
template <int VAR, int RH, int RL, int STAGE, int POS>
__device__ __forceinline__ void dpack_stage(const float h0, const float h1, const float l0, const float l1, DPair& t, float dn, float up, float nup,
                                            unsigned mword, float w0, float w1, float dsig) {
    if constexpr (STAGE == 0) {
        asm volatile("v_fma_f32 %0, %2, %6, %3\n\tv_fma_f32 %1, %4, %6, %5" : "=&v"(t.y0), "=&v"(t.y1) : "v"(l0), "v"(h0), "v"(l1), "v"(h1), "s"(dn));
        if constexpr (VAR == 2) asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1" : "+v"(t.y0), "+v"(t.y1) : "v"(w0), "v"(w1), "v"(dsig));
    } else if constexpr (STAGE == 1) {
        if constexpr (VAR >= 1) {
            int m0, m1;
            asm volatile("v_bfe_i32 %0, %2, %3, 1\n\tv_bfe_i32 %1, %2, %4, 1" : "=&v"(m0), "=&v"(m1) : "v"(mword), "n"(POS), "n"(POS - 1));
            asm volatile("v_and_b32 %0, %0, %2\n\tv_and_b32 %1, %1, %3" : "+v"(t.y0), "+v"(t.y1) : "v"(m0), "v"(m1));
        }
    } else if constexpr (STAGE == 2) {
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=&v"(t.hi) : "v"(t.y0), "v"(t.y1));
    } else if constexpr (STAGE == 3) {
        float Y0, Y1;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(Y0), "=&v"(Y1) : "v"(t.y0), "v"(t.y1), "s"(up));
        asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(t.lo) : "v"(t.hi), "s"(nup), "v"(Y0), "v"(Y1));
    } else if constexpr (STAGE == 4) {
        asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1" ::"v"(t.hi), "v"(t.lo), "n"(RH), "n"(RL));
    }
}
// one 16-byte piece of a delta row: the tile's four values of this lane's point, back at true scale
template <int T>
__device__ __forceinline__ void delta_store(const DPair& e0, const DPair& e1, float* rowp, float inv_s, float& dmax) {
    dmax = __builtin_fmaxf(dmax, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(e0.y0), __builtin_fabsf(e0.y1)), __builtin_fmaxf(__builtin_fabsf(e1.y0), __builtin_fabsf(e1.y1))));
    f32x4 v;
    v[0] = e0.y0 * inv_s; v[1] = e0.y1 * inv_s; v[2] = e1.y0 * inv_s; v[3] = e1.y1 * inv_s;
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2 " F16S_STASH_STORE_FLAGS "\n\ts_nop 1" ::"v"(rowp), "v"(v), "n"(MT * T * 4) : "memory");
}
// the 24-slot schedule of pack_sched, with the row piece in the free sixth slot of a tile's second pair
template <int VAR, int SET, int T, int KS, int SUB>
__device__ __forceinline__ void dpack_sched(const f32x4 (&ph)[NP], const f32x4 (&pl)[NP], DPair (&t)[4], float dn, float up, float nup,
                                            const unsigned (&msh)[NP][4], const f32x4& dwv, const float (&dsig)[NP], float* const (&rowp)[NP], float inv_s, float& dmax) {
    if constexpr (KS >= 1 && KS <= 6 && SUB >= 2) {
        constexpr int n = 4 * (KS - 1) + (SUB - 2), pair = n / 6, stage = n % 6, p = pair >> 1, e = pair & 1;
        constexpr int POS = 31 - 4 * ((2 * T) & 7) - 2 * e;
        if constexpr (stage < 5)
            dpack_stage<VAR, tile_reg(SET, p, T, 0) + e, tile_reg(SET, p, T, 1) + e, stage, POS>(ph[p][2 * e], ph[p][2 * e + 1], pl[p][2 * e], pl[p][2 * e + 1], t[pair],
                                                                                                 dn, up, nup, msh[p][T >> 2], dwv[2 * e], dwv[2 * e + 1], dsig[p]);
        else if constexpr (e == 1) delta_store<T>(t[2 * p], t[2 * p + 1], rowp[p], inv_s, dmax);
    }
}
// a whole pair in one gap (G0's short jobs)
template <int SET, int T, int PAIR>
__device__ __forceinline__ void dpack_pair_block(const f32x4 (&ph)[NP], const f32x4 (&pl)[NP], DPair (&t)[4], float dn, float up, float nup) {
    constexpr int p = PAIR >> 1, e = PAIR & 1;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    static_for<0, 5>([&](auto st_c) __attribute__((always_inline)) {
        dpack_stage<0, tile_reg(SET, p, T, 0) + e, tile_reg(SET, p, T, 1) + e, decltype(st_c)::value, 0>(ph[p][2 * e], ph[p][2 * e + 1], pl[p][2 * e], pl[p][2 * e + 1], t[PAIR],
                                                                                                         dn, up, nup, 0u, z[0], z[0], 0.0f);
    });
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dgrad_f16s_kernel(const DArgs a) {
    constexpr int W = 256;
    constexpr int BIG = 1 << 30;
    constexpr int KG0 = KH / 2;                                // k-steps of G0 (W/2 = 128 inputs)
    asm volatile("" ::: "a255");
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");       // FP16_OVFL: a conversion beyond the f16 range saturates
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cw = (float*)(smem + RING_BYTES_S);                 // colour head [3][W/2], then the density head [W], then a mask block per wave
    float* dwl = cw + 3 * (W / 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, q4 = lane >> 4;
    const float dn = SC_DN, up = SC_UP, nup = -SC_UP;
    for (int i = tid; i < 3 * (W / 2); i += 256) cw[i] = a.color_w[i];
    for (int i = tid; i < W; i += 256) dwl[i] = a.dens_w[i];
    const unsigned mb = (unsigned)__builtin_amdgcn_readfirstlane((int)*a.absmax_bits);
    int ex = (int)((mb >> 23) & 255u) - 127;
    if (mb == 0u || ex < -100) ex = -100;
    if (ex > 100) ex = 100;
    const float sc = __uint_as_float((unsigned)(127 + 7 - ex) << 23), inv_sc = __uint_as_float((unsigned)(127 - 7 + ex) << 23);

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
    for (int i = 0; i < DMA_PER_WAVE; ++i) ring_dma(ring, i);
    ring_next_fetch(ring);
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) ring_dma(ring, i);
    u32x4b aq[NBUF][2];
    ring_advance(ring);                                             // also publishes the head tables (barrier)
#pragma unroll
    for (int i = 0; i < LA; ++i) { aq[i][0] = ring_read(smem, ring, lane, 2 * i); aq[i][1] = ring_read(smem, ring, lane, 2 * i + 1); }

    const unsigned NW = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    auto unit_of = [&](unsigned it) -> unsigned {
        if (a.ppr) { const unsigned blk = it / a.ppr; return (blk * NW + wid) * a.ppr + (it - blk * a.ppr); }
        return it * NW + wid;
    };
    f32x4 ah[NP], al[NP], ph[NP], pl[NP];
    const f32x4 czero = {0.f, 0.f, 0.f, 0.f};
    auto csel0 = [&](int) __attribute__((always_inline)) -> const f32x4& { return czero; };
    const int sh4 = 4 * (q4 >> 1);                                  // this lane's nibble of a mask byte pair (see stash_tile)
    float dmax = 0.0f;                                              // largest scaled gradient this lane stored
    const char* mlds = (const char*)(dwl + W) + wave * (a.D * 1024);
    const unsigned mlds_m0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + RING_BYTES_S + (3 * (W / 2) + W) * 4 + wave * (a.D * 1024);

    for (unsigned it = 0; it < a.n_iter; ++it) {
        const unsigned n_tile = unit_of(it);
        const bool active = n_tile < a.n_wtiles;
        const unsigned tcur = active ? n_tile : a.n_wtiles - 1;
        const unsigned tray = tcur / (unsigned)a.tpr, chunk = tcur - tray * (unsigned)a.tpr;
        long long out_idx[NP];
        float dsig[NP];
        float* rowp[NP];
        unsigned msh[NP][4];
        u32x4b bdh[NP][KG0], bdl[NP][KG0];
        // ---- this tile's ReLU' words, all layers: [D][64 lanes][16 bytes] straight into the wave's LDS block.  The ring's own DMAs are in mid
        // slot here (five of the slot's eight issued: the A pipeline runs three pairs ahead), and they set M0 only at their first and fifth:
        // M0 goes back to what the sixth expects.
        {
            const char* mg = (const char*)a.mask_h + ((size_t)tcur * 64 + lane) * 16;
            const size_t layer_bytes = (size_t)a.n_wtiles * 1024;
            for (int ml = 0; ml < a.D; ++ml) {
                set_m0(mlds_m0 + ml * 1024);
                dma16<0>(mg + ml * layer_bytes);
            }
            set_m0(ring.fetch_lds + 4096);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int sample = (int)chunk * 32 + 16 * p + col;
            long long idx = (long long)tray * a.S + (sample < a.S ? sample : a.S - 1);
            if (idx >= a.n_valid) idx = a.n_valid - 1;
            out_idx[p] = idx;
            // ---- colour head^T and ReLU' of linear_d: this lane's 32 values of delta_d (the k-values of its four B fragments), rows stored
            f32x4 dr = *(const f32x4*)(a.d_raw + idx * 4);
            dr[0] *= sc; dr[1] *= sc; dr[2] *= sc; dr[3] *= sc;
            dsig[p] = dr[3];
            const u32x2b mgv = *(const u32x2b*)(a.mask_g + ((size_t)tcur * 64 + (col + 16 * p) + 32 * (q4 & 1)) * 2);
            const unsigned mgs[2] = {mgv[0] << sh4, mgv[1] << sh4};
            float* drow = a.delta_d + idx * (W / 2) + 4 * q4;
#pragma unroll
            for (int s = 0; s < KG0; ++s) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int tt = 2 * s + jj;                                      // 16-feature tile of the direction layer's output
                    const f32x4 w0 = *(const f32x4*)(cw + MT * tt + 4 * q4);
                    const f32x4 w1 = *(const f32x4*)(cw + W / 2 + MT * tt + 4 * q4);
                    const f32x4 w2 = *(const f32x4*)(cw + W + MT * tt + 4 * q4);
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float x = dr[0] * w0[i];
                        x = __builtin_fmaf(dr[1], w1[i], x);
                        x = __builtin_fmaf(dr[2], w2[i], x);
                        const int pos = 31 - 4 * ((2 * tt) & 7) - i;
                        const int m = ((int)(mgs[tt >> 2] << (31 - pos))) >> 31;      // bit `pos` of the pre-shifted word, sign-extended
                        x = __uint_as_float(__float_as_uint(x) & (unsigned)m);
                        v[4 * jj + i] = x;
                        o[i] = x * inv_sc;
                    }
                    *(f32x4*)(drow + MT * tt) = o;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned hi, lo;
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
                    const float Y0 = v[2 * i] * SC_UP, Y1 = v[2 * i + 1] * SC_UP;
                    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                 : "=&v"(lo) : "v"(hi), "s"(nup), "v"(Y0), "v"(Y1));
                    bdh[p][s][i] = hi; bdl[p][s][i] = lo;
                }
            }
            rowp[p] = a.delta_f + idx * W + 4 * q4;                                // G0's output rows
#pragma unroll
            for (int k = 0; k < 4; ++k) msh[p][k] = 0u;
        }
        // ---- G0: 16 jobs of 4 k-steps over delta_d (VGPR fragments) -> d feature into set 0; no activation --------------------------------
        {
            auto bh = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> const u32x4b& { return bdh[decltype(p_c)::value][decltype(ks_c)::value]; };
            auto bl = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> const u32x4b& { return bdl[decltype(p_c)::value][decltype(ks_c)::value]; };
            static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                DPair pt[4];
                auto hook = [&](auto ks_c, auto s_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, sub = decltype(s_c)::value;
                    if constexpr (t > 0 && ks == 0 && sub >= 2) dpack_pair_block<0, t - 1, sub - 2>(ph, pl, pt, dn, up, nup);
                    if constexpr (t > 0 && ks == 1 && (sub == 2 || sub == 3)) delta_store<t - 1>(pt[2 * (sub - 2)], pt[2 * (sub - 2) + 1], rowp[sub - 2], inv_sc, dmax);
                };
                job<t * KG0, KG0, BIG, 0>(ah, al, csel0, bh, bl, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) { ph[p] = ah[p]; pl[p] = al[p]; }
            });
        }
        // ---- G1 and the trunk: 16 jobs of 8 k-steps; set SIN -> set 1 - SIN.  VARP / VARO: how the PREVIOUS GEMM's last tile (packed by job 0)
        // and this GEMM's own tiles are finished; mask words of the two differ, row pointers too
        f32x4 dwv = czero;
        unsigned mprev[NP][4];
        float* rprev[NP];
        // The ReLU' words of every layer for this tile sit in the wave's LDS block (one 1 KiB LDS-DMA per layer, issued in the prologue: a
        // per-GEMM global load would be waited for by the very next ring advance -- every advance waits for all older vector memory
        // operations -- and cost 0.6 ms of the step).  mask_layer: the layer whose words mask this GEMM's OUTPUT.
        auto gemm = [&](auto sin_c, auto varp_c, auto varo_c, int mask_layer, float* rows_out) __attribute__((always_inline)) {
            constexpr int SIN = decltype(sin_c)::value, SOUT = 1 - SIN, VARP = decltype(varp_c)::value, VARO = decltype(varo_c)::value;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                rprev[p] = rowp[p];
                rowp[p] = rows_out + out_idx[p] * W + 4 * q4;
                const u32x4b mv = *(const u32x4b*)(mlds + mask_layer * 1024 + ((col + 16 * p) + 32 * (q4 & 1)) * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) { mprev[p][k] = msh[p][k]; msh[p][k] = mv[k] << sh4; }
            }
            auto bh = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(SIN, decltype(p_c)::value, decltype(ks_c)::value, 0)>{}; };
            auto bl = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(SIN, decltype(p_c)::value, decltype(ks_c)::value, 1)>{}; };
            static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                DPair pt[4];
                auto hook = [&](auto ks_c, auto s_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, sub = decltype(s_c)::value;
                    if constexpr (t == 0) dpack_sched<VARP, SIN, NT - 1, ks, sub>(ph, pl, pt, dn, up, nup, mprev, dwv, dsig, rprev, inv_sc, dmax);
                    else dpack_sched<VARO, SOUT, t - 1, ks, sub>(ph, pl, pt, dn, up, nup, msh, dwv, dsig, rowp, inv_sc, dmax);
                    // density head^T weights of the tile the NEXT job packs (rank-1 term of G1's output)
                    if constexpr (VARO == 2 && ks == 7 && sub == 2) dwv = *(const f32x4*)(dwl + MT * t + 4 * q4);
                };
                job<t * KH, KH, BIG, 0>(ah, al, csel0, bh, bl, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) { ph[p] = ah[p]; pl[p] = al[p]; }
            });
        };
        // G1: input d feature (set 0, plain), output d trunk out (+ dens_w d sigma, mask of layer D-1) -> set 1
        gemm(IC<0>{}, IC<0>{}, IC<2>{}, a.D - 1, a.delta_h + (size_t)(a.D - 1) * a.P * W);
        // trunk GEMMs l = D-1 .. 1: output = delta_{l-1}, masked by layer l-1's words
        int l = a.D - 1;
        if (l >= 1) { gemm(IC<1>{}, IC<2>{}, IC<1>{}, l - 1, a.delta_h + (size_t)(l - 1) * a.P * W); --l; }
#pragma unroll 1
        for (; l >= 2; l -= 2) {
            gemm(IC<0>{}, IC<1>{}, IC<1>{}, l - 1, a.delta_h + (size_t)(l - 1) * a.P * W);
            gemm(IC<1>{}, IC<1>{}, IC<1>{}, l - 2, a.delta_h + (size_t)(l - 2) * a.P * W);
        }
        if (l == 1) gemm(IC<0>{}, IC<1>{}, IC<1>{}, 0, a.delta_h);
        // ---- the last GEMM's last tile has no job behind it: finish it here (rows only) ---------------------------------------------------
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y = __builtin_fmaf(pl[p][i], SC_DN, ph[p][i]);
                const int pos = 31 - 4 * ((2 * (NT - 1)) & 7) - i;
                const int m = ((int)(msh[p][(NT - 1) >> 2] << (31 - pos))) >> 31;
                o[i] = __uint_as_float(__float_as_uint(y) & (unsigned)m) * inv_sc;
            }
            *(f32x4*)(rowp[p] + MT * (NT - 1)) = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = __builtin_fmaxf(dmax, __shfl_xor(dmax, o, 64));
    if (lane == 0 && dmax == dmax) atomicMax(a.delta_absmax_bits, __float_as_uint(dmax));
}

}  // namespace f16s

// backward-data chain in split precision: the deltas mlp_dgrad_kernel writes (same tensors, true scale), from the masks either training forward leaves
int dgrad_f16s(const mi_nerf_net* net, const void* packed_bwd_f16s_dev, const float* color_w_dev, const float* dens_w_dev, const float* d_raw_dev,
               const unsigned* mask_h, const unsigned* mask_g, float* delta_h, float* delta_f, float* delta_d, int64_t n_rays, int S, long long P_pitch,
               long long n_valid, const unsigned* absmax_dev, hipStream_t st) {
    using namespace f16s;
    if (int rc = check_net(net)) return rc;
    MN_CHECK_ARG(n_rays >= 1 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    MN_CHECK_ARG(packed_bwd_f16s_dev && color_w_dev && dens_w_dev && d_raw_dev && mask_h && mask_g && delta_h && delta_f && delta_d && absmax_dev, "NULL device pointer");
    DArgs a{};
    a.stream = (const char*)packed_bwd_f16s_dev + HEADER_BYTES;
    a.stream_bytes = bwd_stream_bytes_s(net->D);
    a.color_w = color_w_dev; a.dens_w = dens_w_dev;
    a.d_raw = d_raw_dev; a.mask_h = mask_h; a.mask_g = mask_g;
    a.delta_h = delta_h; a.delta_f = delta_f; a.delta_d = delta_d;
    a.P = P_pitch; a.n_valid = n_valid;
    a.S = S; a.tpr = (S + 31) / 32; a.D = net->D;
    const long long n_wtiles = (long long)n_rays * a.tpr;
    MN_CHECK_ARG(n_wtiles < (1LL << 30), "too many points for one launch: %lld rays x %d samples", (long long)n_rays, S);
    a.n_wtiles = (unsigned)n_wtiles;
    a.absmax_bits = absmax_dev;
    a.delta_absmax_bits = const_cast<unsigned*>(absmax_dev) + 1;    // second word of the same slot (zeroed with the first)
    const size_t lds = RING_BYTES_S + (size_t)(3 * 128 + 256) * 4 + (size_t)4 * net->D * 1024;      // ring | heads | per wave: D layers of mask words
    MN_CHECK_ARG(lds <= 160 * 1024, "the split-precision backward-data kernel keeps a tile's ReLU' words of all layers in LDS: D = %d does not fit (D <= 15)", net->D);
    static LdsOptIn opt_in = {};
    if (int rc = ensure_lds_opt_in(opt_in, (const void*)dgrad_f16s_kernel)) return rc;
    const int n_cus = device_cus();
    const long long n_wg = (n_wtiles + 3) / 4;
    const int grid = (int)(n_wg < n_cus ? n_wg : n_cus);
    const long long NW = (long long)grid * 4;
    a.ppr = 0; a.n_iter = (unsigned)((n_wtiles + NW - 1) / NW);
    hipLaunchKernelGGL(dgrad_f16s_kernel, dim3(grid), dim3(256), lds, st, a);
    MN_LAUNCH_CHECK("dgrad_f16s_kernel");
    return MI_NERF_OK;
}
}  // namespace minerf
