// mlp_train.hip -- backward pass of the NeRF MLP on fp32 MFMA (gfx950): what `loss.backward()` does for
// model/NeRF.py:33-52 inside train.py:53-70.
//
// Kernels, fed from what the training forward keeps (mlp_fp32.hip, STASH: row-major activations + ReLU' bit masks):
//   * mlp_dgrad_kernel     backward-data chain.  Same register-resident design as the forward kernel: a wave owns 32
//                          points, the gradient w.r.t. a layer's output sits in the MFMA accumulators, is masked with the
//                          forward's ReLU' bits (one coalesced 16-byte load per lane per layer) and is directly the B
//                          operand of the next (transposed) GEMM.  The TRANSPOSED weights stream through the same LDS
//                          ring (pack.cpp: pack_bwd_fp32).  Every layer's pre-activation gradient ("delta") row is
//                          written by the GEMM that consumes it, one store per k-quad.
//   * wgrad_big_kernel     dW[m][n] = sum_p delta[p][m] * input[p][n] for the W-wide layers: the contraction runs over
//                          POINTS, so both MFMA operands are plain row-major reads (lane = feature, k = point).  One
//                          workgroup accumulates a whole 256x256 block (16 accumulator tiles per wave) over its share of
//                          the points; bias gradients are column sums of the same operand on the VALU under the MFMAs.
//   * wgrad_narrow_kernel  the products with a narrow side (encoded inputs, colour / density gradients): a wave owns
//                          the whole output, the workgroup's four waves add up through LDS.
//   * reduce_partial_kernel  deterministic slice reduction into the parameter-gradient vector.
//   * pack_apply_kernel    device-side re-pack of the weight blobs after optimizer.step().
// Nothing is differentiated w.r.t. the sample positions or view directions: the reference's only trainable
// tensors are the MLP parameters (train.py:149-152).
#include <type_traits>
#include <vector>
#include "mlp_core.h"

namespace minerf {

int stage_embed(const float*, const float*, int64_t, int, int, int, float*, hipStream_t);
int dgrad_f16s(const mi_nerf_net*, const void*, const float*, const float*, const float*, const unsigned*, const unsigned*, float*, float*, float*, int64_t, int,
               long long, long long, const unsigned*, hipStream_t);

// ---------------------------------------------------------------------------------------------
// backward data
// ---------------------------------------------------------------------------------------------
struct DgradArgs {
    const char* stream;       // backward blob + HEADER_BYTES
    unsigned stream_bytes;
    const float* side;        // forward blob's side tables (colour / density head weights)
    unsigned side_floats, o_dens_w, o_color_w;
    const float* d_raw;       // [P][4]  dL/d(rgb_raw, density_raw)
    const unsigned* mask_h;   // [D][n_wtiles][64][4]  ReLU' bits written by the training forward (same tiling, same lanes)
    const unsigned* mask_g;   // [n_wtiles][64][2]
    float* delta_h;           // [D][P][W]   dL/d(pre-activation of trunk layer l)
    float* delta_f;           // [P][W]      dL/d(linear_feat output)
    float* delta_d;           // [P][W/2]    dL/d(pre-activation of linear_d)
    long long P;              // rows per layer in the delta / stash buffers (n_rays * S)
    long long n_valid;        // rows that exist (== P for rays; the embedded-row mode pads its last tile)
    long long n_wtiles;       // n_rays * tpr: the forward's (ray, 32-sample chunk) tiles
    int S, tpr;
    int D;
};

template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[8]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

template <int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_dgrad_kernel(const DgradArgs a) {
    constexpr int NT = W / 32;
    constexpr int HN = W / 2;
    constexpr int AL = 8;               // ring_advance<ALLOW>: row stores are interleaved with the weight DMAs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* side = (float*)(smem + RING_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hh = lane >> 5;

    for (unsigned i = tid * 4; i < a.side_floats; i += 256 * 4) *(f32x4*)(side + i) = *(const f32x4*)(a.side + i);

    const long long n_wg_tiles = (a.n_wtiles + 3) >> 2;
    if ((long long)blockIdx.x >= n_wg_tiles) return;

    WRing ring;
    ring.sbase = a.stream + wave * (4 * QUAD_BYTES);
    ring.voff = lane * 16;
    ring.fetch_off = 0;
    ring.stream_bytes = a.stream_bytes;
    ring.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (4 * QUAD_BYTES);
    ring.lds_hi = ring.lds_lo + RING_BYTES;
    ring.fetch_lds = ring.lds_lo;
    ring.read_slot = NSLOT - 1;
#ifdef MN_DIAG
    ring.dlog = nullptr; ring.dcnt = 0;
#endif
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        if (sl) ring_next_fetch(ring);
        ring_dma<0>(ring); ring_dma<1>(ring); ring_dma<2>(ring); ring_dma<3>(ring);
    }

    f32x16 acc[8];
    f32x4 aq[8];
    float h[HN];
    ring_advance(ring);
#pragma unroll
    for (int t = 0; t < NT; ++t) aq[t] = ring_read(smem, ring, lane, t);

    const float* cw = side + a.o_color_w;
    const float* dw = side + a.o_dens_w;
    for (long long wgt = blockIdx.x; wgt < n_wg_tiles; wgt += gridDim.x) {
        long long wt = wgt * 4 + wave;
        const bool wave_active = wt < a.n_wtiles;
        if (!wave_active) wt = a.n_wtiles - 1;
        const long long ray = wt / a.tpr;
        const int sample = (int)(wt - ray * a.tpr) * 32 + col;
        long long idx = ray * a.S + (sample < a.S ? sample : a.S - 1);
        const bool valid = wave_active && sample < a.S && idx < a.n_valid;
        if (idx >= a.n_valid) idx = a.n_valid - 1;                 // padded rows recompute the last one
        // padding lanes (sample >= S) and inactive tail waves recompute a valid point: their row stores rewrite the same bytes
        const f32x4 dr = *(const f32x4*)(a.d_raw + idx * 4);
        const u32x2 mgv = *(const u32x2*)(a.mask_g + (wt * 64 + lane) * 2);
        u32x4 mhv = *(const u32x4*)(a.mask_h + (((long long)(a.D - 1) * a.n_wtiles + wt) * 64 + lane) * 4);

        // ---- colour head^T (VALU, 3 inputs) and ReLU' of linear_d ----
        float h2[HN / 2];
        {
            const unsigned mg[2] = {mgv[0], mgv[1]};
#pragma unroll
            for (int q = 0; q < HN / 8; ++q) {
                const f32x4 w0 = *(const f32x4*)(cw + 8 * q + 4 * hh);
                const f32x4 w1 = *(const f32x4*)(cw + W / 2 + 8 * q + 4 * hh);
                const f32x4 w2 = *(const f32x4*)(cw + W + 8 * q + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = dr[0] * w0[e];
                    v = __builtin_fmaf(dr[1], w1[e], v);
                    v = __builtin_fmaf(dr[2], w2[e], v);
                    h2[4 * q + e] = mask_apply(v, q, e, mg);
                }
            }
        }
        // every delta row is written by the GEMM that consumes it as its B operand: one 16-byte store per k-quad
        // ---- linear_d^T, feature block: d feature = Wd[:, :W]^T delta_d ----
        acc_zero<NT>(acc);
        {
            float* row = a.delta_d + idx * (W / 2) + 4 * hh;
            auto hook = [&](int kq, int t) __attribute__((always_inline)) { if (t == NT - 1) store_chunk(h2, kq, row); };
            gemm_part<NT, HN / 2, NT, AL>(acc, h2, aq, smem, ring, lane, hook);
        }
        acc_to_b<NT, false>(acc, h);
        // ---- linear_feat^T (+ density head^T, rank 1) -> gradient of the trunk output ----
        acc_zero<NT>(acc);
        {
            float* row = a.delta_f + idx * W + 4 * hh;
            auto hook = [&](int kq, int t) __attribute__((always_inline)) { if (t == NT - 1) store_chunk(h, kq, row); };
            gemm_part<NT, HN, NT, AL>(acc, h, aq, smem, ring, lane, hook);
        }
        // ---- trunk, last layer first.  The ReLU' epilogue of layer l produces delta_l in `h`; the loop body is [GEMM with
        // W_l^T, rows of delta_l stored from its hook] [epilogue of layer l-1].  The first epilogue (the trunk output also feeds
        // the density head: + dens_w * d sigma, rank 1 on the VALU) is peeled, so the loop holds ONE epilogue variant (with both
        // in the loop body hipcc spilled 124 bytes per lane around it).
        {
            const unsigned mw[4] = {mhv[0], mhv[1], mhv[2], mhv[3]};
            if (a.D > 1) mhv = *(const u32x4*)(a.mask_h + (((long long)(a.D - 2) * a.n_wtiles + wt) * 64 + lane) * 4);   // a layer ahead
            const float ds = dr[3];
#pragma unroll
            for (int q = 0; q < 4 * NT; ++q) {
                const f32x4 wv = *(const f32x4*)(dw + 8 * q + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    h[4 * q + e] = mask_apply(__builtin_fmaf(wv[e], ds, acc[q >> 2][4 * (q & 3) + e]), q, e, mw);    // ReLU'
            }
        }
#pragma unroll 1
        for (int l = a.D - 1; l > 0; --l) {
            float* row = a.delta_h + ((long long)l * a.P + idx) * W + 4 * hh;
            acc_zero<NT>(acc);
            auto hook = [&](int kq, int t) __attribute__((always_inline)) { if (t == NT - 1) store_chunk(h, kq, row); };
            gemm_part<NT, HN, NT, AL>(acc, h, aq, smem, ring, lane, hook);          // W_l[:, h-block]^T delta_l
            const unsigned mw[4] = {mhv[0], mhv[1], mhv[2], mhv[3]};
            if (l > 1) mhv = *(const u32x4*)(a.mask_h + (((long long)(l - 2) * a.n_wtiles + wt) * 64 + lane) * 4);       // a layer ahead
#pragma unroll
            for (int q = 0; q < 4 * NT; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) h[4 * q + e] = mask_apply(acc[q >> 2][4 * (q & 3) + e], q, e, mw);              // ReLU'
        }
        store_rows<NT>(h, a.delta_h + idx * W + 4 * hh, valid);      // no GEMM consumes delta_0 (there is no gradient w.r.t. gamma(x))
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// backward weights
// ---------------------------------------------------------------------------------------------
constexpr int WG_MAXB = 12;               // products per launch (an 8-layer net has 9 wide ones)
struct WgradArgs {                        // a BATCH of products of the same point set: blockIdx.y picks the product
    const float* dlt[WG_MAXB]; const float* x[WG_MAXB];     // delta [P, ldd] (columns [0, M)), layer input [P, ldx] (columns [0, N))
    int ldd[WG_MAXB], ldx[WG_MAXB], M[WG_MAXB], N[WG_MAXB]; // M, N <= 256
    int want_bias[WG_MAXB];
    long long P;
    float* partial;                       // [product][slices][256][256]
    float* bpartial;                      // [product][slices][256] column sums of delta
    int slices;
};
struct ReduceBatch {                      // where each product of a batch goes in the flat gradient vector
    float* out[WG_MAXB]; float* bias[WG_MAXB];
    int ldo[WG_MAXB], M[WG_MAXB], N[WG_MAXB];
};

// 256 x 256 block of dW per workgroup: 2x2 waves, each 4x4 MFMA tiles (all 256 accumulator registers).
// Operand fetch: ONE 16-byte load per lane per point pair and operand: lane i takes columns 4i..4i+3, i.e. MFMA tile
// t of this wave covers columns {4i + t}.  Any bijection lane <-> column works as long as the store uses the same one.
// Requires M, N, ldd, ldx multiples of 4 and 16-byte aligned operands (true for every W-wide layer).
//
// What a non-MFMA instruction costs here (tools/mfma_probe5.hip, cycles from a PMC pass): v_mfma_f32_32x32x2_f32 runs on the
// SIMD's fp32 lanes, so VALU work does NOT hide behind it -- every VALU instruction takes its 4 issue cycles out of the matrix
// rate plus ~8 cycles for each MFMA gap that holds any (14 v_add_f32 per 16 MFMAs: 66 -> 73-76 cycles per MFMA; in one gap: 70);
// SALU instructions and s_nop are free.  The first version of this loop spent ~25 VALU per k-step on addresses (64-bit row *
// pitch, the row < P and column < M selects) and the bias sums: 84 % of the MFMA rate in cycles.  Now the loop has NO vector
// address arithmetic: rows come through buffer loads whose descriptor (base = first row of the 12-row group, num_records = the
// bytes of that group that exist) is rebuilt per group on the SALU, the per-lane offsets of the six k-steps are loop constants,
// and the hardware's range check returns zeros for rows at or past P and for lanes whose columns do not exist (offset 2^31).
// The bias sums (4 v_add per k-step) are split between the two waves that hold the same delta columns (even / odd k-steps).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const float* base, long long row0, long long ld, long long P, int rows) {
    long long left = P - row0;                                   // rows of this group that exist (wave-uniform: SALU)
    left = left < 0 ? 0 : (left > rows ? rows : left);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + row0 * ld), 0, (int)(left * ld * 4), 0x00020000);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void wgrad_big_kernel(const WgradArgs a) {
    constexpr int U = 6;                  // k-steps (2 points each) per software-pipeline stage
    __shared__ f32x4 bshare[2][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, kh = lane >> 5;
    // (product, slice) of this workgroup.  Workgroups are dealt round robin over the 8 XCDs (observed, not promised: it only
    // affects speed), so consecutive linear ids would spread every product over every L2: nine operand pairs streaming through
    // each 4 MiB L2 at once, and the rows the two waves of a pair share get evicted between their two reads (FETCH_SIZE 18.4 GB
    // per fine-net batch for 14.1 GB of operands).  Ids that share an XCD (v % 8) are laid out consecutively instead, so an
    // L2 sees at most two products: 15.7 GB at the same speed.  (A barrier per three groups brings it to 14.1 GB, every operand
    // read once, but costs 0.17 % of the step: not shipped, tools/ABLATIONS.md.)
    const unsigned nwg = gridDim.x * gridDim.y, v = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = v & 7, slot = v >> 3;
    const unsigned full = nwg >> 3, rem = nwg & 7;               // XCDs 0..rem-1 host full + 1 workgroups, the others full
    const unsigned flat = xcd * full + (xcd < rem ? xcd : rem) + slot;
    const int b = (int)(flat / gridDim.x);                       // product of the batch
    const unsigned slice = flat - (unsigned)b * gridDim.x;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = wm * 128, n0 = wn * 128;
    const bool aok = m0 + 4 * i < a.M[b], bok = n0 + 4 * i < a.N[b];
    const float* abase = a.dlt[b];
    const float* bbase = a.x[b];
    const long long ldd = a.ldd[b], ldx = a.ldx[b];
    const bool want_bias = a.want_bias[b] != 0;

    f32x16 acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    // The points are dealt to the workgroups in GROUPS of 2U rows, round robin: group g of workgroup s is rows
    // [(g * slices + s) * 2U, ... + 2U).  At any moment the whole chip streams one window of slices x 2U consecutive rows per
    // operand (3 MB) instead of 256 separate regions hundreds of MB apart -- contiguous per-workgroup slices lose 10-25 % to
    // the memory system as P grows (tools/wgrad_probe.py).  Rows at or past P are never fetched (range check, above).
    // Three register sets rotated by NAME (loop unrolled by 3; a copy of a just-requested set forces s_waitcnt vmcnt(0)),
    // and the two loads of a k-step are issued right behind the 16 MFMAs of the same k-step of the current set.
    const long long slices = gridDim.x;
    unsigned voa[U], vob[U];                                      // byte offsets inside a group: row 2u + kh, columns m0 + 4i..
#pragma unroll
    for (int u = 0; u < U; ++u) {
        voa[u] = aok ? (unsigned)(((2 * u + kh) * ldd + m0 + 4 * i) * 4) : 0x80000000u;
        vob[u] = bok ? (unsigned)(((2 * u + kh) * ldx + n0 + 4 * i) * 4) : 0x80000000u;
    }
    long long req_group = slice;                                  // global index of the next group this workgroup requests
    f32x4 ca[U], cb[U], na[U], nb[U], fa[U], fb[U];
    __amdgpu_buffer_rsrc_t ra, rb;
    auto open_group = [&]() __attribute__((always_inline)) {
        const long long row0 = req_group * (2 * U);
        ra = wg_rsrc(abase, row0, ldd, a.P, 2 * U);
        rb = wg_rsrc(bbase, row0, ldx, a.P, 2 * U);
        req_group += slices;
    };
    auto request = [&](int u, f32x4& A, f32x4& B) __attribute__((always_inline)) {
        A = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa[u], 0, 0));
        B = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, vob[u], 0, 0));
    };
    open_group();
#pragma unroll
    for (int u = 0; u < U; ++u) request(u, ca[u], cb[u]);
    open_group();
#pragma unroll
    for (int u = 0; u < U; ++u) request(u, na[u], nb[u]);
    // PAR: which k-steps of a group this wave adds to the bias sums (0 even, 1 odd, -1 none).  A compile-time property of
    // the loop: as a run-time condition hipcc turns it into add + select on every k-step.
    auto step = [&](auto PAR, f32x4 (&A)[U], f32x4 (&B)[U], f32x4 (&FA)[U], f32x4 (&FB)[U]) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value;
        open_group();
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u][tm], B[u][tn], acc[tm][tn], 0, 0, 0);
            if ((u & 1) == par) bsum += A[u];
            request(u, FA[u], FB[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const long long all_groups = (a.P + 2 * U - 1) / (2 * U);
    const long long n_groups = (all_groups + slices - 1) / slices;       // per workgroup; surplus groups multiply zeros
    auto loop = [&](auto PAR) __attribute__((always_inline)) {
        long long g = 0;
        do {
            step(PAR, ca, cb, fa, fb);
            step(PAR, na, nb, ca, cb);
            step(PAR, fa, fb, na, nb);
            g += 3;
        } while (g < n_groups);
    };
    if (!want_bias) loop(std::integral_constant<int, -1>{});
    else if (wn == 0) loop(std::integral_constant<int, 0>{});
    else loop(std::integral_constant<int, 1>{});
    // D[i'][j]: i' = (r&3) + 8*(r>>2) + 4*kh is the A-side lane index, j = lane & 31 the B-side one
    // The lane index is taken afresh (mbcnt) for the read-out: every VGPR is spoken for inside the loop, and a lane-derived value kept
    // alive across it for these few lines was the kernel's one spilled register.
    const int lane_o = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int i_o = lane_o & 31, kh_o = lane_o >> 5;
    float* out = a.partial + ((size_t)b * a.slices + slice) * (256 * 256);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kh_o) + tm;
            f32x4 v;                                             // explicit just-in-time reads out of the AGPR file (see wgrad_narrow_kernel)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[tn]) : "a"(acc[tm][tn][r]));
            *(f32x4*)(out + (size_t)m * 256 + n0 + 4 * i_o) = v;
        }
    if (want_bias) {                                             // uniform over the workgroup
        if (wn == 1) bshare[wm][lane_o] = bsum;                    // odd k-steps of the same columns
        __syncthreads();
        if (wn == 0) {
            bsum += bshare[wm][lane_o];
            f32x4 sv;
#pragma unroll
            for (int e = 0; e < 4; ++e) sv[e] = bsum[e] + __shfl_xor(bsum[e], 32, 64);      // even + odd points of every k-step
            if (kh_o == 0) *(f32x4*)(a.bpartial + ((size_t)b * a.slices + slice) * 256 + m0 + 4 * i_o) = sv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same products in SPLIT PRECISION (the training step's extra mode, beside mlp_f16s.hip's forward): both operands converted on the
// fly to f16 pairs v = hi + lo (lo = f16(v - hi), unscaled) and a product taken as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 into
// ONE fp32 accumulator -- every f16 x f16 product is exact in fp32, the accumulation is the fp32 kernel's.  A sum over points needs
// ABSOLUTE accuracy: an operand element carries an error of at most 2^-25 (the f16 subnormal step) however small it is, 2^-22 relative
// for elements above 2^-3.  Activations are O(1) as they are; the gradient operand is lifted by a power of two s taken from max|d_raw|
// (absmax_bits, written on the device by absmax_kernel) so that its largest entries sit near 2^8 -- seven binades below the f16 maximum
// for what the transposed weights add on the way down; FP16_OVFL makes a conversion beyond that saturate instead of producing inf.  The
// result is scaled back by 1 / s (exact) on the way out.  Same workgroup shape, operand bijection, slices, partials and reduction as
// wgrad_big_kernel; a k-step is 16 points (lane half kh takes rows 8 kh .. 8 kh + 7 as the eight k-values of its 128-bit operand), loaded
// in units of 8 rows, four units (two k-steps, 32 KiB per wave) in flight.  At three times the fp32 matrix rate the kernel is bound by the
// HBM reads of its operands.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out_bits) {
    float m = 0.0f;
    const long long n4 = ((uintptr_t)x & 15) == 0 ? n >> 2 : 0;          // 16-byte pieces, then the tail (or everything) one by one
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)x)[i];
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))), __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = __builtin_fmaxf(m, __builtin_fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m == m) atomicMax(out_bits, __float_as_uint(m));      // non-negative floats order like their bit patterns
}
static void launch_absmax(const float* x, long long n, unsigned* out_bits, hipStream_t st) {
    long long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, out_bits);
}
// two fp32 values -> a packed f16 pair hi and the packed pair lo = f16(v * s - hi); S: scale by s first (gradient operand)
template <bool S>
__device__ __forceinline__ void split_pair(float v0, float v1, float s, unsigned& hi, unsigned& lo) {
    if constexpr (S) { v0 *= s; v1 *= s; }
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(v0), "v"(v1));
}
template <int BIAS>
__device__ __forceinline__ void wgrad_f16s_body(const WgradArgs& a, const unsigned* absmax_bits, f32x4 (*bshare)[64]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const unsigned nwg = gridDim.x * gridDim.y, v = blockIdx.y * gridDim.x + blockIdx.x;     // XCD-aware (product, slice): see wgrad_big_kernel
    const unsigned xcd = v & 7, slot = v >> 3;
    const unsigned full = nwg >> 3, rem = nwg & 7;
    const unsigned flat = xcd * full + (xcd < rem ? xcd : rem) + slot;
    const int b = (int)(flat / gridDim.x);
    const unsigned slice = flat - (unsigned)b * gridDim.x;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = wm * 128, n0 = wn * 128;
    const bool aok = m0 + 4 * i < a.M[b], bok = n0 + 4 * i < a.N[b];
    const float* abase = a.dlt[b];
    const float* bbase = a.x[b];
    const long long ldd = a.ldd[b], ldx = a.ldx[b];
    // scale of the gradient operand: max|d_raw| in [2^e, 2^(e+1)) -> s = 2^(7 - e), its largest entries land in [2^7, 2^8)
    const unsigned mb = __builtin_amdgcn_readfirstlane((int)*absmax_bits);
    int e = (int)((mb >> 23) & 255u) - 127;
    if (mb == 0u || e < -100) e = -100;                                 // all-zero (or denormal) gradients: any scale does
    if (e > 100) e = 100;
    const float sc = __uint_as_float((unsigned)(127 + 7 - e) << 23), inv_sc = __uint_as_float((unsigned)(127 - 7 + e) << 23);

    f32x16 acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    constexpr int GROUP = 32;                                           // rows per group = two k-steps = four load units of 8 rows
    const long long slices = gridDim.x;
    // unit q (0..3) of a group: k-step q >> 1, half h = q & 1; this lane's rows 16 (q >> 1) + 4 h + 8 kh + {0..3}.  One descriptor per unit
    // (base = the unit's first row, num_records = the bytes of its 12 rows that exist, rebuilt on the SALU), the whole per-lane offset in
    // the vector operand: rows at or past P and lanes without columns (offset 2^31) read as zeros by the range check, as in wgrad_big_kernel
    unsigned voa[4], vob[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        voa[r] = aok ? (unsigned)(((8 * kh + r) * ldd + m0 + 4 * i) * 4) : 0x80000000u;
        vob[r] = bok ? (unsigned)(((8 * kh + r) * ldx + n0 + 4 * i) * 4) : 0x80000000u;
    }
    long long req_group = slice;
    struct Unit { f32x4 A[4], B[4]; };
    auto request = [&](int q, Unit& u) __attribute__((always_inline)) {
        const long long row0 = req_group * GROUP + 16 * (q >> 1) + 4 * (q & 1);
        const __amdgpu_buffer_rsrc_t ra = wg_rsrc(abase, row0, ldd, a.P, 12), rb = wg_rsrc(bbase, row0, ldx, a.P, 12);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u.A[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa[r], 0, 0));
            u.B[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, vob[r], 0, 0));
        }
    };
    auto open_group = [&]() __attribute__((always_inline)) { req_group += slices; };
    // operands of the running k-step: [tile column t][dword]: dwords 0, 1 from the k-step's first unit, 2, 3 from its second
    unsigned ahi[4][4], alo[4][4], bhi[4][4], blo[4][4];
    auto convert = [&](int h, const Unit& u) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                split_pair<true>(u.A[2 * pr][t], u.A[2 * pr + 1][t], sc, ahi[t][2 * h + pr], alo[t][2 * h + pr]);
                split_pair<false>(u.B[2 * pr][t], u.B[2 * pr + 1][t], 1.0f, bhi[t][2 * h + pr], blo[t][2 * h + pr]);
            }
        if constexpr (BIAS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bsum += u.A[r];
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
        f16x8 AH[4], AL[4], BH[4], BL[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 w;
            w[0] = ahi[t][0]; w[1] = ahi[t][1]; w[2] = ahi[t][2]; w[3] = ahi[t][3]; AH[t] = __builtin_bit_cast(f16x8, w);
            w[0] = alo[t][0]; w[1] = alo[t][1]; w[2] = alo[t][2]; w[3] = alo[t][3]; AL[t] = __builtin_bit_cast(f16x8, w);
            w[0] = bhi[t][0]; w[1] = bhi[t][1]; w[2] = bhi[t][2]; w[3] = bhi[t][3]; BH[t] = __builtin_bit_cast(f16x8, w);
            w[0] = blo[t][0]; w[1] = blo[t][1]; w[2] = blo[t][2]; w[3] = blo[t][3]; BL[t] = __builtin_bit_cast(f16x8, w);
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[tm], BH[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[tm], BL[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[tm], BH[tn], acc[tm][tn], 0, 0, 0);
            }
    };
    Unit u0, u1, u2, u3;
    request(0, u0); request(1, u1); request(2, u2); request(3, u3);
    const long long all_groups = (a.P + GROUP - 1) / GROUP;
    const long long n_groups = (all_groups + slices - 1) / slices;       // per workgroup; surplus groups multiply zeros
    for (long long g = 0; g < n_groups; ++g) {
        // (the waves of a workgroup fetch each row twice between them: FETCH_SIZE 19.7 GB per fine-net batch for 14.1 GB of operands.  A barrier
        // per group keeps the second fetch in the L2 but costs more than it saves -- step 13.09 -> 13.58 ms; fetching each row once bounds the
        // gain at 13 % of this kernel: profiles/r04_wgrad_f16s_nodup_bound.txt.  Both experiments are closed: tools/ABLATIONS.md.)
        open_group();                                                    // the NEXT group's rows: every unit is requested a whole group ahead
        convert(0, u0); request(0, u0);
        convert(1, u1); request(1, u1);
        mfmas();
        convert(0, u2); request(2, u2);
        convert(1, u3); request(3, u3);
        mfmas();
    }
    const int lane_o = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int i_o = lane_o & 31, kh_o = lane_o >> 5;
    float* out = a.partial + ((size_t)b * a.slices + slice) * (256 * 256);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kh_o) + tm;
            f32x4 vv;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) { float t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[tm][tn][r])); vv[tn] = t * inv_sc; }
            *(f32x4*)(out + (size_t)m * 256 + n0 + 4 * i_o) = vv;
        }
    if (a.want_bias[b] != 0) {                                           // uniform over the workgroup; the wn == 0 waves summed every row
        if constexpr (BIAS) {
            f32x4 sv;
#pragma unroll
            for (int c = 0; c < 4; ++c) sv[c] = bsum[c] + __shfl_xor(bsum[c], 32, 64);
            if (kh_o == 0) *(f32x4*)(a.bpartial + ((size_t)b * a.slices + slice) * 256 + m0 + 4 * i_o) = sv;
        }
    }
    (void)bshare;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void wgrad_f16s_kernel(const WgradArgs a, const unsigned* absmax_bits) {
    // FP16_OVFL (MODE bit 23): f16 conversions that overflow saturate at the f16 maximum instead of returning inf
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((wave & 1) == 0) wgrad_f16s_body<1>(a, absmax_bits, nullptr);    // the two waves of a row pair hold the same gradient columns: one sums them
    else wgrad_f16s_body<0>(a, absmax_bits, nullptr);
}

// Products with one NARROW operand: the encoded inputs gamma(x) (63 columns) and gamma(d) (27), row pitch 90, and the
// 3-wide colour / 1-wide density gradients in d_raw (row pitch 4).  The other operand is a W- or W/2-wide row-major tensor.
// One WAVE owns all of the output, wide (128 WQ columns, fetched like wgrad_big_kernel: lane i takes columns 4i..4i+3 of
// each 128-column group with one 16-byte load) x narrow (32 NN columns, one 4-byte load each), over its own slice of the
// points; 4 WQ NN MFMAs per point pair against 16 WQ + 4 NN bytes per lane.  Output rows = wide index, columns = narrow
// index; the reduction transposes when the gradient operand is the narrow one.
struct NarrowArgs {
    const float* wide; int ldw; int Mw;      // [P, ldw], columns [0, Mw), Mw <= 128 WQ, 16-byte aligned, ldw % 4 == 0
    const float* nar;  int ldn; int Nn;      // [P, ldn], columns [0, Nn), Nn <= 32 NN
    long long P;
    int pps;                                 // points per slice (one slice per wave)
    float* partial;                          // [slices][128 WQ][32 NN]
    float* wsum;                             // [slices][128 WQ]  column sums of the wide operand, or NULL
    float* nsum;                             // [slices][32 NN]   column sums of the narrow operand, or NULL
};

// BW / BN: column sums of the wide / narrow operand wanted (the bias gradient of the layer).  Compile-time: VALU work beside
// fp32 MFMAs is never free (see wgrad_big_kernel), so the sums exist only in the instantiations that return them.
template <int WQ, int NN, bool BW, bool BN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, (WQ * NN == 4) ? 1 : 2)))
void wgrad_narrow_kernel(const NarrowArgs a) {
    constexpr int U = (WQ == 2) ? 4 : 6;      // k-steps per pipeline stage: 3 register sets next to 64 WQ NN accumulators
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, kh = lane >> 5;
    // one point slice per wave; the four waves of a workgroup add their results through LDS (fixed order) before the
    // workgroup writes ONE partial, so the reduction kernel sees a quarter of the slices
    const long long slice = (long long)blockIdx.x * 4 + wave;
    const long long pb = slice * a.pps;
    const long long pe = pb >= a.P ? pb : ((pb + a.pps < a.P) ? pb + a.pps : a.P);   // tail waves: empty slice, still join the barriers
    const long long ldw = a.ldw, ldn = a.ldn;

    // Operands through buffer loads, as in wgrad_big_kernel: the descriptor of a group of 2U rows is rebuilt on the SALU (base = the
    // group's first row, num_records = the bytes of it inside [pb, pe)), the per-lane offsets of the U k-steps are loop constants,
    // and the range check zeroes rows past the slice end and lanes whose column does not exist: no address arithmetic, no
    // row < P selects and no zeroing of over-fetched values on the VALU, and nothing is read outside the operands' P rows.
    unsigned vow[U][WQ], von[U][NN];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int q = 0; q < WQ; ++q) vow[u][q] = (128 * q + 4 * i < a.Mw) ? (unsigned)(((2 * u + kh) * ldw + 128 * q + 4 * i) * 4) : 0x80000000u;
#pragma unroll
        for (int q = 0; q < NN; ++q) von[u][q] = (32 * q + i < a.Nn) ? (unsigned)(((2 * u + kh) * ldn + 32 * q + i) * 4) : 0x80000000u;
    }

    f32x16 acc[4 * WQ][NN];
#pragma unroll
    for (int tm = 0; tm < 4 * WQ; ++tm)
#pragma unroll
        for (int tn = 0; tn < NN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
    f32x4 wsum[WQ];
    float nsum[NN];
#pragma unroll
    for (int q = 0; q < WQ; ++q) wsum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NN; ++q) nsum[q] = 0.0f;

    struct Set { f32x4 w[U][WQ]; float n[U][NN]; };
    Set c, nx, f;
    long long req_row = pb;                  // first row of the next group to request (wave-uniform)
    __amdgpu_buffer_rsrc_t rw, rn;
    auto open_group = [&]() __attribute__((always_inline)) {
        rw = wg_rsrc(a.wide, req_row, ldw, pe, 2 * U);
        rn = wg_rsrc(a.nar, req_row, ldn, pe, 2 * U);
        req_row += 2 * U;
    };
    auto request = [&](Set& S, int u) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < WQ; ++q) S.w[u][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, vow[u][q], 0, 0));
#pragma unroll
        for (int q = 0; q < NN; ++q) S.n[u][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rn, von[u][q], 0, 0));
    };
    open_group();
#pragma unroll
    for (int u = 0; u < U; ++u) request(c, u);
    open_group();
#pragma unroll
    for (int u = 0; u < U; ++u) request(nx, u);
    auto step = [&](Set& A, Set& F) __attribute__((always_inline)) {
        open_group();
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int tm = 0; tm < 4 * WQ; ++tm)
#pragma unroll
                for (int tn = 0; tn < NN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w[u][tm >> 2][tm & 3], A.n[u][tn], acc[tm][tn], 0, 0, 0);
            if (BW) {
#pragma unroll
                for (int q = 0; q < WQ; ++q) wsum[q] += A.w[u][q];
            }
            if (BN) {
#pragma unroll
                for (int q = 0; q < NN; ++q) nsum[q] += A.n[u][q];
            }
            request(F, u);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // (hipcc leaves ~100 bytes of scratch in the <2,2> instantiations: all of it on the edge that bypasses the loop for an empty
    // slice and in the read-out below.  Forcing the loop to run -- do-while, or an assumed n_groups >= 1 -- moves ~150 scratch
    // operations INTO the loop instead.)
    const long long n_groups = (pe - pb + 2 * U - 1) / (2 * U);
    for (long long g = 0; g < n_groups; g += 3) {
        step(c, f);
        step(nx, c);
        step(f, nx);
    }
    // D[i'][j]: i' = (r&3) + 8*(r>>2) + 4*kh <-> wide column 128*(tm>>2) + 4*i' + (tm&3);  j = lane & 31 <-> narrow column 32*tn + j
    constexpr int Wp = 128 * WQ, Np = 32 * NN;
    __shared__ float red[3][16][64];
    // The partial is written through a buffer descriptor: ONE per-lane offset (the lane's part of the element index: 16 kh rows + column
    // i) plus a compile-time scalar offset per (tile, register).  With plain 64-bit addressing hipcc materialised the 256 store
    // addresses of the unrolled read-out as VGPR pairs and spilled accumulators around them (the <2,2> instantiations: 64-100 bytes
    // of scratch, none of it in the loop).
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)blockIdx.x * Wp * Np), 0, Wp * Np * 4, 0x00020000);
    const unsigned vout = (unsigned)((16 * kh * Np + i) * 4);
#pragma unroll
    for (int tm = 0; tm < 4 * WQ; ++tm)
#pragma unroll
        for (int tn = 0; tn < NN; ++tn) {
            // the accumulators are read out of the AGPR file ONE TILE AT A TIME, by an explicit v_accvgpr_read: left to itself hipcc copies
            // all 256 into VGPRs at the loop exit (they are VALU / LDS-store operands from here on) and spills what does not fit
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t[r]) : "a"(acc[tm][tn][r]));
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = t[r];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = ((t[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
                    const int w0 = 128 * (tm >> 2) + 4 * ((r & 3) + 8 * (r >> 2)) + (tm & 3);      // + 16 kh: in the lane offset
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, vout, (w0 * Np + 32 * tn) * 4, 0);
                }
            }
            __syncthreads();
        }
    // column sums (bias gradients): even + odd points of every k-step, then the four waves
    float* sums = &red[0][0][0];             // [4 waves][Wp + Np] floats fit easily
#pragma unroll
    for (int q = 0; q < WQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = wsum[q][e] + __shfl_xor(wsum[q][e], 32, 64);
            if (kh == 0) sums[wave * (Wp + Np) + 128 * q + 4 * i + e] = v;
        }
#pragma unroll
    for (int q = 0; q < NN; ++q) {
        const float v = nsum[q] + __shfl_xor(nsum[q], 32, 64);
        if (kh == 0) sums[wave * (Wp + Np) + Wp + 32 * q + i] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Wp + Np; c += 256) {
        const float v = ((sums[c] + sums[(Wp + Np) + c]) + sums[2 * (Wp + Np) + c]) + sums[3 * (Wp + Np) + c];
        if (c < Wp) { if (a.wsum) a.wsum[(size_t)blockIdx.x * Wp + c] = v; }
        else if (a.nsum) a.nsum[(size_t)blockIdx.x * Np + (c - Wp)] = v;
    }
}

// out[m*ldo + n] = sum_s partial[s][m][n]  (m < M, n < N);  bias[m] = sum_s bpartial[s][m]
// transposed: the partials are stored [n][m] (the gradient operand was the narrow one), i.e. partial[s][n*Mp + m] with
// Mp the row pitch of that layout.
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial,
                                                              int slices, int Mp, int Np, int M, int N, float* __restrict__ out, int ldo,
                                                              float* __restrict__ bias, int bias_pitch, int transposed) {
    // a block sums 64 outputs: 4 groups of 64 threads take every 4th slice (8 loads in flight each), then add up through
    // LDS in a fixed order -- 4 blocks per CU instead of one thread per output walking all slices alone
    __shared__ float red[4][64];
    const int g = threadIdx.x >> 6, o = threadIdx.x & 63;
    const int idx = blockIdx.x * 64 + o;
    const int total = M * N + (bias ? M : 0);
    float acc = 0.0f;
    if (idx < M * N) {
        const int m = idx / N, n = idx - m * N;
        const float* p = partial + (transposed ? (size_t)n * Np + m : (size_t)m * Np + n);
        const size_t stride = (size_t)Mp * Np;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sl = g;
        for (; sl + 28 < slices; sl += 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += p[(size_t)(sl + 4 * k) * stride];
        }
        for (; sl < slices; sl += 4) s[0] += p[(size_t)sl * stride];
        acc = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    } else if (idx < total) {
        const int m = idx - M * N;
        for (int sl = g; sl < slices; sl += 4) acc += bpartial[(size_t)sl * bias_pitch + m];
    }
    red[g][o] = acc;
    __syncthreads();
    if (g == 0 && idx < total) {
        const float v = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
        if (idx < M * N) { const int m = idx / N, n = idx - m * N; out[(size_t)m * ldo + n] = v; }
        else bias[idx - M * N] = v;
    }
}

// the same for a batch of 256x256-block products (blockIdx.y = product; partial [product][slices][256][256])
__global__ __launch_bounds__(256) void reduce_batch_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial, int slices,
                                                            const ReduceBatch rb) {
    __shared__ float red[4][64];
    const int b = blockIdx.y;
    const int M = rb.M[b], N = rb.N[b];
    const int g = threadIdx.x >> 6, o = threadIdx.x & 63;
    const int idx = blockIdx.x * 64 + o;
    const int total = M * N + (rb.bias[b] ? M : 0);
    float acc = 0.0f;
    if (idx < M * N) {
        const int m = idx / N, n = idx - m * N;
        const float* p = partial + (size_t)b * slices * 65536 + (size_t)m * 256 + n;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sl = g;
        for (; sl + 28 < slices; sl += 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += p[(size_t)(sl + 4 * k) * 65536];
        }
        for (; sl < slices; sl += 4) s[0] += p[(size_t)sl * 65536];
        acc = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    } else if (idx < total) {
        const int m = idx - M * N;
        for (int sl = g; sl < slices; sl += 4) acc += bpartial[((size_t)b * slices + sl) * 256 + m];
    }
    red[g][o] = acc;
    __syncthreads();
    if (g == 0 && idx < total) {
        const float v = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
        if (idx < M * N) { const int m = idx / N, n = idx - m * N; rb.out[b][(size_t)m * rb.ldo[b] + n] = v; }
        else rb.bias[b][idx - M * N] = v;
    }
}

// blob[i] = map[i] ? flat[map[i] - 1] : 0      (device-side re-pack after an optimiser step; map from pack_map)
__global__ __launch_bounds__(256) void pack_apply_kernel(const int32_t* __restrict__ map, const float* __restrict__ flat, long long n,
                                                          float* __restrict__ blob) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t m = map[i];
    blob[i] = m > 0 ? flat[m - 1] : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int num_cus_t() { return device_cus(); }

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
// Slack behind the operands whose rows are narrower than 1 KB.  The first weight-gradient kernels ran their load pipelines past the
// end of a point slice and needed it; the current ones fetch through range-checked buffer loads and read nothing outside rows
// [0, P).  Kept: it costs 256 KB and leaves the layout (mi_nerf_train_layout_query) unchanged.
constexpr size_t WGRAD_OVERRUN_PAD = 128 * 1024;

constexpr size_t WGRAD_PARTIAL_FLOATS = (size_t)256 * 256 * 256 + (size_t)256 * 256;   // 256 slices of a 256x256 block + bias rows

int train_layout(const mi_nerf_net* net, int64_t n_rays, int S, mi_nerf_train_layout* L) {
    MN_CHECK_ARG(net && L, "NULL net/layout");
    MN_CHECK_ARG(native_width(net->W), "the training kernels exist for W = 128 and 256 (got %d; inference pads narrower networks)", net->W);
    MN_CHECK_ARG(net->D >= 2 && net->D <= 16, "unsupported depth D=%d", net->D);
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    const size_t p = (size_t)n_rays * S, W = (size_t)net->W, D = (size_t)net->D;
    const size_t n_wtiles = (size_t)n_rays * ((S + 31) / 32);
    const size_t in_all = (size_t)(3 + 6 * net->L_x) + (size_t)(3 + 6 * net->L_d);
    size_t off = 0;
    L->stash_h = off; off += al256(D * p * W * 4);
    L->stash_f = off; off += al256(p * W * 4);
    L->stash_g = off; off += al256(p * (W / 2) * 4) + WGRAD_OVERRUN_PAD;
    L->mask_h = off;  off += al256(D * n_wtiles * 64 * 16);
    L->mask_g = off;  off += al256(n_wtiles * 64 * 8);
    L->stash_bytes = off;
    off = 0;
    L->delta_h = off; off += al256(D * p * W * 4);
    L->delta_f = off; off += al256(p * W * 4);
    L->delta_d = off; off += al256(p * (W / 2) * 4) + WGRAD_OVERRUN_PAD;
    L->emb = off;     off += al256(p * in_all * 4);
    L->partial = off; off += al256(WGRAD_PARTIAL_FLOATS * 4);
    off += 256;                                                  // max|d_raw| of the call (split-precision weight gradients), right behind the partials
    L->work_bytes = off;
    return MI_NERF_OK;
}

// A batch of W- or W/2-wide products over the same points in ONE launch: the CUs are split between the products, so every
// product is cut into num_cus / n slices instead of num_cus -- the partials written and reduced per product (256 KB per
// slice) shrink by the same factor, and so does the fixed cost that kept a lone 256x256 product at ~75 % of the MFMA peak.
struct WideProduct { const float* dlt; int ldd, M; const float* x; int ldx, N; float* out; int ldo; float* bias; };

// absmax: NULL = fp32 MFMA (wgrad_big_kernel); else the device word holding max|d_raw| and the products run in split precision
static int run_wgrad_batch(const WideProduct* pr, int n, long long P, float* partial, hipStream_t st, const unsigned* absmax = nullptr) {
    MN_CHECK_ARG(n >= 1 && n <= WG_MAXB, "%d wide products in one batch (at most WG_MAXB)", n);
    WgradArgs a{};
    ReduceBatch rb{};
    int max_total = 0;
    for (int b = 0; b < n; ++b) {
        const WideProduct& q = pr[b];
        MN_CHECK_ARG(q.M <= 256 && q.N <= 256 && q.M % 4 == 0 && q.N % 4 == 0 && q.ldd % 4 == 0 && q.ldx % 4 == 0 &&
                     ((uintptr_t)q.dlt & 15) == 0 && ((uintptr_t)q.x & 15) == 0,
                     "wgrad operands must be at most 256 wide, 16-byte aligned, with pitches of 4 floats");
        a.dlt[b] = q.dlt; a.x[b] = q.x; a.ldd[b] = q.ldd; a.ldx[b] = q.ldx; a.M[b] = q.M; a.N[b] = q.N; a.want_bias[b] = q.bias != nullptr;
        rb.out[b] = q.out; rb.bias[b] = q.bias; rb.ldo[b] = q.ldo; rb.M[b] = q.M; rb.N[b] = q.N;
        const int total = q.M * q.N + (q.bias ? q.M : 0);
        if (total > max_total) max_total = total;
    }
    int slices = num_cus_t() / n;                              // one workgroup per CU owns all registers
    const long long all_groups = (P + 11) / 12;                // groups of 2 x U points, dealt round robin to the slices
    if (slices > all_groups) slices = (int)all_groups;
    if (slices < 1) slices = 1;
    MN_CHECK_ARG((size_t)n * slices * (65536 + 256) <= WGRAD_PARTIAL_FLOATS, "internal: wgrad partial buffer too small");
    a.P = P; a.slices = slices;
    a.partial = partial;
    a.bpartial = partial + (size_t)n * slices * 65536;
    if (absmax) hipLaunchKernelGGL(wgrad_f16s_kernel, dim3(slices, n), dim3(256), 0, st, a, absmax);
    else hipLaunchKernelGGL(wgrad_big_kernel, dim3(slices, n), dim3(256), 0, st, a);
    MN_LAUNCH_CHECK("wgrad_big_kernel");
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((max_total + 63) / 64, n), dim3(256), 0, st, (const float*)partial, (const float*)a.bpartial,
                       slices, rb);
    MN_LAUNCH_CHECK("reduce_batch_kernel");
    return MI_NERF_OK;
}

// one dW (+ optional bias) = delta^T x input, slice partials reduced into `out`
static int run_wgrad(const float* dlt, int ldd, int M, const float* x, int ldx, int N, long long P, float* out, int ldo, float* bias,
                     float* partial, hipStream_t st) {
    if (M > 64 && N > 64) {                  // W- or W/2-wide on both sides: a batch of one
        const WideProduct q{dlt, ldd, M, x, ldx, N, out, ldo, bias};
        return run_wgrad_batch(&q, 1, P, partial, st);
    }
    // one narrow side: the wide operand is whichever has more columns; the partials come out [wide][narrow]
    const bool delta_is_wide = M >= N;
    NarrowArgs a{};
    a.wide = delta_is_wide ? dlt : x; a.ldw = delta_is_wide ? ldd : ldx; a.Mw = delta_is_wide ? M : N;
    a.nar = delta_is_wide ? x : dlt;  a.ldn = delta_is_wide ? ldx : ldd; a.Nn = delta_is_wide ? N : M;
    a.P = P;
    MN_CHECK_ARG(a.Mw <= 256 && a.Nn <= 64 && a.Mw % 4 == 0 && a.ldw % 4 == 0 && ((uintptr_t)a.wide & 15) == 0,
                 "unsupported narrow wgrad shape %d x %d", M, N);
    const int WQ = a.Mw > 128 ? 2 : 1, NN = a.Nn > 32 ? 2 : 1;
    const int Wp = 128 * WQ, Np = 32 * NN;
    const size_t per_slice = (size_t)Wp * Np + Wp + Np;
    long long wslices = 4LL * num_cus_t();                 // one point slice per wave, one partial per workgroup
    long long pps = (P + wslices - 1) / wslices;
    pps = (pps + 23) / 24 * 24;                          // whole load groups for U = 4 and U = 6
    wslices = (P + pps - 1) / pps;
    const long long slices = (wslices + 3) / 4;
    MN_CHECK_ARG((size_t)slices * per_slice <= WGRAD_PARTIAL_FLOATS, "internal: wgrad partial buffer too small");
    a.pps = (int)pps;
    a.partial = partial;
    float* wsum = partial + (size_t)slices * Wp * Np;
    float* nsum = wsum + (size_t)slices * Wp;
    a.wsum = (bias && delta_is_wide) ? wsum : nullptr;
    a.nsum = (bias && !delta_is_wide) ? nsum : nullptr;
    const dim3 grid((unsigned)slices);
    const bool bw = a.wsum != nullptr, bn = a.nsum != nullptr;
#define MN_NARROW(WQ_, NN_)                                                                                                   \
    do {                                                                                                                      \
        if (bw) hipLaunchKernelGGL((wgrad_narrow_kernel<WQ_, NN_, true, false>), grid, dim3(256), 0, st, a);                  \
        else if (bn) hipLaunchKernelGGL((wgrad_narrow_kernel<WQ_, NN_, false, true>), grid, dim3(256), 0, st, a);             \
        else hipLaunchKernelGGL((wgrad_narrow_kernel<WQ_, NN_, false, false>), grid, dim3(256), 0, st, a);                    \
    } while (0)
    if (WQ == 2 && NN == 2) MN_NARROW(2, 2);
    else if (WQ == 2) MN_NARROW(2, 1);
    else if (NN == 2) MN_NARROW(1, 2);
    else MN_NARROW(1, 1);
#undef MN_NARROW
    MN_LAUNCH_CHECK("wgrad_narrow_kernel");
    const int total = M * N + (bias ? M : 0);
    // reduce: rows of the partial are the wide index.  delta wide: out[m][n] = partial[m][n];  delta narrow: out[m][n] = partial[n][m]
    hipLaunchKernelGGL(reduce_partial_kernel, dim3((total + 63) / 64), dim3(256), 0, st, (const float*)partial,
                       (const float*)(delta_is_wide ? wsum : nsum), (int)slices, Wp, Np, M, N, out, ldo, bias, delta_is_wide ? Wp : Np,
                       delta_is_wide ? 0 : 1);
    MN_LAUNCH_CHECK("reduce_partial_kernel");
    return MI_NERF_OK;
}

size_t wgrad_scratch_bytes() { return WGRAD_PARTIAL_FLOATS * 4 + 256; }      // partials + the max|delta| word of the split-precision entry

// n (<= 12) products over the SAME P points -- how the backward pass itself runs the nine 256 x 256 products of a network
// (run_wgrad_batch): the WIDE ones (both sides wider than 64 columns) share ONE launch, the CUs shared out between them, so a product is
// cut into num_cus / n point slices and writes / reduces 1 / n of the partials a launch of its own does; a product with a narrow side
// (gamma(x), gamma(d), the heads) goes through wgrad_narrow_kernel in a launch of its own, in list order behind the wide batch.
// f16s: the products in split precision (wgrad_f16s_kernel; wide products only); the gradient operands are scaled from the largest |delta|
// entry of the batch, found on the device first (one pass over the deltas: the training step takes it from d_raw instead)
int wgrad_products(int n, const float* const* dlt, const int* ldd, const int* M, const float* const* x, const int* ldx, const int* N, int64_t P,
                   float* const* out, const int* ldo, float* const* bias, void* scratch, size_t scratch_bytes, hipStream_t st, bool f16s) {
    MN_CHECK_ARG(n >= 1 && n <= WG_MAXB, "between 1 and %d products per launch (got %d)", WG_MAXB, n);
    MN_CHECK_ARG(P >= 1 && dlt && ldd && M && x && ldx && N && out && ldo && scratch, "bad sizes / NULL pointer");
    MN_CHECK_ARG(scratch_bytes >= wgrad_scratch_bytes(), "scratch too small: %zu < %zu", scratch_bytes, wgrad_scratch_bytes());
    WideProduct pr[WG_MAXB];
    int n_wide = 0;
    for (int b = 0; b < n; ++b) {
        MN_CHECK_ARG(dlt[b] && x[b] && out[b] && M[b] >= 1 && N[b] >= 1 && ldd[b] >= M[b] && ldx[b] >= N[b] && ldo[b] >= N[b],
                     "product %d: bad sizes M=%d N=%d (pitches %d / %d / %d) or NULL pointer", b, M[b], N[b], ldd[b], ldx[b], ldo[b]);
        const bool wide = M[b] > 64 && N[b] > 64;
        MN_CHECK_ARG(wide || !f16s, "product %d: the split-precision entry takes wide products (more than 64 columns on both sides), M=%d N=%d", b, M[b], N[b]);
        if (wide) pr[n_wide++] = WideProduct{dlt[b], ldd[b], M[b], x[b], ldx[b], N[b], out[b], ldo[b], bias ? bias[b] : nullptr};
    }
    unsigned* absmax = nullptr;
    if (f16s) {
        absmax = (unsigned*)((char*)scratch + WGRAD_PARTIAL_FLOATS * 4);
        MN_HIP(hipMemsetAsync(absmax, 0, 4, st));
        for (int b = 0; b < n; ++b) {
            MN_CHECK_ARG(ldd[b] == M[b], "product %d: the split-precision entry takes gradient operands without row padding (ldd == M)", b);
            launch_absmax(dlt[b], (long long)P * M[b], absmax, st);
        }
        MN_LAUNCH_CHECK("absmax_kernel");
    }
    if (n_wide)
        if (int rc = run_wgrad_batch(pr, n_wide, P, (float*)scratch, st, absmax)) return rc;
    for (int b = 0; b < n; ++b)                                  // the same scratch, in stream order
        if (!(M[b] > 64 && N[b] > 64))
            if (int rc = run_wgrad(dlt[b], ldd[b], M[b], x[b], ldx[b], N[b], P, out[b], ldo[b], bias ? bias[b] : nullptr, (float*)scratch, st)) return rc;
    return MI_NERF_OK;
}

template <int W>
static int launch_dgrad(const DgradArgs& a, hipStream_t st) {
    const size_t lds = RING_BYTES + (size_t)a.side_floats * 4;
    MN_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded: %zu bytes", lds);
    auto kern = mlp_dgrad_kernel<W>;
    static LdsOptIn opt_in = {};
    if (int rc = ensure_lds_opt_in(opt_in, (const void*)kern)) return rc;
    const long long n_wg = (a.n_wtiles + 3) / 4;
    const int grid = (int)(n_wg < (long long)num_cus_t() ? n_wg : (long long)num_cus_t());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    MN_LAUNCH_CHECK("mlp_dgrad_kernel");
    return MI_NERF_OK;
}

// d_raw [P,4] -> flat parameter gradient (ParamOffsets order).  stash: written by mlp_rays_fp32_stash for the same
// rays/z; work: scratch of train_layout().work_bytes.  stage 0: everything; 1: stop after the backward-data kernel
// (deltas stay in `work` for inspection).
// x_dev != NULL: the embedded-row mode (the forward was mlp_embedded_fp32_stash over n_rows rows): rays / z are unused, the
// layer inputs gamma(x), gamma(d) are the caller's rows, and (n_rays, S) must be (ceil(n_rows / 32), 32).
int mlp_backward_fp32(const mi_nerf_net* net, const void* packed_fwd, const void* packed_bwd, const float* rays, const float* z,
                      int64_t n_rays, int S, const float* d_raw, const void* stash, void* work, size_t work_bytes, float* grads,
                      int stage, hipStream_t st, const float* x_dev, int64_t n_rows, int mode) {
    MN_CHECK_ARG(net != nullptr, "net is NULL");
    MN_CHECK_ARG(net->L_x >= 0 && net->L_x <= KERNEL_LX && net->L_d >= 0 && net->L_d <= KERNEL_LD, "unsupported encoding L_x=%d L_d=%d", net->L_x, net->L_d);
    MN_CHECK_ARG(net->skip >= -1, "bad skip=%d", net->skip);
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    const long long Ppad = (long long)n_rays * S;                  // rows per layer in the stash / delta buffers
    const long long P = x_dev ? n_rows : Ppad;                     // rows that exist
    MN_CHECK_ARG(!x_dev || (S == 32 && n_rows >= 0 && n_rays == (n_rows + 31) / 32), "embedded mode wants (n_rays, S) = (ceil(n / 32), 32)");
    mi_nerf_train_layout L;
    if (int rc = train_layout(net, n_rays, S, &L)) return rc;
    if (P == 0) {                                                  // an empty batch: the gradient of nothing is zero (what autograd gives), never the caller's uninitialised buffer
        if (grads && stage != 1) {
            MN_CHECK_ARG(net->D >= 1 && net->D <= 16 && (net->W == 128 || net->W == 256), "unsupported network D=%d W=%d", net->D, net->W);
            MN_HIP(hipMemsetAsync(grads, 0, (size_t)make_param_offsets(net->D, net->W, net->skip, net->L_x, net->L_d).total * 4, st));
        }
        if ((mode & 3) && work && work_bytes >= L.work_bytes)      // ... and the split-precision range words say "nothing seen", not whatever the allocator left there
            MN_HIP(hipMemsetAsync((char*)work + L.partial + al256(WGRAD_PARTIAL_FLOATS * 4), 0, 8, st));
        return MI_NERF_OK;
    }
    MN_CHECK_ARG(packed_fwd && packed_bwd && (x_dev || (rays && z)) && d_raw && stash && work && grads, "NULL device pointer");
    MN_CHECK_ARG(work_bytes >= L.work_bytes, "workspace too small: %zu < %zu", work_bytes, L.work_bytes);
    const int D = net->D, W = net->W;
    const int in_x = 3 + 6 * net->L_x, in_d = 3 + 6 * net->L_d, in_all = in_x + in_d;
    const BlobLayout BL = make_layout(D, W, net->skip, net->L_x, net->L_d);
    const ParamOffsets po = make_param_offsets(D, W, net->skip, net->L_x, net->L_d);
    const float* stash_h = (const float*)((const char*)stash + L.stash_h);
    const float* stash_f = (const float*)((const char*)stash + L.stash_f);
    const float* stash_g = (const float*)((const char*)stash + L.stash_g);
    float* delta_h = (float*)((char*)work + L.delta_h);
    float* delta_f = (float*)((char*)work + L.delta_f);
    float* delta_d = (float*)((char*)work + L.delta_d);
    float* emb = (float*)((char*)work + L.emb);
    float* partial = (float*)((char*)work + L.partial);

    DgradArgs a{};
    a.stream = (const char*)packed_bwd + HEADER_BYTES;
    a.stream_bytes = bwd_stream_bytes(D, W);
    a.side = (const float*)((const char*)packed_fwd + BL.side_off);
    a.side_floats = BL.side_floats; a.o_dens_w = BL.dens_w; a.o_color_w = BL.color_w;
    a.d_raw = d_raw;
    a.mask_h = (const unsigned*)((const char*)stash + L.mask_h);
    a.mask_g = (const unsigned*)((const char*)stash + L.mask_g);
    a.delta_h = delta_h; a.delta_f = delta_f; a.delta_d = delta_d;
    a.S = S; a.tpr = (S + 31) / 32;
    a.P = Ppad; a.n_valid = P; a.n_wtiles = (long long)n_rays * a.tpr; a.D = D;
    // mode bit 0: the wide weight-gradient products, bit 1: the backward-data chain in split precision (packed_bwd is then the blob of
    // mi_nerf_pack_weights_bwd_f16s); both scale their gradient operands from max|d_raw| of this call, taken on the device
    unsigned* absmax = nullptr;
    if (mode & 3) {
        MN_CHECK_ARG(W == 256 && !x_dev, "the split-precision backward is built for W=256 and the ray entry point");
        absmax = (unsigned*)((char*)work + L.partial + al256(WGRAD_PARTIAL_FLOATS * 4));
        MN_HIP(hipMemsetAsync(absmax, 0, 8, st));              // [0] max|d_raw|, [1] max|delta * s| (written by the split-precision backward-data kernel)
        launch_absmax(d_raw, (long long)P * 4, absmax, st);
        MN_LAUNCH_CHECK("absmax_kernel");
    }
    if (mode & 2) {
        if (int rc = dgrad_f16s(net, packed_bwd, a.side + a.o_color_w, a.side + a.o_dens_w, d_raw, a.mask_h, a.mask_g, delta_h, delta_f, delta_d, n_rays, S,
                                Ppad, P, absmax, st)) return rc;
    } else {
        if (int rc = (W == 256 ? launch_dgrad<256>(a, st) : launch_dgrad<128>(a, st))) return rc;
    }
    if (stage == 1) return MI_NERF_OK;

    // layer inputs gamma(x), gamma(d) as rows (nerf_process.py:69-85), or the caller's own rows
    const float* embc = x_dev;
    if (!x_dev) {
        if (int rc = stage_embed(rays, z, n_rays, S, net->L_x, net->L_d, emb, st)) return rc;
        embc = emb;
    }
    const size_t PW = (size_t)Ppad * W;
    if (!(mode & 1)) absmax = nullptr;                          // the wide products stay on the fp32 matrix pipe
    // wide products (both sides W or W/2 wide) in one launch: trunk layers 1..D-1 (activation part), linear_feat, linear_d (feature part)
    const float* h_last = stash_h + (size_t)(D - 1) * PW;
    {
        WideProduct pr[WG_MAXB];
        int n = 0;
        auto flush = [&]() -> int {
            const int rc = n ? run_wgrad_batch(pr, n, P, partial, st, absmax) : MI_NERF_OK;
            n = 0;
            return rc;
        };
        for (int l = 1; l < D; ++l) {
            const bool cat = po.in_l[l] != W;
            pr[n++] = WideProduct{delta_h + (size_t)l * PW, W, W, stash_h + (size_t)(l - 1) * PW, W, W, grads + po.w_x[l] + (cat ? in_x : 0),
                                  po.in_l[l], grads + po.b_x[l]};
            if (n == WG_MAXB) if (int rc = flush()) return rc;
        }
        pr[n++] = WideProduct{delta_f, W, W, h_last, W, W, grads + po.w_feat, W, grads + po.b_feat};
        if (n == WG_MAXB) if (int rc = flush()) return rc;
        pr[n++] = WideProduct{delta_d, W / 2, W / 2, stash_f, W, W, grads + po.w_d, W + in_d, grads + po.b_d};
        if (int rc = flush()) return rc;
    }
    // products with a narrow side: gamma(x) into layer 0 and the skip layer, gamma(d) into linear_d, the density and colour heads
    if (int rc = run_wgrad(delta_h, W, W, embc, in_all, in_x, P, grads + po.w_x[0], in_x, grads + po.b_x[0], partial, st)) return rc;
    for (int l = 1; l < D; ++l)
        if (po.in_l[l] != W)
            if (int rc = run_wgrad(delta_h + (size_t)l * PW, W, W, embc, in_all, in_x, P, grads + po.w_x[l], po.in_l[l], nullptr, partial, st)) return rc;
    if (int rc = run_wgrad(d_raw + 3, 4, 1, h_last, W, W, P, grads + po.w_dens, W, grads + po.b_dens, partial, st)) return rc;
    if (int rc = run_wgrad(delta_d, W / 2, W / 2, embc + in_x, in_all, in_d, P, grads + po.w_d + W, W + in_d, nullptr, partial, st)) return rc;
    if (int rc = run_wgrad(d_raw, 4, 3, stash_g, W / 2, W / 2, P, grads + po.w_color, W / 2, grads + po.b_color, partial, st)) return rc;
    return MI_NERF_OK;
}

int pack_apply(const int32_t* map_dev, const float* flat_dev, size_t blob_bytes, void* blob_dev, hipStream_t st) {
    MN_CHECK_ARG(map_dev && flat_dev && blob_dev, "NULL device pointer");
    MN_CHECK_ARG(blob_bytes % 4 == 0, "blob size must be a multiple of 4");
    const long long n = (long long)(blob_bytes / 4);
    if (n == 0) return MI_NERF_OK;
    hipLaunchKernelGGL(pack_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, map_dev, flat_dev, n, (float*)blob_dev);
    MN_LAUNCH_CHECK("pack_apply_kernel");
    return MI_NERF_OK;
}

}  // namespace minerf
