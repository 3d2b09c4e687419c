// mlp_fp32_wide.hip -- the fused positional-encoding + NeRF MLP forward for networks WIDER than 256 (257 <= netWidth <= 512, config.py:57;
// padded to 384 or 512 by the packer, layout.h kernel_width), fp32 MFMA, inference only.
//
// Replaces the same reference code as mlp_fp32.hip (nerf_process.py:69-85, :190-194 / :206-209; model/NeRF.py:33-52).
//
// Why a second kernel.  mlp_fp32.hip keeps a wave's 32 points in v_mfma_f32_32x32x2_f32 accumulator layout: W / 2 registers of B operand
// plus W / 2 of accumulators per lane -- at W = 512 that is the whole 512-entry register file before a single weight quad is held.  Here a
// wave owns 16 points on v_mfma_f32_16x16x4_f32 (the same 64 FLOP / clock / SIMD): W / 4 registers of B operand + W / 4 of accumulators = the
// budget of the 256-wide kernel.  Everything else is that kernel's design: one persistent workgroup per CU, one wave per SIMD, the
// accumulator of layer l IS the B operand of layer l + 1 (register r of output tile t on lane quarter q holds feature 16t + 4q + r = the
// four features of k-step 4t + r), weights streamed L2 -> LDS by LDS-DMA through mlp_core.h's 4 x 16 KiB ring in consumption order
// (pack.cpp: the W16 stream), gamma(x) in registers, the view-direction block of linear_d hoisted to a per-ray bias, heads on the VALU.
// Two differences: the 8-pass MFMA needs 40 cycles before its accumulator can be chained, so a group interleaves TWO output tiles (each
// accumulator every 64 cycles); and a k-quad of a 512-wide layer is 32 quads = two ring slots, so the A-operand pipeline is a rotating
// file of 8 quads read 8 positions ahead instead of one register per output tile.
#include <stdlib.h>
#include "common.h"
#include "layout.h"
#include "mlp_core.h"

namespace minerf {

struct WideArgs {
    const char* stream;
    const float* side;
    const float* rays;        // MODE 0: [n_rays, 6]
    const float* z;           // MODE 0: [n_rays, S]
    const float* x;           // MODE 1: [n_pts, in_x + in_d], the network's own widths
    int Lx_net, Ld_net;
    float* out;               // [n_pts, 4]
    long long n_wtiles;       // 16-point wave tiles
    long long n_pts;          // MODE 1
    long long n_rays, walk_ray, walk_carry, n_iter;      // MODE 0 tile walk (mlp_fp32.hip)
    int walk_chunk, ray_major;
    int S, tpr;               // MODE 0: wave tiles per ray = ceil(S / 16)
    int D, skip_layer;
    unsigned stream_bytes, side_floats;
    unsigned o_bias_trunk, o_bias_feat, o_bias_d, o_dens_w, o_dens_b, o_color_w, o_color_b, o_wdir_t;
};

// ---- building blocks in the 16x16 layout (lane = (col = lane & 15: the point, q4 = lane >> 4: the quarter)) ----------------------------------
template <int NTP>
__device__ __forceinline__ void acc_init16(f32x4 (&acc)[32], const float* vec_lds, int q4) {
#pragma unroll
    for (int t = 0; t < NTP; ++t) acc[t] = *(const f32x4*)(vec_lds + 16 * t + 4 * q4);
}
template <int NTP, bool RELU, int NB>
__device__ __forceinline__ void acc_to_b16(const f32x4 (&acc)[32], float (&h)[NB]) {
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[4 * t + r] = RELU ? relu_pinned(acc[t][r]) : acc[t][r];
}
// sum over this lane's quarter of the features: h[4t + r] * w[16t + 4q + r]; w natural order in LDS
template <int N, int NB>
__device__ __forceinline__ float dot_quarter(const float (&h)[NB], const float* w_lds, int q4) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int t = 0; t < N / 4; ++t) {
        const f32x4 w = *(const f32x4*)(w_lds + 16 * t + 4 * q4);
        s0 = __builtin_fmaf(h[4 * t + 0], w[0], s0);
        s1 = __builtin_fmaf(h[4 * t + 1], w[1], s1);
        s2 = __builtin_fmaf(h[4 * t + 2], w[2], s2);
        s3 = __builtin_fmaf(h[4 * t + 3], w[3], s3);
    }
    return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ float quarter_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// One GEMM part: acc[0 .. NTP) += A(stream) x B, B = KS per-lane registers.  `a` is the rotating file of 8 A quads: quad i of the part sits in
// a[i % 8] and is replaced, once its four MFMAs are issued, by quad i + 8 (every part is a multiple of 8 quads long and slot aligned, so the
// next part's first quads take over where this one's leave off: NEXT_Q = how many of them exist, at most 8).
template <int NTP, int KS, int NEXT_Q, int NB>
__device__ __forceinline__ void gemm_part16(f32x4 (&acc)[32], const float (&b)[NB], f32x4 (&a)[8], const char* smem, WRing& ring, int lane) {
    static_assert(KS % 4 == 0 && KS <= NB && NTP % 2 == 0 && (NTP * (KS / 4)) % 8 == 0, "parts are whole k-quads of tile pairs");
    constexpr int KQ = KS / 4, QUADS = KQ * NTP;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
#pragma unroll
        for (int tp = 0; tp < NTP; tp += 2) {
            const int i0 = kq * NTP + tp;                      // quad index of tile tp (tile tp + 1: i0 + 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) {                      // two independent accumulator chains: 64 cycles between links (40 needed)
                acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i0 % 8][j], b[4 * kq + j], acc[tp], 0, 0, 0);
                acc[tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i0 + 1) % 8][j], b[4 * kq + j], acc[tp + 1], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int nq = i0 + e + 8;                     // the quad that takes the slot just consumed
                if (nq < QUADS) {
                    if (nq % SLOT_QUADS == 0) ring_advance<4>(ring);
                    a[nq % 8] = ring_read(smem, ring, lane, nq % SLOT_QUADS);
                } else if (nq - QUADS < NEXT_Q) {
                    if (nq == QUADS) ring_advance<4>(ring);    // the next part starts a fresh slot
                    a[(nq - QUADS) % 8] = ring_read(smem, ring, lane, nq - QUADS);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                 // pin the (8 MFMA, 2 LDS read) group order
        }
    }
}

// encoded-input registers: k-step s < ceil(3L / 2) holds, on quarter q, sin (q & 1 == 0) or cos of 2^k p_c for combination m = 2s + (q >> 1)
template <int L, bool SLOW, int NPE>
__device__ __forceinline__ void encode_regs16(float (&pe)[NPE], const float (&p)[3], int q4) {
    constexpr int COMBOS = 3 * L, STEPS = (COMBOS + 1) / 2;
    const bool upper = (q4 >> 1) != 0;
    const int fn = q4 & 1;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int mA = 2 * s, mB = 2 * s + 1;
        const float yA = p[mA % 3] * (float)(1 << (mA / 3));
        const float yB = mB < COMBOS ? p[mB % 3] * (float)(1 << (mB / 3)) : 0.0f;
        const float y = upper ? yB : yA;
        const float v = SLOW ? sin_cos_slow(y, fn) : sin_cos_fast(y, fn);
        pe[s] = (upper && mB >= COMBOS) ? 0.0f : v;
    }
    pe[STEPS] = q4 == 0 ? p[0] : (q4 == 1 ? p[1] : (q4 == 2 ? p[2] : 0.0f));
#pragma unroll
    for (int s = STEPS + 1; s < NPE; ++s) pe[s] = 0.0f;
}
// the same registers from a pre-embedded row holding L_net <= L frequencies
template <int L, int NPE>
__device__ __forceinline__ void gather_regs16(float (&pe)[NPE], const float* row, int q4, int L_net) {
    constexpr int COMBOS = 3 * L, STEPS = (COMBOS + 1) / 2;
    const bool upper = (q4 >> 1) != 0;
    const int fn = q4 & 1;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int mA = 2 * s, mB = 2 * s + 1;
        const int m = upper ? mB : mA;
        const bool have = m < COMBOS && (m / 3) < L_net;
        pe[s] = have ? row[have ? 3 + 6 * (m / 3) + (m % 3) + 3 * fn : 0] : 0.0f;       // never index past a row with fewer frequencies
    }
    pe[STEPS] = q4 < 3 ? row[q4] : 0.0f;
#pragma unroll
    for (int s = STEPS + 1; s < NPE; ++s) pe[s] = 0.0f;
}

template <int W, int MODE, int LX, int LD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_fp32_wide_kernel(const WideArgs a) {
    constexpr int NT = W / 16;            // output tiles of a W-wide layer (32)
    constexpr int HN = W / 4;             // activation registers per lane (128)
    constexpr int KPE = pe_ksteps16(LX);  // 16
    constexpr int KDE = pe_ksteps16(LD);  // 8
    constexpr int IN_D = 3 + 6 * LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* side = (float*)(smem + RING_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, q4 = lane >> 4;
    float* scratch = side + a.side_floats + wave * (W / 2);   // per-wave hoisted direction bias

    for (unsigned i = tid * 4; i < a.side_floats; i += 256 * 4) *(f32x4*)(side + i) = *(const f32x4*)(a.side + i);
    if ((long long)blockIdx.x >= ((a.n_wtiles + 3) >> 2)) return;     // host never launches such a block

    WRing ring;
    ring.sbase = a.stream + wave * (4 * QUAD_BYTES);
    ring.voff = lane * 16;
    ring.fetch_off = 0;
    ring.stream_bytes = a.stream_bytes;
    ring.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (4 * QUAD_BYTES);
    ring.lds_hi = ring.lds_lo + RING_BYTES;
    ring.fetch_lds = ring.lds_lo;
    ring.read_slot = NSLOT - 1;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        if (sl) ring_next_fetch(ring);
        ring_dma<0>(ring); ring_dma<1>(ring); ring_dma<2>(ring); ring_dma<3>(ring);
    }

    f32x4 acc[32];
    f32x4 aq[8];
    float h[HN];
    float pe[KPE];
    ring_advance(ring);                   // also publishes the side tables (barrier)
#pragma unroll
    for (int i = 0; i < 8; ++i) aq[i] = ring_read(smem, ring, lane, i);

    long long t_ray = 0;
    int t_chunk = 0;
    long long bias_ray = -1;
    float nx_o[3] = {0.f, 0.f, 0.f}, nx_d[3] = {0.f, 0.f, 0.f}, nx_z = 0.f;
    auto tile_inputs = [&](long long ray, int chunk) __attribute__((always_inline)) {
        const int sample = chunk * 16 + col;
        const int sc = sample < a.S ? sample : a.S - 1;
        const float* rp = a.rays + ray * 6;
        nx_o[0] = rp[0]; nx_o[1] = rp[1]; nx_o[2] = rp[2]; nx_d[0] = rp[3]; nx_d[1] = rp[4]; nx_d[2] = rp[5];
        nx_z = a.z[ray * a.S + sc];
    };
    const long long wid = (long long)blockIdx.x * 4 + wave;
    if constexpr (MODE == 0) {
        if (a.ray_major) { t_ray = wid; t_chunk = 0; }
        else { t_ray = wid / a.tpr; t_chunk = (int)(wid - t_ray * a.tpr); }
        const bool in = t_ray < a.n_rays;
        tile_inputs(in ? t_ray : a.n_rays - 1, in ? t_chunk : a.tpr - 1);
    }
    for (long long it = 0; it < a.n_iter; ++it) {
        bool valid;
        long long out_idx;
        float de[KDE];
        if constexpr (MODE == 0) {
            const bool wave_active = t_ray < a.n_rays;
            const long long ray = wave_active ? t_ray : a.n_rays - 1;          // the tail recomputes the last tile
            const int chunk = wave_active ? t_chunk : a.tpr - 1;
            const int sample = chunk * 16 + col;
            valid = wave_active && sample < a.S;
            out_idx = ray * a.S + (sample < a.S ? sample : a.S - 1);
            const float ox = nx_o[0], oy = nx_o[1], oz = nx_o[2], dx = nx_d[0], dy = nx_d[1], dz = nx_d[2];
            const float zv = nx_z;
            t_ray += a.walk_ray;
            t_chunk += a.walk_chunk;
            if (t_chunk >= a.tpr) { t_chunk -= a.tpr; t_ray += a.walk_carry; }
            {
                const bool in = t_ray < a.n_rays;
                tile_inputs(in ? t_ray : a.n_rays - 1, in ? t_chunk : a.tpr - 1);
            }
            const float p[3] = {ox + dx * zv, oy + dy * zv, oz + dz * zv};     // nerf_process.py:69-70, no contraction
            const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p[0]), __builtin_fabsf(p[1])), __builtin_fabsf(p[2])) * (float)(1 << (LX - 1));
            if (__builtin_expect(amax < SINCOS_FAST_LIMIT, 1)) encode_regs16<LX, false>(pe, p, q4);
            else encode_regs16<LX, true>(pe, p, q4);
            if (ray != bias_ray) {        // hoisted view-direction term of linear_d, per ray: scratch[n] = b_d[n] + sum_f Wd[n][W + f] gamma(d / |d|)[f]
                bias_ray = ray;
                const float nrm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
                const float v[3] = {dx / nrm, dy / nrm, dz / nrm};
                float g[IN_D];
                g[0] = v[0]; g[1] = v[1]; g[2] = v[2];
#pragma unroll
                for (int k = 0; k < LD; ++k)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float y = v[c] * (float)(1 << k);
                        g[3 + 6 * k + c] = sin_cos_fast(y, 0);
                        g[3 + 6 * k + 3 + c] = sin_cos_fast(y, 1);
                    }
                const float* wdt = side + a.o_wdir_t;
                const float* bd = side + a.o_bias_d;
#pragma unroll
                for (int n0 = 0; n0 < W / 2; n0 += 64) {
                    const int n = n0 + lane;
                    float sacc = bd[n];
#pragma unroll
                    for (int f = 0; f < IN_D; ++f) sacc = __builtin_fmaf(wdt[f * (W / 2) + n], g[f], sacc);
                    scratch[n] = sacc;
                }
            }
        } else {
            long long wt = it * ((long long)gridDim.x * 4) + wid;
            const bool wave_active = wt < a.n_wtiles;
            if (!wave_active) wt = a.n_wtiles - 1;
            const long long p0 = wt * 16 + col;
            valid = wave_active && p0 < a.n_pts;
            out_idx = p0 < a.n_pts ? p0 : a.n_pts - 1;
            const int in_x_net = 3 + 6 * a.Lx_net;
            const float* row = a.x + out_idx * (in_x_net + 3 + 6 * a.Ld_net);
            gather_regs16<LX>(pe, row, q4, a.Lx_net);
            gather_regs16<LD>(de, row + in_x_net, q4, a.Ld_net);
        }

        // ---- trunk ----
        acc_init16<NT>(acc, side + a.o_bias_trunk, q4);
        gemm_part16<NT, KPE, 8>(acc, pe, aq, smem, ring, lane);
#pragma unroll 1
        for (int l = 1; l < a.D; ++l) {
            acc_to_b16<NT, true>(acc, h);
            acc_init16<NT>(acc, side + a.o_bias_trunk + l * W, q4);
            if (l == a.skip_layer) gemm_part16<NT, KPE, 8>(acc, pe, aq, smem, ring, lane);       // cat([gamma(x), h]) order
            gemm_part16<NT, HN, 8>(acc, h, aq, smem, ring, lane);
        }
        acc_to_b16<NT, true>(acc, h);
        const float dens = quarter_sum(dot_quarter<HN>(h, side + a.o_dens_w, q4)) + side[a.o_dens_b];
        // ---- feature layer (no activation) ----
        acc_init16<NT>(acc, side + a.o_bias_feat, q4);
        gemm_part16<NT, HN, 8>(acc, h, aq, smem, ring, lane);
        acc_to_b16<NT, false>(acc, h);
        // ---- view-direction layer ----
        if constexpr (MODE == 0) {
            acc_init16<NT / 2>(acc, scratch, q4);
            gemm_part16<NT / 2, HN, 8>(acc, h, aq, smem, ring, lane);
        } else {
            acc_init16<NT / 2>(acc, side + a.o_bias_d, q4);
            gemm_part16<NT / 2, HN, 8>(acc, h, aq, smem, ring, lane);
            gemm_part16<NT / 2, KDE, 8>(acc, de, aq, smem, ring, lane);
        }
        float h2[HN / 2];
        acc_to_b16<NT / 2, true>(acc, h2);
        // ---- colour head ----
        const float* cw = side + a.o_color_w;
        const float r0 = quarter_sum(dot_quarter<HN / 2>(h2, cw, q4)) + side[a.o_color_b + 0];
        const float r1 = quarter_sum(dot_quarter<HN / 2>(h2, cw + W / 2, q4)) + side[a.o_color_b + 1];
        const float r2 = quarter_sum(dot_quarter<HN / 2>(h2, cw + W, q4)) + side[a.o_color_b + 2];
        if (valid && q4 == 0) {
            f32x4 o; o[0] = r0; o[1] = r1; o[2] = r2; o[3] = dens;     // cat([rgb, density]) NeRF.py:51
            *(f32x4*)(a.out + out_idx * 4) = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring runs 3 slots ahead: let the last prefetches land before the LDS is released
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int W, int MODE>
static int launch_wide(const WideArgs& args_in, long long n_wtiles, hipStream_t st) {
    WideArgs args = args_in;
    const size_t lds = RING_BYTES + (size_t)args.side_floats * 4 + 4 * (W / 2) * 4;
    MN_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded: %zu bytes (a %d-deep network of width > 256)", lds, args.D);
    auto kern = mlp_fp32_wide_kernel<W, MODE, KERNEL_LX, KERNEL_LD>;
    static LdsOptIn opt_in = {};
    if (int rc = ensure_lds_opt_in(opt_in, (const void*)kern)) return rc;
    const long long n_wg = (n_wtiles + 3) / 4;
    const int grid = (int)(n_wg < (long long)device_cus() ? n_wg : (long long)device_cus());
    const long long NW = (long long)grid * 4;
    const long long it_ray = (args.n_rays + NW - 1) / NW * args.tpr, it_flat = (n_wg + grid - 1) / grid;
    args.ray_major = (MODE == 0 && args.n_rays >= NW && it_ray <= it_flat) ? 1 : 0;
    if (args.ray_major) {
        args.walk_ray = 0; args.walk_chunk = 1; args.walk_carry = NW;
        args.n_iter = it_ray;
    } else {
        const long long tpr = MODE == 0 ? args.tpr : 1;
        args.walk_ray = NW / tpr; args.walk_chunk = (int)(NW % tpr); args.walk_carry = 1;
        args.n_iter = it_flat;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, args);
    MN_LAUNCH_CHECK("mlp_fp32_wide_kernel");
    return MI_NERF_OK;
}

static void fill_wide(WideArgs& a, const mi_nerf_net* net, const void* packed_dev, bool full_stream) {
    const BlobLayout L = make_layout(net->D, net->W, net->skip, net->L_x, net->L_d);
    a.stream = (const char*)packed_dev + L.stream_off;
    a.side = (const float*)((const char*)packed_dev + L.side_off);
    a.D = net->D;
    a.Lx_net = net->L_x; a.Ld_net = net->L_d;
    a.skip_layer = (net->skip >= 0 && net->skip + 1 < net->D) ? net->skip + 1 : -1;
    a.stream_bytes = full_stream ? L.stream_bytes_full : L.stream_bytes_hoist;
    a.side_floats = L.side_floats;
    a.o_bias_trunk = L.bias_trunk; a.o_bias_feat = L.bias_feat; a.o_bias_d = L.bias_d;
    a.o_dens_w = L.dens_w; a.o_dens_b = L.dens_b; a.o_color_w = L.color_w; a.o_color_b = L.color_b;
    a.o_wdir_t = L.wdir_t;
}

// callers (mlp_fp32.hip) have checked the net and the pointers
int mlp_rays_fp32_wide(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev, int64_t n_rays, int S,
                       float* raw_dev, hipStream_t st) {
    WideArgs a{};
    fill_wide(a, net, packed_dev, false);
    a.rays = rays_dev; a.z = z_dev; a.out = raw_dev; a.S = S; a.tpr = (S + 15) / 16;
    a.n_wtiles = (long long)n_rays * a.tpr; a.n_rays = n_rays;
    return kernel_width(net->W) == 384 ? launch_wide<384, 0>(a, a.n_wtiles, st) : launch_wide<512, 0>(a, a.n_wtiles, st);
}

int mlp_embedded_fp32_wide(const mi_nerf_net* net, const void* packed_dev, const float* x_dev, int64_t n, float* out_dev, hipStream_t st) {
    WideArgs a{};
    fill_wide(a, net, packed_dev, true);
    a.x = x_dev; a.out = out_dev; a.n_pts = n;
    a.n_wtiles = (n + 15) / 16;
    return kernel_width(net->W) == 384 ? launch_wide<384, 1>(a, a.n_wtiles, st) : launch_wide<512, 1>(a, a.n_wtiles, st);
}

}  // namespace minerf
