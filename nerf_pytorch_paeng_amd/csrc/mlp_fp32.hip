// mlp_fp32.hip -- fused positional-encoding + NeRF MLP forward on fp32 MFMA (gfx950).
//
// Replaces: nerf_process.py:69-85 (point build + posenc), :190-194 / :206-209 (chunked model eval),
//           model/NeRF.py:33-52 (NeRFModule.forward), model/PositionalEncoding.py:29-30.
//
// Design (MI355X-first):
//   * one persistent 256-thread workgroup per CU, ONE wave per SIMD with the whole 512-entry register file;
//   * a wave owns 32 points (MFMA column j = lane & 31) and keeps all their activations in registers in
//     v_mfma_f32_32x32x2_f32 accumulator layout; the accumulator of layer l IS the B operand of layer l+1
//     (bias + ReLU applied in place) -- activations never touch LDS or HBM;
//   * weights are the A operand, pre-permuted on the host into consumption order ("stream" of 1 KiB quads,
//     layout.h) and streamed L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into a 4-slot x 16 KiB ring
//     shared by the 4 waves; one barrier per slot = per 64 MFMAs per wave;
//   * gamma(x) is computed in registers per lane (sin on lanes 0-31, cos on lanes 32-63 of each k-step);
//     in fused mode the view-direction block of linear_d is hoisted to a per-ray bias (it is constant
//     over a ray's samples), density and colour heads run on the VALU as register dot products.
#include <stdlib.h>
#include <vector>
#include "common.h"
#include "layout.h"
#include "mlp_core.h"


namespace minerf {


struct MlpArgs {
    const char* stream;       // device: blob + stream_off
    const float* side;        // device: blob + side_off
    const float* rays;        // MODE 0: [n_rays, 6]
    const float* z;           // MODE 0: [n_rays, S]
    const float* x;           // MODE 1: [n_pts, in_x + in_d] with the NETWORK's own widths in_x = 3 + 6 Lx_net, in_d = 3 + 6 Ld_net
    int Lx_net, Ld_net;       // the network's frequencies (<= the kernel's LX / LD; the missing channels carry zero weights, layout.h)
    float* out;               // [n_pts, 4]
    long long n_wtiles;       // 32-point wave tiles
    long long n_pts;          // MODE 1
    // MODE 0 tile walk of one wave: start at wave id w (ray-major: ray w, chunk 0; tile-major: tile w), then per step
    // (ray, chunk) += (walk_ray, walk_chunk), chunk overflow carries walk_carry rays.  n_iter steps for every wave.
    long long n_rays, walk_ray, walk_carry, n_iter;
    int walk_chunk, ray_major;
    int S;                    // MODE 0
    int tpr;                  // MODE 0: wave tiles per ray = ceil(S / 32)
    int D;
    int skip_layer;           // trunk layer index that consumes [gamma(x), h]; -1: none
    unsigned stream_bytes;    // hoisted or full length, multiple of SLOT_BYTES
    unsigned side_floats;
    unsigned o_bias_trunk, o_bias_feat, o_bias_d, o_dens_w, o_dens_b, o_color_w, o_color_b, o_wdir_t;
    unsigned long long* diag;  // MN_DIAG builds only: per-wave segment cycle sums + one tile's k-quad stamps
    // STASH kernels (training forward): row-major activations kept for the backward pass, P = stash_rows points
    float* stash_h;           // [D][P][W]   post-ReLU output of trunk layer l
    float* stash_f;           // [P][W]      linear_feat output
    float* stash_g;           // [P][W/2]    post-ReLU linear_d output
    unsigned* mask_h;         // [D][n_wtiles][64 lanes][4]  ReLU' bits of stash_h in the kernel's register order
    unsigned* mask_g;         // [n_wtiles][64 lanes][2]     ... of stash_g
    long long stash_rows;
};


// encoded-input registers for k-step s: level k = s/3, axis c = s%3
template <int L, bool SLOW, int NPE>
__device__ __forceinline__ void encode_regs(float (&pe)[NPE], const float (&p)[3], int hh) {
#pragma unroll
    for (int s = 0; s < 3 * L; ++s) {
        const float y = p[s % 3] * (float)(1 << (s / 3));      // exact: power-of-two scale
        pe[s] = SLOW ? sin_cos_slow(y, hh) : sin_cos_fast(y, hh);
    }
    pe[3 * L] = hh ? p[1] : p[0];
    pe[3 * L + 1] = hh ? 0.0f : p[2];
#pragma unroll
    for (int s = 3 * L + 2; s < NPE; ++s) pe[s] = 0.0f;
}

// gather the same registers from a pre-embedded row (MODE 1); the row holds L_net <= L frequencies: the others read as zero
template <int L, int NPE>
__device__ __forceinline__ void gather_regs(float (&pe)[NPE], const float* row, int hh, bool valid, int L_net) {
#pragma unroll
    for (int s = 0; s < 3 * L; ++s) {
        const int ch = 3 + 6 * (s / 3) + (s % 3) + 3 * hh;
        const bool have = valid && (s / 3) < L_net;
        pe[s] = have ? row[have ? ch : 0] : 0.0f;
    }
    pe[3 * L] = valid ? row[hh] : 0.0f;
    pe[3 * L + 1] = (valid && !hh) ? row[2] : 0.0f;
#pragma unroll
    for (int s = 3 * L + 2; s < NPE; ++s) pe[s] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// the kernel.  MODE 0: rays + z (fused posenc, hoisted view-direction bias).  MODE 1: embedded rows.
// ---------------------------------------------------------------------------------------------
template <int W, int MODE, int LX, int LD, bool STASH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_fp32_kernel(const MlpArgs a) {
    constexpr int NT = W / 32;          // output tiles of a W-wide layer
    constexpr int HN = W / 2;           // activation registers per lane
    constexpr int KPE = pe_ksteps(LX);  // 32
    constexpr int KDE = pe_ksteps(LD);  // 16
    constexpr int IN_D = 3 + 6 * LD;
    // groups (t) of a k-quad that carry the training hooks: ReLU' bits, their merge, the row store
    constexpr int HK_B = NT - 3, HK_M = NT - 2, HK_S = NT - 1;
    constexpr int AL = STASH ? 8 : 4;         // ring_advance<ALLOW>: the training forward interleaves row stores with the DMAs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* side = (float*)(smem + RING_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hh = lane >> 5;
    float* scratch = side + a.side_floats + wave * (W / 2);   // per-wave hoisted direction bias

    for (unsigned i = tid * 4; i < a.side_floats; i += 256 * 4) *(f32x4*)(side + i) = *(const f32x4*)(a.side + i);

    const long long n_wg_tiles = (a.n_wtiles + 3) >> 2;
    if ((long long)blockIdx.x >= n_wg_tiles) return;     // host never launches such a block

    WRing ring;
    ring.sbase = a.stream + wave * (4 * QUAD_BYTES);
    ring.voff = lane * 16;
    ring.fetch_off = 0;
    ring.stream_bytes = a.stream_bytes;
    ring.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (4 * QUAD_BYTES);
    ring.lds_hi = ring.lds_lo + RING_BYTES;
    ring.fetch_lds = ring.lds_lo;
    ring.read_slot = NSLOT - 1;         // first ring_advance moves to slot 0
    // slots 0..2 in bursts; from then on slot p+3 streams in, one DMA per group, while slot p is consumed
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        if (sl) ring_next_fetch(ring);
        ring_dma<0>(ring); ring_dma<1>(ring); ring_dma<2>(ring); ring_dma<3>(ring);
    }

    f32x16 acc[8];
    f32x4 aq[8];                        // A-operand pipeline (gemm_part)
    float h[HN];
    float pe[KPE];
    ring_advance(ring);                 // also publishes the side tables (barrier)
#pragma unroll
    for (int t = 0; t < NT; ++t) aq[t] = ring_read(smem, ring, lane, t);

#ifdef MN_DIAG
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = mn_stamp();
#endif
    // MODE 0 tile walk without a 64-bit division per tile.  Large batches walk RAY-MAJOR: a wave takes whole rays (ray w,
    // w + #waves, ...) and their 32-sample chunks back to back, so the per-ray work of the prologue (view-direction
    // encoding and its hoisted linear_d term, ~550 VALU per lane) is done once per ray instead of once per tile.  Small
    // batches keep the tile-major walk (tile w, w + #waves, ...), which spreads few rays over more workgroups.
    // The inputs of the NEXT tile (six ray scalars, one depth per lane) are loaded a whole tile ahead.
    long long t_ray = 0;
    int t_chunk = 0;
    long long bias_ray = -1;            // ray whose hoisted direction term sits in `scratch`
    float nx_o[3] = {0.f, 0.f, 0.f}, nx_d[3] = {0.f, 0.f, 0.f}, nx_z = 0.f;
    auto tile_inputs = [&](long long ray, int chunk) __attribute__((always_inline)) {
        const int sample = chunk * 32 + col;
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
#ifdef MN_DIAG
        ring.dlog = (a.diag && it == 2 && blockIdx.x < 4) ? a.diag + (size_t)gridDim.x * 32 + ((size_t)blockIdx.x * 4 + wave) * 512 : nullptr;
        ring.dcnt = 0;
#endif
        bool wave_active;
        long long wt;
        bool valid;
        long long out_idx;
        float de[KDE];
        if constexpr (MODE == 0) {
            wave_active = t_ray < a.n_rays;
            const long long ray = wave_active ? t_ray : a.n_rays - 1;          // the tail recomputes the last tile
            const int chunk = wave_active ? t_chunk : a.tpr - 1;
            wt = ray * a.tpr + chunk;
            const int sample = chunk * 32 + col;
            valid = wave_active && sample < a.S;
            const int sc = sample < a.S ? sample : a.S - 1;
            out_idx = ray * a.S + sc;
            const float ox = nx_o[0], oy = nx_o[1], oz = nx_o[2], dx = nx_d[0], dy = nx_d[1], dz = nx_d[2];
            const float zv = nx_z;
            // advance to this wave's next tile and start loading its inputs now
            t_ray += a.walk_ray;
            t_chunk += a.walk_chunk;
            if (t_chunk >= a.tpr) { t_chunk -= a.tpr; t_ray += a.walk_carry; }
            {
                const bool in = t_ray < a.n_rays;
                tile_inputs(in ? t_ray : a.n_rays - 1, in ? t_chunk : a.tpr - 1);
            }
            // pts = rays_o + rays_d * z : separate multiply and add (nerf_process.py:69-70), no contraction
            const float p[3] = {ox + dx * zv, oy + dy * zv, oz + dz * zv};
            const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p[0]), __builtin_fabsf(p[1])), __builtin_fabsf(p[2])) * (float)(1 << (LX - 1));
            if (__builtin_expect(amax < SINCOS_FAST_LIMIT, 1)) encode_regs<LX, false>(pe, p, hh);
            else encode_regs<LX, true>(pe, p, hh);               // huge or non-finite coordinates: libm path
            // hoisted view-direction term of linear_d: scratch[n] = b_d[n] + sum_f Wd[n][W+f] * gamma(d/|d|)[f]
            // (per RAY: kept in this wave's LDS scratch while the wave stays on the ray)
            if (ray != bias_ray) {
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
                        g[3 + 6 * k + c] = sin_cos_fast(y, 0);       // |y| <= 2^(LD-1): no fallback needed
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
            wt = it * ((long long)gridDim.x * 4) + wid;
            wave_active = wt < a.n_wtiles;
            if (!wave_active) wt = a.n_wtiles - 1;
            const long long p0 = wt * 32 + col;
            valid = wave_active && p0 < a.n_pts;
            out_idx = p0 < a.n_pts ? p0 : a.n_pts - 1;
            const int in_x_net = 3 + 6 * a.Lx_net;
            const float* row = a.x + out_idx * (in_x_net + 3 + 6 * a.Ld_net);
            gather_regs<LX>(pe, row, hh, true, a.Lx_net);
            gather_regs<LD>(de, row + in_x_net, hh, true, a.Ld_net);
        }

        MN_STAMP(0);   // prologue
        // ---- trunk ----
        // STASH (training forward): every activation row is written while the NEXT GEMM consumes the same registers as
        // its B operand -- one 16-byte store per k-quad from the group hook -- and the ReLU' bit masks are packed there too.
        acc_init<NT>(acc, side + a.o_bias_trunk, hh);
        gemm_part<NT, KPE, NT, AL>(acc, pe, aq, smem, ring, lane);
        MN_STAMP(1);   // layer 0
#pragma unroll 1
        for (int l = 1; l < a.D; ++l) {
            acc_to_b<NT, true>(acc, h);
            acc_init<NT>(acc, side + a.o_bias_trunk + l * W, hh);
            if (l == a.skip_layer) gemm_part<NT, KPE, NT, AL>(acc, pe, aq, smem, ring, lane);   // cat([gamma(x), h]) order
            if constexpr (STASH) {
                float* row = a.stash_h + ((long long)(l - 1) * a.stash_rows + out_idx) * W + 4 * hh;
                unsigned mw[4] = {0u, 0u, 0u, 0u};
                unsigned mb[4];
                auto hook = [&](int kq, int t) __attribute__((always_inline)) {
                    if (t == HK_B) mask_bits_chunk(h, kq, mb);
                    if (t == HK_M) mask_merge_chunk(mb, kq, mw);
                    if (t == HK_S) store_chunk(h, kq, row);
                };
                gemm_part<NT, HN, NT, AL>(acc, h, aq, smem, ring, lane, hook);
                if (wave_active) {
                    u32x4 m; m[0] = mw[0]; m[1] = mw[1]; m[2] = mw[2]; m[3] = mw[3];
                    *(u32x4*)(a.mask_h + (((long long)(l - 1) * a.n_wtiles + wt) * 64 + lane) * 4) = m;
                }
            } else {
                gemm_part<NT, HN, NT, AL>(acc, h, aq, smem, ring, lane);
            }
        }
        acc_to_b<NT, true>(acc, h);
        MN_STAMP(2);   // trunk layers 1..D-1
        // ---- density head (VALU dot over the trunk output) ----
        const float dens = xhalf_sum(dot_half<HN>(h, side + a.o_dens_w, hh)) + side[a.o_dens_b];
        // ---- feature layer (no activation) ----
        acc_init<NT>(acc, side + a.o_bias_feat, hh);
        if constexpr (STASH) {
            float* row = a.stash_h + ((long long)(a.D - 1) * a.stash_rows + out_idx) * W + 4 * hh;
            unsigned mw[4] = {0u, 0u, 0u, 0u};
            unsigned mb[4];
            auto hook = [&](int kq, int t) __attribute__((always_inline)) {
                if (t == HK_B) mask_bits_chunk(h, kq, mb);
                if (t == HK_M) mask_merge_chunk(mb, kq, mw);
                if (t == HK_S) store_chunk(h, kq, row);
            };
            gemm_part<NT, HN, NT / 2, AL>(acc, h, aq, smem, ring, lane, hook);
            if (wave_active) {
                u32x4 m; m[0] = mw[0]; m[1] = mw[1]; m[2] = mw[2]; m[3] = mw[3];
                *(u32x4*)(a.mask_h + (((long long)(a.D - 1) * a.n_wtiles + wt) * 64 + lane) * 4) = m;
            }
        } else {
            gemm_part<NT, HN, NT / 2, AL>(acc, h, aq, smem, ring, lane);
        }
        acc_to_b<NT, false>(acc, h);
        MN_STAMP(3);   // density head + feature layer
        // ---- view-direction layer ----
        if constexpr (MODE == 0) {
            acc_init<NT / 2>(acc, scratch, hh);
            if constexpr (STASH) {
                float* row = a.stash_f + out_idx * W + 4 * hh;
                auto hook = [&](int kq, int t) __attribute__((always_inline)) {
                    if (t == NT / 2 - 1) store_chunk(h, kq, row);
                };
                gemm_part<NT / 2, HN, NT, AL>(acc, h, aq, smem, ring, lane, hook);
            } else {
                gemm_part<NT / 2, HN, NT, AL>(acc, h, aq, smem, ring, lane);
            }
        } else {
            acc_init<NT / 2>(acc, side + a.o_bias_d, hh);
            if constexpr (STASH) {
                float* row = a.stash_f + out_idx * W + 4 * hh;
                auto hook = [&](int kq, int t) __attribute__((always_inline)) {
                    if (t == NT / 2 - 1) store_chunk(h, kq, row);
                };
                gemm_part<NT / 2, HN, NT / 2, AL>(acc, h, aq, smem, ring, lane, hook);
            } else {
                gemm_part<NT / 2, HN, NT / 2, AL>(acc, h, aq, smem, ring, lane);
            }
            gemm_part<NT / 2, KDE, NT, AL>(acc, de, aq, smem, ring, lane);
        }
        float h2[HN / 2];
        acc_to_b<NT / 2, true>(acc, h2);
        if constexpr (STASH) {      // nothing consumes h2 as a B operand: 4*NT/2 row stores and its mask in one go
            store_rows<NT / 2>(h2, a.stash_g + out_idx * (W / 2) + 4 * hh, valid);
            unsigned mg[2] = {0u, 0u};
#pragma unroll
            for (int q = 0; q < HN / 8; ++q) mask_pack_chunk(h2, q, mg);
            if (wave_active) {
                u32x2 m; m[0] = mg[0]; m[1] = mg[1];
                *(u32x2*)(a.mask_g + (wt * 64 + lane) * 2) = m;
            }
        }
        MN_STAMP(4);   // view-direction layer
        // ---- colour head ----
        const float* cw = side + a.o_color_w;
        const float r0 = xhalf_sum(dot_half<HN / 2>(h2, cw, hh)) + side[a.o_color_b + 0];
        const float r1 = xhalf_sum(dot_half<HN / 2>(h2, cw + W / 2, hh)) + side[a.o_color_b + 1];
        const float r2 = xhalf_sum(dot_half<HN / 2>(h2, cw + W, hh)) + side[a.o_color_b + 2];
        if (valid && hh == 0) {
            f32x4 o; o[0] = r0; o[1] = r1; o[2] = r2; o[3] = dens;     // cat([rgb, density]) NeRF.py:51
            *(f32x4*)(a.out + out_idx * 4) = o;
        }
        MN_STAMP(5);   // colour head + store
    }
#ifdef MN_DIAG
    if (a.diag && lane == 0) {
        unsigned long long* d = a.diag + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = seg[i];
    }
#endif
    // the ring runs 3 slots ahead: let the last prefetches land before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int check_net(const mi_nerf_net* net) {
    MN_CHECK_ARG(net != nullptr, "net is NULL");
    MN_CHECK_ARG(net->W >= 2 && net->W <= MAX_KERNEL_WIDTH, "unsupported width W=%d (the fp32 inference kernels run 2 <= W <= %d, padded to 128 / 256 / 384 / 512)", net->W, MAX_KERNEL_WIDTH);
    MN_CHECK_ARG(net->D >= 2 && net->D <= 16, "unsupported depth D=%d", net->D);
    MN_CHECK_ARG(net->L_x >= 0 && net->L_x <= KERNEL_LX && net->L_d >= 0 && net->L_d <= KERNEL_LD,
                 "unsupported encoding L_x=%d L_d=%d (the kernels evaluate up to %d / %d frequencies)", net->L_x, net->L_d, KERNEL_LX, KERNEL_LD);
    MN_CHECK_ARG(net->skip >= -1, "bad skip=%d", net->skip);
    return MI_NERF_OK;
}

static int num_cus() { return device_cus(); }

// networks wider than 256 (padded to 512): mlp_fp32_wide.hip
int mlp_rays_fp32_wide(const mi_nerf_net*, const void*, const float*, const float*, int64_t, int, float*, hipStream_t);
int mlp_embedded_fp32_wide(const mi_nerf_net*, const void*, const float*, int64_t, float*, hipStream_t);

template <int W, int MODE, bool STASH = false>
static int launch(const MlpArgs& args_in, long long n_wtiles, hipStream_t st) {
    MlpArgs args = args_in;
    const size_t lds = RING_BYTES + (size_t)args.side_floats * 4 + 4 * (W / 2) * 4;
    MN_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded: %zu bytes", lds);
    auto kern = mlp_fp32_kernel<W, MODE, 10, 4, STASH>;
    static LdsOptIn opt_in = {};                              // per template instantiation, per device
    if (int rc = ensure_lds_opt_in(opt_in, (const void*)kern)) return rc;
    const long long n_wg = (n_wtiles + 3) / 4;
    const int grid = (int)(n_wg < (long long)num_cus() ? n_wg : (long long)num_cus());
    {   // the tile walk (see the kernel): ray-major once every wave of the grid gets at least one whole ray
        const long long NW = (long long)grid * 4;
        // ... and dealing whole rays does not cost a round more than dealing tiles (1500 rays x 6 tiles on 1024 waves: 12 steps
        // ray-major, 9 tile-major)
        const long long it_ray = (args.n_rays + NW - 1) / NW * args.tpr, it_flat = (n_wg + grid - 1) / grid;
        args.ray_major = (MODE == 0 && args.n_rays >= NW && it_ray <= it_flat) ? 1 : 0;
        if (args.ray_major) {
            args.walk_ray = 0; args.walk_chunk = 1; args.walk_carry = NW;
            args.n_iter = (args.n_rays + NW - 1) / NW * args.tpr;
        } else {
            const long long tpr = MODE == 0 ? args.tpr : 1;
            args.walk_ray = NW / tpr; args.walk_chunk = (int)(NW % tpr); args.walk_carry = 1;
            args.n_iter = (n_wg + grid - 1) / grid;
        }
    }
#ifdef MN_DIAG
    {   // diagnostic build: run once with stamps and print the per-segment averages (cycles per tile per wave)
        MlpArgs da = args;
        unsigned long long* dbuf = nullptr;
        const size_t n = (size_t)grid * 4 * 8 + 16 * 512;
        MN_HIP(hipMalloc(&dbuf, n * 8));
        MN_HIP(hipMemsetAsync(dbuf, 0, n * 8, st));
        da.diag = dbuf;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, da);
        MN_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> hbuf(n);
        MN_HIP(hipMemcpy(hbuf.data(), dbuf, n * 8, hipMemcpyDeviceToHost));
        (void)hipFree(dbuf);
        const double tiles = (double)((n_wg + grid - 1) / grid);
        static const char* names[6] = {"prologue", "layer0", "trunk", "dens+feature", "viewdir", "colour+store"};
        double tot = 0;
        fprintf(stderr, "[mn_diag] W=%d MODE=%d grid=%d tiles/wg=%.1f  cycles per tile (mean over waves):\n", W, MODE, grid, tiles);
        for (int sgi = 0; sgi < 6; ++sgi) {
            double sum = 0;
            for (size_t w = 0; w < (size_t)grid * 4; ++w) sum += (double)hbuf[w * 8 + sgi];
            const double per = sum / ((double)grid * 4) / tiles;
            tot += per;
            fprintf(stderr, "[mn_diag]   %-14s %10.0f\n", names[sgi], per);
        }
        fprintf(stderr, "[mn_diag]   %-14s %10.0f\n", "total", tot);
        if (const char* e = getenv("MN_DIAG_KQ")) {      // per-k-quad stamp deltas of tile #2, block 0..3 wave e
            const int wsel = atoi(e);
            const unsigned long long* lg = hbuf.data() + (size_t)grid * 32 + (size_t)wsel * 512;
            fprintf(stderr, "[mn_diag] k-quad deltas (cycles; ideal %d) of wave %d, tile 2:", 4 * 64 * (W / 32), wsel);
            for (int i = 1; i < 500 && lg[i]; ++i) fprintf(stderr, "%s%llu", (i % 16 == 1) ? "\n[mn_diag]   " : " ", lg[i] - lg[i - 1]);
            fprintf(stderr, "\n");
        }
        return MI_NERF_OK;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, args);
    MN_LAUNCH_CHECK("mlp_fp32_kernel");
    return MI_NERF_OK;
}

static void fill_common(MlpArgs& a, const mi_nerf_net* net, const void* packed_dev, bool full_stream) {
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

int mlp_rays_fp32(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev,
                  int64_t n_rays, int S, float* raw_dev, hipStream_t st) {
    if (int rc = check_net(net)) return rc;
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    if (n_rays == 0) return MI_NERF_OK;
    MN_CHECK_ARG(packed_dev && rays_dev && z_dev && raw_dev, "NULL device pointer");
    if (wide_kernel_width(kernel_width(net->W))) return mlp_rays_fp32_wide(net, packed_dev, rays_dev, z_dev, n_rays, S, raw_dev, st);
    MlpArgs a{};
    fill_common(a, net, packed_dev, false);
    a.rays = rays_dev; a.z = z_dev; a.out = raw_dev; a.S = S; a.tpr = (S + 31) / 32;
    a.n_wtiles = (long long)n_rays * a.tpr; a.n_rays = n_rays;
    return kernel_width(net->W) == 256 ? launch<256, 0>(a, a.n_wtiles, st) : launch<128, 0>(a, a.n_wtiles, st);
}

// training forward: same kernel, additionally keeping every layer's activations row-major for the backward pass
int mlp_rays_fp32_stash(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev,
                        int64_t n_rays, int S, float* raw_dev, float* stash_h, float* stash_f, float* stash_g, unsigned* mask_h,
                        unsigned* mask_g, hipStream_t st) {
    if (int rc = check_net(net)) return rc;
    MN_CHECK_ARG(native_width(net->W), "the training kernels exist for W = 128 and 256 (got %d)", net->W);
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    if (n_rays == 0) return MI_NERF_OK;
    MN_CHECK_ARG(packed_dev && rays_dev && z_dev && raw_dev && stash_h && stash_f && stash_g && mask_h && mask_g, "NULL device pointer");
    MlpArgs a{};
    fill_common(a, net, packed_dev, false);
    a.rays = rays_dev; a.z = z_dev; a.out = raw_dev; a.S = S; a.tpr = (S + 31) / 32;
    a.n_wtiles = (long long)n_rays * a.tpr; a.n_rays = n_rays;
    a.stash_h = stash_h; a.stash_f = stash_f; a.stash_g = stash_g; a.stash_rows = (long long)n_rays * S;
    a.mask_h = mask_h; a.mask_g = mask_g;
    return net->W == 256 ? launch<256, 0, true>(a, a.n_wtiles, st) : launch<128, 0, true>(a, a.n_wtiles, st);
}

// training forward over pre-embedded rows (model/NeRF.py:33-52 called directly with gradients enabled): flat 32-row tiles;
// the stash and masks are laid out for ceil(n / 32) "rays" of 32 samples (train_layout(net, ceil(n/32), 32))
int mlp_embedded_fp32_stash(const mi_nerf_net* net, const void* packed_dev, const float* x_dev, int64_t n, float* out_dev, float* stash_h,
                            float* stash_f, float* stash_g, unsigned* mask_h, unsigned* mask_g, hipStream_t st) {
    if (int rc = check_net(net)) return rc;
    MN_CHECK_ARG(native_width(net->W), "the training kernels exist for W = 128 and 256 (got %d)", net->W);
    MN_CHECK_ARG(n >= 0, "bad n=%lld", (long long)n);
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(packed_dev && x_dev && out_dev && stash_h && stash_f && stash_g && mask_h && mask_g, "NULL device pointer");
    MlpArgs a{};
    fill_common(a, net, packed_dev, true);
    a.x = x_dev; a.out = out_dev; a.n_pts = n;
    a.n_wtiles = (n + 31) / 32;
    a.stash_h = stash_h; a.stash_f = stash_f; a.stash_g = stash_g; a.stash_rows = a.n_wtiles * 32;
    a.mask_h = mask_h; a.mask_g = mask_g;
    return net->W == 256 ? launch<256, 1, true>(a, a.n_wtiles, st) : launch<128, 1, true>(a, a.n_wtiles, st);
}

int mlp_embedded_fp32(const mi_nerf_net* net, const void* packed_dev, const float* x_dev, int64_t n, float* out_dev,
                      hipStream_t st) {
    if (int rc = check_net(net)) return rc;
    MN_CHECK_ARG(n >= 0, "bad n=%lld", (long long)n);
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(packed_dev && x_dev && out_dev, "NULL device pointer");
    if (wide_kernel_width(kernel_width(net->W))) return mlp_embedded_fp32_wide(net, packed_dev, x_dev, n, out_dev, st);
    MlpArgs a{};
    fill_common(a, net, packed_dev, true);
    a.x = x_dev; a.out = out_dev; a.n_pts = n;
    a.n_wtiles = (n + 31) / 32;
    return kernel_width(net->W) == 256 ? launch<256, 1>(a, a.n_wtiles, st) : launch<128, 1>(a, a.n_wtiles, st);
}

}  // namespace minerf
