// stage_dev.h -- the per-ray device functions of the sampling / compositing stages (one wavefront per ray), shared by the stage kernels
// (stages.hip) and by the bf16 network kernel's small-launch epilogue (mlp_bf16.hip: a workgroup that owns whole rays composites them and
// draws their fine depths itself instead of leaving that to a launch of its own).  One definition, so "fused" and "staged" are the same
// instructions on the same operands: bit-identical results by construction (tests/test_gpu_parity.py).
#pragma once
#include "common.h"

namespace minerf {

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// exclusive prefix product / sum across the 64 lanes (Kogge-Stone on __shfl_up)
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(inc, d, 64);
        if (lane >= d) inc *= o;
    }
    const float e = __shfl_up(inc, 1, 64);
    return lane == 0 ? 1.0f : e;
}
__device__ __forceinline__ float wave_excl_sum(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    const float e = __shfl_up(inc, 1, 64);
    return lane == 0 ? 0.0f : e;
}

// ------------------------------------------------------------------------------------------------
// alpha compositing: one wavefront per ray, lane l owns samples [l*C, (l+1)*C)
// ------------------------------------------------------------------------------------------------
// One ray (this wave).  w_lds (optional): the weights also go to this LDS array.
template <int C>
__device__ __forceinline__ void composite_ray(const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays,
                                              int ray_stride, long long ray, int S, int lane, float* __restrict__ rgb_o,
                                              float* __restrict__ disp_o, float* __restrict__ acc_o, float* __restrict__ w_o,
                                              float* __restrict__ depth_o, float* w_lds) {
    const float* dp = rays + ray * ray_stride + (ray_stride == 6 ? 3 : 0);
    const float dx = dp[0], dy = dp[1], dz = dp[2];
    const float dnorm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);     // nerf_process.py:101
    const float* zr = z + ray * S;
    const f32x4* rr = (const f32x4*)(raw + ray * S * 4);

    float alpha[C], zv[C], cr[C], cg[C], cb[C];
    float local = 1.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane * C + c;
        const bool in = s < S;
        const int sc = in ? s : S - 1;
        const f32x4 v = rr[sc];
        zv[c] = zr[sc];
        float dist = (s + 1 < S) ? (zr[s + 1] - zv[c]) : 1e10f;           // :93-97
        dist = dist * dnorm;                                               // :101
        const float sig = __builtin_fmaxf(v[3], 0.0f);                     // relu, :91
        float a = 1.0f - expf(-sig * dist);                                // :92
        // S == 1: the reference's `dists[..., :1]` slice of an empty tensor leaves NO samples at all (white, acc 0)
        if (!in || S == 1) a = 0.0f;
        alpha[c] = a;
        cr[c] = 1.0f / (1.0f + expf(-v[0]));                               // sigmoid, :104
        cg[c] = 1.0f / (1.0f + expf(-v[1]));
        cb[c] = 1.0f / (1.0f + expf(-v[2]));
        local *= in ? (1.0f - a + 1e-10f) : 1.0f;                          // :110
    }
    float T = wave_excl_prod(local, lane);                                 // transmittance entering this lane's chunk
    float sw = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane * C + c;
        const float w = alpha[c] * T;                                      // :111
        if (s < S) {
            if (w_o) w_o[ray * S + s] = w;
            if (w_lds) w_lds[s] = w;
            sw += w; sr += w * cr[c]; sg += w * cg[c]; sb += w * cb[c]; sd += w * zv[c];
        }
        T *= (1.0f - alpha[c] + 1e-10f);
    }
    sw = wave_sum(sw); sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd);
    if (lane == 0) {
        const float q = sd / sw;                                           // depth / acc
        const float m = (q != q) ? q : __builtin_fmaxf(1e-10f, q);         // torch.max propagates NaN (:124)
        float disp = 1.0f / m;
        if (disp != disp) disp = 0.0f;                                     // :126
        if (disp > 5.0f) disp = 5.0f;                                      // :132-134
        const float bg = 1.0f - sw;                                        // :138 white background, always
        rgb_o[ray * 3 + 0] = sr + bg; rgb_o[ray * 3 + 1] = sg + bg; rgb_o[ray * 3 + 2] = sb + bg;
        disp_o[ray] = disp;
        if (acc_o) acc_o[ray] = sw;
        if (depth_o) depth_o[ray] = sd;
    }
}

// ------------------------------------------------------------------------------------------------
// inverse-CDF sampling, one wavefront per ray.  cdf/bins live in this wave's LDS slice.
// ------------------------------------------------------------------------------------------------
// Build cdf[0..B) from weights w[0..B-1) (nerf_process.py:150-154).  Lane l owns entries [l*C, (l+1)*C).
__device__ __forceinline__ void build_cdf(const float* __restrict__ w, int nw, float* cdf, int lane) {
    const int C = (nw + 63) / 64;
    float part = 0.f;
    for (int c = 0; c < C; ++c) {
        const int k = lane * C + c;
        if (k < nw) part += w[k] + 1e-5f;                                  // :150
    }
    const float total = wave_sum(part);
    float run = 0.f, lsum = 0.f;
    for (int c = 0; c < C; ++c) {
        const int k = lane * C + c;
        if (k < nw) lsum += (w[k] + 1e-5f) / total;                        // pdf, :151
    }
    run = wave_excl_sum(lsum, lane);
    if (lane == 0) cdf[0] = 0.0f;                                          // :154
    for (int c = 0; c < C; ++c) {
        const int k = lane * C + c;
        if (k < nw) { run += (w[k] + 1e-5f) / total; cdf[k + 1] = run; }   // cumsum, :152
    }
}

__device__ __forceinline__ float invert_cdf(const float* cdf, const float* bins, int B, float u) {
    // searchsorted(cdf, u, right=True): number of entries <= u   (:167)
    int lo = 0, len = B;
    while (len > 0) {
        const int half = len >> 1;
        const bool go = cdf[lo + half] <= u;
        lo = go ? lo + half + 1 : lo;
        len = go ? len - half - 1 : half;
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;                             // :168
    const int above = lo < B - 1 ? lo : B - 1;                             // :169
    const float c0 = cdf[below], c1 = cdf[above];
    float denom = c1 - c0;                                                 // :178
    if (denom < 1e-5f) denom = 1.0f;                                       // :179
    const float t = (u - c0) / denom;                                      // :180
    const float b0 = bins[below], b1 = bins[above];
    return b0 + t * (b1 - b0);                                             // :181
}

__device__ __forceinline__ float det_u(int j, int N) {                     // torch.linspace(0,1,N)[j], :158
    if (N == 1) return 0.0f;
    const float step = 1.0f / (float)(N - 1);
    return (j < N / 2) ? step * (float)j : 1.0f - step * (float)(N - 1 - j);
}

// Bitonic sorting network over n2 = 64 R values held R per lane (v[r] = element R * lane + r; no NaNs), ascending.  A compare-exchange
// whose partner lies among the lane's own R elements is a register operation; the others fetch the partner lane's register through
// ds_bpermute (__shfl_xor) -- no LDS array, no fence per stage.  64 + 128 depths (R = 4): 15 in-lane and 21 cross-lane stages.
template <int R>
__device__ __forceinline__ void bitonic_sort_regs(float (&v)[R], int lane) {
    constexpr int n2 = 64 * R;
#pragma unroll
    for (int k = 2; k <= n2; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= R) {
                const int m = j / R;                                       // partner: the same register of lane ^ m
                const bool lower = (lane & m) == 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool up = (((R * lane + r) & k) == 0);
                    const float x = v[r], y = __shfl_xor(x, m, 64);
                    const bool lt = x < y;
                    const float lo = lt ? x : y, hi = lt ? y : x;
                    v[r] = (lower == up) ? lo : hi;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if ((r & j) == 0) {
                        const bool up = (((R * lane + r) & k) == 0);
                        const float x = v[r], y = v[r | j];
                        const bool lt = x < y;
                        const float lo = lt ? x : y, hi = lt ? y : x;
                        v[r] = up ? lo : hi;
                        v[r | j] = up ? hi : lo;
                    }
            }
        }
    }
}
// all[0 .. St) (this wave's LDS slice, complete and visible to the wave) -> z_f row, sorted, NaNs last
template <int R>
__device__ __forceinline__ void sort_row_regs(const float* all, int St, float* __restrict__ zrow, int lane) {
    float v[R];
    int nan_here = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = R * lane + r;
        const float x = e < St ? all[e] : __builtin_inff();
        const bool isn = x != x;
        nan_here += isn ? 1 : 0;
        v[r] = isn ? __builtin_inff() : x;
    }
    int n_nan = nan_here;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_nan += __shfl_xor(n_nan, o, 64);
    bitonic_sort_regs<R>(v, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = R * lane + r;
        if (e < St) zrow[e] = e < St - n_nan ? v[r] : __builtin_nanf("");
    }
}

// fine branch: bins = mid(z_c), weights = weights_c[1:-1], then sort(cat(z_c, samples))   (:63-67)
// wr: this ray's Sc coarse weights (global or LDS); lds: this wave's slice of 2 (Sc - 1) + n2 floats.
__device__ __forceinline__ void fine_z_ray(const float* __restrict__ z_c, const float* wr, long long ray, int Sc, int Nf, int n2, int det,
                                           const Jitter& u, float* __restrict__ z_f, float* __restrict__ z_samp, float* lds,
                                           int lane) {
    const int B = Sc - 1, St = Sc + Nf;
    float* cdf = lds;                                       // n2: Sc + Nf rounded up to a power of two (the sort network)
    float* bn = cdf + B;
    float* all = bn + B;
    const float* zr = z_c + ray * Sc;
    for (int k = lane; k < Sc; k += 64) all[k] = zr[k];
    for (int k = lane; k < B; k += 64) bn[k] = 0.5f * (zr[k + 1] + zr[k]);                 // :63
    build_cdf(wr + 1, Sc - 2, cdf, lane);                                                   // weights[..., 1:-1]
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int j = lane; j < Nf; j += 64) {
        const float uu = det ? det_u(j, Nf) : jitter_at(u, ray, j, Nf);
        const float s = invert_cdf(cdf, bn, B, uu);
        all[Sc + j] = s;
        if (z_samp) z_samp[ray * Nf + j] = s;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // sort(cat(z_c, samples)) (nerf_process.py:67): only the sorted VALUES are returned, so any correct sort gives the reference's
    // tensor.  Bitonic network over the wave's LDS slice, padded to a power of two with +inf: log2(n2)(log2(n2)+1)/2 stages of
    // n2/2 compare-exchanges (36 stages of 2 per lane for 64 + 128 samples; the rank sort this replaces did St compares for each
    // of St/64 elements per lane -- 48 k cycles per ray, 23 us per launch however few the rays).  NaN depths (a diverged
    // network: NaN weights -> NaN samples) are sorted as +inf and written back as NaN in the last slots, where torch.sort
    // places them; every slot of z_f (torch.empty) is written.  Up to 512 depths the network runs in registers (bitonic_sort_regs).
    if (n2 == 256) return sort_row_regs<4>(all, St, z_f + ray * St, lane);
    if (n2 == 128) return sort_row_regs<2>(all, St, z_f + ray * St, lane);
    if (n2 == 64) return sort_row_regs<1>(all, St, z_f + ray * St, lane);
    if (n2 == 512) return sort_row_regs<8>(all, St, z_f + ray * St, lane);
    int nan_here = 0;
    for (int e = lane; e < n2; e += 64) {
        const float v = e < St ? all[e] : __builtin_inff();
        const bool isn = v != v;
        nan_here += isn ? 1 : 0;
        if (isn || e >= St) all[e] = __builtin_inff();
    }
    int n_nan = nan_here;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_nan += __shfl_xor(n_nan, o, 64);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // t with a 0 inserted at bit log2(j)
                const int l = i | j;
                const float x = all[i], y = all[l];
                const bool lt = x < y;
                const float lo = lt ? x : y, hi = lt ? y : x;
                const bool up = (i & k) == 0;
                all[i] = up ? lo : hi;
                all[l] = up ? hi : lo;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    for (int e = lane; e < St; e += 64) z_f[ray * St + e] = e < St - n_nan ? all[e] : __builtin_nanf("");
}

}  // namespace minerf
