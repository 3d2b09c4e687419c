// stages.hip -- the non-GEMM stages of the render path as HBM-bound / wavefront kernels (gfx950).
//
//   make_o_d            rays.py:20-34               one thread per pixel, coalesced AoS stores
//   ndc_rays            nerf_process.py:8-28        one thread per ray
//   fill_uniform        stand-in for torch.rand at nerf_process.py:58,162 (counter-based, shard invariant)
//   stratified_z        nerf_process.py:42-60       one thread per sample
//   embed               nerf_process.py:36-39,69-85 + model/PositionalEncoding.py:29-30, one work item per (point, band) into an
//                       LDS tile of 64 points, contiguous 16-byte stores (the tensor is write-only traffic: 360 B/point)
//   composite           nerf_process.py:89-140      one 64-lane wavefront per ray: per-lane chunk product,
//                       Kogge-Stone exclusive prefix product across lanes, butterfly sums
//   sample_pdf / fine_z nerf_process.py:144-182, :62-67  one wavefront per ray: prefix-sum CDF in LDS,
//                       branch-free upper_bound, bitonic sort network over the merged depths
//
// Arithmetic follows the reference's operation order in fp32 with IEEE division and no FMA contraction
// (the file is compiled with -ffp-contract=off); FMAs appear only where written explicitly.
#include "common.h"
#include "stage_dev.h"

namespace minerf {

// ------------------------------------------------------------------------------------------------
// ray generation
// ------------------------------------------------------------------------------------------------
struct CamArgs {
    float fx, fy, cx, cy;
    float r[9];     // rotation, row-major
    float t[3];     // translation
};

__device__ __forceinline__ void pixel_ray(const CamArgs& c, int x, int y, float (&d)[3]) {
    const float dx = ((float)x - c.cx) / c.fx;          // rays.py:28
    const float dy = -((float)y - c.cy) / c.fy;         // rays.py:29
    const float dz = -1.0f;                              // rays.py:30
#pragma unroll
    for (int i = 0; i < 3; ++i)                          // dirs @ R^T  (rays.py:32)
        d[i] = __builtin_fmaf(dz, c.r[3 * i + 2], __builtin_fmaf(dy, c.r[3 * i + 1], dx * c.r[3 * i + 0]));
}

__global__ __launch_bounds__(256) void make_o_d_kernel(CamArgs c, int W, int row0, long long n, float* __restrict__ o,
                                                        float* __restrict__ dout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int y = (int)(i / W) + row0, x = (int)(i % W);
    float d[3];
    pixel_ray(c, x, y, d);
    dout[3 * i + 0] = d[0]; dout[3 * i + 1] = d[1]; dout[3 * i + 2] = d[2];
    if (o) { o[3 * i + 0] = c.t[0]; o[3 * i + 1] = c.t[1]; o[3 * i + 2] = c.t[2]; }
}

__global__ __launch_bounds__(256) void make_o_d_pixels_kernel(CamArgs c, int W, const long long* __restrict__ pix,
                                                               long long n, float* __restrict__ o, float* __restrict__ dout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long p = pix[i];
    float d[3];
    pixel_ray(c, (int)(p % W), (int)(p / W), d);
    dout[3 * i + 0] = d[0]; dout[3 * i + 1] = d[1]; dout[3 * i + 2] = d[2];
    if (o) { o[3 * i + 0] = c.t[0]; o[3 * i + 1] = c.t[1]; o[3 * i + 2] = c.t[2]; }
}

static CamArgs cam_args(const float k4[4], const float pose12[12]) {
    CamArgs c;
    c.fx = k4[0]; c.fy = k4[1]; c.cx = k4[2]; c.cy = k4[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.r[3 * i + j] = pose12[4 * i + j];
        c.t[i] = pose12[4 * i + 3];
    }
    return c;
}

// ------------------------------------------------------------------------------------------------
// NDC warp
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ndc_kernel(float sx, float sy, float near_, float two_near, const float* __restrict__ oin,
                                                   long long os, const float* __restrict__ din, long long ds, long long n,
                                                   float* __restrict__ oo, float* __restrict__ dd) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float ox = oin[i * os + 0], oy = oin[i * os + 1], oz = oin[i * os + 2];
    const float dx = din[i * ds + 0], dy = din[i * ds + 1], dz = din[i * ds + 2];
    const float t = -(near_ + oz) / dz;                  // nerf_process.py:11
    ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz; // :12
    const float ox_oz = ox / oz, oy_oz = oy / oz;
    oo[3 * i + 0] = sx * ox / oz;                        // :15  (scale*o_x)/o_z
    oo[3 * i + 1] = sy * oy / oz;                        // :16
    oo[3 * i + 2] = 1.0f + two_near / oz;                // :17
    dd[3 * i + 0] = sx * (dx / dz - ox_oz);              // :19-20
    dd[3 * i + 1] = sy * (dy / dz - oy_oz);              // :21-22
    dd[3 * i + 2] = -two_near / oz;                      // :23
}

// ------------------------------------------------------------------------------------------------
// uniforms and stratified depths
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_uniform_kernel(uint32_t seed, uint32_t stream_id, long long ray0, long long total,
                                                            int S, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long r = i / S;
    out[i] = counter_uniform(seed, stream_id, (uint32_t)(ray0 + r), (uint32_t)(i - r * S));
}

__global__ __launch_bounds__(256) void stratified_kernel(long long total, int S, float near_, float far_, float step,
                                                          Jitter t_rand, float* __restrict__ z) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long ray = idx / S;
    const int i = (int)(idx - ray * S);
    z[idx] = stratified_depth(ray, i, S, step, near_, far_, t_rand);
}

// ------------------------------------------------------------------------------------------------
// network-input assembly (unfused path, for pre_process parity)
// ------------------------------------------------------------------------------------------------
// A block encodes EMB_PTS points into an LDS tile and writes the [points][channels] rows out as one contiguous run of 16-byte
// words.  Work item = (point, band): the three components' sin AND cos of one frequency share the argument reduction and the
// polynomial pair; the ray lookup and the direction norm are done once per item, not once per output float.  (The first
// version ran one thread per output element -- a 64-bit division, a reduction and half a polynomial pair each: 1.1 TB/s of
// output on a kernel whose only traffic is its output; it cost the training step 0.33 ms.)
constexpr int EMB_PTS = 64;
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ rays, const float* __restrict__ z, long long n_pts,
                                                     int S, int L_x, int L_d, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float emb_tile[];
    const int in_x = 3 + 6 * L_x, ch_total = in_x + 3 + 6 * L_d;
    const int items = L_x + L_d + 2;                                     // per point: x, its L_x bands, d/|d|, its L_d bands
    const long long p0 = (long long)blockIdx.x * EMB_PTS;
    const int npt = (int)((n_pts - p0 < EMB_PTS) ? n_pts - p0 : EMB_PTS);
    for (int w = threadIdx.x; w < npt * items; w += 256) {
        const int lp = w / items, it = w - lp * items;
        const long long pt = p0 + lp;
        const float* rp = rays + (pt / S) * 6;
        const bool is_d = it > L_x;
        float b[3];
        if (!is_d) {
            const float zz = z[pt];
#pragma unroll
            for (int c = 0; c < 3; ++c) b[c] = rp[c] + rp[3 + c] * zz;       // nerf_process.py:69-70
        } else {
            const float dx = rp[3], dy = rp[4], dz = rp[5];
            const float nrm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);  // :39
#pragma unroll
            for (int c = 0; c < 3; ++c) b[c] = rp[3 + c] / nrm;
        }
        float* row = emb_tile + lp * ch_total + (is_d ? in_x : 0);
        const int k = (is_d ? it - L_x - 1 : it) - 1;                        // -1: the identity channels
        if (k < 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) row[c] = b[c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {                                   // PositionalEncoding.py:20-24
                const float y = b[c] * (float)(1 << k);
                const bool fast = __builtin_fabsf(y) < SINCOS_FAST_LIMIT;
                row[3 + 6 * k + c] = fast ? sin_cos_fast(y, 0) : sin_cos_slow(y, 0);
                row[3 + 6 * k + 3 + c] = fast ? sin_cos_fast(y, 1) : sin_cos_slow(y, 1);
            }
        }
    }
    __syncthreads();
    float* o = out + p0 * ch_total;
    const int n = npt * ch_total;
    if (((uintptr_t)o & 15) == 0) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) ((f32x4*)o)[i] = ((const f32x4*)emb_tile)[i];
        for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) o[i] = emb_tile[i];
    } else {
        for (int i = threadIdx.x; i < n; i += 256) o[i] = emb_tile[i];
    }
}

// gamma(x) for arbitrary 3-vectors: the closure returned by get_positional_encoder (PositionalEncoding.py:33-36)
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ x, long long n, int L, float* __restrict__ out) {
    const int ch_total = 3 + 6 * L;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * ch_total) return;
    const long long pt = idx / ch_total;
    const int ch = (int)(idx - pt * ch_total);
    const int c = (ch < 3) ? ch : (ch - 3) % 3;
    const float base = x[pt * 3 + c];
    float v = base;
    if (ch >= 3) {
        const int k = (ch - 3) / 6, is_cos = ((ch - 3) % 6) >= 3;
        const float y = base * (float)(1 << k);
        v = (__builtin_fabsf(y) < SINCOS_FAST_LIMIT) ? sin_cos_fast(y, is_cos) : sin_cos_slow(y, is_cos);
    }
    out[idx] = v;
}

template <int C>
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                         const float* __restrict__ rays, int ray_stride, long long n, int S,
                                                         float* __restrict__ rgb_o, float* __restrict__ disp_o,
                                                         float* __restrict__ acc_o, float* __restrict__ w_o,
                                                         float* __restrict__ depth_o) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n) return;
    composite_ray<C>(raw, z, rays, ray_stride, ray, S, lane, rgb_o, disp_o, acc_o, w_o, depth_o, nullptr);
}

// exclusive SUFFIX sum across the 64 lanes (sum of the lanes above this one), Kogge-Stone on __shfl_down
__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_down(inc, d, 64);
        if (lane + d < 64) inc += o;
    }
    const float e = __shfl_down(inc, 1, 64);
    return lane == 63 ? 0.0f : e;
}

// ------------------------------------------------------------------------------------------------
// alpha compositing, backward: d rgb_map [n,3] -> d raw [n,S,4]   (what autograd does for
// nerf_process.py:89-140 when the loss reads rgb_map only, train.py:59-66; depths are constants:
// the coarse ones carry no graph and the fine ones are detached at nerf_process.py:66).
//   rgb_map = sum_i w_i c_i + 1 - sum_i w_i,  w_i = a_i T_i,  T_i = prod_{k<i} u_k,  u_k = 1 - a_k + 1e-10
//   q_i := dL/dw_i = sum_ch G_ch (c_i,ch - 1)
//   dL/da_i = q_i T_i - (sum_{k>i} q_k w_k) / u_i
//   da_i/dsigma_i = dist_i exp(-relu(sigma_i) dist_i) for sigma_i > 0, else 0;  dc/draw = c (1 - c)
// Same wave-per-ray decomposition as the forward kernel; the forward quantities are recomputed.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                             const float* __restrict__ rays, int ray_stride, long long n, int S,
                                                             const float* __restrict__ d_rgb, float* __restrict__ d_raw) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n) return;
    const float* dp = rays + ray * ray_stride + (ray_stride == 6 ? 3 : 0);
    const float dx = dp[0], dy = dp[1], dz = dp[2];
    const float dnorm = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    const float* zr = z + ray * S;
    const f32x4* rr = (const f32x4*)(raw + ray * S * 4);
    f32x4* out = (f32x4*)(d_raw + ray * S * 4);
    const float Gr = d_rgb[ray * 3 + 0], Gg = d_rgb[ray * 3 + 1], Gb = d_rgb[ray * 3 + 2];

    float alpha[C], dads[C], cr[C], cg[C], cb[C];
    float local = 1.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane * C + c;
        const bool in = s < S;
        const int sc = in ? s : S - 1;
        const f32x4 v = rr[sc];
        float dist = (s + 1 < S) ? (zr[s + 1] - zr[sc]) : 1e10f;
        dist = dist * dnorm;
        const float sig = __builtin_fmaxf(v[3], 0.0f);
        const float e = expf(-sig * dist);
        float a = 1.0f - e;
        float ds = (v[3] > 0.0f) ? dist * e : 0.0f;
        if (!in || S == 1) { a = 0.0f; ds = 0.0f; }
        alpha[c] = a;
        dads[c] = ds;
        cr[c] = 1.0f / (1.0f + expf(-v[0]));
        cg[c] = 1.0f / (1.0f + expf(-v[1]));
        cb[c] = 1.0f / (1.0f + expf(-v[2]));
        local *= in ? (1.0f - a + 1e-10f) : 1.0f;
    }
    float T = wave_excl_prod(local, lane);
    float Tc[C], q[C], w[C];
    float lsum = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        Tc[c] = T;
        w[c] = alpha[c] * T;
        q[c] = Gr * (cr[c] - 1.0f) + Gg * (cg[c] - 1.0f) + Gb * (cb[c] - 1.0f);
        lsum += q[c] * w[c];
        T *= (1.0f - alpha[c] + 1e-10f);
    }
    float R = wave_excl_suffix_sum(lsum, lane);     // sum over the samples owned by higher lanes
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {
        const int s = lane * C + c;
        const float u = 1.0f - alpha[c] + 1e-10f;
        const float dLda = q[c] * Tc[c] - R / u;
        R += q[c] * w[c];
        if (s < S) {
            f32x4 o;
            o[0] = Gr * w[c] * cr[c] * (1.0f - cr[c]);
            o[1] = Gg * w[c] * cg[c] * (1.0f - cg[c]);
            o[2] = Gb * w[c] * cb[c] * (1.0f - cb[c]);
            o[3] = dLda * dads[c];
            out[s] = o;
        }
    }
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ weights,
                                                          long long n, int B, int N, int det, const float* __restrict__ u,
                                                          float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * 4 + wv;
    if (ray >= n) return;
    float* cdf = lds + wv * 2 * B;
    float* bn = cdf + B;
    for (int k = lane; k < B; k += 64) bn[k] = bins[ray * B + k];
    build_cdf(weights + ray * (B - 1), B - 1, cdf, lane);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int j = lane; j < N; j += 64) {
        const float uu = det ? det_u(j, N) : u[ray * N + j];
        out[ray * N + j] = invert_cdf(cdf, bn, B, uu);
    }
}

__global__ __launch_bounds__(256) void fine_z_kernel(const float* __restrict__ z_c, const float* __restrict__ w_c, long long n,
                                                      int Sc, int Nf, int n2, int det, Jitter u,
                                                      float* __restrict__ z_f, float* __restrict__ z_samp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * 4 + wv;
    if (ray >= n) return;
    fine_z_ray(z_c, w_c + ray * Sc, ray, Sc, Nf, n2, det, u, z_f, z_samp, lds + wv * (2 * (Sc - 1) + n2), lane);
}

// render_rays' middle (nerf_process.py:198-203) in ONE launch: composite the coarse pass, resample from its weights, merge-sort.  Both
// halves are one wave per ray; the weights go from the compositing registers to the sampler through the wave's LDS slice (and to the
// workspace, which the staged parity checks read).  At a 512-ray shard a launch is ~4 us of dispatch for ~1 us of work: the fused
// kernel is one launch where there were two (three with the uniforms' own).
template <int C>
__global__ __launch_bounds__(256) void composite_fine_z_kernel(const float* __restrict__ raw, const float* __restrict__ z_c,
                                                                const float* __restrict__ rays, long long n, int Sc, int Nf, int n2, int det,
                                                                Jitter u, float* __restrict__ rgb_o, float* __restrict__ disp_o,
                                                                float* __restrict__ w_o, float* __restrict__ z_f) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * 4 + wv;
    if (ray >= n) return;
    float* mine = lds + wv * (Sc + 2 * (Sc - 1) + n2);
    composite_ray<C>(raw, z_c, rays, 6, ray, Sc, lane, rgb_o, disp_o, nullptr, w_o, nullptr, mine);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    fine_z_ray(z_c, mine, ray, Sc, Nf, n2, det, u, z_f, nullptr, mine + Sc, lane);
}

// ------------------------------------------------------------------------------------------------
// host entry points (called from api.cpp)
// ------------------------------------------------------------------------------------------------
constexpr int MAX_LDS_FLOATS_PER_RAY = 64 * 1024 / 4 / 4;      // 4 rays per block share 64 KB: 4096 floats per ray
static inline unsigned blocks_for(long long n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

int stage_make_o_d(int W, int H, const float k4[4], const float pose12[12], int row0, int n_rows, float* o, float* d,
                   hipStream_t st) {
    MN_CHECK_ARG(W > 0 && H > 0 && row0 >= 0 && n_rows >= 0 && row0 + n_rows <= H, "bad image window W=%d H=%d rows [%d,+%d)", W, H, row0, n_rows);
    MN_CHECK_ARG(d != nullptr, "rays_d output is NULL");
    const long long n = (long long)n_rows * W;
    if (n == 0) return MI_NERF_OK;
    hipLaunchKernelGGL(make_o_d_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, cam_args(k4, pose12), W, row0, n, o, d);
    MN_LAUNCH_CHECK("make_o_d_kernel");
    return MI_NERF_OK;
}

int stage_make_o_d_pixels(int W, int H, const float k4[4], const float pose12[12], const int64_t* pix, int64_t n, float* o,
                          float* d, hipStream_t st) {
    MN_CHECK_ARG(W > 0 && H > 0 && n >= 0 && d != nullptr && (pix != nullptr || n == 0), "bad arguments");
    if (n == 0) return MI_NERF_OK;
    hipLaunchKernelGGL(make_o_d_pixels_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, cam_args(k4, pose12), W,
                       (const long long*)pix, (long long)n, o, d);
    MN_LAUNCH_CHECK("make_o_d_pixels_kernel");
    return MI_NERF_OK;
}

int stage_ndc(int H, int W, float focal, float near_, const float* o_in, int64_t os, const float* d_in, int64_t ds, int64_t n,
              float* o_out, float* d_out, hipStream_t st) {
    MN_CHECK_ARG(H > 0 && W > 0 && n >= 0, "bad sizes");
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(o_in && d_in && o_out && d_out, "NULL pointer");
    // python-float (double) scale factors, rounded once to fp32 where they meet the tensors (nerf_process.py:15-23)
    const float sx = (float)(-1.0 / ((double)W / (2.0 * (double)focal)));
    const float sy = (float)(-1.0 / ((double)H / (2.0 * (double)focal)));
    const float two_near = (float)(2.0 * (double)near_);
    hipLaunchKernelGGL(ndc_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, sx, sy, near_, two_near, o_in, (long long)os, d_in,
                       (long long)ds, (long long)n, o_out, d_out);
    MN_LAUNCH_CHECK("ndc_kernel");
    return MI_NERF_OK;
}

int stage_fill_uniform(uint32_t seed, uint32_t stream_id, int64_t ray0, int64_t n_rays, int S, float* out, hipStream_t st) {
    MN_CHECK_ARG(n_rays >= 0 && S >= 0 && (out || n_rays * S == 0), "bad arguments");
    const long long total = (long long)n_rays * S;
    if (total == 0) return MI_NERF_OK;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, st, seed, stream_id, (long long)ray0, total, S, out);
    MN_LAUNCH_CHECK("fill_uniform_kernel");
    return MI_NERF_OK;
}

// t_rand == NULL: the uniforms are drawn in the kernel from (seed, stream 0, ray0 + ray, sample)
int stage_stratified(int64_t n_rays, int S, float near_, float far_, const float* t_rand, uint32_t seed, int64_t ray0, float* z, hipStream_t st) {
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes");
    const long long total = (long long)n_rays * S;
    if (total == 0) return MI_NERF_OK;
    MN_CHECK_ARG(z, "NULL pointer");
    const float step = S > 1 ? 1.0f / (float)(S - 1) : 0.0f;
    hipLaunchKernelGGL(stratified_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, st, total, S, near_, far_, step,
                       Jitter{t_rand, seed, 0u, (long long)ray0}, z);
    MN_LAUNCH_CHECK("stratified_kernel");
    return MI_NERF_OK;
}

int stage_embed(const float* rays, const float* z, int64_t n_rays, int S, int L_x, int L_d, float* out, hipStream_t st) {
    MN_CHECK_ARG(n_rays >= 0 && S >= 1 && L_x >= 0 && L_x <= 20 && L_d >= 0 && L_d <= 20, "bad sizes");
    const long long n_pts = (long long)n_rays * S;
    const long long total = n_pts * (6 + 6 * L_x + 6 * L_d);
    if (total == 0) return MI_NERF_OK;
    MN_CHECK_ARG(rays && z && out, "NULL pointer");
    hipLaunchKernelGGL(embed_kernel, dim3(blocks_for(n_pts, EMB_PTS)), dim3(256), (size_t)EMB_PTS * (6 + 6 * L_x + 6 * L_d) * sizeof(float), st,
                       rays, z, n_pts, S, L_x, L_d, out);
    MN_LAUNCH_CHECK("embed_kernel");
    return MI_NERF_OK;
}

int stage_posenc(const float* x, int64_t n, int L, float* out, hipStream_t st) {
    MN_CHECK_ARG(n >= 0 && L >= 0 && L <= 20, "bad sizes");
    const long long total = (long long)n * (3 + 6 * L);
    if (total == 0) return MI_NERF_OK;
    MN_CHECK_ARG(x && out, "NULL pointer");
    hipLaunchKernelGGL(posenc_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, st, x, (long long)n, L, out);
    MN_LAUNCH_CHECK("posenc_kernel");
    return MI_NERF_OK;
}

int stage_composite(const float* raw, const float* z, const float* rays, int ray_stride, int64_t n, int S, float* rgb, float* disp,
                    float* acc, float* weights, float* depth, hipStream_t st) {
    MN_CHECK_ARG(n >= 0 && S >= 1 && S <= 1024, "bad sizes (n=%lld S=%d)", (long long)n, S);
    MN_CHECK_ARG(ray_stride == 3 || ray_stride == 6, "ray_stride must be 3 or 6");
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(raw && z && rays && rgb && disp, "NULL pointer");
    const dim3 grid(blocks_for(n, 4)), block(256);
    const int C = (S + 63) / 64;
#define MN_COMP(CC) hipLaunchKernelGGL(composite_kernel<CC>, grid, block, 0, st, raw, z, rays, ray_stride, (long long)n, S, rgb, disp, acc, weights, depth)
    if (C == 1) MN_COMP(1); else if (C == 2) MN_COMP(2); else if (C == 3) MN_COMP(3); else if (C == 4) MN_COMP(4);
    else if (C <= 8) MN_COMP(8); else MN_COMP(16);
#undef MN_COMP
    MN_LAUNCH_CHECK("composite_kernel");
    return MI_NERF_OK;
}

int stage_composite_backward(const float* raw, const float* z, const float* rays, int ray_stride, int64_t n, int S,
                             const float* d_rgb, float* d_raw, hipStream_t st) {
    MN_CHECK_ARG(n >= 0 && S >= 1 && S <= 1024, "bad sizes (n=%lld S=%d)", (long long)n, S);
    MN_CHECK_ARG(ray_stride == 3 || ray_stride == 6, "ray_stride must be 3 or 6");
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(raw && z && rays && d_rgb && d_raw, "NULL pointer");
    const dim3 grid(blocks_for(n, 4)), block(256);
    const int C = (S + 63) / 64;
#define MN_COMPB(CC) hipLaunchKernelGGL(composite_bwd_kernel<CC>, grid, block, 0, st, raw, z, rays, ray_stride, (long long)n, S, d_rgb, d_raw)
    if (C == 1) MN_COMPB(1); else if (C == 2) MN_COMPB(2); else if (C == 3) MN_COMPB(3); else if (C == 4) MN_COMPB(4);
    else if (C <= 8) MN_COMPB(8); else MN_COMPB(16);
#undef MN_COMPB
    MN_LAUNCH_CHECK("composite_bwd_kernel");
    return MI_NERF_OK;
}

int stage_sample_pdf(const float* bins, const float* weights, int64_t n, int B, int N, int det, const float* u, float* out,
                     hipStream_t st) {
    // 4 rays per block, 2B floats each, within the 64 KB of dynamic LDS a launch gets without opting in
    MN_CHECK_ARG(n >= 0 && B >= 2 && B <= MAX_LDS_FLOATS_PER_RAY / 2 && N >= 1, "bad sizes (B=%d N=%d; at most %d bins)", B, N, MAX_LDS_FLOATS_PER_RAY / 2);
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(bins && weights && out && (det || u), "NULL pointer");
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(blocks_for(n, 4)), dim3(256), (size_t)4 * 2 * B * sizeof(float), st, bins, weights,
                       (long long)n, B, N, det, u, out);
    MN_LAUNCH_CHECK("sample_pdf_kernel");
    return MI_NERF_OK;
}

// u == NULL (and not det): the uniforms are drawn in the kernel from (seed, stream 1, ray0 + ray, sample)
int stage_fine_z(const float* z_c, const float* w_c, int64_t n, int Sc, int Nf, int det, const float* u, uint32_t seed, int64_t ray0,
                 float* z_f, float* z_samp, hipStream_t st) {
    // 4 rays per block, 2(Sc-1) + pow2(Sc + Nf) floats each, within the 64 KB of dynamic LDS a launch gets without opting in
    MN_CHECK_ARG(n >= 0 && Sc >= 3 && Nf >= 1 && Sc + Nf <= MAX_LDS_FLOATS_PER_RAY, "bad sizes (Sc=%d Nf=%d)", Sc, Nf);
    int n2 = 2;
    while (n2 < Sc + Nf) n2 <<= 1;
    MN_CHECK_ARG(2 * (Sc - 1) + n2 <= MAX_LDS_FLOATS_PER_RAY, "bad sizes (Sc=%d Nf=%d: 2*(Sc-1) + %d (Sc+Nf rounded up to a power of two) must not exceed %d)",
                 Sc, Nf, n2, MAX_LDS_FLOATS_PER_RAY);
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(z_c && w_c && z_f, "NULL pointer");
    const size_t lds = (size_t)4 * (2 * (Sc - 1) + n2) * sizeof(float);
    hipLaunchKernelGGL(fine_z_kernel, dim3(blocks_for(n, 4)), dim3(256), lds, st, z_c, w_c, (long long)n, Sc, Nf, n2, det,
                       Jitter{u, seed, 1u, (long long)ray0}, z_f, z_samp);
    MN_LAUNCH_CHECK("fine_z_kernel");
    return MI_NERF_OK;
}

// composite of the coarse pass + resampling + merge-sort in one launch (what mi_nerf_render_rays runs between its two network passes)
int stage_composite_fine_z(const float* raw_c, const float* z_c, const float* rays, int64_t n, int Sc, int Nf, int det, const float* u,
                           uint32_t seed, int64_t ray0, float* rgb_c, float* disp_c, float* w_c, float* z_f, hipStream_t st) {
    MN_CHECK_ARG(n >= 0 && Sc >= 3 && Sc <= 1024 && Nf >= 1 && Sc + Nf <= MAX_LDS_FLOATS_PER_RAY, "bad sizes (Sc=%d Nf=%d)", Sc, Nf);
    int n2 = 2;
    while (n2 < Sc + Nf) n2 <<= 1;
    // (tighter than stage_fine_z's 2 (Sc - 1) + n2, but never the binding limit of mi_nerf_render_rays: its fine compositing takes at most
    // 1024 depths per ray (stage_composite), so Sc + Nf <= 1024, n2 <= 1024 and 3 Sc - 2 + n2 <= 4094 -- every sample count the staged
    // sequence rendered, the fused launch renders too; tests/test_gpu_parity.py test_largest_sample_counts)
    MN_CHECK_ARG(Sc + 2 * (Sc - 1) + n2 <= MAX_LDS_FLOATS_PER_RAY, "bad sizes (Sc=%d Nf=%d: 3 Sc - 2 + %d must not exceed %d)", Sc, Nf, n2,
                 MAX_LDS_FLOATS_PER_RAY);
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(raw_c && z_c && rays && rgb_c && disp_c && w_c && z_f, "NULL pointer");
    const size_t lds = (size_t)4 * (Sc + 2 * (Sc - 1) + n2) * sizeof(float);
    const dim3 grid(blocks_for(n, 4)), block(256);
    const Jitter j{u, seed, 1u, (long long)ray0};
    const int C = (Sc + 63) / 64;
#define MN_CFZ(CC) hipLaunchKernelGGL(composite_fine_z_kernel<CC>, grid, block, lds, st, raw_c, z_c, rays, (long long)n, Sc, Nf, n2, det, j, rgb_c, disp_c, w_c, z_f)
    if (C == 1) MN_CFZ(1); else if (C == 2) MN_CFZ(2); else if (C == 3) MN_CFZ(3); else if (C == 4) MN_CFZ(4);
    else if (C <= 8) MN_CFZ(8); else MN_CFZ(16);
#undef MN_CFZ
    MN_LAUNCH_CHECK("composite_fine_z_kernel");
    return MI_NERF_OK;
}

}  // namespace minerf
