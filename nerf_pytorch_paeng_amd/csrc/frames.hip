// frames.hip -- the device ops either side of the path in the reference's callers:
//   image_metrics   img2mse + mse2psnr            utils.py:18-23, test.py:64-68
//   nanmax, to8b    disp / nanmax(disp), to8b     test.py:55-56, utils.py:15
//   rays_rgb        global-batch ray precompute   main.py:92-101 (get_rays_np rays.py:7-17 for every training image,
//                                                 concatenated with the pixels, flattened to [N*H*W, 3, 3])
//   permute_rows    np.random.shuffle(rays_rgb)   main.py:102, utils.py:47-52 (gather by a permutation)
// All HBM-bound streaming kernels: one pass, coalesced, grid-stride; reductions are two-stage and deterministic.
#include "common.h"

namespace minerf {

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;                                   // valid on thread 0
}

__global__ __launch_bounds__(256) void sqerr_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                             double* __restrict__ partial) {
    __shared__ double sh[4];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = a[i] - b[i];            // fp32 difference and square, like torch (utils.py:18)
        s += (double)(d * d);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void mse_final_kernel(const double* __restrict__ partial, int nb, long long n, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        const float mse = (float)(s / (double)n);
        out[0] = mse;
        out[1] = -10.0f * logf(mse) / logf(10.0f);              // mse2psnr, utils.py:21-23
    }
}

__global__ __launch_bounds__(256) void nanmax_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float m = -INFINITY;
    bool any = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        if (v == v) { m = v > m ? v : m; any = true; }
    }
    if (!any) m = __builtin_nanf("");                               // all-NaN slice: np.nanmax returns NaN
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const float o = __shfl_xor(m, k, 64);
        m = (o == o && (m != m || o > m)) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = sh[0];
        for (int i = 1; i < 4; ++i) { const float o = sh[i]; r = (o == o && (r != r || o > r)) ? o : r; }
        partial[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(256) void nanmax_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    __shared__ float sh[4];
    float m = __builtin_nanf("");
    for (int i = threadIdx.x; i < nb; i += 256) { const float o = partial[i]; m = (o == o && (m != m || o > m)) ? o : m; }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const float o = __shfl_xor(m, k, 64);
        m = (o == o && (m != m || o > m)) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = sh[0];
        for (int i = 1; i < 4; ++i) { const float o = sh[i]; r = (o == o && (r != r || o > r)) ? o : r; }
        out[0] = r;
    }
}

// to8b(x / divisor): (255 * clip(v, 0, 1)).astype(uint8)  -- fp32 product, truncation (utils.py:15); NaN -> 0
__global__ __launch_bounds__(256) void to8b_kernel(const float* __restrict__ x, long long n, const float* __restrict__ divisor,
                                                    unsigned char* __restrict__ out) {
    const float dv = divisor ? divisor[0] : 1.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = x[i];
        if (divisor) v = v / dv;
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);                // np.clip; NaN falls through both comparisons
        const float s = 255.0f * v;
        out[i] = (s == s) ? (unsigned char)s : (unsigned char)0;
    }
}

// rays_rgb[(img*H*W + pix)][3][3] = (origin, direction, pixel): main.py:92-101 for all images in one launch.
// poses [n_img][12] row-major 3x4; images [n_img][H*W][3]; k4 = fx, fy, cx, cy.
__global__ __launch_bounds__(256) void rays_rgb_kernel(int W, int H, float fx, float fy, float cx, float cy, const float* __restrict__ poses,
                                                        const float* __restrict__ images, long long n_img, float* __restrict__ out) {
    // a block owns 256 consecutive rays: each thread builds one 9-float row in LDS, then the block streams the 2304 floats
    // out as consecutive 16-byte stores (a thread writing its own 36-byte row touches 9 cache lines per wave instruction)
    __shared__ __attribute__((aligned(16))) float tile[256 * 9];
    const long long hw = (long long)H * W;
    const long long total = n_img * hw;
    const long long nblk = (total + 255) / 256;
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long long idx = blk * 256 + threadIdx.x;
        if (idx < total) {
            const long long img = idx / hw;
            const long long pix = idx - img * hw;
            const int py = (int)(pix / W), px = (int)(pix - (long long)py * W);
            const float* c = poses + img * 12;
            const float dx = ((float)px - cx) / fx, dy = -((float)py - cy) / fy, dz = -1.0f;     // rays.py:10-11
            float* o = tile + threadIdx.x * 9;
            o[0] = c[3]; o[1] = c[7]; o[2] = c[11];                                               // rays.py:16
            // np.sum(dirs[..., None, :] * c2w[:3,:3], -1): left-to-right fp32 sum of three products (rays.py:14)
            o[3] = (dx * c[0] + dy * c[1]) + dz * c[2];
            o[4] = (dx * c[4] + dy * c[5]) + dz * c[6];
            o[5] = (dx * c[8] + dy * c[9]) + dz * c[10];
            const float* p = images + idx * 3;
            o[6] = p[0]; o[7] = p[1]; o[8] = p[2];
        }
        __syncthreads();
        const long long base = blk * 256 * 9;                      // multiple of 4 floats: 16-byte aligned
        const long long lim = (total * 9 - base < 2304) ? total * 9 - base : 2304;
        for (int q = threadIdx.x * 4; q < lim; q += 1024) {
            if (q + 4 <= lim) *(f32x4*)(out + base + q) = *(const f32x4*)(tile + q);
            else for (int e = q; e < lim; ++e) out[base + e] = tile[e];
        }
        __syncthreads();
    }
}

// dst[i] = src[perm[i]], i < n, for rows of `row_floats` floats: a gather of n rows by index (np.random.shuffle of the leading axis == the gather
// of ALL rows by a permutation, main.py:102; one training batch == the gather of B rows, train.py:29).  A block owns 256 consecutive OUTPUT rows:
// their 256 indices come in as one coalesced 8-byte load per thread and sit in LDS; then the block walks the 256 * RF output floats in order --
// lane -> (row, column) by a 32-bit divide by the compile-time RF -- so the writes are consecutive floats and the RF lanes of one row read one 36-byte
// piece of a random 128-byte line.  The reads are what bounds a whole-table shuffle: a random 36-byte row costs a 128-byte HBM request (two for
// a quarter of the rows), ~160 B fetched per 36 B used, so 80 algorithmic bytes per row move ~204 (profiles/r06_permute_rows_bound.txt).
template <int RF>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ perm, long long n,
                                                           int row_floats, float* __restrict__ dst) {
    __shared__ long long row_of[256];
    const int rf = RF > 0 ? RF : row_floats;
    const long long nblk = (n + 255) / 256;
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long long row0 = blk * 256;
        const int rows = (int)(n - row0 < 256 ? n - row0 : 256);
        if ((int)threadIdx.x < rows) row_of[threadIdx.x] = perm[row0 + threadIdx.x];
        __syncthreads();
        float* out = dst + row0 * rf;
        const int total = rows * rf;
        for (int q = threadIdx.x; q < total; q += 256) {
            const int lr = q / rf, c = q - lr * rf;
            out[q] = src[row_of[lr] * rf + c];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
static unsigned grid_for(long long n, int cap) {
    long long b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int frames_image_metrics(const float* pred, const float* target, int64_t n, float* out2, void* scratch, size_t scratch_bytes, hipStream_t st) {
    MN_CHECK_ARG(n >= 1, "image_metrics needs at least one element (n=%lld)", (long long)n);
    MN_CHECK_ARG(pred && target && out2 && scratch, "NULL pointer");
    MN_CHECK_ARG(scratch_bytes >= RED_BLOCKS * sizeof(double), "scratch too small: %zu < %zu", scratch_bytes, RED_BLOCKS * sizeof(double));
    const unsigned nb = grid_for(n, RED_BLOCKS);
    hipLaunchKernelGGL(sqerr_partial_kernel, dim3(nb), dim3(256), 0, st, pred, target, (long long)n, (double*)scratch);
    MN_LAUNCH_CHECK("sqerr_partial_kernel");
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, st, (const double*)scratch, (int)nb, (long long)n, out2);
    MN_LAUNCH_CHECK("mse_final_kernel");
    return MI_NERF_OK;
}

int frames_nanmax(const float* x, int64_t n, float* out, void* scratch, size_t scratch_bytes, hipStream_t st) {
    MN_CHECK_ARG(n >= 1, "nanmax needs at least one element (n=%lld)", (long long)n);
    MN_CHECK_ARG(x && out && scratch, "NULL pointer");
    MN_CHECK_ARG(scratch_bytes >= RED_BLOCKS * sizeof(double), "scratch too small: %zu < %zu", scratch_bytes, RED_BLOCKS * sizeof(double));
    const unsigned nb = grid_for(n, RED_BLOCKS);
    hipLaunchKernelGGL(nanmax_partial_kernel, dim3(nb), dim3(256), 0, st, x, (long long)n, (float*)scratch);
    MN_LAUNCH_CHECK("nanmax_partial_kernel");
    hipLaunchKernelGGL(nanmax_final_kernel, dim3(1), dim3(256), 0, st, (const float*)scratch, (int)nb, out);
    MN_LAUNCH_CHECK("nanmax_final_kernel");
    return MI_NERF_OK;
}

int frames_to8b(const float* x, int64_t n, const float* divisor, unsigned char* out, hipStream_t st) {
    MN_CHECK_ARG(n >= 0, "bad n=%lld", (long long)n);
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(x && out, "NULL pointer");
    hipLaunchKernelGGL(to8b_kernel, dim3(grid_for(n, 65536)), dim3(256), 0, st, x, (long long)n, divisor, out);
    MN_LAUNCH_CHECK("to8b_kernel");
    return MI_NERF_OK;
}

int frames_rays_rgb(int W, int H, const float k4[4], const float* poses, const float* images, int64_t n_img, float* out, hipStream_t st) {
    MN_CHECK_ARG(W >= 1 && H >= 1 && n_img >= 0 && k4, "bad sizes W=%d H=%d n_img=%lld", W, H, (long long)n_img);
    if (n_img == 0) return MI_NERF_OK;
    MN_CHECK_ARG(poses && images && out, "NULL pointer");
    const long long total = (long long)n_img * H * W;
    hipLaunchKernelGGL(rays_rgb_kernel, dim3(grid_for(total, 65536)), dim3(256), 0, st, W, H, k4[0], k4[1], k4[2], k4[3], poses, images,
                       (long long)n_img, out);
    MN_LAUNCH_CHECK("rays_rgb_kernel");
    return MI_NERF_OK;
}

int frames_permute_rows(const float* src, const int64_t* perm, int64_t n, int row_floats, float* dst, hipStream_t st) {
    MN_CHECK_ARG(n >= 0 && row_floats >= 1 && row_floats <= (1 << 20), "bad sizes n=%lld row_floats=%d", (long long)n, row_floats);
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(src && perm && dst && src != dst, "NULL pointer or in-place permutation");
    const dim3 grid(grid_for((long long)n, 65536));               // one block per 256 output rows (grid-stride beyond 65536 blocks)
    if (row_floats == 9)                                            // the [3][3] rows of rays_rgb (main.py:97-100)
        hipLaunchKernelGGL(gather_rows_kernel<9>, grid, dim3(256), 0, st, src, (const long long*)perm, (long long)n, row_floats, dst);
    else
        hipLaunchKernelGGL(gather_rows_kernel<0>, grid, dim3(256), 0, st, src, (const long long*)perm, (long long)n, row_floats, dst);
    MN_LAUNCH_CHECK("gather_rows_kernel");
    return MI_NERF_OK;
}

}  // namespace minerf
