// api.hip -- extern "C" surface of libmi_nerf.so (include/mi_nerf.h): argument checks, error text,
// the fused render_rays launch sequence, the hipEvent timing hook and the MFMA layout self test.
#include <string>
#include "common.h"
#include "layout.h"

namespace minerf {

// ---- error plumbing --------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return MI_NERF_EHIP;
}

// ---- per-device launch state -----------------------------------------------------------------------
int device_cus() {
    static std::atomic<int> cus[MN_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::atomic<int>& slot = cus[dev & (MN_MAX_DEVICES - 1)];
    int c = slot.load(std::memory_order_relaxed);
    if (c <= 0) {
        hipDeviceProp_t prop;
        c = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        slot.store(c, std::memory_order_relaxed);
    }
    return c;
}
int ensure_lds_opt_in(LdsOptIn& state, const void* kernel) {
    int dev = 0;
    MN_HIP(hipGetDevice(&dev));
    std::atomic<bool>& done = state.done[dev & (MN_MAX_DEVICES - 1)];
    if (!done.load(std::memory_order_acquire)) {
        MN_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        done.store(true, std::memory_order_release);
    }
    return MI_NERF_OK;
}

// ---- implemented in the other translation units --------------------------------------------------
int pack_fp32(const mi_nerf_net*, const mi_nerf_params*, void*, size_t);
int pack_bf16(const mi_nerf_net*, const mi_nerf_params*, void*, size_t);
size_t packed_bytes_bf16(const mi_nerf_net*);
size_t pack_map_bf16_len(const mi_nerf_net*);
int pack_map_bf16(const mi_nerf_net*, int32_t*, size_t);
int pack_apply_bf16(const mi_nerf_net*, const int32_t*, const float*, void*, size_t, hipStream_t);
int mlp_rays_fp32(const mi_nerf_net*, const void*, const float*, const float*, int64_t, int, float*, hipStream_t);
int mlp_embedded_fp32(const mi_nerf_net*, const void*, const float*, int64_t, float*, hipStream_t);
int mlp_rays_bf16(const mi_nerf_net*, const void*, const float*, const float*, int64_t, int, float*, hipStream_t, int points_per_wave, const StratDraw* strat,
                  FineDraw* fine = nullptr);
size_t packed_bytes_f16s(const mi_nerf_net*);
int pack_f16s(const mi_nerf_net*, const mi_nerf_params*, void*, size_t);
int mlp_rays_f16s(const mi_nerf_net*, const void*, const float*, const float*, int64_t, int, float*, hipStream_t);
int mlp_rays_f16s_stash(const mi_nerf_net*, const void*, const float*, const float*, int64_t, int, float*, float*, float*, float*, unsigned*, unsigned*,
                        hipStream_t);
size_t pack_map_f16s_len(const mi_nerf_net*);
int pack_map_f16s(const mi_nerf_net*, int32_t*, size_t);
int pack_apply_f16s(const mi_nerf_net*, const int32_t*, const float*, void*, size_t, unsigned*, hipStream_t);
size_t packed_bytes_bwd_f16s(const mi_nerf_net*);
int pack_bwd_f16s(const mi_nerf_net*, const mi_nerf_params*, void*, size_t);
size_t pack_map_bwd_f16s_len(const mi_nerf_net*);
int pack_map_bwd_f16s(const mi_nerf_net*, int32_t*, size_t);
int pack_apply_bwd_f16s(const mi_nerf_net*, const int32_t*, const float*, void*, size_t, unsigned*, hipStream_t);
// MI_NERF_MODE_* (mi_nerf_render_cfg.mode / mi_nerf_time_mlp_rays) -> launch shape of the bf16 kernel (0: chosen per launch)
static inline int bf16_points_per_wave(int mode) { return mode == MI_NERF_MODE_BF16_64 ? 64 : (mode == MI_NERF_MODE_BF16_32 ? 32 : (mode == 4 ? 832 : 0)); }
static inline bool mode_is_bf16(int mode) { return mode >= MI_NERF_MODE_BF16 && mode <= 4; }
int wgrad_products(int, const float* const*, const int*, const int*, const float* const*, const int*, const int*, int64_t, float* const*, const int*,
                   float* const*, void*, size_t, hipStream_t, bool);
int mlp_rays_fp32_stash(const mi_nerf_net*, const void*, const float*, const float*, int64_t, int, float*, float*, float*, float*, unsigned*,
                        unsigned*, hipStream_t);
int mlp_backward_fp32(const mi_nerf_net*, const void*, const void*, const float*, const float*, int64_t, int, const float*, const void*, void*,
                      size_t, float*, int, hipStream_t, const float*, int64_t, int);
int mlp_embedded_fp32_stash(const mi_nerf_net*, const void*, const float*, int64_t, float*, float*, float*, float*, unsigned*, unsigned*,
                            hipStream_t);
int train_layout(const mi_nerf_net*, int64_t, int, mi_nerf_train_layout*);
size_t wgrad_scratch_bytes();
int pack_apply(const int32_t*, const float*, size_t, void*, hipStream_t);
int pack_bwd_fp32(const mi_nerf_net*, const mi_nerf_params*, void*, size_t);
size_t packed_bytes_bwd(const mi_nerf_net*);
int pack_map(const mi_nerf_net*, int, int32_t*, size_t);
int frames_image_metrics(const float*, const float*, int64_t, float*, void*, size_t, hipStream_t);
int frames_nanmax(const float*, int64_t, float*, void*, size_t, hipStream_t);
int frames_to8b(const float*, int64_t, const float*, unsigned char*, hipStream_t);
int frames_rays_rgb(int, int, const float*, const float*, const float*, int64_t, float*, hipStream_t);
int frames_permute_rows(const float*, const int64_t*, int64_t, int, float*, hipStream_t);
int stage_make_o_d(int, int, const float*, const float*, int, int, float*, float*, hipStream_t);
int stage_make_o_d_pixels(int, int, const float*, const float*, const int64_t*, int64_t, float*, float*, hipStream_t);
int stage_ndc(int, int, float, float, const float*, int64_t, const float*, int64_t, int64_t, float*, float*, hipStream_t);
int stage_fill_uniform(uint32_t, uint32_t, int64_t, int64_t, int, float*, hipStream_t);
int stage_stratified(int64_t, int, float, float, const float*, uint32_t, int64_t, float*, hipStream_t);
int stage_embed(const float*, const float*, int64_t, int, int, int, float*, hipStream_t);
int stage_posenc(const float*, int64_t, int, float*, hipStream_t);
int stage_composite(const float*, const float*, const float*, int, int64_t, int, float*, float*, float*, float*, float*, hipStream_t);
int stage_composite_backward(const float*, const float*, const float*, int, int64_t, int, const float*, float*, hipStream_t);
int stage_sample_pdf(const float*, const float*, int64_t, int, int, int, const float*, float*, hipStream_t);
int stage_fine_z(const float*, const float*, int64_t, int, int, int, const float*, uint32_t, int64_t, float*, float*, hipStream_t);
int stage_composite_fine_z(const float*, const float*, const float*, int64_t, int, int, int, const float*, uint32_t, int64_t, float*, float*, float*,
                           float*, hipStream_t);

static int check_net_basic(const mi_nerf_net* net) {
    MN_CHECK_ARG(net != nullptr, "net is NULL");
    MN_CHECK_ARG(net->W >= 2 && net->W <= MAX_KERNEL_WIDTH, "unsupported width W=%d (the fp32 inference kernels run 2 <= W <= %d, padded to 128 / 256 / 384 / 512; "
                 "training, bf16 and split precision: 128 / 256 and 256)", net->W, MAX_KERNEL_WIDTH);
    MN_CHECK_ARG(net->D >= 2 && net->D <= 16, "unsupported depth D=%d", net->D);
    MN_CHECK_ARG(net->L_x >= 0 && net->L_x <= 10 && net->L_d >= 0 && net->L_d <= 4,
                 "unsupported encoding L_x=%d L_d=%d (the kernels evaluate up to 10 / 4 frequencies; fewer run with zero weights on the rest)", net->L_x, net->L_d);
    MN_CHECK_ARG(net->skip >= -1, "bad skip=%d", net->skip);
    return MI_NERF_OK;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int workspace_layout(const mi_nerf_render_cfg* cfg, int64_t n, mi_nerf_workspace_layout* L) {
    MN_CHECK_ARG(cfg && L, "NULL cfg/layout");
    MN_CHECK_ARG(n >= 0 && cfg->Sc >= 1 && cfg->Nf >= 0, "bad sizes n=%lld Sc=%d Nf=%d", (long long)n, cfg->Sc, cfg->Nf);
    MN_CHECK_ARG(cfg->Nf == 0 || cfg->Sc >= 3, "hierarchical sampling needs at least 3 coarse samples");
    size_t off = 0;
    const size_t nn = (size_t)n, Sc = (size_t)cfg->Sc, St = (size_t)(cfg->Sc + cfg->Nf);
    L->z_c = off;       off += align256(nn * Sc * 4);
    L->raw_c = off;     off += align256(nn * Sc * 16);
    L->weights_c = off; off += align256(nn * Sc * 4);
    L->z_f = off;       off += cfg->Nf > 0 ? align256(nn * St * 4) : 0;
    L->raw_f = off;     off += cfg->Nf > 0 ? align256(nn * St * 16) : 0;
    L->total = off;
    return MI_NERF_OK;
}

// ---- MFMA fragment-layout self test ------------------------------------------------------------
// D[32][32] = A[32][8] * B[8][32] through four v_mfma_f32_32x32x2_f32, using exactly the operand and
// result maps layout.h documents; integer-valued asymmetric data so any transposition shows up.
__global__ void mfma_selftest_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = 2 * s + (lane >> 5);
        const float a = A[(lane & 31) * 8 + k];     // A[i = lane&31][k]
        const float b = B[k * 32 + (lane & 31)];    // B[k][j = lane&31]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        D[i * 32 + (lane & 31)] = acc[r];
    }
}

}  // namespace minerf

using namespace minerf;

extern "C" {

int mi_nerf_abi_version(void) { return MI_NERF_ABI_VERSION; }
const char* mi_nerf_last_error(void) { return g_err; }

size_t mi_nerf_packed_bytes(const mi_nerf_net* net) {
    if (check_net_basic(net)) return 0;
    return make_layout(net->D, net->W, net->skip, net->L_x, net->L_d).total_bytes;
}
int mi_nerf_pack_weights(const mi_nerf_net* net, const mi_nerf_params* params, void* blob, size_t bytes) {
    if (int rc = check_net_basic(net)) return rc;
    MN_CHECK_ARG(params && blob, "NULL params/blob");
    MN_CHECK_ARG(params->linear_x_w && params->linear_x_b && params->linear_density_w && params->linear_density_b &&
                 params->linear_feat_w && params->linear_feat_b && params->linear_d_w && params->linear_d_b &&
                 params->linear_color_w && params->linear_color_b, "NULL parameter pointer");
    for (int l = 0; l < net->D; ++l) MN_CHECK_ARG(params->linear_x_w[l] && params->linear_x_b[l], "NULL trunk layer %d", l);
    return pack_fp32(net, params, blob, bytes);
}
size_t mi_nerf_packed_bytes_bf16(const mi_nerf_net* net) {
    if (check_net_basic(net)) return 0;
    return packed_bytes_bf16(net);
}
size_t mi_nerf_packed_bytes_f16s(const mi_nerf_net* net) {
    return packed_bytes_f16s(net);
}
int mi_nerf_pack_weights_f16s(const mi_nerf_net* net, const mi_nerf_params* params, void* blob, size_t bytes) {
    MN_CHECK_ARG(net && params && blob, "NULL argument");
    return pack_f16s(net, params, blob, bytes);
}
int mi_nerf_pack_weights_bf16(const mi_nerf_net* net, const mi_nerf_params* params, void* blob, size_t bytes) {
    if (int rc = check_net_basic(net)) return rc;
    MN_CHECK_ARG(params && blob, "NULL params/blob");
    return pack_bf16(net, params, blob, bytes);
}

int mi_nerf_make_o_d(int W, int H, const float k4[4], const float pose12[12], int row0, int n_rows, float* o, float* d, void* st) {
    MN_CHECK_ARG(k4 && pose12, "NULL camera");
    return stage_make_o_d(W, H, k4, pose12, row0, n_rows, o, d, (hipStream_t)st);
}
int mi_nerf_make_o_d_pixels(int W, int H, const float k4[4], const float pose12[12], const int64_t* pix, int64_t n, float* o,
                            float* d, void* st) {
    MN_CHECK_ARG(k4 && pose12, "NULL camera");
    return stage_make_o_d_pixels(W, H, k4, pose12, pix, n, o, d, (hipStream_t)st);
}
int mi_nerf_ndc_rays(int H, int W, float focal, float near_, const float* o_in, int64_t os, const float* d_in, int64_t ds, int64_t n,
                     float* o_out, float* d_out, void* st) {
    return stage_ndc(H, W, focal, near_, o_in, os, d_in, ds, n, o_out, d_out, (hipStream_t)st);
}
int mi_nerf_fill_uniform(uint32_t seed, uint32_t stream_id, int64_t ray0, int64_t n_rays, int S, float* out, void* st) {
    return stage_fill_uniform(seed, stream_id, ray0, n_rays, S, out, (hipStream_t)st);
}
int mi_nerf_stratified_z(int64_t n_rays, int S, float near_, float far_, const float* t_rand, float* z, void* st) {
    MN_CHECK_ARG(t_rand != nullptr || n_rays == 0, "t_rand is NULL");
    return stage_stratified(n_rays, S, near_, far_, t_rand, 0, 0, z, (hipStream_t)st);
}
int mi_nerf_sample_pdf(const float* bins, const float* weights, int64_t n, int B, int N, int det, const float* u, float* out, void* st) {
    return stage_sample_pdf(bins, weights, n, B, N, det, u, out, (hipStream_t)st);
}
int mi_nerf_fine_z(const float* z_c, const float* w_c, int64_t n, int Sc, int Nf, int det, const float* u, float* z_f, float* z_s,
                   void* st) {
    MN_CHECK_ARG(det || u != nullptr || n == 0, "u is NULL (and det == 0)");
    return stage_fine_z(z_c, w_c, n, Sc, Nf, det, u, 0, 0, z_f, z_s, (hipStream_t)st);
}
int mi_nerf_embed(const float* rays, const float* z, int64_t n_rays, int S, int L_x, int L_d, float* out, void* st) {
    return stage_embed(rays, z, n_rays, S, L_x, L_d, out, (hipStream_t)st);
}
int mi_nerf_posenc(const float* x, int64_t n, int L, float* out, void* st) { return stage_posenc(x, n, L, out, (hipStream_t)st); }
int mi_nerf_mlp_embedded(const mi_nerf_net* net, const void* packed, const float* x, int64_t n, float* out, void* st) {
    return mlp_embedded_fp32(net, packed, x, n, out, (hipStream_t)st);
}
int mi_nerf_mlp_rays(const mi_nerf_net* net, const void* packed, const float* rays, const float* z, int64_t n_rays, int S, float* raw,
                     void* st) {
    return mlp_rays_fp32(net, packed, rays, z, n_rays, S, raw, (hipStream_t)st);
}
int mi_nerf_mlp_rays_bf16(const mi_nerf_net* net, const void* packed, const float* rays, const float* z, int64_t n_rays, int S,
                          float* raw, void* st) {
    MN_CHECK_ARG(z != nullptr || n_rays == 0, "z is NULL");
    return mlp_rays_bf16(net, packed, rays, z, n_rays, S, raw, (hipStream_t)st, 0, nullptr);
}
int mi_nerf_mlp_rays_f16s(const mi_nerf_net* net, const void* packed, const float* rays, const float* z, int64_t n_rays, int S, float* raw,
                          void* st) {
    return mlp_rays_f16s(net, packed, rays, z, n_rays, S, raw, (hipStream_t)st);
}
int mi_nerf_mlp_rays_bf16_shape(const mi_nerf_net* net, const void* packed, const float* rays, const float* z, int64_t n_rays, int S,
                                float* raw, int points_per_wave, void* st) {
    MN_CHECK_ARG(z != nullptr || n_rays == 0, "z is NULL");
    return mlp_rays_bf16(net, packed, rays, z, n_rays, S, raw, (hipStream_t)st, points_per_wave, nullptr);
}
int mi_nerf_composite(const float* raw, const float* z, const float* rays, int ray_stride, int64_t n, int S, float* rgb, float* disp,
                      float* acc, float* weights, float* depth, void* st) {
    return stage_composite(raw, z, rays, ray_stride, n, S, rgb, disp, acc, weights, depth, (hipStream_t)st);
}

int mi_nerf_composite_backward(const float* raw, const float* z, const float* rays, int ray_stride, int64_t n, int S,
                               const float* d_rgb, float* d_raw, void* st) {
    return stage_composite_backward(raw, z, rays, ray_stride, n, S, d_rgb, d_raw, (hipStream_t)st);
}

size_t mi_nerf_param_count(const mi_nerf_net* net) {
    if (check_net_basic(net)) return 0;
    return make_param_offsets(net->D, net->W, net->skip, net->L_x, net->L_d).total;
}
size_t mi_nerf_packed_bytes_bwd(const mi_nerf_net* net) {
    if (check_net_basic(net)) return 0;
    return packed_bytes_bwd(net);
}
int mi_nerf_pack_weights_bwd(const mi_nerf_net* net, const mi_nerf_params* params, void* host_blob, size_t blob_bytes) {
    if (int rc = check_net_basic(net)) return rc;
    MN_CHECK_ARG(params && host_blob, "NULL params/blob");
    return pack_bwd_fp32(net, params, host_blob, blob_bytes);
}
int mi_nerf_pack_map(const mi_nerf_net* net, int kind, int32_t* map_host, size_t map_len) {
    if (int rc = check_net_basic(net)) return rc;
    return pack_map(net, kind, map_host, map_len);
}
size_t mi_nerf_pack_map_bf16_len(const mi_nerf_net* net) {
    if (!net) return 0;
    return pack_map_bf16_len(net);
}
int mi_nerf_pack_map_bf16(const mi_nerf_net* net, int32_t* map_host, size_t map_len) {
    MN_CHECK_ARG(net && map_host, "NULL pointer");
    return pack_map_bf16(net, map_host, map_len);
}
int mi_nerf_pack_apply_bf16(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_dev, void* blob_dev, size_t blob_bytes, void* st) {
    MN_CHECK_ARG(net, "net is NULL");
    return pack_apply_bf16(net, map_dev, flat_dev, blob_dev, blob_bytes, (hipStream_t)st);
}
int mi_nerf_pack_apply(const int32_t* map_dev, const float* flat_dev, size_t blob_bytes, void* blob_dev, void* st) {
    return pack_apply(map_dev, flat_dev, blob_bytes, blob_dev, (hipStream_t)st);
}
int mi_nerf_train_layout_query(const mi_nerf_net* net, int64_t n_rays, int S, mi_nerf_train_layout* out) {
    if (int rc = check_net_basic(net)) return rc;
    return train_layout(net, n_rays, S, out);
}
int mi_nerf_mlp_rays_train(const mi_nerf_net* net, const void* packed, const float* rays, const float* z, int64_t n_rays, int S,
                           float* raw, void* stash, size_t stash_bytes, void* st) {
    if (int rc = check_net_basic(net)) return rc;
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    mi_nerf_train_layout L;
    if (int rc = train_layout(net, n_rays, S, &L)) return rc;
    if (n_rays == 0) return MI_NERF_OK;
    MN_CHECK_ARG(stash != nullptr && stash_bytes >= L.stash_bytes, "stash too small: %zu < %zu", stash_bytes, L.stash_bytes);
    return mlp_rays_fp32_stash(net, packed, rays, z, n_rays, S, raw, (float*)((char*)stash + L.stash_h), (float*)((char*)stash + L.stash_f),
                               (float*)((char*)stash + L.stash_g), (unsigned*)((char*)stash + L.mask_h), (unsigned*)((char*)stash + L.mask_g),
                               (hipStream_t)st);
}
int mi_nerf_mlp_rays_train_f16s(const mi_nerf_net* net, const void* packed_f16s, const float* rays, const float* z, int64_t n_rays, int S,
                                float* raw, void* stash, size_t stash_bytes, void* st) {
    if (int rc = check_net_basic(net)) return rc;
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    mi_nerf_train_layout L;
    if (int rc = train_layout(net, n_rays, S, &L)) return rc;
    if (n_rays == 0) return MI_NERF_OK;
    MN_CHECK_ARG(stash != nullptr && stash_bytes >= L.stash_bytes, "stash too small: %zu < %zu", stash_bytes, L.stash_bytes);
    return mlp_rays_f16s_stash(net, packed_f16s, rays, z, n_rays, S, raw, (float*)((char*)stash + L.stash_h), (float*)((char*)stash + L.stash_f),
                               (float*)((char*)stash + L.stash_g), (unsigned*)((char*)stash + L.mask_h), (unsigned*)((char*)stash + L.mask_g),
                               (hipStream_t)st);
}
size_t mi_nerf_pack_map_f16s_len(const mi_nerf_net* net) {
    if (!net) return 0;
    return pack_map_f16s_len(net);
}
int mi_nerf_pack_map_f16s(const mi_nerf_net* net, int32_t* map_host, size_t map_len) {
    MN_CHECK_ARG(net && map_host, "NULL pointer");
    return pack_map_f16s(net, map_host, map_len);
}
size_t mi_nerf_packed_bytes_bwd_f16s(const mi_nerf_net* net) {
    if (!net) return 0;
    return packed_bytes_bwd_f16s(net);
}
int mi_nerf_pack_weights_bwd_f16s(const mi_nerf_net* net, const mi_nerf_params* params, void* host_blob, size_t blob_bytes) {
    MN_CHECK_ARG(net && params && host_blob, "NULL pointer");
    return pack_bwd_f16s(net, params, host_blob, blob_bytes);
}
size_t mi_nerf_pack_map_bwd_f16s_len(const mi_nerf_net* net) {
    if (!net) return 0;
    return pack_map_bwd_f16s_len(net);
}
int mi_nerf_pack_map_bwd_f16s(const mi_nerf_net* net, int32_t* map_host, size_t map_len) {
    MN_CHECK_ARG(net && map_host, "NULL pointer");
    return pack_map_bwd_f16s(net, map_host, map_len);
}
int mi_nerf_pack_apply_bwd_f16s(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_dev, void* blob_dev, size_t blob_bytes,
                                uint32_t* out_of_range_dev, void* st) {
    MN_CHECK_ARG(net, "net is NULL");
    return pack_apply_bwd_f16s(net, map_dev, flat_dev, blob_dev, blob_bytes, out_of_range_dev, (hipStream_t)st);
}
int mi_nerf_pack_apply_f16s(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_dev, void* blob_dev, size_t blob_bytes,
                            uint32_t* out_of_range_dev, void* st) {
    MN_CHECK_ARG(net, "net is NULL");
    return pack_apply_f16s(net, map_dev, flat_dev, blob_dev, blob_bytes, out_of_range_dev, (hipStream_t)st);
}
int mi_nerf_mlp_backward(const mi_nerf_net* net, const void* packed, const void* packed_bwd, const float* rays, const float* z,
                         int64_t n_rays, int S, const float* d_raw, const void* stash, void* work, size_t work_bytes, float* grads,
                         int stage, void* st) {
    if (int rc = check_net_basic(net)) return rc;
    return mlp_backward_fp32(net, packed, packed_bwd, rays, z, n_rays, S, d_raw, stash, work, work_bytes, grads, stage, (hipStream_t)st, nullptr, -1, 0);
}
int mi_nerf_mlp_backward_mode(const mi_nerf_net* net, const void* packed, const void* packed_bwd, const float* rays, const float* z,
                              int64_t n_rays, int S, const float* d_raw, const void* stash, void* work, size_t work_bytes, float* grads,
                              int stage, int mode, void* st) {
    if (int rc = check_net_basic(net)) return rc;
    MN_CHECK_ARG((mode & ~3) == 0, "unknown backward mode %d", mode);
    return mlp_backward_fp32(net, packed, packed_bwd, rays, z, n_rays, S, d_raw, stash, work, work_bytes, grads, stage, (hipStream_t)st, nullptr, -1, mode);
}
int mi_nerf_mlp_embedded_train(const mi_nerf_net* net, const void* packed, const float* x, int64_t n, float* out, void* stash, size_t stash_bytes,
                               void* st) {
    if (int rc = check_net_basic(net)) return rc;
    MN_CHECK_ARG(n >= 0, "bad n=%lld", (long long)n);
    mi_nerf_train_layout L;
    if (int rc = train_layout(net, (n + 31) / 32, 32, &L)) return rc;
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(stash != nullptr && stash_bytes >= L.stash_bytes, "stash too small: %zu < %zu", stash_bytes, L.stash_bytes);
    return mlp_embedded_fp32_stash(net, packed, x, n, out, (float*)((char*)stash + L.stash_h), (float*)((char*)stash + L.stash_f),
                                   (float*)((char*)stash + L.stash_g), (unsigned*)((char*)stash + L.mask_h), (unsigned*)((char*)stash + L.mask_g),
                                   (hipStream_t)st);
}
int mi_nerf_mlp_embedded_backward(const mi_nerf_net* net, const void* packed, const void* packed_bwd, const float* x, int64_t n, const float* d_out,
                                  const void* stash, void* work, size_t work_bytes, float* grads, void* st) {
    if (int rc = check_net_basic(net)) return rc;
    MN_CHECK_ARG(n >= 0, "bad n=%lld", (long long)n);
    MN_CHECK_ARG(x != nullptr || n == 0, "NULL device pointer");          // n == 0: mlp_backward_fp32 zero-fills grads (the gradient of nothing)
    static const float no_rows = 0.0f;                                     // never dereferenced: a non-NULL x selects the embedded mode
    return mlp_backward_fp32(net, packed, packed_bwd, nullptr, nullptr, (n + 31) / 32, 32, d_out, stash, work, work_bytes, grads, 0, (hipStream_t)st,
                             x ? x : &no_rows, n, 0);
}

int mi_nerf_image_metrics(const float* pred, const float* target, int64_t n, float* out2, void* scratch, size_t scratch_bytes, void* st) {
    return frames_image_metrics(pred, target, n, out2, scratch, scratch_bytes, (hipStream_t)st);
}
int mi_nerf_nanmax(const float* x, int64_t n, float* out, void* scratch, size_t scratch_bytes, void* st) {
    return frames_nanmax(x, n, out, scratch, scratch_bytes, (hipStream_t)st);
}
int mi_nerf_to8b(const float* x, int64_t n, const float* divisor, uint8_t* out, void* st) {
    return frames_to8b(x, n, divisor, out, (hipStream_t)st);
}
int mi_nerf_rays_rgb(int W, int H, const float k4[4], const float* poses, const float* images, int64_t n_img, float* out, void* st) {
    return frames_rays_rgb(W, H, k4, poses, images, n_img, out, (hipStream_t)st);
}
int mi_nerf_permute_rows(const float* src, const int64_t* perm, int64_t n, int row_floats, float* dst, void* st) {
    return frames_permute_rows(src, perm, n, row_floats, dst, (hipStream_t)st);
}

size_t mi_nerf_render_workspace_bytes(const mi_nerf_render_cfg* cfg, int64_t n_rays) {
    mi_nerf_workspace_layout L;
    if (workspace_layout(cfg, n_rays, &L)) return 0;
    return L.total;
}
int mi_nerf_render_workspace_layout(const mi_nerf_render_cfg* cfg, int64_t n_rays, mi_nerf_workspace_layout* out) {
    return workspace_layout(cfg, n_rays, out);
}

int mi_nerf_render_rays(const mi_nerf_net* net, const void* packed_c, const void* packed_f, const mi_nerf_render_cfg* cfg,
                        const float* rays, int64_t n, const float* t_rand, const float* u, void* ws, size_t ws_bytes,
                        float* rgb_c, float* disp_c, float* rgb_f, float* disp_f, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    mi_nerf_workspace_layout L;
    if (int rc = workspace_layout(cfg, n, &L)) return rc;
    if (n == 0) return MI_NERF_OK;
    MN_CHECK_ARG(ws_bytes >= L.total, "workspace too small: %zu < %zu", ws_bytes, L.total);
    MN_CHECK_ARG(rays && rgb_c && disp_c && packed_c && (ws || L.total == 0), "NULL pointer");
    MN_CHECK_ARG(cfg->Nf == 0 || (packed_f && rgb_f && disp_f), "fine pass needs packed_fine and outputs");
    MN_CHECK_ARG(cfg->reserved == 0, "mi_nerf_render_cfg.reserved must be 0");
    char* w = (char*)ws;
    float* z_c = (float*)(w + L.z_c);
    float* raw_c = (float*)(w + L.raw_c);
    float* wts_c = (float*)(w + L.weights_c);
    const int Sc = cfg->Sc, St = cfg->Sc + cfg->Nf;
    // 1-a) stratified depths; 2-a) coarse net; 3-a) composite          (nerf_process.py:187-198)
    // t_rand / u NULL: the jitter is drawn inside the consuming kernels (cfg->seed, cfg->ray_offset + ray, sample)
    MN_CHECK_ARG(cfg->mode >= MI_NERF_MODE_F32 && cfg->mode <= MI_NERF_MODE_F16S_BF16, "mode must be one of MI_NERF_MODE_* (0..6; got %d)", cfg->mode);
    const int ppw = bf16_points_per_wave(cfg->mode);
    // per network: which kernel family evaluates it (MI_NERF_MODE_F16S_BF16: coarse in split precision, fine in bf16)
    const bool coarse_f16s = cfg->mode == MI_NERF_MODE_F16S || cfg->mode == MI_NERF_MODE_F16S_BF16;
    const bool fine_f16s = cfg->mode == MI_NERF_MODE_F16S;
    const bool coarse_bf16 = mode_is_bf16(cfg->mode);
    const bool fine_bf16 = mode_is_bf16(cfg->mode) || cfg->mode == MI_NERF_MODE_F16S_BF16;
    FineDraw fd{};
    if (coarse_bf16) {
        // the bf16 kernel draws the stratified depths in its own prologue and writes z_c (one launch fewer: at a 512-ray shard a
        // launch is ~4 us of a ~130 us step)
        // ... and, for a small shard (one 32-point unit per wave: <= 512 rays on 256 CUs), render_rays' middle as well: `fd.taken`
        const StratDraw sd{cfg->near_, cfg->far_, t_rand, cfg->seed, cfg->ray_offset, z_c};
        if (cfg->Nf > 0) {
            MN_CHECK_ARG(packed_f && rgb_f && disp_f, "fine pass needs packed_fine and outputs");
            fd = FineDraw{cfg->Nf, cfg->det, u, cfg->seed, cfg->ray_offset, rgb_c, disp_c, wts_c, (float*)(w + L.z_f), false};
        }
        if (int rc = mlp_rays_bf16(net, packed_c, rays, nullptr, n, Sc, raw_c, st, ppw, &sd, cfg->Nf > 0 && ppw == 0 ? &fd : nullptr)) return rc;
    } else {
        if (int rc = stage_stratified(n, Sc, cfg->near_, cfg->far_, t_rand, cfg->seed, cfg->ray_offset, z_c, st)) return rc;
        if (int rc = coarse_f16s ? mlp_rays_f16s(net, packed_c, rays, z_c, n, Sc, raw_c, st) : mlp_rays_fp32(net, packed_c, rays, z_c, n, Sc, raw_c, st)) return rc;
    }
    if (cfg->Nf == 0) return stage_composite(raw_c, z_c, rays, 6, n, Sc, rgb_c, disp_c, nullptr, wts_c, nullptr, st);
    {
        // 3-a) + 1-b) composite, resample + merge in one launch; 2-b) fine net over all Sc+Nf depths; 3-b) composite   (:198-213)
        float* z_f = (float*)(w + L.z_f);
        float* raw_f = (float*)(w + L.raw_f);
        if (!fd.taken)        // (a small bf16 coarse launch has done this in its epilogue)
            if (int rc = stage_composite_fine_z(raw_c, z_c, rays, n, Sc, cfg->Nf, cfg->det, u, cfg->seed, cfg->ray_offset, rgb_c, disp_c, wts_c, z_f, st))
                return rc;
        if (int rc = fine_f16s ? mlp_rays_f16s(net, packed_f, rays, z_f, n, St, raw_f, st)
                          : fine_bf16 ? mlp_rays_bf16(net, packed_f, rays, z_f, n, St, raw_f, st, ppw, nullptr)
                                      : mlp_rays_fp32(net, packed_f, rays, z_f, n, St, raw_f, st)) return rc;
        if (int rc = stage_composite(raw_f, z_f, rays, 6, n, St, rgb_f, disp_f, nullptr, nullptr, nullptr, st)) return rc;
    }
    return MI_NERF_OK;
}

int mi_nerf_time_mlp_rays(const mi_nerf_net* net, const void* packed, const float* rays, const float* z, int64_t n_rays, int S,
                          float* raw, int iters, int mode, float* avg_ms, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    MN_CHECK_ARG(iters >= 1 && avg_ms, "bad iters / NULL output");
    MN_CHECK_ARG(mode >= MI_NERF_MODE_F32 && mode <= MI_NERF_MODE_F16S, "mode must be MI_NERF_MODE_F32 .. MI_NERF_MODE_F16S: one network, one kernel family (got %d)", mode);
    hipEvent_t e0, e1;
    MN_HIP(hipEventCreate(&e0));
    MN_HIP(hipEventCreate(&e1));
    int rc = MI_NERF_OK;
    MN_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters && rc == MI_NERF_OK; ++i)
        rc = mode == MI_NERF_MODE_F16S ? mlp_rays_f16s(net, packed, rays, z, n_rays, S, raw, st)
             : mode ? mlp_rays_bf16(net, packed, rays, z, n_rays, S, raw, st, bf16_points_per_wave(mode), nullptr)
                        : mlp_rays_fp32(net, packed, rays, z, n_rays, S, raw, st);
    MN_HIP(hipEventRecord(e1, st));
    MN_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MN_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)iters;
    return rc;
}

size_t mi_nerf_wgrad_scratch_bytes(void) { return wgrad_scratch_bytes(); }
// launches `body()` iters times on `st`; *avg_ms (optional) = average device time by hipEvents on that stream
extern "C++" template <typename F>
static int timed_launches(int iters, float* avg_ms, hipStream_t st, F body) {
    MN_CHECK_ARG(iters >= 1, "bad iters");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (avg_ms) {
        MN_HIP(hipEventCreate(&e0));
        MN_HIP(hipEventCreate(&e1));
        MN_HIP(hipEventRecord(e0, st));
    }
    int rc = MI_NERF_OK;
    for (int i = 0; i < iters && rc == MI_NERF_OK; ++i) rc = body();
    if (avg_ms) {
        MN_HIP(hipEventRecord(e1, st));
        MN_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        MN_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms = ms / (float)iters;
    }
    return rc;
}
int mi_nerf_wgrad_products(int n, const float* const* delta, const int* ldd, const int* M, const float* const* x, const int* ldx, const int* N,
                           int64_t P, float* const* out, const int* ldo, float* const* bias, void* scratch, size_t scratch_bytes, int iters,
                           float* avg_ms, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    return timed_launches(iters, avg_ms, st, [&] { return wgrad_products(n, delta, ldd, M, x, ldx, N, P, out, ldo, bias, scratch, scratch_bytes, st, false); });
}
int mi_nerf_wgrad_products_f16s(int n, const float* const* delta, const int* ldd, const int* M, const float* const* x, const int* ldx, const int* N,
                                int64_t P, float* const* out, const int* ldo, float* const* bias, void* scratch, size_t scratch_bytes, int iters,
                                float* avg_ms, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    return timed_launches(iters, avg_ms, st, [&] { return wgrad_products(n, delta, ldd, M, x, ldx, N, P, out, ldo, bias, scratch, scratch_bytes, st, true); });
}

int mi_nerf_selftest_mfma(void* stream) {
    hipStream_t st = (hipStream_t)stream;
    float hA[32 * 8], hB[8 * 32], hD[32 * 32], ref[32 * 32];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 8; ++k) hA[i * 8 + k] = (float)((i * 3 + k * 7) % 11 - 5);
    for (int k = 0; k < 8; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 5 + j * 2 + (j > 9)) % 13 - 6);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0.f;
        for (int k = 0; k < 8; ++k) s += hA[i * 8 + k] * hB[k * 32 + j];
        ref[i * 32 + j] = s;
    }
    float *dA, *dB, *dD;
    MN_HIP(hipMalloc(&dA, sizeof(hA)));
    MN_HIP(hipMalloc(&dB, sizeof(hB)));
    MN_HIP(hipMalloc(&dD, sizeof(hD)));
    MN_HIP(hipMemcpyAsync(dA, hA, sizeof(hA), hipMemcpyHostToDevice, st));
    MN_HIP(hipMemcpyAsync(dB, hB, sizeof(hB), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, st, dA, dB, dD);
    MN_LAUNCH_CHECK("mfma_selftest_kernel");
    MN_HIP(hipMemcpyAsync(hD, dD, sizeof(hD), hipMemcpyDeviceToHost, st));
    MN_HIP(hipStreamSynchronize(st));
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
    for (int i = 0; i < 32 * 32; ++i)
        if (hD[i] != ref[i]) {
            set_error("MFMA layout mismatch at D[%d][%d]: got %g want %g", i / 32, i % 32, hD[i], ref[i]);
            return MI_NERF_EINVAL;
        }
    return MI_NERF_OK;
}

}  // extern "C"
