/*
 * mi_nerf.h -- C ABI of libmi_nerf.so: the MI355X (gfx950) NeRF volume-rendering hot path.
 *
 * The reference (nuggy875/NeRF_pytorch_paeng) has NO native/FFI layer: its hot path is the Python
 * function surface of nerf_process.py / rays.py / model/ (SURVEY.md section 8(b)).  This header is the
 * flat boundary a binding for that surface talks to: raw device pointers (tensor.data_ptr()), explicit
 * sizes, scalars, a hipStream_t passed as void*, int status return (0 = ok).  No torch types, no
 * exceptions, no hidden allocation: the caller allocates every output.  Error text: mi_nerf_last_error().
 *
 * Each entry point cites the reference code it replaces (file:line relative to the reference tree).
 * All tensors are fp32, contiguous, row-major unless stated.  "dev" = device pointer, "host" = host pointer.
 */
#ifndef MI_NERF_H
#define MI_NERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_NERF_ABI_VERSION 4   /* 2: mi_nerf_render_cfg grew seed / reserved / ray_offset (in-kernel jitter)
                                   3: mi_nerf_wgrad_product removed, mi_nerf_wgrad_products takes narrow products and needs
                                      mi_nerf_wgrad_scratch_bytes(); the RCCL tile-gather helpers (mi_nerf_comm_*, mi_nerf_all_gather_tiles)
                                   4: mi_nerf_render_cfg.use_bf16 is called what it is, `mode` (MI_NERF_MODE_*; same offset, same values: a v3
                                      caller's struct is read unchanged, and C11 / C++ callers keep the old member name for this one version);
                                      MI_NERF_MODE_F16S_BF16; mi_nerf_permute_rows documented as the row gather it is (n = rows of dst) */

/* Precision mode of an MLP launch (mi_nerf_render_cfg.mode, mi_nerf_time_mlp_rays): which kernel family evaluates the networks, and
 * therefore which packer made the blobs handed in. */
#define MI_NERF_MODE_F32 0          /* fp32 MFMA (default): blobs of mi_nerf_pack_weights                                              */
#define MI_NERF_MODE_BF16 1         /* bf16 MFMA, launch shape chosen per launch: blobs of mi_nerf_pack_weights_bf16                   */
#define MI_NERF_MODE_BF16_64 2      /* ... 64 points per wave pinned (A/B measurements)                                               */
#define MI_NERF_MODE_BF16_32 3      /* ... 32 points per wave pinned                                                                  */
                                    /* 4: retired (an 8-wave launch shape that was measured and removed); refused                      */
#define MI_NERF_MODE_F16S 5         /* f16 split precision, fp32-grade results: blobs of mi_nerf_pack_weights_f16s                     */
#define MI_NERF_MODE_F16S_BF16 6    /* mi_nerf_render_rays only: COARSE network in f16 split precision (packed_coarse from
                                       mi_nerf_pack_weights_f16s), FINE network in bf16 (packed_fine from mi_nerf_pack_weights_bf16):
                                       the fine sample positions come out fp32-grade, the 3/4 of the evaluations that the fine network
                                       makes run at the bf16 rate                                                                     */

/* status codes */
#define MI_NERF_OK 0
#define MI_NERF_EINVAL 1   /* bad argument / unsupported shape */
#define MI_NERF_EHIP 2     /* HIP runtime error (launch, etc.) */
#define MI_NERF_ERCCL 3    /* RCCL not loadable, or an RCCL call failed (mi_nerf_comm_*, mi_nerf_all_gather_tiles only) */

int mi_nerf_abi_version(void);
/* Thread-local text of the last error on this thread ("" if none). */
const char* mi_nerf_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Network description.  One NeRFModule: D trunk layers of width W, skip-concat of the encoded
 * position after trunk layer `skip` (so trunk layer skip+1 takes [gamma(x), h]), density / feature /
 * view-direction / colour heads (model/NeRF.py:10-52).  Supported: fp32 inference 2 <= W <= 512 -- the kernels are instantiated for
 * W = 128, 256 (32 points per wave) and 384, 512 (16 points per wave), and mi_nerf_pack_weights / mi_nerf_pack_map lay a network of any other width out for the next of them, with zero
 * weights and biases for the hidden units it does not have (exactly 0 through the ReLU: the W-wide network's result at the padded width's
 * cost; linear_d is W // 2 wide, model/NeRF.py:28); the training path W in {128, 256}; the bf16 and split-precision variants W = 256;
 * 2 <= D <= 16, L_x <= 10, L_d <= 4 (the kernels evaluate gamma_10 / gamma_4; gamma_L is a prefix of them in the reference's
 * channel order, PositionalEncoding.py:18-24, so a network with fewer frequencies is packed with zero weights on the rest and
 * gives the same result), at most one skip (skip = -1: none; a skip index >= D-1 never fires, exactly like the reference's
 * `range(D-1)` construction at model/NeRF.py:25).  Pre-embedded rows (mi_nerf_mlp_embedded) have the network's own width.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mi_nerf_net {
    int32_t D;      /* trunk depth             (opts.netDepth, config.py:56) */
    int32_t W;      /* trunk width             (opts.netWidth, config.py:57) */
    int32_t skip;   /* skips=[4] -> 4; -1 none (model/NeRF.py:11,25,40)      */
    int32_t L_x;    /* position frequencies    (config.py:54) -> 3+6*L_x input channels */
    int32_t L_d;    /* direction frequencies   (config.py:55) */
} mi_nerf_net;

/* Host pointers to one NeRFModule's parameters in the reference checkpoint layout
 * ([out,in] row-major weights; model/NeRF.py:24-30).  linear_x_w / linear_x_b are arrays of D pointers. */
typedef struct mi_nerf_params {
    const float* const* linear_x_w;   /* [D] : W x (in_x | W | W+in_x) */
    const float* const* linear_x_b;   /* [D] : W                        */
    const float* linear_density_w;    /* 1 x W     */
    const float* linear_density_b;    /* 1         */
    const float* linear_feat_w;       /* W x W     */
    const float* linear_feat_b;       /* W         */
    const float* linear_d_w;          /* W/2 x (W + in_d), input order [feature, gamma(d)] (NeRF.py:46) */
    const float* linear_d_b;          /* W/2       */
    const float* linear_color_w;      /* 3 x W/2   */
    const float* linear_color_b;      /* 3         */
} mi_nerf_params;

/* Packed-weights blob: the kernels' streaming order (MFMA A-operand fragments in consumption order,
 * then bias / head side tables).  Pack on the host, copy the blob to the device once, pass the device
 * pointer to the mlp / render entry points.  Replaces nothing in the reference (state_dict ingest:
 * train.py:105-114, test.py:20-21 load the same tensors). */
size_t mi_nerf_packed_bytes(const mi_nerf_net* net);
int mi_nerf_pack_weights(const mi_nerf_net* net, const mi_nerf_params* params, void* host_blob, size_t blob_bytes);
/* dtype 1: bf16 stream for the bf16 MFMA variant (fp32 side tables). */
size_t mi_nerf_packed_bytes_bf16(const mi_nerf_net* net);
int mi_nerf_pack_weights_bf16(const mi_nerf_net* net, const mi_nerf_params* params, void* host_blob, size_t blob_bytes);

/* ------------------------------------------------------------------------------------------------
 * a1  make_o_d(img_w, img_h, img_k, pose)                                           rays.py:20-34
 * Rays for image rows [row0, row0+n_rows) (whole image: 0, H).  k4 = {fx, fy, cx, cy} already rounded
 * to fp32 (torch rounds the float64 K entries to fp32 when they meet the fp32 pixel grid); pose12 =
 * row-major 3x4 camera-to-world.  Outputs [n_rows*W, 3] each; rays_o_dev may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int mi_nerf_make_o_d(int W, int H, const float k4[4], const float pose12[12], int row0, int n_rows,
                     float* rays_o_dev, float* rays_d_dev, void* stream);
/* Same maths for an explicit list of flat pixel indices (int64, y*W+x): the 4096-ray benchmark batch. */
int mi_nerf_make_o_d_pixels(int W, int H, const float k4[4], const float pose12[12], const int64_t* pix_dev,
                            int64_t n, float* rays_o_dev, float* rays_d_dev, void* stream);

/* a3  ndc_rays(H, W, focal, near, rays_o, rays_d)                          nerf_process.py:8-28
 * o_in/d_in: [n,3] with arbitrary row strides in floats (stride 0 = broadcast origin, rays.py:33). */
int mi_nerf_ndc_rays(int H, int W, float focal, float near_, const float* o_in_dev, int64_t o_stride,
                     const float* d_in_dev, int64_t d_stride, int64_t n, float* o_out_dev, float* d_out_dev,
                     void* stream);

/* Counter-based U[0,1) generator: out[r, s] depends only on (seed, stream_id, ray0 + r, s), so any
 * sharding/chunking of the rays sees the same numbers.  Stands in for torch.rand at
 * nerf_process.py:58 (stream_id 0: t_rand) and :162 (stream_id 1: u). */
int mi_nerf_fill_uniform(uint32_t seed, uint32_t stream_id, int64_t ray0, int64_t n_rays, int n_samples,
                         float* out_dev, void* stream);

/* a6  stratified coarse depths                                             nerf_process.py:42-60
 * z[n, S] = lower + (upper - lower) * t_rand, bins linear in depth between near and far. */
int mi_nerf_stratified_z(int64_t n_rays, int S, float near_, float far_, const float* t_rand_dev, float* z_dev,
                         void* stream);

/* a7  sample_pdf(bins, weights, N_samples, det)                          nerf_process.py:144-182
 * bins [n, B], weights [n, B-1], u [n, N] (ignored / may be NULL when det != 0) -> samples [n, N]. */
int mi_nerf_sample_pdf(const float* bins_dev, const float* weights_dev, int64_t n, int B, int N, int det,
                       const float* u_dev, float* samples_dev, void* stream);
/* a7  fine branch of pre_process                                          nerf_process.py:62-67
 * z_c [n, Sc] (sorted), weights_c [n, Sc] -> z_fine [n, Sc+Nf] = sort(cat(z_c, sample_pdf(mid(z_c),
 * weights_c[1:-1], Nf))); z_samples_dev (optional, [n, Nf]) receives the unsorted new samples.
 * NaN depths (NaN weights) are placed last, as torch.sort does.  Size limit (one wave's LDS slice): 2 (Sc - 1) + the next
 * power of two >= Sc + Nf must not exceed 4096 floats; larger sample counts are refused with MI_NERF_EINVAL. */
int mi_nerf_fine_z(const float* z_c_dev, const float* weights_c_dev, int64_t n, int Sc, int Nf, int det,
                   const float* u_dev, float* z_fine_dev, float* z_samples_dev, void* stream);

/* a8  network input assembly                                    nerf_process.py:36-39,69-85
 * rays [n,6] (o,d), z [n,S] -> embedded [n*S, (3+6L_x)+(3+6L_d)] = [gamma(o+d z), gamma(d/|d|)]
 * (model/PositionalEncoding.py:7-36). */
int mi_nerf_embed(const float* rays_dev, const float* z_dev, int64_t n_rays, int S, int L_x, int L_d,
                  float* embedded_dev, void* stream);

/* gamma(x): x [n,3] -> [n, 3+6L]  (the closure of get_positional_encoder, model/PositionalEncoding.py:33-36) */
int mi_nerf_posenc(const float* x_dev, int64_t n, int L, float* out_dev, void* stream);

/* a9  model(x)                                    model/NeRF.py:33-52,70-78; nerf_process.py:190-192
 * x [n, in_x+in_d] pre-embedded rows -> out [n,4] = (rgb_raw, density_raw). */
int mi_nerf_mlp_embedded(const mi_nerf_net* net, const void* packed_dev, const float* x_dev, int64_t n,
                         float* out_dev, void* stream);
/* a8+a9 fused: positional encoding computed in registers, no [n_pts, 90] tensor in HBM.
 * rays [n,6], z [n,S] -> raw [n,S,4]. */
int mi_nerf_mlp_rays(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev,
                     int64_t n_rays, int S, float* raw_dev, void* stream);
/* bf16-MFMA variant of the fused entry (packed blob from mi_nerf_pack_weights_bf16). */
int mi_nerf_mlp_rays_bf16(const mi_nerf_net* net, const void* packed_bf16_dev, const float* rays_dev,
                          const float* z_dev, int64_t n_rays, int S, float* raw_dev, void* stream);
/* The same with the launch shape pinned: points_per_wave 64 (standard: 256 points per workgroup and pass of the weight
 * stream), 32 (small-launch shape: 128) or 0 (chosen per launch, as mi_nerf_mlp_rays_bf16 does: the 32-point shape only
 * where the 64-point one would leave SIMDs idle, e.g. a 512-ray shard).  Results are identical bit for bit. */
int mi_nerf_mlp_rays_bf16_shape(const mi_nerf_net* net, const void* packed_bf16_dev, const float* rays_dev,
                                const float* z_dev, int64_t n_rays, int S, float* raw_dev, int points_per_wave, void* stream);

/* SPLIT-PRECISION variant of the fused entry: fp32-grade results on the f16 matrix pipe (gfx950 has no xf32 / TF32; its f32-input MFMA
 * runs at 1/16 of the f16 rate).  Weights and activations travel as f16 pairs x = hi + lo * 2^-11, a product is three
 * v_mfma_f32_16x16x32_f16 with fp32 accumulation (hi.hi, hi.lo, lo.hi; the dropped lo.lo term is 2^-22 of the product): the error
 * against an fp64 evaluation is that of the fp32 kernel.  W = 256; |weights| and |activations| must stay STRICTLY below the f16 maximum
 * (65 504; both packers refuse / count |w| >= 65 504 and NaN).  Range contract: the host packer refuses larger weights (the device packer counts them: out_of_range_dev below); an
 * ACTIVATION at or beyond 65 520 comes out as NaN in every output that depends on it -- all four raw values of the point for a trunk
 * unit, the three colours for a linear_feat / linear_d unit -- never as a finite value (the kernels' ReLU is the NaN-propagating
 * maximum); a pre-activation <= -65 520 in front of a ReLU is exact (the unit is off, as in fp32).  An extra precision mode like the
 * bf16 variant, not the default path. */
size_t mi_nerf_packed_bytes_f16s(const mi_nerf_net* net);
int mi_nerf_pack_weights_f16s(const mi_nerf_net* net, const mi_nerf_params* params, void* host_blob, size_t blob_bytes);
int mi_nerf_mlp_rays_f16s(const mi_nerf_net* net, const void* packed_f16s_dev, const float* rays_dev, const float* z_dev,
                          int64_t n_rays, int S, float* raw_dev, void* stream);
/* ... as the TRAINING forward (replaces mi_nerf_mlp_rays_train, i.e. model/NeRF.py:33-52 with the graph autograd would record): the same
 * outputs plus the same activation stash in the same layouts, so mi_nerf_mlp_backward runs unchanged behind it.  The weights change
 * every step: mi_nerf_pack_apply_f16s re-packs on the device from the flat parameter vector (module.parameters() order) through a
 * gather map built once (mi_nerf_pack_map_f16s; mi_nerf_pack_map_f16s_len entries).  out_of_range_dev (may be NULL): incremented once
 * per stream element whose weight is NaN or beyond the f16 range -- what the host packer refuses. */
int mi_nerf_mlp_rays_train_f16s(const mi_nerf_net* net, const void* packed_f16s_dev, const float* rays_dev, const float* z_dev,
                                int64_t n_rays, int S, float* raw_dev, void* stash_dev, size_t stash_bytes, void* stream);
size_t mi_nerf_pack_map_f16s_len(const mi_nerf_net* net);
int mi_nerf_pack_map_f16s(const mi_nerf_net* net, int32_t* map_host, size_t map_len);
int mi_nerf_pack_apply_f16s(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_dev, void* blob_dev, size_t blob_bytes,
                            uint32_t* out_of_range_dev, void* stream);
/* the TRANSPOSED weights for the split-precision backward-data chain (mi_nerf_mlp_backward_mode, mode bit 1): host packer, and the
 * device-side re-pack through a gather map like the forward blob's */
size_t mi_nerf_packed_bytes_bwd_f16s(const mi_nerf_net* net);
int mi_nerf_pack_weights_bwd_f16s(const mi_nerf_net* net, const mi_nerf_params* params, void* host_blob, size_t blob_bytes);
size_t mi_nerf_pack_map_bwd_f16s_len(const mi_nerf_net* net);
int mi_nerf_pack_map_bwd_f16s(const mi_nerf_net* net, int32_t* map_host, size_t map_len);
int mi_nerf_pack_apply_bwd_f16s(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_dev, void* blob_dev, size_t blob_bytes,
                                uint32_t* out_of_range_dev, void* stream);

/* a10 post_process(outputs, z_vals, rays_d)                              nerf_process.py:89-140
 * raw [n,S,4], z [n,S], rays [n, ray_stride] with the direction at floats 3..5 when ray_stride == 6, or a
 * bare [n,3] direction tensor when ray_stride == 3.  Any of acc/weights/depth may be NULL. */
int mi_nerf_composite(const float* raw_dev, const float* z_dev, const float* rays_dev, int ray_stride, int64_t n,
                      int S, float* rgb_dev, float* disp_dev, float* acc_dev, float* weights_dev,
                      float* depth_dev, void* stream);

/* a5  render_rays(rays, model, posenc, opts)                            nerf_process.py:185-216
 * Whole coarse(+fine) pipeline for n rays on one stream, no host synchronisation.
 * Randomness (the reference draws unseeded torch.rand inside pre_process / sample_pdf, nerf_process.py:58-60,162-163): pass
 * t_rand [n,Sc] / u [n,Nf] to inject explicit uniforms, or NULL to have the consuming kernels draw them from the counter-based
 * generator keyed on (cfg->seed, cfg->ray_offset + ray index, sample index) -- exactly the values mi_nerf_fill_uniform(seed,
 * stream 0 / 1, ray_offset, ...) writes, without the tensors.  u is ignored when det != 0.
 * Launches: stratified depths (bf16: drawn in the coarse net kernel) | coarse net | composite + resample + merge (bf16, at most four rays
 * per CU and 33..64 coarse samples: done in the coarse net kernel's epilogue by the waves that own the rays) | fine net | composite.
 * Workspace (caller-allocated, mi_nerf_render_workspace_bytes): z_c, raw_c, weights_c, z_f, raw_f.
 * Outputs: rgb_c [n,3], disp_c [n]; rgb_f [n,3], disp_f [n] when Nf > 0 (else may be NULL). */
typedef struct mi_nerf_render_cfg {
    float near_, far_;     /* opts.near / opts.far   (nerf_process.py:44,47) */
    int32_t Sc, Nf;        /* opts.N_samples_c / _f  (config.py:72-73)       */
    int32_t det;           /* opts.perturb == 0.     (nerf_process.py:65)    */
#if defined(__cplusplus) || (defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L)
    union {
        int32_t mode;      /* MI_NERF_MODE_*                                                        */
        int32_t use_bf16;  /* the member's name up to ABI 3 (same storage); goes away with ABI 5    */
    };
#else
    int32_t mode;          /* MI_NERF_MODE_*  (C99: no anonymous union, so no alias for the old name) */
#endif
    uint32_t seed;         /* in-kernel jitter (t_rand / u NULL): generator seed ...             */
    uint32_t reserved;     /* must be 0                                                          */
    int64_t ray_offset;    /* ... and the GLOBAL index of ray 0 (chunk / shard invariant frames) */
} mi_nerf_render_cfg;
size_t mi_nerf_render_workspace_bytes(const mi_nerf_render_cfg* cfg, int64_t n_rays);
int mi_nerf_render_rays(const mi_nerf_net* net, const void* packed_coarse_dev, const void* packed_fine_dev,
                        const mi_nerf_render_cfg* cfg, const float* rays_dev, int64_t n_rays,
                        const float* t_rand_dev, const float* u_dev, void* workspace_dev, size_t workspace_bytes,
                        float* rgb_c_dev, float* disp_c_dev, float* rgb_f_dev, float* disp_f_dev, void* stream);
/* Offsets (bytes) of the intermediates inside the workspace, for staged parity checks. */
typedef struct mi_nerf_workspace_layout {
    size_t z_c, raw_c, weights_c, z_f, raw_f, total;
} mi_nerf_workspace_layout;
int mi_nerf_render_workspace_layout(const mi_nerf_render_cfg* cfg, int64_t n_rays, mi_nerf_workspace_layout* out);

/* ------------------------------------------------------------------------------------------------
 * Training path: what `loss.backward()` does for the hot path in
 * train.py:53-70.  Only rgb_map carries gradient (the loss reads pred_rgb_c / pred_rgb_f); depths are
 * constants (coarse: no graph; fine: detached at nerf_process.py:66).
 * ---------------------------------------------------------------------------------------------- */
/* backward of mi_nerf_composite w.r.t. raw: d_rgb [n,3] -> d_raw [n,S,4]   (autograd of nerf_process.py:89-140) */
int mi_nerf_composite_backward(const float* raw_dev, const float* z_dev, const float* rays_dev, int ray_stride,
                               int64_t n, int S, const float* d_rgb_dev, float* d_raw_dev, void* stream);

/* Flat parameter vector of one NeRFModule, in module.parameters() order of model/NeRF.py:24-30:
 * linear_x[0..D).{weight,bias}, linear_d.{weight,bias}, linear_feat, linear_density, linear_color
 * (each weight [out,in] row-major).  Gradients come back in the same order. */
size_t mi_nerf_param_count(const mi_nerf_net* net);

/* Backward-data blob: the transposed weights in the order the backward chain consumes them (host packer). */
size_t mi_nerf_packed_bytes_bwd(const mi_nerf_net* net);
int mi_nerf_pack_weights_bwd(const mi_nerf_net* net, const mi_nerf_params* params, void* host_blob, size_t blob_bytes);
/* Device-side re-pack (training re-packs after every optimizer.step(), train.py:70): a gather map, built once on
 * the host, from the flat parameter vector to a blob.  kind 0: forward blob (mi_nerf_packed_bytes), 1: backward-data
 * blob (mi_nerf_packed_bytes_bwd).  map_len = blob bytes / 4 entries; entry = 1 + flat index, 0 = constant zero. */
int mi_nerf_pack_map(const mi_nerf_net* net, int kind, int32_t* map_host, size_t map_len);
int mi_nerf_pack_apply(const int32_t* map_dev, const float* flat_params_dev, size_t blob_bytes, void* blob_dev, void* stream);
/* The same for the bf16 blob (mi_nerf_packed_bytes_bf16): one map entry per bf16 stream element followed by one per fp32
 * side-table float (mi_nerf_pack_map_bf16_len entries); the apply kernel rounds the stream to bf16 (nearest even) and writes
 * the blob header.  A model whose parameters live on the device is packed without a host round trip (weights.py). */
size_t mi_nerf_pack_map_bf16_len(const mi_nerf_net* net);
int mi_nerf_pack_map_bf16(const mi_nerf_net* net, int32_t* map_host, size_t map_len);
int mi_nerf_pack_apply_bf16(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_params_dev, void* blob_dev,
                            size_t blob_bytes, void* stream);

/* Activation stash written by the training forward and the scratch the backward needs (byte offsets). */
typedef struct mi_nerf_train_layout {
    size_t stash_h, stash_f, stash_g;                            /* [D][P][W], [P][W], [P][W/2] post-activation rows, P = n_rays*S */
    size_t mask_h, mask_g, stash_bytes;                          /* ReLU' bit masks in the kernels' (ray, 32-sample tile, lane) order */
    size_t delta_h, delta_f, delta_d, emb, partial, work_bytes;  /* pre-activation gradients, encoded inputs, wgrad partials */
} mi_nerf_train_layout;
int mi_nerf_train_layout_query(const mi_nerf_net* net, int64_t n_rays, int S, mi_nerf_train_layout* out);

/* Training forward: mi_nerf_mlp_rays that also keeps every layer's activations (model/NeRF.py:33-52 with the
 * autograd graph the reference builds implicitly).  stash: mi_nerf_train_layout.stash_bytes for (n_rays, S). */
int mi_nerf_mlp_rays_train(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev,
                           int64_t n_rays, int S, float* raw_dev, void* stash_dev, size_t stash_bytes, void* stream);
/* Backward of the above: d_raw [n_rays*S, 4] -> grads [mi_nerf_param_count] (overwritten, not accumulated).
 * stage 0: full backward; 1: backward-data only (per-layer deltas left in work, for staged parity checks). */
int mi_nerf_mlp_backward(const mi_nerf_net* net, const void* packed_dev, const void* packed_bwd_dev, const float* rays_dev,
                         const float* z_dev, int64_t n_rays, int S, const float* d_raw_dev, const void* stash_dev,
                         void* work_dev, size_t work_bytes, float* grads_dev, int stage, void* stream);
/* ... with a mode (bits may be combined; 0 is mi_nerf_mlp_backward): bit 1 = the backward-data chain in SPLIT PRECISION
 * (dgrad_f16s_kernel: packed_bwd_dev is then the blob of mi_nerf_pack_weights_bwd_f16s / mi_nerf_pack_apply_bwd_f16s, W = 256);
 * bit 0 = the nine W-wide weight-gradient products of a network in SPLIT PRECISION (operands converted on the fly to
 * f16 hi + lo pairs, three f16 MFMAs per product into one fp32 accumulator, the gradient operand scaled by a power of two taken from
 * max|d_raw| on the device): fp32-grade gradients, the products bound by their HBM reads instead of the fp32 matrix rate.  mode 0 is
 * mi_nerf_mlp_backward. */
int mi_nerf_mlp_backward_mode(const mi_nerf_net* net, const void* packed_dev, const void* packed_bwd_dev, const float* rays_dev,
                              const float* z_dev, int64_t n_rays, int S, const float* d_raw_dev, const void* stash_dev, void* work_dev,
                              size_t work_bytes, float* grads_dev, int stage, int mode, void* stream);

/* The same pair for pre-embedded rows x [n, in_x + in_d]: what autograd does when model(embedded, is_fine) is called directly
 * with gradients enabled, as the reference's own render_rays does (nerf_process.py:190-192,206-207; model/NeRF.py:70-78) --
 * for callers that keep the reference's pipeline and swap only the model.  Buffers: mi_nerf_train_layout_query(net,
 * ceil(n/32), 32).  d_out [n,4] -> grads [mi_nerf_param_count]; nothing is differentiated w.r.t. x. */
int mi_nerf_mlp_embedded_train(const mi_nerf_net* net, const void* packed_dev, const float* x_dev, int64_t n, float* out_dev,
                               void* stash_dev, size_t stash_bytes, void* stream);
int mi_nerf_mlp_embedded_backward(const mi_nerf_net* net, const void* packed_dev, const void* packed_bwd_dev, const float* x_dev,
                                  int64_t n, const float* d_out_dev, const void* stash_dev, void* work_dev, size_t work_bytes,
                                  float* grads_dev, void* stream);

/* Weight-gradient products of the backward pass on their own: n (1..12) products over the SAME P points,
 * out[b][M_b, ldo_b] = delta[b][P, ldd_b]^T x[b][P, ldx_b] (first M_b / N_b columns), bias[b][M_b] = column sums of delta[b] -- what
 * autograd computes for one nn.Linear each (model/NeRF.py:24-30).  Products with both sides wider than 64 columns (at most 256) share
 * ONE launch -- the form the backward pass itself uses for the nine 256 x 256 products of an 8 x 256 network: the CUs are shared out
 * between them, so each is cut into (CUs / n) point slices and writes / reduces 1 / n of the partial sums a launch of its own does
 * (nine 256 x 256 products over 786 432 points: 0.925 of the fp32 MFMA peak; n = 1: 0.78, which is why there is no single-product
 * entry any more -- round 4).  A product with a side of at most 64 columns (gamma(x), gamma(d), the heads) runs in a launch of its
 * own behind the wide batch.  Operands wider than 64 columns must be 16-byte aligned with pitches of 4 floats; no operand is read outside
 * its P rows (the load pipelines run ahead of the data through range-checked buffer loads: requests past row P - 1 return zeros without
 * touching memory).  Arrays of n HOST entries (device pointers, pitches, sizes); bias_dev may be NULL, or hold NULL entries.
 * scratch: mi_nerf_wgrad_scratch_bytes() bytes.  iters launches back to back; avg_ms_out (may be NULL) = their average device time by
 * hipEvents on `stream` (bench.py's roofline leg for the training kernels; synchronises the stream when given). */
size_t mi_nerf_wgrad_scratch_bytes(void);
int mi_nerf_wgrad_products(int n, const float* const* delta_dev, const int* ldd, const int* M, const float* const* x_dev, const int* ldx,
                           const int* N, int64_t P, float* const* out_dev, const int* ldo, float* const* bias_dev, void* scratch_dev,
                           size_t scratch_bytes, int iters, float* avg_ms_out, void* stream);
/* ... in SPLIT PRECISION (wgrad_f16s_kernel: operands converted on the fly to f16 hi + lo pairs, fp32 accumulate, fp32-grade results; the
 * gradient operands are scaled by a power of two from their largest entry, found by one extra pass over them; ldd[b] == M[b]).  The products
 * are then bound by the HBM reads of their operands, not by the fp32 matrix rate.  The training step uses the same kernel through
 * mi_nerf_mlp_backward_mode. */
int mi_nerf_wgrad_products_f16s(int n, const float* const* delta_dev, const int* ldd, const int* M, const float* const* x_dev, const int* ldx,
                                const int* N, int64_t P, float* const* out_dev, const int* ldo, float* const* bias_dev, void* scratch_dev,
                                size_t scratch_bytes, int iters, float* avg_ms, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Either side of the path in the reference's callers.  HBM-bound streaming ops.
 * `scratch`: caller-allocated device buffer of at least MI_NERF_REDUCE_SCRATCH_BYTES.
 * ---------------------------------------------------------------------------------------------- */
#define MI_NERF_REDUCE_SCRATCH_BYTES 8192
/* img2mse + mse2psnr (utils.py:18-23; test.py:64-68): out2_dev = { mean((pred-target)^2), -10 log10(mse) } over n floats */
int mi_nerf_image_metrics(const float* pred_dev, const float* target_dev, int64_t n, float* out2_dev, void* scratch_dev,
                          size_t scratch_bytes, void* stream);
/* np.nanmax (test.py:56, test.py:157): maximum ignoring NaN (NaN if every element is NaN) */
int mi_nerf_nanmax(const float* x_dev, int64_t n, float* out_dev, void* scratch_dev, size_t scratch_bytes, void* stream);
/* to8b(x) or to8b(x / divisor_dev[0]) (utils.py:15; test.py:55-56): (255 * clip(v, 0, 1)).astype(uint8); divisor_dev may be NULL */
int mi_nerf_to8b(const float* x_dev, int64_t n, const float* divisor_dev, uint8_t* out_dev, void* stream);
/* Global-batch precompute (main.py:92-101): for n_img training images, rays_rgb [n_img*H*W, 3, 3] = (origin, direction,
 * pixel) per ray; get_rays_np (rays.py:7-17) for every pose in one launch.  poses [n_img,12] row-major 3x4 c2w,
 * images [n_img,H,W,3], k4 = {fx, fy, cx, cy}. */
int mi_nerf_rays_rgb(int W, int H, const float k4[4], const float* poses_dev, const float* images_dev, int64_t n_img,
                     float* rays_rgb_dev, void* stream);
/* Row gather by index: dst[i] = src[perm[i]] for i < n, rows of row_floats floats; perm int64, every value a row of src (not checked: the
 * caller's table).  With perm a permutation of all n rows this is np.random.shuffle(rays_rgb) (main.py:102; utils.py:47-52); with perm the
 * B indices of one step it is the batch rays_rgb[i_batch - B : i_batch] of a shuffled table that is never materialised (train.py:29). */
int mi_nerf_permute_rows(const float* src_dev, const int64_t* perm_dev, int64_t n, int row_floats, float* dst_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU frame assembly (SURVEY.md section 8(b) "thin RCCL helpers", 8(e)).  The reference is single-GPU (main.py:166-170): no
 * counterpart.  One process per GPU; every rank renders the contiguous row block
 *     rows(r) = H / world + (r < H % world),   row0(r) = r * (H / world) + min(r, H % world)
 * of an H x W frame (for a flat ray batch: H = rays, W = 1), and the ONLY exchange is one all-gather of the [rows(r) * W, C] fp32
 * tiles -- C = 4: rgb + disp, 1.28 MB per rank for 800 x 800 over 8 GPUs.  librccl is resolved at FIRST USE (dlopen of librccl.so.1:
 * the copy the process already holds -- PyTorch's -- else ROCm's; MI_NERF_RCCL_LIB overrides), so a one-GPU user needs no RCCL; when it
 * cannot be loaded these entries return MI_NERF_ERCCL.  The collective is enqueued on `stream` -- pass the stream the render ran on: no
 * host synchronisation separates the last composite launch from the gather.
 *   bootstrap: rank 0 calls mi_nerf_comm_unique_id and hands the MI_NERF_COMM_ID_BYTES to every rank by any channel (MPI, a TCP store,
 *   torch.distributed); each rank calls mi_nerf_comm_init_rank with its device current (hipSetDevice) -- collective over all ranks.
 * ---------------------------------------------------------------------------------------------- */
#define MI_NERF_COMM_ID_BYTES 128
int mi_nerf_rccl_available(void);            /* MI_NERF_OK when librccl is resolved (loads it on first call); else MI_NERF_ERCCL + the loader's text */
int mi_nerf_comm_unique_id(void* id_host);
int mi_nerf_comm_init_rank(const void* id_host, int world, int rank, void** comm_out);
int mi_nerf_comm_info(void* comm, int* world_out, int* rank_out, int* device_out);      /* any out pointer may be NULL */
int mi_nerf_comm_destroy(void* comm);
/* tile [rows_local * W, C] (rows_local must be rows(rank)) -> frame [H * W, C] on every rank.  H % world == 0: ONE ncclAllGather straight
 * into frame_dev (in place, no copy at all, when tile_dev == frame_dev + row0 * W * C: render into the frame); staging unused (may be
 * NULL).  Ragged split (fern's 378 rows over 8 ranks: 48, 48, 47 x 6): the tiles are padded to the largest block inside `staging`
 * (caller-allocated, mi_nerf_all_gather_staging_bytes, 16-byte aligned), gathered in place there and un-padded into frame_dev by one copy
 * kernel (mi_nerf_unpad_tiles, exported so that a one-GPU box can check the ragged geometry for any world size). */
size_t mi_nerf_all_gather_staging_bytes(int world, int H, int W, int C);                /* 0 when H % world == 0 */
int mi_nerf_all_gather_tiles(void* comm, const float* tile_dev, int rows_local, int H, int W, int C, float* frame_dev, void* staging_dev,
                             size_t staging_bytes, void* stream);
int mi_nerf_unpad_tiles(const float* staging_dev, int world, int H, int W, int C, float* frame_dev, void* stream);

/* Timing hook used by bench.py: average device time (ms) of `iters` back-to-back launches of the fused MLP
 * kernel on `stream`, measured with hipEvents recorded on that same stream (torch.cuda.Event only sees
 * torch's current stream).  Synchronises the stream. */
int mi_nerf_time_mlp_rays(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev,
                          int64_t n_rays, int S, float* raw_dev, int iters, int mode, float* avg_ms_out,
                          void* stream);

/* MFMA fragment-layout self test: runs a 32x32x(2k) product through the kernel's operand maps with
 * asymmetric integer data and compares on the host.  Returns 0 when the maps hold. */
int mi_nerf_selftest_mfma(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_NERF_H */
