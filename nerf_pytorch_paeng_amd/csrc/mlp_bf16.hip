// mlp_bf16.hip -- bf16-MFMA variant of the fused positional-encoding + NeRF MLP forward (BASELINE config #5).
//
// Same idea as mlp_fp32.hip -- activations never leave registers, the accumulator of layer l becomes the B operand of
// layer l+1, weights stream L2 -> LDS in consumption order -- rebuilt around what bf16 MFMA makes expensive: at 16x the
// fp32 MFMA rate a 256-wide layer is only 4096 matrix cycles per 32 points, so everything that is NOT an MFMA
// (accumulator -> bf16 packing, biases, gamma(x), heads), the weight stream and -- above all -- the POWER the chip
// has to spend per FLOP decide the speed.  (Round 1's kernel, with the fp32 kernel's k-outer order: 5.8 VALU
// instructions per MFMA, all exposed at the layer boundaries; 7.5 GB of L2 -> LDS weight traffic per launch; 42 % MFMA
// busy at 2.0 GHz -- profiles/r02_bf16_old_pmc*.json.)
//
//   * OUTPUT-TILE-MAJOR order ("t-outer").  A layer is 16 jobs; job t computes output features 16t..16t+15 over ALL
//     k-steps with ONE small accumulator per point tile.  Tiles 2s, 2s+1 of layer l are exactly B fragment s of layer
//     l+1, which that layer does not touch before its k-step s -- so the packing of job j's accumulators
//     (v_cvt_pk_bf16_f32 + ReLU as v_pk_max_i16 on the packed pair) is dealt out over the groups of job j+1, across
//     layer boundaries as well: there IS no layer boundary.  The bias is the C operand of a job's first MFMA.
//   * 64 POINTS PER WAVE (four point tiles of 16, one wave per SIMD): every A fragment (one ds_read_b128 per lane)
//     feeds four MFMAs (64 matrix cycles), so LDS reads and the L2 -> LDS stream are half of the 32-points-per-wave
//     design per FLOP (256 points per workgroup per pass over the 1.2 MB stream).  A second instantiation with 32 points
//     per wave (NP = 2) serves launches too small to give every SIMD a 64-point unit (see NP below).
//   * v_mfma_f32_16x16x32_bf16, not 32x32x16: the kernel is POWER limited (a build stripped to MFMAs + A-fragment
//     reads runs at ~1.7 GHz, profiles/r02_bf16_ablation.json), and at equal cycles per FLOP the chip holds a higher
//     clock on the 16x16x32 shape (tools/mfma_probe4.hip: +9 % FLOP/s for this operand pattern; MI355X_MICROARCH.md,
//     DVFS give-back item 7).  It also quarters the accumulator registers (4 per tile instead of 16).
//   * THE FRAGMENT FILE: both ping-pong sets of B fragments (2 x 4 point tiles x 8 fragments x 4 registers = all 256
//     AGPRs) are managed by hand with explicit register numbers; hipcc allocates only the VGPR side (see below).
//   * HEADS ON THE MATRIX PIPE.  Density (256 -> 1) is row 3 of an extra 16-row output tile over the trunk output (8
//     k-steps, right after the feature layer), colour (128 -> 3) rows 0..2 of another over the view-direction layer's
//     output (4 k-steps): (r, g, b) and the density land in registers 0..2 / 3 of lanes 0..15, one 16-byte store per
//     point; no VALU dot products, no cross-lane shuffles.  (+1 % MFMAs, -500 VALU per 32 points.)
//   * gamma(x) by ANGLE DOUBLING: one accurate sin/cos per axis (Cody-Waite + Cephes, as the fp32 kernel), then
//     s' = 2sc, c' = 1 - 2s^2 for the nine higher octaves.  The recurrence doubles the error per octave (<= 2^9 * 1e-7 =
//     5e-5 at the top octave), two orders below the bf16 rounding (2^-9 relative) the values get next.  The fp32
//     kernel keeps one full-precision evaluation per channel; this is the bf16 variant's own accuracy contract
//     (checked against an oracle with the same rounding points, and as PSNR against the fp32 path).
//
//   * SMALL COARSE LAUNCHES DO render_rays' MIDDLE THEMSELVES (round 6).  One 32-point unit per wave and 33..64 coarse samples = two units per
//     ray: a workgroup's four waves hold two rays whole, so its epilogue composites them and draws their fine depths (composite_ray +
//     fine_z_ray of stage_dev.h -- the stage kernel's own device functions: bit-identical) and mi_nerf_render_rays skips that launch:
//     -2.2 us of a ~125 us step at the 512-ray shard of an 8-GPU split (profiles/r06_bf16_fused_stages_ab.txt).  From 513 to 1024 rays a wave's
//     one 64-point unit IS a ray, and every wave does the middle for its own ray (-1.8 us of ~205 at 1024 rays).
//
//   A fragment: lane l (i = l&15, q = l>>4) holds A[i][k = 8q + j], j = 0..7  (8 bf16 = 16 B = one ds_read_b128)
//   B fragment: lane l holds B[k = 8q + j][col = l&15]
//   D (4 registers): col = l&15, row = 4q + r
// so the accumulators of output tiles 2s (elements j = 0..3) and 2s+1 (j = 4..7), packed pairwise, are the B fragment of
// k-step s whose element j on lane quarter q is feature 16(2s + (j>>2)) + 4q + (j&3); the weights are packed in that
// order on the host.  Encoded inputs: slot u = 32 ks + 8q + j is channel u of gamma(x) (zero weight beyond the last).
//
// Stream (1 KiB quads = the A fragment of FOUR MFMAs; 32 KiB slots, 3-slot ring, LDS-DMA two slots ahead):
//   layer 0:   for T in 0..15: 2 gamma(x) k-steps                                   32 quads
//   layer l:   for T in 0..15: 8 activation k-steps [2 gamma(x) k-steps if skip]    128 | 160 quads
//   tail:      feature layer (16 x 8) | density tile over the trunk output (8) | view-direction layer (8 x 8)
//              | colour tile over the view-direction output (4) | 20 quads of padding   224 quads
#include <string.h>
#include <type_traits>
#include <vector>
#include "common.h"
#include "layout.h"
#include "stage_dev.h"

namespace minerf {

// where the next unit's inputs are requested in the view-direction layer: (job, k-step)
constexpr int BF16_PF_T = 3, BF16_PF_KS = 2;

typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

// Point tiles (16 points each) per wave: 4 in the standard shape (64 points per wave, 256 per workgroup and pass of the weight
// stream), 2 in the SMALL-LAUNCH shape (32 / 128): twice the LDS reads and weight stream per FLOP, chosen by the launcher only
// where the 64-point shape would leave SIMDs idle (a 512-ray shard of BASELINE config #5's 8-GPU split: 128 + 384 workgroup
// passes on 256 CUs become 256 + 768 half-size ones).  NP is a template parameter of everything below.
constexpr int MT = 16;                                     // output features per job
constexpr int KF = 32;                                     // k per MFMA
constexpr int DA = 4;                                      // A-operand pipeline depth (fragments in flight)
constexpr int BSLOT_QUADS = 32;
constexpr int BSLOT_BYTES = BSLOT_QUADS * QUAD_BYTES;      // 32 KiB
constexpr int BNSLOT = 3;
constexpr int BRING_BYTES = BNSLOT * BSLOT_BYTES;
__host__ __device__ constexpr int bdma_of(int nwv) { return BSLOT_QUADS / nwv; }      // DMAs per wave per slot (NWV waves per workgroup)
constexpr int TAIL_USED = 128 + 8 + 64 + 4;                // quads of the tail body that carry weights
constexpr int TAIL_QUADS = 224;                            // ... padded to whole slots

__host__ __device__ constexpr int enc_ksteps32(int L) { return (3 + 6 * L + KF - 1) / KF; }

struct BlobLayoutBf16 {
    uint32_t stream_off, stream_bytes, side_off, side_floats;
    uint32_t bias_trunk, bias_feat, bias_d, head_b, wdir_t, total_bytes;
};

static BlobLayoutBf16 make_layout_bf16(int D, int W, int skip, int /*L_x*/, int /*L_d*/) {
    constexpr int L_x = KERNEL_LX, L_d = KERNEL_LD;          // the kernel's layout; a network with fewer frequencies gets zero weights (layout.h)
    BlobLayoutBf16 b{};
    const int NT = W / MT, in_d = 3 + 6 * L_d;
    const uint32_t pe_q = (uint32_t)enc_ksteps32(L_x) * NT, h_q = (uint32_t)(W / KF) * NT;
    uint32_t quads = pe_q;
    for (int l = 1; l < D; ++l) quads += h_q + ((skip >= 0 && l == skip + 1) ? pe_q : 0);
    quads += TAIL_QUADS;
    b.stream_off = HEADER_BYTES;
    b.stream_bytes = quads * QUAD_BYTES;
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

// ---------------------------------------------------------------------------------------------
// host: packer
// ---------------------------------------------------------------------------------------------
static inline uint16_t f32_to_bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// The stream is emitted either as bf16 (the blob) or as the float itself (the gather map of the device-side packer, built by
// running this packer over parameters whose values are their own flat index + 1).
template <typename T> static inline T stream_elem(float x);
template <> inline uint16_t stream_elem<uint16_t>(float x) { return f32_to_bf16_rne(x); }
template <> inline float stream_elem<float>(float x) { return x; }

// One quad: the A fragment of output rows row0..row0+15 for the 32 input columns cols[q*8 + j] (-1: zero).
// rowmap (optional, 16 entries): weight-matrix row feeding output row i of the tile, -1: zero row.
template <typename T>
static void emit_quad(std::vector<T>& st, const float* Wm, int n_out, int n_in, int row0, const int* rowmap, const int* cols) {
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const int col = cols[(lane >> 4) * 8 + j];
            const int n = rowmap ? rowmap[lane & 15] : row0 + (lane & 15);
            st.push_back((col >= 0 && n >= 0 && n < n_out) ? stream_elem<T>(Wm[(size_t)n * n_in + col]) : (T)0);
        }
}
static std::vector<int> enc_cols32(int L, int base) {           // L: the network's own frequencies; the k-step count is the kernel's
    const int nch = 3 + 6 * L, KS = enc_ksteps32(KERNEL_LX);
    std::vector<int> c(KS * KF);
    for (int u = 0; u < KS * KF; ++u) c[u] = u < nch ? base + u : -1;
    return c;
}
// input columns in the order the packed accumulators present them: fragment s, lane quarter q, element j
static std::vector<int> act_cols32(int W, int base) {
    std::vector<int> c;
    for (int s = 0; s < W / KF; ++s)
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 8; ++j) c.push_back(base + MT * (2 * s + (j >> 2)) + 4 * q + (j & 3));
    return c;
}
// a layer in output-tile-major order: for every tile, all its k-steps
template <typename T>
static void emit_layer(std::vector<T>& st, const float* Wm, int n_out, int n_in, int NT, const std::vector<int>& cols) {
    const int KS = (int)cols.size() / KF;
    for (int tile = 0; tile < NT; ++tile)
        for (int ks = 0; ks < KS; ++ks) emit_quad(st, Wm, n_out, n_in, MT * tile, nullptr, cols.data() + KF * ks);
}

static int check_net_bf16(const mi_nerf_net* net) {
    MN_CHECK_ARG(net != nullptr, "net is NULL");
    MN_CHECK_ARG(net->W == 256, "the bf16 variant is built for W=256 only (got %d)", net->W);
    MN_CHECK_ARG(net->D >= 2 && net->D <= 16 && net->L_x >= 0 && net->L_x <= KERNEL_LX && net->L_d >= 0 && net->L_d <= KERNEL_LD && net->skip >= -1,
                 "unsupported network for bf16 (D=%d L_x=%d L_d=%d skip=%d)", net->D, net->L_x, net->L_d, net->skip);
    return MI_NERF_OK;
}

size_t packed_bytes_bf16(const mi_nerf_net* net) {
    if (check_net_bf16(net)) return 0;
    return make_layout_bf16(net->D, net->W, net->skip, net->L_x, net->L_d).total_bytes;
}

// the weight stream in consumption order
template <typename T>
static int build_stream(const mi_nerf_net* net, const mi_nerf_params* p, const BlobLayoutBf16& L, std::vector<T>& st) {
    const int D = net->D, W = net->W, NT = W / MT;
    const int in_x = 3 + 6 * net->L_x, in_d = 3 + 6 * net->L_d;
    const size_t per_quad = QUAD_BYTES / 2;
    st.reserve(L.stream_bytes / 2);
    emit_layer(st, p->linear_x_w[0], W, in_x, NT, enc_cols32(net->L_x, 0));
    for (int l = 1; l < D; ++l) {
        const bool cat = (net->skip >= 0 && l == net->skip + 1);
        std::vector<int> cols = act_cols32(W, cat ? in_x : 0);          // input columns are cat([gamma(x), h]), NeRF.py:41 ...
        if (cat) {                                                      // ... consumed activations first, gamma(x) last
            const std::vector<int> enc = enc_cols32(net->L_x, 0);
            cols.insert(cols.end(), enc.begin(), enc.end());
        }
        emit_layer(st, p->linear_x_w[l], W, cat ? W + in_x : W, NT, cols);
    }
    // tail: feature layer | density tile over the trunk output | view-direction layer | colour tile
    emit_layer(st, p->linear_feat_w, W, W, NT, act_cols32(W, 0));
    int rowmap[MT];
    {
        const std::vector<int> act = act_cols32(W, 0);
        for (int i = 0; i < MT; ++i) rowmap[i] = (i == 3) ? 0 : -1;     // output row 3 <- linear_density row 0
        for (int ks = 0; ks < W / KF; ++ks) emit_quad(st, p->linear_density_w, 1, W, 0, rowmap, act.data() + KF * ks);
    }
    emit_layer(st, p->linear_d_w, W / 2, W + in_d, NT / 2, act_cols32(W, 0));
    {
        const std::vector<int> act = act_cols32(W / 2, 0);
        for (int i = 0; i < MT; ++i) rowmap[i] = (i < 3) ? i : -1;      // output rows 0..2 <- linear_color rows 0..2
        for (int ks = 0; ks < W / 2 / KF; ++ks) emit_quad(st, p->linear_color_w, 3, W / 2, 0, rowmap, act.data() + KF * ks);
    }
    st.resize(st.size() + (size_t)(TAIL_QUADS - TAIL_USED) * per_quad, (T)0);
    MN_CHECK_ARG(st.size() * 2 == L.stream_bytes, "internal: bf16 stream %zu != %u", st.size() * 2, L.stream_bytes);
    return MI_NERF_OK;
}
// the fp32 side tables
static void fill_side(const mi_nerf_net* net, const mi_nerf_params* p, const BlobLayoutBf16& L, float* side) {
    const int D = net->D, W = net->W, in_d = 3 + 6 * net->L_d;
    for (int l = 0; l < D; ++l) memcpy(side + L.bias_trunk + (size_t)l * W, p->linear_x_b[l], W * 4);
    memcpy(side + L.bias_feat, p->linear_feat_b, W * 4);
    memcpy(side + L.bias_d, p->linear_d_b, (W / 2) * 4);
    memcpy(side + L.head_b, p->linear_color_b, 3 * 4);
    side[L.head_b + 3] = p->linear_density_b[0];
    for (int f = 0; f < in_d; ++f)
        for (int n = 0; n < W / 2; ++n) side[L.wdir_t + (size_t)f * (W / 2) + n] = p->linear_d_w[(size_t)n * (W + in_d) + W + f];
}
static void fill_header(const mi_nerf_net* net, const BlobLayoutBf16& L, uint32_t* hdr) {
    memset(hdr, 0, HEADER_BYTES);
    hdr[0] = BLOB_MAGIC; hdr[1] = 3; hdr[2] = net->D; hdr[3] = net->W; hdr[4] = (uint32_t)net->skip; hdr[5] = KERNEL_LX; hdr[6] = KERNEL_LD;   // the layout's
    hdr[13] = net->L_x; hdr[14] = net->L_d;                                                                                                    // the network's
    hdr[7] = L.stream_off; hdr[8] = L.stream_bytes; hdr[9] = L.stream_bytes; hdr[10] = L.side_off; hdr[11] = L.side_floats;
    hdr[12] = 2;   // stream element bytes
}

int pack_bf16(const mi_nerf_net* net, const mi_nerf_params* p, void* blob, size_t blob_bytes) {
    if (int rc = check_net_bf16(net)) return rc;
    const BlobLayoutBf16 L = make_layout_bf16(net->D, net->W, net->skip, net->L_x, net->L_d);
    MN_CHECK_ARG(blob_bytes >= L.total_bytes, "blob too small: %zu < %u", blob_bytes, L.total_bytes);
    memset(blob, 0, L.total_bytes);
    std::vector<uint16_t> st;
    if (int rc = build_stream(net, p, L, st)) return rc;
    fill_header(net, L, (uint32_t*)blob);
    memcpy((char*)blob + L.stream_off, st.data(), L.stream_bytes);
    fill_side(net, p, L, (float*)((char*)blob + L.side_off));
    return MI_NERF_OK;
}

// Device-side packing of the bf16 blob (a model whose parameters live on the device is re-packed for every call: weights.py):
// gather map from the flat parameter vector (mi_nerf_param_count order), one entry per stream ELEMENT followed by one per side
// float; entry = 1 + flat index, 0 = constant zero.  Built like pack_map (pack.cpp): this packer run over index-valued parameters.
size_t pack_map_bf16_len(const mi_nerf_net* net) {
    if (check_net_bf16(net)) return 0;
    const BlobLayoutBf16 L = make_layout_bf16(net->D, net->W, net->skip, net->L_x, net->L_d);
    return (size_t)L.stream_bytes / 2 + L.side_floats;
}
int pack_map_bf16(const mi_nerf_net* net, int32_t* map, size_t map_len) {
    if (int rc = check_net_bf16(net)) return rc;
    const int D = net->D, W = net->W;
    const BlobLayoutBf16 L = make_layout_bf16(D, W, net->skip, net->L_x, net->L_d);
    const ParamOffsets po = make_param_offsets(D, W, net->skip, net->L_x, net->L_d);
    MN_CHECK_ARG(po.total < (1u << 24), "network too large for the index map (%u parameters)", po.total);
    const size_t n_stream = (size_t)L.stream_bytes / 2;
    MN_CHECK_ARG(map && map_len >= n_stream + L.side_floats, "map too small: %zu entries for %zu", map_len, n_stream + L.side_floats);
    std::vector<float> flat(po.total);
    for (uint32_t i = 0; i < po.total; ++i) flat[i] = (float)(i + 1);
    std::vector<const float*> wx(D), bx(D);
    for (int l = 0; l < D; ++l) { wx[l] = flat.data() + po.w_x[l]; bx[l] = flat.data() + po.b_x[l]; }
    mi_nerf_params p{};
    p.linear_x_w = wx.data(); p.linear_x_b = bx.data();
    p.linear_density_w = flat.data() + po.w_dens; p.linear_density_b = flat.data() + po.b_dens;
    p.linear_feat_w = flat.data() + po.w_feat; p.linear_feat_b = flat.data() + po.b_feat;
    p.linear_d_w = flat.data() + po.w_d; p.linear_d_b = flat.data() + po.b_d;
    p.linear_color_w = flat.data() + po.w_color; p.linear_color_b = flat.data() + po.b_color;
    std::vector<float> st;
    if (int rc = build_stream(net, &p, L, st)) return rc;
    std::vector<float> side(L.side_floats, 0.0f);
    fill_side(net, &p, L, side.data());
    for (size_t i = 0; i < n_stream; ++i) map[i] = (int32_t)st[i];
    for (size_t i = 0; i < L.side_floats; ++i) map[n_stream + i] = (int32_t)side[i];
    return MI_NERF_OK;
}

struct HeaderWords { uint32_t w[HEADER_BYTES / 4]; };
__global__ __launch_bounds__(256) void pack_apply_bf16_kernel(const int32_t* __restrict__ map, const float* __restrict__ flat, unsigned n_stream,
                                                               unsigned n_side, unsigned stream_off, unsigned side_off, HeaderWords hdr,
                                                               char* __restrict__ blob) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < HEADER_BYTES / 4) ((uint32_t*)blob)[i] = hdr.w[i];
    if (i < n_stream) {
        const int32_t m = map[i];
        unsigned u = m ? __float_as_uint(flat[m - 1]) : 0u;
        u = ((u & 0x7FFFFFFFu) > 0x7F800000u) ? ((u >> 16) | 0x40u) : ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);      // f32_to_bf16_rne
        ((uint16_t*)(blob + stream_off))[i] = (uint16_t)u;
    } else if (i < n_stream + n_side) {
        const int32_t m = map[i];
        ((float*)(blob + side_off))[i - n_stream] = m ? flat[m - 1] : 0.0f;
    }
}
int pack_apply_bf16(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_dev, void* blob_dev, size_t blob_bytes, hipStream_t st) {
    if (int rc = check_net_bf16(net)) return rc;
    const BlobLayoutBf16 L = make_layout_bf16(net->D, net->W, net->skip, net->L_x, net->L_d);
    MN_CHECK_ARG(map_dev && flat_dev && blob_dev, "NULL device pointer");
    MN_CHECK_ARG(blob_bytes >= L.total_bytes && ((uintptr_t)blob_dev & 15) == 0, "blob too small (%zu < %u) or not 16-byte aligned", blob_bytes, L.total_bytes);
    HeaderWords h;
    fill_header(net, L, h.w);
    const unsigned n_stream = L.stream_bytes / 2, total = n_stream + L.side_floats;
    hipLaunchKernelGGL(pack_apply_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, st, map_dev, flat_dev, n_stream, L.side_floats, L.stream_off,
                       L.side_off, h, (char*)blob_dev);
    MN_LAUNCH_CHECK("pack_apply_bf16_kernel");
    return MI_NERF_OK;
}

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
struct PhaseB {
    unsigned tile0, tile_end;   // this phase's range of 32-point tiles; tiles are numbered ray * tpr + chunk over the whole call
    unsigned n_iter;            // units per wave (the same for every wave: the ring barriers are workgroup-wide); a unit = NP / 2 tiles
    unsigned ppr;               // ray-major walk: units per ray (tpr / (NP / 2)); 0: flat walk
};
struct MlpArgsB {
    const char* stream;
    const float* side;
    const float* rays;
    const float* z;             // [n_rays, S] depths; NULL: the coarse pass draws its own stratified depths (strat_*) and writes them to z_out
    float* z_out;
    float strat_near, strat_far, strat_step;
    Jitter strat_jitter;
    float* out;
    PhaseB ph[2];               // one or two phases (see run_phase); the kernel's template arguments say how many and of which shape
    unsigned n_rays;
    int S, tpr, D, skip_layer;
    unsigned stream_bytes, side_floats;
    unsigned o_bias_trunk, o_bias_feat, o_bias_d, o_head_b, o_wdir_t;
    unsigned long long* diag;   // MN_DIAG builds only: per-wave cycle sums of the kernel's segments
    // small coarse launches (one 32-point unit per wave, two units per ray: a workgroup's four waves hold rays 2b and 2b + 1 whole): the
    // workgroup composites its two rays and draws their fine depths in the kernel's epilogue (stage_dev.h), fz_on != 0
    int fz_on, fz_Nf, fz_n2, fz_det;
    Jitter fz_u;
    float *fz_rgb, *fz_disp, *fz_w, *fz_zf;
};

#ifdef MN_DIAG
// diagnostic build only (never shipped, never timed): s_memtime stamps around the kernel's segments (read SHARES, not totals)
__device__ __forceinline__ unsigned long long bstamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define BSTAMP(i) do { const unsigned long long t_ = bstamp(); seg[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define BSTAMP(i) do {} while (0)
#endif

struct BRing {
    const char* sbase;      // stream + wave's 8 KiB share
    unsigned voff;          // lane*16
    unsigned fetch_off, stream_bytes;
    unsigned fetch_lds, lds_lo, lds_hi;
    unsigned read_slot;
};

// LDS-DMA of the weight stream.  One global_load_lds_dwordx4 moves 64 lanes x 16 B = one 1 KiB quad: global address = per-lane
// VGPR pair + instruction offset, LDS destination = M0 + instruction offset + lane * 16.  M0 is written twice per slot (each
// wave's 8 KiB share = two 4 KiB halves, the 13-bit offset reaches 4 KiB) and is NOT saved / restored around each DMA: nothing
// else in this kernel touches M0 (hipcc uses it only for LDS-direct / GWS / sendmsg / movrel instructions, none of which occur
// here; tests/test_packing_cpu.py disassembles the object and checks that every M0 write is ours).
__device__ __forceinline__ void bdma_set_m0(unsigned lds_in) {
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);      // wave-uniform by construction; pin to an SGPR
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_addr) : "memory");
}
template <int IMM>
__device__ __forceinline__ void bdma16(const char* gaddr_lane) {
    asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gaddr_lane), "i"(IMM) : "memory");
}
// DMA number i (0 .. BSLOT_QUADS / NWV - 1) of the slot being fetched: a wave's share of a slot is 8 KiB (4 waves per workgroup: two
// 4 KiB halves, M0 set twice) or 4 KiB (8 waves)
template <int NWV>
__device__ __forceinline__ void bring_dma(const BRing& r, int i) {
    const char* g = r.sbase + r.fetch_off + r.voff + (i >= 4 ? 4096 : 0);
    if (i == 0) bdma_set_m0(r.fetch_lds);
    if (i == 4) bdma_set_m0(r.fetch_lds + 4096);
    if ((i & 3) == 0) bdma16<0>(g);
    else if ((i & 3) == 1) bdma16<1024>(g);
    else if ((i & 3) == 2) bdma16<2048>(g);
    else bdma16<3072>(g);
}
__device__ __forceinline__ void bring_next_fetch(BRing& r) {
    r.fetch_off += BSLOT_BYTES;
    if (r.fetch_off >= r.stream_bytes) r.fetch_off = 0;
    r.fetch_lds += BSLOT_BYTES;
    if (r.fetch_lds >= r.lds_hi) r.fetch_lds = r.lds_lo;
}
// consume the next slot: everything but the DMAs issued during the phase that ends here has landed (slot p+1 was
// issued two phases ago); barrier; slot p+2 streams into ring[(p+2)%3] == ring[(p-1)%3] during the new phase.
// Other vector-memory operations of the wave (input prefetches, result stores) share the counter and retire in order:
// they can only make this wait stricter.
template <int NWV>
__device__ __forceinline__ void bring_advance(BRing& r) {
    if constexpr (NWV == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    bring_next_fetch(r);
    r.read_slot = (r.read_slot + 1 == BNSLOT) ? 0 : r.read_slot + 1;
}
// fragment at slot position qs; positions 1 .. BSLOT_QUADS / NWV also issue one of the slot's DMAs (never a burst)
template <int NWV>
__device__ __forceinline__ u32x4b bring_read(const char* smem, const BRing& r, int lane, int qs) {
    if (qs >= 1 && qs <= bdma_of(NWV)) bring_dma<NWV>(r, qs - 1);
    return *(const u32x4b*)(smem + r.read_slot * BSLOT_BYTES + lane * 16 + qs * QUAD_BYTES);
}

// two floats -> one dword of a B fragment (round to nearest even).  Pinned where it is written.
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    unsigned d;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
    return d;
}

// ---------------------------------------------------------------------------------------------
// THE FRAGMENT FILE: all 256 AGPRs, managed by hand.
//
// Two sets (ping-pong between consecutive layers) x NP point tiles x 8 B fragments x 4 registers = 256 (NP = 4) = the whole
// accumulation-register file (half of it at NP = 2).  With most of the wave's 512 registers live, hipcc's allocator could not place these tuples
// (it treats MFMA operands as "either file" values and thrashed: fragments copied AGPR -> VGPR in front of every MFMA,
// accumulators spilled around the packing, up to 300 spilled registers) -- so the fragments never become compiler values at
// all: they are written with v_accvgpr_write_b32 a[N] and read as the MFMA's B operand a[N:N+3] with N a compile-time
// constant, and the compiler allocates only the VGPR side.  It must keep out of the AGPRs entirely: this file is built with
// -mllvm -amdgpu-spill-vgpr-to-agpr=0 and contains no MFMA builtin; one clobber of a255 makes the kernel descriptor reserve
// the whole file (tests/test_packing_cpu.py checks the object).
//
// Consequence: the MFMAs are asm statements and hipcc inserts NO hazard wait states around them.  They hold by construction:
//   * the four accumulation chains of a job are interleaved, a chain's next MFMA is four issues (64 cycles) behind;
//   * every other reader of an MFMA result (the packing of a finished tile, the final store) is at least four MFMA issues
//     behind the MFMA that wrote it -- the packing starts in the NEXT job's second group, the store and the density read-out
//     are preceded by whole groups / explicit s_nops;
//   * a fragment register is written at least one whole group (>= 64 cycles) before the MFMA that reads it and never while an
//     MFMA that reads it can be in flight (a layer writes the OTHER set; the half fragment packed across a layer boundary is
//     the last one that layer reads);
//   * the C operand of a job's first MFMAs is kept allocated until the next group (the matrix pipe reads it after issue);
//   * VGPR operands (A fragments, biases, gamma(x)) come from LDS reads the compiler tracks (s_waitcnt before the asm).
// ---------------------------------------------------------------------------------------------
template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}
__host__ __device__ constexpr int frag_reg(int np, int set, int p, int f) { return ((set * np + p) * 8 + f) * 4; }

// first MFMA of a job (C operand = bias) / accumulate; B operand from the fragment file (IC<R>) or from a VGPR fragment
template <int R>
__device__ __forceinline__ void mfma_first(f32x4& acc, const u32x4b& afrag, IC<R>, const f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%3:%4], %2" : "=&v"(acc) : "v"(afrag), "v"(c), "n"(R), "n"(R + 3));
}
__device__ __forceinline__ void mfma_first(f32x4& acc, const u32x4b& afrag, const u32x4b& bfrag, const f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(afrag), "v"(bfrag), "v"(c));
}
template <int R>
__device__ __forceinline__ void mfma_acc(f32x4& acc, const u32x4b& afrag, IC<R>) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(afrag), "n"(R), "n"(R + 3));
}
__device__ __forceinline__ void mfma_acc(f32x4& acc, const u32x4b& afrag, const u32x4b& bfrag) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(afrag), "v"(bfrag));
}

// A finished 16x16 tile (4 accumulator registers per lane) -> two packed dwords -> registers R, R + 1 of the fragment file
// (tile t of a layer is dwords 2(t&1), 2(t&1)+1 of fragment t>>1).  ReLU on the packed pair: as signed 16-bit integers every
// negative bf16 (and -0.0) is below zero.  In three stages of two independent instructions each, one stage per MFMA gap (a 16-cycle
// MFMA leaves room for two VALU issues), or as one statement where there are more gaps than work.
template <bool RELU, int R, int STAGE>
__device__ __forceinline__ void pack_stage(const f32x4& acc, unsigned (&t)[2]) {
    if constexpr (STAGE == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5" : "=&v"(t[0]), "=&v"(t[1]) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
    else if constexpr (STAGE == 1) { if (RELU) asm volatile("v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0" : "+v"(t[0]), "+v"(t[1])); }
    else asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1" ::"v"(t[0]), "v"(t[1]), "n"(R), "n"(R + 1));
}
template <bool RELU, int R>
__device__ __forceinline__ void pack_whole(const f32x4& acc) {
    unsigned t[2];
    pack_stage<RELU, R, 0>(acc, t); pack_stage<RELU, R, 1>(acc, t); pack_stage<RELU, R, 2>(acc, t);
}
// register of the fragment file that receives tile T of a layer written into set SET, for point tile P
__host__ __device__ constexpr int tile_reg(int np, int set, int p, int t) { return frag_reg(np, set, p, t >> 1) + 2 * (t & 1); }

// The standard schedule of a job's packing work (jobs of >= 8 k-steps).  The gap behind MFMA 0 of a group carries the ring
// bookkeeping, so packing rides in the other gaps.
//   NP = 4: the previous job's point tile pp is packed in group pp + 1, stages 0 / 1 / 2 in the gaps behind MFMAs 1 / 2 / 3.
//           Done by group 4.
//   NP = 2: one gap per group: tile 0 in groups 1, 2, 3, tile 1 in groups 3, 4, 5 (group 3 carries two stages).  Done by group 5.
// Either way the half fragment packed across a layer boundary (it feeds k-step 7) is written >= 2 MFMA issues before its reader,
// and a tile's first stage is >= 5 MFMA issues (80 cycles) behind the MFMA that finished it.
template <int NP, bool RELU, int SET, int T, int KS, int P>
__device__ __forceinline__ void pack_sched(const f32x4 (&prev)[NP], unsigned (&t)[NP][2]) {
    if constexpr (NP == 4) {
        if constexpr (KS >= 1 && KS <= NP && P >= 1) pack_stage<RELU, tile_reg(NP, SET, KS - 1, T), P - 1>(prev[KS - 1], t[KS - 1]);
    } else {
        static_assert(NP == 2, "packing schedules exist for 4 and 2 point tiles per wave");
        if constexpr (P == 1 && KS >= 1 && KS <= 3) pack_stage<RELU, tile_reg(NP, SET, 0, T), KS - 1>(prev[0], t[0]);
        if constexpr (P == 1 && KS >= 3 && KS <= 5) pack_stage<RELU, tile_reg(NP, SET, 1, T), KS - 3>(prev[1], t[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// One job: output tile of 16 features x NP point tiles over KS k-steps, stream quads Q0..Q0+KS-1 of the current body
// (bodies start on a slot boundary, so every ring position below is a compile-time constant).
// csel(p): C operand of point tile p's first MFMA (the bias).  bsrc(p_c, ks_c): B operand -- IC<register> (fragment file) or a
// VGPR fragment.  Group ks = NP sub-groups [MFMA of point tile p on fragment a[(Q0+ks) % DA]] [p == 0: ring bookkeeping -- advance,
// one DMA, the A-pipeline refill of the register the PREVIOUS group consumed] [hook(ks_c, p_c)], each pinned: in-order issue lets
// only ~two VALU instructions ride behind a 16-cycle MFMA before the wave blocks on the next one.
// QEND/QPAD: stream positions >= QEND skip QPAD quads (the padding at the end of the tail body).
// ---------------------------------------------------------------------------------------------
template <int NP, int NWV, int Q0, int KS, int QEND, int QPAD, typename CSel, typename BSrc, typename Hook>
__device__ __forceinline__ void job(f32x4 (&acc)[NP], CSel csel, BSrc bsrc, u32x4b (&a)[DA], const char* smem, BRing& ring, int lane, Hook hook) {
    static_for<0, KS>([&](auto ks_c) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_c)::value;
        constexpr int q0 = Q0 + ks + DA - 1;                              // stream position being read into register q0 % DA
        constexpr int qn = (q0 >= QEND) ? q0 + QPAD : q0;
        static_for<0, NP>([&](auto p_c) __attribute__((always_inline)) {
            constexpr int p = decltype(p_c)::value;
            if constexpr (ks == 0) mfma_first(acc[p], a[(Q0 + ks) % DA], bsrc(p_c, ks_c), csel(p));
            else mfma_acc(acc[p], a[(Q0 + ks) % DA], bsrc(p_c, ks_c));
            if constexpr (p == 0) {
                if constexpr (qn % BSLOT_QUADS == 0) bring_advance<NWV>(ring);
                a[q0 % DA] = bring_read<NWV>(smem, ring, lane, qn % BSLOT_QUADS);
            }
            hook(ks_c, p_c);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        });
        // the C operands are dead for the compiler once the first MFMAs are issued, but the matrix pipe reads C for up to 7 wait states
        // after the issue of an 8-pass MFMA (ISA guide 4.5: XDL read srcC -> VALU write): keep their registers out of the allocator's
        // hands until the END OF GROUP 1 -- >= 2 MFMA issues = 8 wait states behind the last MFMA of group 0.  (Until round 4 this
        // sat behind group 0, zero wait states after its last MFMA: at the 32-point shape hipcc put the ring's `v_add_u32 fetch_off`
        // into the bias register right there -- tools/mfma_hazard_check.py; no wrong result was ever observed.)
        static_assert(KS >= 2, "a job has at least two k-steps");
        if constexpr (ks == 1) {
            asm volatile("" ::"v"(csel(0)), "v"(csel(1)));
            if constexpr (NP == 4) asm volatile("" ::"v"(csel(2)), "v"(csel(3)));
        }
    });
}

// One PHASE of a launch: ph.n_iter units of NP point tiles per wave over the tiles [ph.tile0, ph.tile_end).  A launch is one phase
// (one shape) or two (whole rounds of the 64-point shape, then the remainder as one round of the 32-point shape): the weight
// ring and the A-fragment pipeline run on across the phase boundary (every unit ends at stream position 0).
template <int W, int LX, int LD, int NP, int NWV>
__device__ __forceinline__ void run_phase(const MlpArgsB& a, const PhaseB ph, char* smem, float* side, float* scr_base, char* pe_base, BRing& ring,
                                          u32x4b (&aq)[DA], const int lane, const int wave
#ifdef MN_DIAG
                                          , unsigned long long (&seg)[8], unsigned long long& tprev
#endif
                                          ) {
    constexpr int NT = W / MT, KH = W / KF, KPE = enc_ksteps32(LX), IN_X = 3 + 6 * LX, IN_D = 3 + 6 * LD;
    constexpr int NTL = NP / 2;                              // 32-sample tiles per unit of work (point tile p: tile p >> 1, half p & 1)
    static_assert(KPE == 2 && NT == 16 && KH == 8 && (NP == 4 || NP == 2), "stream positions and the fragment file are laid out for 63 -> 64 encoded channels, W = 256, 4 or 2 point tiles");
    constexpr int BIG = 1 << 30;
    const int col = lane & 15, q4 = lane >> 4;               // point of a 16-point tile; lane quarter
    const int pq = q4 & (NP - 1);                            // the point tile whose point `col` this lane encodes (NP = 2: quarters 2, 3 duplicate 0, 1)
    float* scratch = scr_base + wave * (NTL * (W / 2));      // per wave, per 32-sample tile: hoisted direction bias
    char* pe_wave = pe_base + wave * (NP * KPE * QUAD_BYTES);    // this wave's parked gamma(x) fragments
    char* pe_lds = pe_wave + lane * 16;
    // ---- tile walk: a wave takes UNITS of NTL consecutive 32-sample tiles = NP 16-point MFMA tiles (point tile p: 32-sample tile
    // p >> 1, half p & 1; a pair of tiles at NP = 4, one tile at NP = 2).  Ray-major (ppr > 0): a wave walks whole rays, so the
    // hoisted view-direction term is computed once per ray; flat otherwise.  Inputs of the next unit are loaded a unit ahead.
    const unsigned NW = gridDim.x * NWV, wid = blockIdx.x * NWV + wave;
    auto pair_of = [&](unsigned it) -> unsigned {
        if (ph.ppr) { const unsigned blk = it / ph.ppr; return (blk * NW + wid) * ph.ppr + (it - blk * ph.ppr); }
        return it * NW + wid;
    };
    // gamma(x) is computed ONCE per point: lane (q4, col) owns point `col` of point tile q4 (32-sample tile q4 >> 1, half q4 & 1),
    // encodes it and writes its fragments' dwords where the other lane quarters read them (the fragments are parked in LDS for the
    // skip layer anyway).  Each lane therefore loads one depth; the rays of both 32-sample tiles are loaded by every lane (the
    // hoisted view-direction term is computed per tile by the whole wave).
    unsigned n_tile[NTL];  float nx_r[NTL][6], nx_z;
    auto load_inputs = [&](unsigned it) __attribute__((always_inline)) {
        const unsigned pr = pair_of(it);
        unsigned my_ray = 0, my_chunk = 0;                        // the tile this lane's own point belongs to (selected, not branched on: a lane-
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) {                        // dependent branch inside the MFMA stream costs a lone wave more than the select)
            unsigned t = ph.tile0 + (unsigned)NTL * pr + tl;
            n_tile[tl] = t;
            if (t >= ph.tile_end) t = ph.tile_end - 1;            // inactive: recompute the last tile, store nothing
            const unsigned ray = t / (unsigned)a.tpr, chunk = t - ray * (unsigned)a.tpr;
            const float* rp = a.rays + (size_t)ray * 6;
#pragma unroll
            for (int e = 0; e < 6; ++e) nx_r[tl][e] = rp[e];
            if (tl == 0 || tl == (pq >> 1)) { my_ray = ray; my_chunk = chunk; }
        }
        {
            const int sample = (int)my_chunk * 32 + 16 * (pq & 1) + col;
            const int sc = sample < a.S ? sample : a.S - 1;
            if (a.z) nx_z = a.z[(size_t)my_ray * a.S + sc];
            else {              // the coarse pass of render_rays: stratified depth drawn here (nerf_process.py:42-60), kept for the compositing
                nx_z = stratified_depth((long long)my_ray, sc, a.S, a.strat_step, a.strat_near, a.strat_far, a.strat_jitter);
                if (q4 < NP) a.z_out[(size_t)my_ray * a.S + sc] = nx_z;      // inactive / clamped lanes rewrite an existing element with its own value
            }
        }
    };
    load_inputs(0);
    unsigned bias_ray[NTL];
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) bias_ray[tl] = ~0u;

    f32x4 acc[NP], prev[NP];
    f32x4 cin, cnext;                                        // bias of the current / next job (shared by the point tiles)
    auto csel1 = [&](int) __attribute__((always_inline)) -> const f32x4& { return cin; };
    u32x4b peb[NP][KPE];

    // ---- bodies (straight-line code, everything static) ------------------------------------------------------------------------
    // A trunk layer reads fragment set SIN (the second half of fragment 7 is still being packed from `prev` when it starts),
    // writes set 1 - SIN, leaves its last tile in `prev`; the bias of the NEXT job is read while a job's last groups compute.
    auto trunk_layer = [&](auto skip_c, auto sin_c, const float* bias, const float* next_bias) __attribute__((always_inline)) {
        constexpr bool SKIP = decltype(skip_c)::value;
        constexpr int SIN = decltype(sin_c)::value, SOUT = 1 - SIN;
        constexpr int KS = SKIP ? KH + KPE : KH;
        // skip layer: the gamma(x) fragments were parked in LDS by the prologue (32 registers that would otherwise stay live
        // through every layer); each job re-reads them just in time, under its own activation k-steps
        u32x4b per[NP][KPE];
        auto bsrc = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> decltype(auto) {
            constexpr int p = decltype(p_c)::value, ks = decltype(ks_c)::value;
            if constexpr (ks >= KH) return (const u32x4b&)per[p][ks - KH];
            else return IC<frag_reg(NP, SIN, p, ks)>{};
        };
        static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
            constexpr int t = decltype(t_c)::value;
            unsigned pt[NP][2];
            auto hook = [&](auto ks_c, auto p_c) __attribute__((always_inline)) {
                constexpr int ks = decltype(ks_c)::value, p = decltype(p_c)::value;
                // previous tile -> (half a) fragment of the layer that follows it
                if constexpr (t == 0) pack_sched<NP, true, SIN, NT - 1, ks, p>(prev, pt);
                else pack_sched<NP, true, SOUT, t - 1, ks, p>(prev, pt);
                if constexpr (ks == 5 && p == 1) {              // bias of the next job (C operand of its first MFMAs)
                    const float* v = (t + 1 < NT) ? bias + MT * (t + 1) + 4 * q4 : next_bias + 4 * q4;
                    cnext = *(const f32x4*)v;
                }
                if constexpr (SKIP && ks >= KH - 2 && ks < KH - 2 + KPE)         // gamma(x) fragment of k-step ks + 2, one point tile per gap
                    per[p][ks - (KH - 2)] = *(const u32x4b*)(pe_lds + (p * KPE + (ks - (KH - 2))) * QUAD_BYTES);
            };
            job<NP, NWV, t * KS, KS, BIG, 0>(acc, csel1, bsrc, aq, smem, ring, lane, hook);
#pragma unroll
            for (int p = 0; p < NP; ++p) prev[p] = acc[p];
            cin = cnext;
        });
    };

    for (unsigned it = 0; it < ph.n_iter; ++it) {
        // ---- prologue: this pair's points, gamma(x) fragments, hoisted view-direction bias ------------------------------------
        unsigned tray[NTL]; bool valid[NP]; size_t out_idx[NP];
        float in_o[NTL][3], in_d[NTL][3];
        const float in_z = nx_z;
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) {
            const bool active = n_tile[tl] < ph.tile_end;
            const unsigned t = active ? n_tile[tl] : ph.tile_end - 1;
            const unsigned ray = t / (unsigned)a.tpr, chunk = t - ray * (unsigned)a.tpr;
            tray[tl] = ray;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                       // results of point tile p = 2 tl + h land on lane quarter 0, lane = point
                const int sample = (int)chunk * 32 + 16 * h + col;
                valid[2 * tl + h] = active && sample < a.S;
                out_idx[2 * tl + h] = (size_t)ray * a.S + (sample < a.S ? sample : a.S - 1);
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) { in_o[tl][e] = nx_r[tl][e]; in_d[tl][e] = nx_r[tl][3 + e]; }
        }
        {
            const bool t1 = (pq >> 1) != 0;                     // this lane's point belongs to the second 32-sample tile (NP = 4 only)
            auto mine = [&](const float (&v)[NTL][3], int c) __attribute__((always_inline)) -> float {
                if constexpr (NTL == 2) return t1 ? v[1][c] : v[0][c];
                else return v[0][c];
            };
            // pts = rays_o + rays_d * z (nerf_process.py:69-70)
            const float pt[3] = {mine(in_o, 0) + mine(in_d, 0) * in_z, mine(in_o, 1) + mine(in_d, 1) * in_z, mine(in_o, 2) + mine(in_d, 2) * in_z};
            float sn[LX][3], cs[LX][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // octave 0: Cody-Waite + Cephes as in the fp32 kernel.  Beyond 4e6 rad the multiple count is no longer exact (and
                // an fp32 argument with ulp >= 0.25 rad has no meaningful sine): the remainder is clamped so that finite inputs
                // give finite, bounded encodings; NaN / Inf still come out as NaN.  (The fp32 kernel takes the libm path there.)
                float r, sp, cp; int q;
                sc_reduce(pt[c], 0, r, q);
                r = __builtin_fminf(__builtin_fmaxf(r, -0.8f), 0.8f) + (r - r);      // (r - r): 0, or NaN for a non-finite remainder
                sc_poly(r, sp, cp);
                sn[0][c] = sc_select(sp, cp, q);
                cs[0][c] = sc_select(sp, cp, q + 1);
#pragma unroll
                for (int k = 1; k < LX; ++k) {                  // angle doubling
                    const float s2 = sn[k - 1][c] + sn[k - 1][c];
                    sn[k][c] = s2 * cs[k - 1][c];
                    cs[k][c] = __builtin_fmaf(-s2, sn[k - 1][c], 1.0f);
                }
            }
            auto chan = [&](int u) __attribute__((always_inline)) -> float {     // channel u of gamma(x); u is a constant at every use
                if (u >= IN_X) return 0.0f;
                if (u < 3) return pt[u];
                const int k = (u - 3) / 6, r = (u - 3) % 6;
                return r < 3 ? sn[k][r] : cs[k][r - 3];
            };
            // fragment (point tile pq, k-step ks): lane quarter qq reads channels 32 ks + 8 qq + j of point `col` at
            // [fragment][(qq * 16 + col) * 16 bytes]: this lane writes those 16 bytes for every qq (at NP = 2 lane quarters 2, 3
            // write what quarters 0, 1 write: same bytes, same addresses)
            char* wr = pe_wave + (pq * KPE) * QUAD_BYTES + col * 16;
#pragma unroll
            for (int ks = 0; ks < KPE; ++ks)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    u32x4b v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = pack2(chan(KF * ks + 8 * qq + 2 * i), chan(KF * ks + 8 * qq + 2 * i + 1));
                    *(u32x4b*)(wr + ks * QUAD_BYTES + qq * 256) = v;
                }
        }
        // LDS operations of one wave execute in order: the fragments written above are complete when these reads return
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ks = 0; ks < KPE; ++ks) peb[p][ks] = *(const u32x4b*)(pe_lds + (p * KPE + ks) * QUAD_BYTES);
        // hoisted view-direction term of linear_d (fp32), per 32-sample tile: scratch[tl][n] = b_d[n] + sum_f Wd[n][W+f] * gamma(d/|d|)[f]
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) {
            float* sc_t = scratch + tl * (W / 2);
            if (tray[tl] != bias_ray[tl]) {
                bias_ray[tl] = tray[tl];
                if (tl == 1 && tray[NTL - 1] == tray[0]) {      // both tiles on one ray: copy (same wave: no barrier needed)
#pragma unroll
                    for (int n0 = 0; n0 < W / 2; n0 += 64) sc_t[n0 + lane] = scratch[n0 + lane];
                } else {
                    const float dx = in_d[tl][0], dy = in_d[tl][1], dz = in_d[tl][2];
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
                        sc_t[n] = s;
                    }
                }
            }
        }
        BSTAMP(0);   // prologue
        // ---- layer 0: 16 jobs of 2 k-steps over gamma(x) (VGPR fragments), output into set 0 -----------------------------------------
        {
            const float* b0 = side + a.o_bias_trunk + 4 * q4;
            cin = *(const f32x4*)b0;
            auto bsrc = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> const u32x4b& { return peb[decltype(p_c)::value][decltype(ks_c)::value]; };
            static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                auto hook = [&](auto ks_c, auto p_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, p = decltype(p_c)::value;
                    // 2 NP MFMAs per job and NP tiles to pack: a whole statement per gap (these jobs run VALU bound; 3 % of the MFMAs).
                    // A tile is packed >= 4 MFMA issues behind the MFMA that finished it.
                    if constexpr (NP == 4) {
                        if constexpr (t > 0 && ks == 0 && p >= 1) pack_whole<true, tile_reg(NP, 0, p - 1, t - 1)>(prev[p - 1]);
                        if constexpr (t > 0 && ks == 1 && p == 1) pack_whole<true, tile_reg(NP, 0, 3, t - 1)>(prev[3]);
                    } else {
                        if constexpr (t > 0 && ks == 1) pack_whole<true, tile_reg(NP, 0, p, t - 1)>(prev[p]);
                    }
                    if constexpr ((NP == 4 && ks == 1 && p == 2) || (NP == 2 && ks == 0 && p == 1)) {
                        const float* v = (t + 1 < NT) ? b0 + MT * (t + 1) : side + a.o_bias_trunk + W + 4 * q4;
                        cnext = *(const f32x4*)v;
                    }
                };
                job<NP, NWV, t * KPE, KPE, BIG, 0>(acc, csel1, bsrc, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) prev[p] = acc[p];
                cin = cnext;
            });
        }
        BSTAMP(1);   // layer 0
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
        int l = 1;                                               // layer 0 wrote set 0 (its last tile is still in `prev`)
#pragma unroll 1
        for (; l + 1 < a.D; l += 2) { layer_01(l); layer_10(l + 1); }
        // ---- tail: feature layer, density tile, view-direction layer, colour tile, store ---------------------------------------------
        auto tail = [&](auto sin_c) __attribute__((always_inline)) {
            constexpr int SIN = decltype(sin_c)::value, SOUT = 1 - SIN;
            f32x4 hd[NP], hc[NP], cind[NP], cnextd[NP], cinh;            // density / colour tiles; per-point-tile direction bias
            float dens[NP];
            auto cseld = [&](int p) __attribute__((always_inline)) -> const f32x4& { return cind[p]; };
            auto cselh = [&](int) __attribute__((always_inline)) -> const f32x4& { return cinh; };
            const float* bf = side + a.o_bias_feat + 4 * q4;
            auto bsrc_in = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(NP, SIN, decltype(p_c)::value, decltype(ks_c)::value)>{}; };
            auto bsrc_out = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(NP, SOUT, decltype(p_c)::value, decltype(ks_c)::value)>{}; };
            // feature layer: no activation on its outputs; its first job still packs the trunk's last tile (ReLU)
            static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                unsigned pt[NP][2];
                auto hook = [&](auto ks_c, auto p_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, p = decltype(p_c)::value;
                    if constexpr (t == 0) pack_sched<NP, true, SIN, NT - 1, ks, p>(prev, pt);
                    else pack_sched<NP, false, SOUT, t - 1, ks, p>(prev, pt);
                    if constexpr (ks == 5 && p == 1) {
                        if constexpr (t + 1 < NT) cnext = *(const f32x4*)(bf + MT * (t + 1));
                        else {                                          // density tile: row 3 = density bias (lane quarter 0 only)
                            const float db = side[a.o_head_b + 3];
                            cnext[0] = 0.0f; cnext[1] = 0.0f; cnext[2] = 0.0f; cnext[3] = q4 == 0 ? db : 0.0f;
                        }
                    }
                };
                job<NP, NWV, t * KH, KH, BIG, 0>(acc, csel1, bsrc_in, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) prev[p] = acc[p];
                cin = cnext;
            });
            // density tile over the trunk output (row 3); packs the feature layer's last tile; reads the direction bias of tile 0
            {
                unsigned pt[NP][2];
                auto hook = [&](auto ks_c, auto p_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, p = decltype(p_c)::value;
                    pack_sched<NP, false, SOUT, NT - 1, ks, p>(prev, pt);
                    if constexpr (ks == 5) cnextd[p] = *(const f32x4*)(scratch + (p >> 1) * (W / 2) + 4 * q4);
                };
                job<NP, NWV, 128, KH, BIG, 0>(hd, csel1, bsrc_in, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) cind[p] = cnextd[p];
            }
            // view-direction layer: 8 jobs over the feature layer's output; ReLU'd tiles go into fragments 0..3 of set SIN (the
            // trunk output is dead once the density tile has run)
            static_for<0, NT / 2>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                unsigned pt[NP][2];
                auto hook = [&](auto ks_c, auto p_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, p = decltype(p_c)::value;
                    if constexpr (t > 0) pack_sched<NP, true, SIN, t - 1, ks, p>(prev, pt);
                    if constexpr (t == 0 && ks == 1 && p == NP - 1) {      // the density tile's last MFMA is >= 4 issues back: its tuples may go (keep_tuple, common.h)
#pragma unroll
                        for (int pp = 0; pp < NP; ++pp) keep_tuple(hd[pp]);
                    }
                    if constexpr (t == 0 && ks == 6)            // the density tile finished >= 24 MFMAs ago: keep its one useful register
                        asm volatile("v_mov_b32 %0, %1" : "=v"(dens[p]) : "v"(hd[p][3]));
                    // next pair's rays and depths, two quads BEHIND a ring advance (tail position 162 = slot 5, quad 2): an advance waits for every
                    // older vector memory operation, and two quads before one (position 158, where this sat) is the worst place for a load.  A/B: 0.2 %;
                    // the loads and their index arithmetic cost the kernel 2 % in all (ablation build without them).
                    if constexpr (t == BF16_PF_T && ks == BF16_PF_KS && p == 1) load_inputs(it + 1 < ph.n_iter ? it + 1 : it);
                    if constexpr (ks == 5) {
                        if constexpr (t + 1 < NT / 2) cnextd[p] = *(const f32x4*)(scratch + (p >> 1) * (W / 2) + MT * (t + 1) + 4 * q4);
                        else if constexpr (p == 1) {                    // colour tile: rows 0..2 = colour bias (lane quarter 0 only)
                            const f32x4 hb4 = *(const f32x4*)(side + a.o_head_b);
                            cinh[0] = q4 == 0 ? hb4[0] : 0.0f; cinh[1] = q4 == 0 ? hb4[1] : 0.0f; cinh[2] = q4 == 0 ? hb4[2] : 0.0f; cinh[3] = 0.0f;
                        }
                    }
                };
                job<NP, NWV, 136 + t * KH, KH, BIG, 0>(acc, cseld, bsrc_out, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) { prev[p] = acc[p]; cind[p] = cnextd[p]; }
            });
            // colour tile over the view-direction output (rows 0..2): 4 k-steps; the last direction tile (second half of fragment 3) is
            // packed in its first groups, a whole statement per gap
            {
                auto hook = [&](auto ks_c, auto p_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value, p = decltype(p_c)::value;
                    if constexpr (NP == 4) {
                        if constexpr (ks == 0 && p >= 1) pack_whole<true, tile_reg(NP, SIN, p - 1, NT / 2 - 1)>(prev[p - 1]);
                        if constexpr (ks == 1 && p == 1) pack_whole<true, tile_reg(NP, SIN, 3, NT / 2 - 1)>(prev[3]);
                    } else {                                    // >= 4 MFMA issues behind the tile's last MFMA, two groups ahead of k-step 3
                        if constexpr (ks == 1) pack_whole<true, tile_reg(NP, SIN, p, NT / 2 - 1)>(prev[p]);
                    }
                };
                job<NP, NWV, 200, KH / 2, TAIL_USED, TAIL_QUADS - TAIL_USED>(hc, cselh, bsrc_in, aq, smem, ring, lane, hook);
            }
            // the MFMAs are asm statements: hipcc does not know that `hc` is still in flight (XDL write -> vector-memory read)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int p = 0; p < NP; ++p) keep_tuple(hc[p]);     // element 3 is never read: whole tuples stay allocated until here
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (valid[p] && q4 == 0) {                      // cat([rgb, density]) NeRF.py:51
                    f32x4 o; o[0] = hc[p][0]; o[1] = hc[p][1]; o[2] = hc[p][2]; o[3] = dens[p];
                    *(f32x4*)(a.out + out_idx[p] * 4) = o;
                }
        };
        if (l < a.D) { layer_01(l); BSTAMP(2); tail(IC<1>{}); }
        else { BSTAMP(2); tail(IC<0>{}); }
        BSTAMP(3);   // tail
    }
}

// NWV waves per workgroup: 4 (one wave per SIMD: the 64-point shape needs the whole register file) or 8 (two waves per SIMD, 32-point
// shape only: 120 VGPRs + the 128 AGPRs of its fragment file fit twice; the partner wave's MFMAs fill the issue slots a wave loses
// to its DMA issues, packing and LDS waits, at the 64-point shape's 256 points per pass of the weight stream).
template <int W, int LX, int LD, int NPA, int NPB, int NWV>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(NWV / 4, NWV / 4)))
void mlp_bf16_kernel(const MlpArgsB a) {
    static_assert(W == 256, "bf16 variant: W = 256");
    static_assert(NWV == 4 || (NWV == 8 && NPA == 2 && NPB == 0), "two waves per SIMD: the 32-point shape only");
    constexpr int NPM = NPA > NPB ? NPA : NPB;
    // reserve the fragment file (see THE FRAGMENT FILE): the whole accumulation-register file, or its lower half
    if constexpr (NWV == 4) asm volatile("" ::: "a255");
    else asm volatile("" ::: "a127");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* side = (float*)(smem + BRING_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (unsigned i = tid * 4; i < a.side_floats; i += 64 * NWV * 4) *(f32x4*)(side + i) = *(const f32x4*)(a.side + i);
    float* scr_base = side + a.side_floats;
    char* pe_base = (char*)(scr_base + NWV * (NPM / 2) * (W / 2));

    BRing ring;
    ring.sbase = a.stream + wave * (bdma_of(NWV) * QUAD_BYTES);
    ring.voff = lane * 16;
    ring.fetch_off = 0;
    ring.stream_bytes = a.stream_bytes;
    ring.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (bdma_of(NWV) * QUAD_BYTES);
    ring.lds_hi = ring.lds_lo + BRING_BYTES;
    ring.fetch_lds = ring.lds_lo;
    ring.read_slot = BNSLOT - 1;
#pragma unroll
    for (int i = 0; i < bdma_of(NWV); ++i) bring_dma<NWV>(ring, i);       // slot 0
    bring_next_fetch(ring);
#pragma unroll
    for (int i = 0; i < bdma_of(NWV); ++i) bring_dma<NWV>(ring, i);       // slot 1; slot p+2 streams in while slot p is consumed

    u32x4b aq[DA];
    bring_advance<NWV>(ring);                                // also publishes the side tables (barrier)
#pragma unroll
    for (int i = 0; i < DA - 1; ++i) aq[i] = bring_read<NWV>(smem, ring, lane, i);      // position q is read while group q - (DA - 1) computes

#ifdef MN_DIAG
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = bstamp();
    run_phase<W, LX, LD, NPA, NWV>(a, a.ph[0], smem, side, scr_base, pe_base, ring, aq, lane, wave, seg, tprev);
    if constexpr (NPB != 0) run_phase<W, LX, LD, NPB, NWV>(a, a.ph[1], smem, side, scr_base, pe_base, ring, aq, lane, wave, seg, tprev);
#else
    run_phase<W, LX, LD, NPA, NWV>(a, a.ph[0], smem, side, scr_base, pe_base, ring, aq, lane, wave);
    if constexpr (NPB != 0) run_phase<W, LX, LD, NPB, NWV>(a, a.ph[1], smem, side, scr_base, pe_base, ring, aq, lane, wave);
#endif
#ifdef MN_DIAG
    if (a.diag && lane == 0) {
        unsigned long long* d = a.diag + ((size_t)blockIdx.x * NWV + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = seg[i];
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- small coarse launch: render_rays' middle for the two rays this workgroup owns (nerf_process.py:198-203) -----------------------------
    // One 32-point unit per wave, flat walk, 64 samples = two units per ray: waves 4b .. 4b + 3 computed rays 2b and 2b + 1 whole.  Their raw
    // outputs and depths are in global memory (the stores above have completed: vmcnt(0); same CU, same L1: nothing to invalidate at workgroup
    // scope), the weight ring is quiescent (its last DMAs have landed) and becomes the two waves' scratch.  Waves 0 and 1 each run the SAME
    // device functions the stage kernel runs (composite_fine_z_kernel, stages.hip): identical results, one launch and ~4 us fewer per step
    // at the 512-ray shard of an 8-GPU split.  Larger launches leave it to the stage kernel (a wave per ray there, thousands in flight).
    if constexpr (NPA == 2 && NPB == 0 && NWV == 4) {
        if (a.fz_on) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const long long ray = 2ll * blockIdx.x + wave;
            if (wave < 2 && ray < (long long)a.n_rays) {
                float* mine = (float*)smem + wave * (a.S + 2 * (a.S - 1) + a.fz_n2);
                composite_ray<1>(a.out, a.z_out, a.rays, 6, ray, a.S, lane, a.fz_rgb, a.fz_disp, nullptr, a.fz_w, nullptr, mine);
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                fine_z_ray(a.z_out, mine, ray, a.S, a.fz_Nf, a.fz_n2, a.fz_det, a.fz_u, a.fz_zf, nullptr, mine + a.S, lane);
            }
        }
    }
    // The same one step up in size (513..1024 rays on 256 CUs): one 64-point unit per wave and 33..64 coarse samples = one unit per RAY, so every wave
    // owns the ray it computed and does the middle for it (all four waves busy; each its own slice of the quiescent ring).
    if constexpr (NPA == 4 && NPB == 0 && NWV == 4) {
        if (a.fz_on) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const long long ray = 4ll * blockIdx.x + wave;
            if (ray < (long long)a.n_rays) {
                float* mine = (float*)smem + wave * (a.S + 2 * (a.S - 1) + a.fz_n2);
                composite_ray<1>(a.out, a.z_out, a.rays, 6, ray, a.S, lane, a.fz_rgb, a.fz_disp, nullptr, a.fz_w, nullptr, mine);
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                fine_z_ray(a.z_out, mine, ray, a.S, a.fz_Nf, a.fz_n2, a.fz_det, a.fz_u, a.fz_zf, nullptr, mine + a.S, lane);
            }
        }
    }
}

// The walk of one phase: shape NP over the tiles [tile0, tile_end) on a grid of `grid` workgroups.
template <int NP, int NWV>
static PhaseB make_phase(const MlpArgsB& a, long long tile0, long long tile_end, int grid) {
    constexpr int NTL = NP / 2;
    PhaseB ph{};
    ph.tile0 = (unsigned)tile0; ph.tile_end = (unsigned)tile_end;
    const long long NW = (long long)grid * NWV;
    const long long n_units = (tile_end - tile0 + NTL - 1) / NTL;
    const long long n_rays = (tile_end + a.tpr - 1) / a.tpr;     // rays this phase touches when it starts at tile 0
    const long long it_flat = (n_units + NW - 1) / NW, it_ray = ((n_rays + NW - 1) / NW) * (a.tpr / NTL);
    if (tile0 == 0 && a.tpr % NTL == 0 && n_rays >= NW && it_ray <= it_flat) {
        ph.ppr = (unsigned)(a.tpr / NTL);                    // ray-major: every wave gets whole rays (tiles >= tile_end are skipped), unless
        ph.n_iter = (unsigned)it_ray;                        // dealing whole rays would cost a round more than dealing units
    } else {
        ph.ppr = 0;
        ph.n_iter = (unsigned)it_flat;
    }
    return ph;
}

// One launch: phase A of shape NPA over [0, split), then (NPB != 0) phase B of shape NPB over [split, n_wtiles).
template <int NPA, int NPB, int NWV>
static int launch_bf16(MlpArgsB a, long long split, long long n_wtiles, hipStream_t st) {
    constexpr int NPM = NPA > NPB ? NPA : NPB;
    const size_t lds = BRING_BYTES + (size_t)a.side_floats * 4 + (size_t)NWV * (NPM / 2) * (256 / 2) * 4 + (size_t)NWV * NPM * enc_ksteps32(10) * QUAD_BYTES;
    MN_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded: %zu bytes", lds);
    auto kern = mlp_bf16_kernel<256, 10, 4, NPA, NPB, NWV>;
    static LdsOptIn opt_in = {};
    if (int rc = ensure_lds_opt_in(opt_in, (const void*)kern)) return rc;
    const int n_cus = device_cus();
    const long long wg_a = ((split + NPA / 2 - 1) / (NPA / 2) + NWV - 1) / NWV;
    constexpr int TPB = NPB ? NPB / 2 : 1;                    // tiles per unit of the second phase (1: no second phase, wg_b unused)
    const long long wg_b = NPB ? ((n_wtiles - split + TPB - 1) / TPB + NWV - 1) / NWV : 0;
    const long long n_wg = wg_a > wg_b ? wg_a : wg_b;
    const int grid = (int)(n_wg < n_cus ? n_wg : n_cus);
    a.ph[0] = make_phase<NPA, NWV>(a, 0, split, grid);
    if constexpr (NPB != 0) a.ph[1] = make_phase<NPB, NWV>(a, split, n_wtiles, grid);
#ifdef MN_DIAG
    {   // diagnostic build: run once with stamps and print the per-segment averages (cycles per unit per wave; single-phase launches)
        unsigned long long* dbuf = nullptr;
        const size_t n = (size_t)grid * NWV * 8;
        MN_HIP(hipMalloc(&dbuf, n * 8));
        MN_HIP(hipMemsetAsync(dbuf, 0, n * 8, st));
        a.diag = dbuf;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWV), lds, st, a);
        MN_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> hbuf(n);
        MN_HIP(hipMemcpy(hbuf.data(), dbuf, n * 8, hipMemcpyDeviceToHost));
        (void)hipFree(dbuf);
        static const char* names[4] = {"prologue", "layer0", "trunk", "tail"};
        const double ideal[4] = {0, 512.0 * NPA, (7 * 2048 + 512.0) * NPA, 204 * 16.0 * NPA};
        double tot = 0;
        const unsigned n_iter = a.ph[0].n_iter + (NPB ? a.ph[1].n_iter : 0);
        fprintf(stderr, "[mn_diag bf16] NP=%d(+%d) grid=%d units/wave=%u  cycles per unit (mean over waves; ideal MFMA cycles of the first shape in brackets):\n",
                NPA, NPB, grid, n_iter);
        for (int sgi = 0; sgi < 4; ++sgi) {
            double sum = 0;
            for (size_t w = 0; w < (size_t)grid * NWV; ++w) sum += (double)hbuf[w * 8 + sgi];
            const double per = sum / ((double)grid * NWV) / (double)n_iter;
            tot += per;
            fprintf(stderr, "[mn_diag bf16]   %-10s %10.0f  [%6.0f]\n", names[sgi], per, ideal[sgi]);
        }
        fprintf(stderr, "[mn_diag bf16]   %-10s %10.0f  [%6.0f]\n", "total", tot, ideal[1] + ideal[2] + ideal[3]);
        return MI_NERF_OK;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWV), lds, st, a);
    MN_LAUNCH_CHECK("mlp_bf16_kernel");
    return MI_NERF_OK;
}

// points_per_wave: 0 = chosen per launch (pick_np), 64 / 32 = forced (A/B measurements, parity tests of each shape)
// z_dev == NULL (strat != NULL): the kernel draws the stratified depths of render_rays' coarse pass itself and writes them to strat->z_out
int mlp_rays_bf16(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev, int64_t n_rays, int S,
                  float* raw_dev, hipStream_t st, int points_per_wave, const StratDraw* strat, FineDraw* fine) {
    if (fine) fine->taken = false;
    if (int rc = check_net_bf16(net)) return rc;
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    MN_CHECK_ARG(points_per_wave == 0 || points_per_wave == 32 || points_per_wave == 64 || points_per_wave == 832,
                 "points_per_wave must be 0 (auto), 32, 64 or 832 (8 waves of 32) (got %d)", points_per_wave);
    if (n_rays == 0) return MI_NERF_OK;
    MN_CHECK_ARG(packed_dev && rays_dev && raw_dev && (z_dev || (strat && strat->z_out)), "NULL device pointer");
    const BlobLayoutBf16 L = make_layout_bf16(net->D, net->W, net->skip, net->L_x, net->L_d);
    MlpArgsB a{};
    if (!z_dev) {
        a.z_out = strat->z_out; a.strat_near = strat->near_; a.strat_far = strat->far_;
        a.strat_step = S > 1 ? 1.0f / (float)(S - 1) : 0.0f;
        a.strat_jitter = Jitter{strat->t_rand, strat->seed, 0u, (long long)strat->ray0};
    }
    a.stream = (const char*)packed_dev + L.stream_off;
    a.side = (const float*)((const char*)packed_dev + L.side_off);
    a.rays = rays_dev; a.z = z_dev; a.out = raw_dev;
    a.S = S; a.tpr = (S + 31) / 32;
    const long long n_wtiles = (long long)n_rays * a.tpr;
    MN_CHECK_ARG(n_wtiles < (1LL << 30), "too many points for one launch: %lld rays x %d samples", (long long)n_rays, S);
    a.n_rays = (unsigned)n_rays;
    a.D = net->D;
    a.skip_layer = (net->skip >= 0 && net->skip + 1 < net->D) ? net->skip + 1 : -1;
    a.stream_bytes = L.stream_bytes; a.side_floats = L.side_floats;
    a.o_bias_trunk = L.bias_trunk; a.o_bias_feat = L.bias_feat; a.o_bias_d = L.bias_d; a.o_head_b = L.head_b; a.o_wdir_t = L.wdir_t;
    if (points_per_wave == 64) return launch_bf16<4, 0, 4>(a, n_wtiles, n_wtiles, st);
    if (points_per_wave == 32) return launch_bf16<2, 0, 4>(a, n_wtiles, n_wtiles, st);
    // (An 8-wave shape -- 32 points per wave, two waves per SIMD -- was measured SLOWER than the 64-point shape at every size: 4096 rays, fine
    // launch 638 vs 603 us, profiles/r03_bf16_two_waves_per_simd.txt.  The kernel is not short of latency hiding, it is short of power: twice
    // the LDS reads per FLOP cost more clock than the interleaving wins back.  Its instantiation is gone: tools/ABLATIONS.md.)
    MN_CHECK_ARG(points_per_wave != 832, "the 8-wave shape (832) was an experiment and is not built (tools/ABLATIONS.md)");
    // The launch plan.  A pass of the 64-point shape takes the same time whatever the number of active CUs (the kernel is bound by
    // what ONE CU does per pass), a pass of the 32-point shape ~0.65 of it (half the matrix work, the same weight stream:
    // profiles/r03_bf16_small_launch_shape.txt).  So: whole rounds of the 64-point shape (every wave of the chip a pair of tiles),
    // then the remainder as ONE round of the 32-point shape if it fits one (every wave one tile), else as one more 64-point round;
    // both phases in ONE launch (the weight ring streams on across the phase boundary).
    // A 512-ray shard of BASELINE config #5 (8 GPUs): coarse 1024 tiles = one 32-point round (was half the chip for a full pass);
    // fine 3072 tiles = one 64-point round + one 32-point round (was two full passes, the second half empty).
    const long long round4 = (long long)device_cus() * 4 * 2, round2 = (long long)device_cus() * 4;
    // render_rays' middle in the coarse launch's epilogue (`fine` offered; a coarse pass of 33..64 samples; one unit per wave, flat walk -- which is what
    // the epilogue's ray numbering assumes): `slices` waves of a workgroup each take a ray and a slice of the ring as scratch
    auto take_middle = [&](int slices) {
        if (!(fine && !z_dev && a.tpr == 2 && fine->Nf >= 1 && S >= 3)) return;
        int n2 = 2;
        while (n2 < S + fine->Nf) n2 <<= 1;
        const size_t scratch = (size_t)slices * (S + 2 * (S - 1) + n2) * sizeof(float);
        if (scratch > (size_t)BRING_BYTES || n2 > 512) return;
        a.fz_on = 1; a.fz_Nf = fine->Nf; a.fz_n2 = n2; a.fz_det = fine->det;
        a.fz_u = Jitter{fine->u, fine->seed, 1u, (long long)fine->ray0};
        a.fz_rgb = fine->rgb_c; a.fz_disp = fine->disp_c; a.fz_w = fine->w_c; a.fz_zf = fine->z_f;
        fine->taken = true;
    };
    const long long main_tiles = (n_wtiles / round4) * round4, rem = n_wtiles - main_tiles;
    if (rem == 0 || rem > round2) {
        // at most one 64-point unit per wave and one unit per ray (33..64 samples): every wave owns the ray it computes (513..1024 rays on 256 CUs)
        if (n_wtiles <= round4) take_middle(4);
        return launch_bf16<4, 0, 4>(a, n_wtiles, n_wtiles, st);
    }
    if (main_tiles == 0) {
        // one round of 32-point units, one unit per wave.  A coarse pass of 33..64 samples is two units per ray, so a workgroup's four waves hold
        // two rays whole: it takes render_rays' middle for them too when the caller offers it (`fine`; the kernel's epilogue).  The grid is one
        // workgroup per four tiles and the walk flat (n_iter == 1: make_phase), which is what the epilogue's ray = 2 * block + wave assumes.
        take_middle(2);
        return launch_bf16<2, 0, 4>(a, n_wtiles, n_wtiles, st);
    }
    return launch_bf16<4, 2, 4>(a, main_tiles, n_wtiles, st);
}

}  // namespace minerf
