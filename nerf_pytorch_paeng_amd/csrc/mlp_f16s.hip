// mlp_f16s.hip -- SPLIT-PRECISION variant of the fused positional-encoding + NeRF MLP forward: fp32-grade results on the f16 matrix pipe.
//
// gfx950 has no reduced-precision fast path for f32 inputs (no xf32): the f32-input MFMA runs at the fp32 vector rate, 1/16 of the
// f16 / bf16 one.  This kernel buys most of that factor back WITHOUT giving up fp32 accuracy: every operand is carried as an f16 pair
//      x = x_hi + x_lo * 2^-11,        x_hi = f16(x),   x_lo = f16((x - x_hi) * 2^11)
// (11 + 11 significand bits; the scale keeps the low part out of the f16 subnormals), and a product W x is three MFMAs
//      acc_hi += W_hi x_hi;      acc_lo += W_hi x_lo;      acc_lo += W_lo x_hi;         y = acc_hi + acc_lo * 2^-11
// with fp32 accumulation -- every f16 x f16 product is exact in fp32, the dropped W_lo x_lo term is 2^-22 of the product.  Measured
// against an fp64 evaluation of the same network the result is as close as the fp32 kernel's (oracle/restate.py mlp_forward_f16split is
// the CPU restatement; tests/test_gpu_parity.py holds the kernel to the fp32 path's own bars).  It is an EXTRA precision mode like the
// bf16 variant, never the headline: bench.py reports it as its own leg against both peaks.
//
// Structure: csrc/mlp_bf16.hip's (read its header first) -- output-tile-major jobs, activations never leave registers, the packed
// accumulators of layer l are the B fragments of layer l+1, the whole AGPR file is a hand-numbered fragment file, weights stream
// L2 -> LDS through a 3 x 32 KiB ring by LDS-DMA, heads on the matrix pipe -- with these differences:
//   * two fragments per operand (hi, lo): the fragment file holds 2 sets x 2 point tiles x 8 fragments x 2 parts x 4 registers = all
//     256 AGPRs, so a wave carries 32 points (2 point tiles of 16), a workgroup 128 per pass of the stream;
//   * the stream holds (hi, lo) quad PAIRS (2.4 MB per network); a group is one k-step = six MFMAs
//         [hi.hi p0] [hi.hi p1] [hi.lo p0] [hi.lo p1] [lo.hi p0] [lo.hi p1]
//     the first two gaps carry the ring work (one A read each, the slot's DMAs), the other four the packing;
//   * packing a pair of accumulator elements (12 VALU): y = fma(lo, 2^-11, hi); Y = fma(hi, 2^11, lo) (= 2^11 y); hi_pk =
//     v_cvt_pk_f16_f32(y0, y1) [v_pk_maximum3_f16 for ReLU, Y = maximum(Y, 0): NaN-propagating]; lo halves by v_fma_mixlo/hi_f16(hi_half, -2^11, Y): the
//     residual (y - hi) 2^11 computed exactly and rounded once; two v_accvgpr_write;
//   * gamma(x) is evaluated in full precision per channel (Cody-Waite + Cephes like the fp32 kernel, libm beyond 4e6 rad), not by
//     angle doubling: this variant's contract is fp32-grade output.
// Ranges: |weights| and |activations| must stay below the f16 maximum (65 504).  The packer refuses larger weights; an activation beyond it
// comes out as NaN in every output that depends on it, never as a finite value (NaN-propagating ReLU: mlp_f16s_core.h "RANGE CONTRACT").

#include "mlp_f16s_core.h"

namespace minerf {
namespace f16s {

// ---------------------------------------------------------------------------------------------
// host: packer
// ---------------------------------------------------------------------------------------------
static inline uint16_t f32_to_f16_rne(float x) {            // finite |x| < 65520 (checked by the caller)
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);                           // >= 65536: inf (not reached)
    if (a < 0x38800000u) {                                                              // below 2^-14: subnormal half (or zero)
        if (a < 0x33000000u) return (uint16_t)sign;                                     // below 2^-25: zero
        const uint32_t m = (a & 0x7FFFFFu) | 0x800000u;
        const int shift = 126 - (int)(a >> 23);                                         // 14 .. 24
        const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t r = a + 0xFFFu + ((a >> 13) & 1u);                                   // round to nearest even at bit 13
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}
static inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { int k = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++k; } u = sign | ((uint32_t)(113 - k) << 23) | ((mm & 0x3FFu) << 13); }
    } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// the two halves of a weight: w = hi + lo * 2^-11
static inline void split_weight(float w, uint16_t& hi, uint16_t& lo) {
    hi = f32_to_f16_rne(w);
    lo = f32_to_f16_rne((w - f16_to_f32(hi)) * SC_UP);
}

// One quad PAIR: the (hi, lo) A fragments of output rows row0..row0+15 for the 32 input columns cols[q*8 + j] (-1: zero).
// rowmap (optional, 16 entries): weight-matrix row feeding output row i of the tile, -1: zero row.
// The stream is built as VALUES first (the source weight at its hi AND its lo position: element i of a 1024-element pair block is a hi
// half for i < 512, a lo half otherwise), so that the same routine over index-valued weights yields the device packer's gather map.
static void emit_pair(std::vector<float>& st, const float* Wm, int n_out, int n_in, int row0, const int* rowmap, const int* cols) {
    const size_t base = st.size();
    st.resize(base + 2 * 512, 0.0f);
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const int col = cols[(lane >> 4) * 8 + j];
            const int n = rowmap ? rowmap[lane & 15] : row0 + (lane & 15);
            const float w = (col >= 0 && n >= 0 && n < n_out) ? Wm[(size_t)n * n_in + col] : 0.0f;
            st[base + lane * 8 + j] = w;
            st[base + 512 + lane * 8 + j] = w;
        }
}
static std::vector<int> enc_cols(int L, int base) {           // L: the network's own frequencies; the k-step count is the kernel's
    const int nch = 3 + 6 * L;
    std::vector<int> c(KPE * KF);
    for (int u = 0; u < KPE * KF; ++u) c[u] = u < nch ? base + u : -1;
    return c;
}
// input columns in the order the packed accumulators present them: fragment s, lane quarter q, element j
static std::vector<int> act_cols(int W, int base) {
    std::vector<int> c;
    for (int s = 0; s < W / KF; ++s)
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 8; ++j) c.push_back(base + MT * (2 * s + (j >> 2)) + 4 * q + (j & 3));
    return c;
}
static void emit_layer(std::vector<float>& st, const float* Wm, int n_out, int n_in, int nt, const std::vector<int>& cols) {
    const int KS = (int)cols.size() / KF;
    for (int tile = 0; tile < nt; ++tile)
        for (int ks = 0; ks < KS; ++ks) emit_pair(st, Wm, n_out, n_in, MT * tile, nullptr, cols.data() + KF * ks);
}

static int check_weights(const float* w, size_t n, const char* name) {
    for (size_t i = 0; i < n; ++i) MN_CHECK_ARG(w[i] == w[i] && (w[i] < 0 ? -w[i] : w[i]) < 65504.0f, "%s[%zu] = %g does not fit the f16-split variant", name, i, (double)w[i]);
    return MI_NERF_OK;
}

}  // namespace f16s

size_t packed_bytes_f16s(const mi_nerf_net* net) {
    if (f16s::check_net(net)) return 0;
    return f16s::make_layout(net->D, net->W, net->skip).total_bytes;
}

namespace f16s {
// value stream of a network (see emit_pair) and its side tables
static int build_stream(const mi_nerf_net* net, const mi_nerf_params* p, const BlobLayoutS& L, std::vector<float>& st) {
    const int D = net->D, W = net->W;
    const int in_x = 3 + 6 * net->L_x, in_d = 3 + 6 * net->L_d;
    st.clear();
    st.reserve(L.stream_bytes / 2);
    emit_layer(st, p->linear_x_w[0], W, in_x, NT, enc_cols(net->L_x, 0));
    for (int l = 1; l < D; ++l) {
        const bool cat = (net->skip >= 0 && l == net->skip + 1);
        std::vector<int> cols = act_cols(W, cat ? in_x : 0);             // input columns are cat([gamma(x), h]), NeRF.py:41 ...
        if (cat) {                                                       // ... consumed activations first, gamma(x) last
            const std::vector<int> enc = enc_cols(net->L_x, 0);
            cols.insert(cols.end(), enc.begin(), enc.end());
        }
        emit_layer(st, p->linear_x_w[l], W, cat ? W + in_x : W, NT, cols);
    }
    // tail: feature layer | density tile over the trunk output | view-direction layer | colour tile | padding
    emit_layer(st, p->linear_feat_w, W, W, NT, act_cols(W, 0));
    int rowmap[MT];
    {
        const std::vector<int> act = act_cols(W, 0);
        for (int i = 0; i < MT; ++i) rowmap[i] = (i == 3) ? 0 : -1;     // output row 3 <- linear_density row 0
        for (int ks = 0; ks < W / KF; ++ks) emit_pair(st, p->linear_density_w, 1, W, 0, rowmap, act.data() + KF * ks);
    }
    emit_layer(st, p->linear_d_w, W / 2, W + in_d, NT / 2, act_cols(W, 0));
    {
        const std::vector<int> act = act_cols(W / 2, 0);
        for (int i = 0; i < MT; ++i) rowmap[i] = (i < 3) ? i : -1;      // output rows 0..2 <- linear_color rows 0..2
        for (int ks = 0; ks < W / 2 / KF; ++ks) emit_pair(st, p->linear_color_w, 3, W / 2, 0, rowmap, act.data() + KF * ks);
    }
    st.resize(st.size() + (size_t)(TAIL_PAIRS - TAIL_USED_P) * 2 * 512, 0.0f);
    MN_CHECK_ARG(st.size() * 2 == L.stream_bytes, "internal: f16-split stream %zu != %u", st.size() * 2, L.stream_bytes);
    return MI_NERF_OK;
}
static void fill_side(const mi_nerf_net* net, const mi_nerf_params* p, const BlobLayoutS& L, float* side) {
    const int D = net->D, W = net->W, in_d = 3 + 6 * net->L_d;
    for (int l = 0; l < D; ++l) memcpy(side + L.bias_trunk + (size_t)l * W, p->linear_x_b[l], W * 4);
    memcpy(side + L.bias_feat, p->linear_feat_b, W * 4);
    memcpy(side + L.bias_d, p->linear_d_b, (W / 2) * 4);
    memcpy(side + L.head_b, p->linear_color_b, 3 * 4);
    side[L.head_b + 3] = p->linear_density_b[0];
    for (int f = 0; f < in_d; ++f)
        for (int n = 0; n < W / 2; ++n) side[L.wdir_t + (size_t)f * (W / 2) + n] = p->linear_d_w[(size_t)n * (W + in_d) + W + f];
}
static void fill_header(const mi_nerf_net* net, const BlobLayoutS& L, uint32_t* hdr) {
    memset(hdr, 0, HEADER_BYTES);
    hdr[0] = BLOB_MAGIC; hdr[1] = 4; hdr[2] = net->D; hdr[3] = net->W; hdr[4] = (uint32_t)net->skip; hdr[5] = KERNEL_LX; hdr[6] = KERNEL_LD;
    hdr[7] = L.stream_off; hdr[8] = L.stream_bytes; hdr[9] = L.stream_bytes; hdr[10] = L.side_off; hdr[11] = L.side_floats;
    hdr[12] = 2; hdr[13] = net->L_x; hdr[14] = net->L_d;
}
}  // namespace f16s

int pack_f16s(const mi_nerf_net* net, const mi_nerf_params* p, void* blob, size_t blob_bytes) {
    using namespace f16s;
    if (int rc = check_net(net)) return rc;
    const int D = net->D, W = net->W;
    const int in_x = 3 + 6 * net->L_x, in_d = 3 + 6 * net->L_d;
    const BlobLayoutS L = make_layout(D, W, net->skip);
    MN_CHECK_ARG(blob_bytes >= L.total_bytes, "blob too small: %zu < %u", blob_bytes, L.total_bytes);
    for (int l = 0; l < D; ++l) {
        const int n_in = l == 0 ? in_x : ((net->skip >= 0 && l == net->skip + 1) ? W + in_x : W);
        if (int rc = check_weights(p->linear_x_w[l], (size_t)W * n_in, "linear_x.weight")) return rc;
    }
    if (int rc = check_weights(p->linear_feat_w, (size_t)W * W, "linear_feat.weight")) return rc;
    if (int rc = check_weights(p->linear_density_w, W, "linear_density.weight")) return rc;
    if (int rc = check_weights(p->linear_d_w, (size_t)(W / 2) * (W + in_d), "linear_d.weight")) return rc;
    if (int rc = check_weights(p->linear_color_w, (size_t)3 * (W / 2), "linear_color.weight")) return rc;
    memset(blob, 0, L.total_bytes);
    std::vector<float> st;
    if (int rc = build_stream(net, p, L, st)) return rc;
    fill_header(net, L, (uint32_t*)blob);
    uint16_t* out = (uint16_t*)((char*)blob + L.stream_off);
    for (size_t i = 0; i < st.size(); ++i) {
        uint16_t hi, lo;
        split_weight(st[i], hi, lo);
        out[i] = (i & 1023) < 512 ? hi : lo;
    }
    fill_side(net, p, L, (float*)((char*)blob + L.side_off));
    return MI_NERF_OK;
}

// Device-side re-pack (the training path packs after every optimizer step): gather map over the FLAT parameter vector
// (module.parameters() order, layout.h make_param_offsets), built by running the routines above over index-valued weights.
// map[i], i < stream elements: 1 + flat index feeding stream element i (0: zero); then the side table's floats.
size_t pack_map_f16s_len(const mi_nerf_net* net) {
    if (f16s::check_net(net)) return 0;
    const f16s::BlobLayoutS L = f16s::make_layout(net->D, net->W, net->skip);
    return (size_t)L.stream_bytes / 2 + L.side_floats;
}
int pack_map_f16s(const mi_nerf_net* net, int32_t* map, size_t map_len) {
    using namespace f16s;
    if (int rc = check_net(net)) return rc;
    const int D = net->D, W = net->W;
    const BlobLayoutS L = make_layout(D, W, net->skip);
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

// ---------------------------------------------------------------------------------------------
// backward-data stream (dgrad_f16s_kernel): the TRANSPOSED weights as (hi, lo) quad pairs in the order the chain
//   d hidden -> linear_d^T (feature block) -> linear_feat^T -> linear_x[D-1]^T ... linear_x[1]^T (activation block)
// consumes them (pack.cpp pack_bwd_fp32 is the fp32 counterpart).  Blob = header | stream; the colour / density head weights are read from
// the fp32 forward blob's side tables like mlp_dgrad_kernel does.
// ---------------------------------------------------------------------------------------------
namespace f16s {
// quad pair of a transposed GEMM: output rows = forward INPUT columns in_base + row0 .. +15, k = forward OUTPUT rows cols[..] (-1: zero)
static void emit_pair_t(std::vector<float>& st, const float* Wm, int n_out_fwd, int n_in_fwd, int in_base, int row0, const int* cols) {
    const size_t base = st.size();
    st.resize(base + 2 * 512, 0.0f);
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const int col = cols[(lane >> 4) * 8 + j];
            const int n = in_base + row0 + (lane & 15);
            const float w = (col >= 0 && col < n_out_fwd && n < n_in_fwd) ? Wm[(size_t)col * n_in_fwd + n] : 0.0f;
            st[base + lane * 8 + j] = w;
            st[base + 512 + lane * 8 + j] = w;
        }
}
static void emit_layer_t(std::vector<float>& st, const float* Wm, int n_out_fwd, int n_in_fwd, int in_base, const std::vector<int>& cols) {
    const int KS = (int)cols.size() / KF;
    for (int tile = 0; tile < NT; ++tile)
        for (int ks = 0; ks < KS; ++ks) emit_pair_t(st, Wm, n_out_fwd, n_in_fwd, in_base, MT * tile, cols.data() + KF * ks);
}
static int build_stream_bwd(const mi_nerf_net* net, const mi_nerf_params* p, std::vector<float>& st) {
    const int D = net->D, W = net->W;
    const int in_x = 3 + 6 * net->L_x, in_d = 3 + 6 * net->L_d;
    st.clear();
    st.reserve(bwd_stream_bytes_s(D) / 2);
    emit_layer_t(st, p->linear_d_w, W / 2, W + in_d, 0, act_cols(W / 2, 0));         // d feature = Wd[:, :W]^T d hidden
    emit_layer_t(st, p->linear_feat_w, W, W, 0, act_cols(W, 0));
    for (int l = D - 1; l >= 1; --l) {
        const bool cat = (net->skip >= 0 && l == net->skip + 1);
        emit_layer_t(st, p->linear_x_w[l], W, cat ? W + in_x : W, cat ? in_x : 0, act_cols(W, 0));
    }
    MN_CHECK_ARG(st.size() * 2 == bwd_stream_bytes_s(D), "internal: f16-split backward stream %zu != %u", st.size() * 2, bwd_stream_bytes_s(D));
    return MI_NERF_OK;
}
static void fill_header_bwd(const mi_nerf_net* net, uint32_t* hdr) {
    memset(hdr, 0, HEADER_BYTES);
    hdr[0] = BLOB_MAGIC; hdr[1] = 5; hdr[2] = net->D; hdr[3] = net->W; hdr[4] = (uint32_t)net->skip; hdr[5] = net->L_x; hdr[6] = net->L_d;
    hdr[7] = HEADER_BYTES; hdr[8] = bwd_stream_bytes_s(net->D);
}
}  // namespace f16s
size_t packed_bytes_bwd_f16s(const mi_nerf_net* net) {
    if (f16s::check_net(net)) return 0;
    return HEADER_BYTES + (size_t)f16s::bwd_stream_bytes_s(net->D);
}
int pack_bwd_f16s(const mi_nerf_net* net, const mi_nerf_params* p, void* blob, size_t blob_bytes) {
    using namespace f16s;
    if (int rc = check_net(net)) return rc;
    const size_t total = packed_bytes_bwd_f16s(net);
    MN_CHECK_ARG(blob_bytes >= total, "blob too small: %zu < %zu", blob_bytes, total);
    std::vector<float> st;
    if (int rc = build_stream_bwd(net, p, st)) return rc;
    for (size_t i = 0; i < st.size(); ++i) MN_CHECK_ARG(st[i] == st[i] && (st[i] < 0 ? -st[i] : st[i]) < 65504.0f, "a weight (%g) does not fit the f16-split variant", (double)st[i]);
    fill_header_bwd(net, (uint32_t*)blob);
    uint16_t* out = (uint16_t*)((char*)blob + HEADER_BYTES);
    for (size_t i = 0; i < st.size(); ++i) {
        uint16_t hi, lo;
        split_weight(st[i], hi, lo);
        out[i] = (i & 1023) < 512 ? hi : lo;
    }
    return MI_NERF_OK;
}
size_t pack_map_bwd_f16s_len(const mi_nerf_net* net) {
    if (f16s::check_net(net)) return 0;
    return (size_t)f16s::bwd_stream_bytes_s(net->D) / 2;
}
int pack_map_bwd_f16s(const mi_nerf_net* net, int32_t* map, size_t map_len) {
    using namespace f16s;
    if (int rc = check_net(net)) return rc;
    const int D = net->D, W = net->W;
    const ParamOffsets po = make_param_offsets(D, W, net->skip, net->L_x, net->L_d);
    MN_CHECK_ARG(po.total < (1u << 24), "network too large for the index map (%u parameters)", po.total);
    const size_t n_stream = pack_map_bwd_f16s_len(net);
    MN_CHECK_ARG(map && map_len >= n_stream, "map too small: %zu entries for %zu", map_len, n_stream);
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
    if (int rc = build_stream_bwd(net, &p, st)) return rc;
    for (size_t i = 0; i < n_stream; ++i) map[i] = (int32_t)st[i];
    return MI_NERF_OK;
}

namespace f16s {
struct HeaderWordsS { uint32_t w[HEADER_BYTES / 4]; };
__global__ __launch_bounds__(256) void pack_apply_f16s_kernel(const int32_t* __restrict__ map, const float* __restrict__ flat, unsigned n_stream,
                                                               unsigned n_side, unsigned stream_off, unsigned side_off, HeaderWordsS hdr,
                                                               char* __restrict__ blob, unsigned* __restrict__ bad) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < HEADER_BYTES / 4) ((uint32_t*)blob)[i] = hdr.w[i];
    if (i < n_stream) {
        const int32_t m = map[i];
        const float w = m ? flat[m - 1] : 0.0f;
        if (!(__builtin_fabsf(w) < 65504.0f) && bad) atomicAdd(bad, 1u);          // NaN or beyond the f16 range: the host packer refuses these
        const _Float16 hi = (_Float16)w;                                          // round to nearest even, like f32_to_f16_rne
        const _Float16 lo = (_Float16)((w - (float)hi) * SC_UP);
        ((_Float16*)(blob + stream_off))[i] = (i & 1023u) < 512u ? hi : lo;
    } else if (i < n_stream + n_side) {
        const int32_t m = map[i];
        ((float*)(blob + side_off))[i - n_stream] = m ? flat[m - 1] : 0.0f;
    }
}
}  // namespace f16s
int pack_apply_f16s(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_dev, void* blob_dev, size_t blob_bytes, unsigned* bad_dev, hipStream_t st) {
    using namespace f16s;
    if (int rc = check_net(net)) return rc;
    const BlobLayoutS L = make_layout(net->D, net->W, net->skip);
    MN_CHECK_ARG(map_dev && flat_dev && blob_dev, "NULL device pointer");
    MN_CHECK_ARG(blob_bytes >= L.total_bytes && ((uintptr_t)blob_dev & 15) == 0, "blob too small (%zu < %u) or not 16-byte aligned", blob_bytes, L.total_bytes);
    HeaderWordsS h;
    fill_header(net, L, h.w);
    const unsigned n_stream = L.stream_bytes / 2, total = n_stream + L.side_floats;
    hipLaunchKernelGGL(pack_apply_f16s_kernel, dim3((total + 255) / 256), dim3(256), 0, st, map_dev, flat_dev, n_stream, L.side_floats, L.stream_off,
                       L.side_off, h, (char*)blob_dev, bad_dev);
    MN_LAUNCH_CHECK("pack_apply_f16s_kernel");
    return MI_NERF_OK;
}

int pack_apply_bwd_f16s(const mi_nerf_net* net, const int32_t* map_dev, const float* flat_dev, void* blob_dev, size_t blob_bytes, unsigned* bad_dev, hipStream_t st) {
    using namespace f16s;
    if (int rc = check_net(net)) return rc;
    const size_t total_bytes = packed_bytes_bwd_f16s(net);
    MN_CHECK_ARG(map_dev && flat_dev && blob_dev, "NULL device pointer");
    MN_CHECK_ARG(blob_bytes >= total_bytes && ((uintptr_t)blob_dev & 15) == 0, "blob too small (%zu < %zu) or not 16-byte aligned", blob_bytes, total_bytes);
    HeaderWordsS h;
    fill_header_bwd(net, h.w);
    const unsigned n_stream = bwd_stream_bytes_s(net->D) / 2;
    hipLaunchKernelGGL(pack_apply_f16s_kernel, dim3((n_stream + 255) / 256), dim3(256), 0, st, map_dev, flat_dev, n_stream, 0u, (unsigned)HEADER_BYTES,
                       (unsigned)total_bytes, h, (char*)blob_dev, bad_dev);
    MN_LAUNCH_CHECK("pack_apply_f16s_kernel");
    return MI_NERF_OK;
}

int mlp_rays_f16s(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev, int64_t n_rays, int S,
                  float* raw_dev, hipStream_t st) {
    return launch_f16s<false>(net, packed_dev, rays_dev, z_dev, n_rays, S, raw_dev, nullptr, st);
}
}  // namespace minerf
