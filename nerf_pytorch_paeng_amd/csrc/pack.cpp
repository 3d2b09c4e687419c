// pack.cpp -- host-side weight packer: reference checkpoint layout -> kernel streaming order.
//
// Input: one NeRFModule's parameters as the reference stores them ([out,in] row-major fp32;
// model/NeRF.py:24-30; checkpoint keys model_{coarse,fine}.* from train.py:105-114).
// Output: the blob described in layout.h (header | MFMA A-operand stream | natural-order side tables).
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
#include "common.h"
#include "layout.h"

namespace minerf {

namespace {

struct KStep { int lo, hi; };   // input column fed by lane half 0 / 1; -1 = zero pad

// k-steps of the KERNEL's encoding (LK frequencies) over a network input gamma_L(.) (L <= LK) occupying columns [base, base+3+6L):
// the frequencies the network does not have get zero weights
std::vector<KStep> enc_ksteps(int LK, int L, int base) {
    std::vector<KStep> ks;
    for (int s = 0; s < 3 * LK; ++s) {
        const int k = s / 3, c = s % 3;
        if (k < L) ks.push_back({base + 3 + 6 * k + c, base + 3 + 6 * k + 3 + c});   // (sin, cos) of 2^k p_c
        else ks.push_back({-1, -1});
    }
    ks.push_back({base + 0, base + 1});
    ks.push_back({base + 2, -1});
    while ((int)ks.size() < pe_ksteps(LK)) ks.push_back({-1, -1});
    return ks;
}

// k-steps over a W-wide activation held in accumulator layout, columns [base, base+W); of those only the first n_real exist in the
// network (narrower than the kernel: layout.h kernel_width) -- the others are the padded units, whose weights are zero
std::vector<KStep> act_ksteps(int W, int base, int n_real = -1) {
    if (n_real < 0) n_real = W;
    std::vector<KStep> ks;
    for (int t = 0; t < W / 32; ++t)
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * t + (r & 3) + 8 * (r >> 2);
            ks.push_back({f < n_real ? base + f : -1, f + 4 < n_real ? base + f + 4 : -1});
        }
    return ks;
}

// ---- the W16 stream order (mlp_fp32_wide.hip, v_mfma_f32_16x16x4_f32): a k-step feeds four input columns, one per lane quarter -------------
struct KStep4 { int c[4]; };    // input column fed by lane quarter 0..3; -1 = zero pad

std::vector<KStep4> enc_ksteps16(int LK, int L, int base) {
    std::vector<KStep4> ks;
    const int combos = 3 * LK;
    for (int s = 0; s < (combos + 1) / 2; ++s) {
        KStep4 k{};
        for (int q = 0; q < 4; ++q) {
            const int m = 2 * s + (q >> 1);
            k.c[q] = (m < combos && m / 3 < L) ? base + 3 + 6 * (m / 3) + (m % 3) + 3 * (q & 1) : -1;
        }
        ks.push_back(k);
    }
    ks.push_back(KStep4{{base + 0, base + 1, base + 2, -1}});
    while ((int)ks.size() < pe_ksteps16(LK)) ks.push_back(KStep4{{-1, -1, -1, -1}});
    return ks;
}

// a W-wide activation in the 16x16 accumulator layout: register r of output tile t on lane quarter q holds feature 16t + 4q + r, and k-step
// 4t + r of the next layer multiplies those four
std::vector<KStep4> act_ksteps16(int W, int base, int n_real) {
    std::vector<KStep4> ks;
    for (int t = 0; t < W / 16; ++t)
        for (int r = 0; r < 4; ++r) {
            KStep4 k{};
            for (int q = 0; q < 4; ++q) {
                const int f = 16 * t + 4 * q + r;
                k.c[q] = f < n_real ? base + f : -1;
            }
            ks.push_back(k);
        }
    return ks;
}

// quad(T, kq): float [64 lanes][4]: lane l = (i = l & 15: output feature 16T + i; quarter l >> 4), element j: k-step 4kq + j
void emit_part16(std::vector<float>& stream, const float* Wm, int n_out, int n_in, int NT, const std::vector<KStep4>& ks) {
    const int KQ = (int)ks.size() / 4;
    for (int kq = 0; kq < KQ; ++kq)
        for (int T = 0; T < NT; ++T)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int col = ks[4 * kq + j].c[lane >> 4];
                    const int n = 16 * T + (lane & 15);
                    stream.push_back((col >= 0 && n < n_out) ? Wm[(size_t)n * n_in + col] : 0.0f);
                }
    const size_t slot_floats = SLOT_BYTES / 4;
    while (stream.size() % slot_floats) stream.push_back(0.0f);
}

// Append one GEMM part to the stream: quads in (kq, T) order, padded to a whole number of slots.
void emit_part(std::vector<float>& stream, const float* Wm, int n_out, int n_in, int NT, const std::vector<KStep>& ks) {
    const int KQ = (int)ks.size() / 4;
    for (int kq = 0; kq < KQ; ++kq)
        for (int T = 0; T < NT; ++T)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const KStep& k = ks[4 * kq + j];
                    const int col = (lane >> 5) ? k.hi : k.lo;
                    const int n = 32 * T + (lane & 31);
                    stream.push_back((col >= 0 && n < n_out) ? Wm[(size_t)n * n_in + col] : 0.0f);
                }
    const size_t slot_floats = SLOT_BYTES / 4;
    while (stream.size() % slot_floats) stream.push_back(0.0f);
}


// Transposed part for the backward-data chain: out = Wm[:, col_base : col_base + n_feat]^T x delta.
// The k index runs over Wm's ROWS (the layer's outputs, held in accumulator layout: ks), the MFMA output
// rows over its input columns.
void emit_part_t(std::vector<float>& stream, const float* Wm, int n_rows, int ld, int col_base, int n_feat, int NT,
                 const std::vector<KStep>& ks) {
    const int KQ = (int)ks.size() / 4;
    for (int kq = 0; kq < KQ; ++kq)
        for (int T = 0; T < NT; ++T)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const KStep& k = ks[4 * kq + j];
                    const int row = (lane >> 5) ? k.hi : k.lo;
                    const int n = 32 * T + (lane & 31);
                    stream.push_back((row >= 0 && row < n_rows && n < n_feat) ? Wm[(size_t)row * ld + col_base + n] : 0.0f);
                }
    const size_t slot_floats = SLOT_BYTES / 4;
    while (stream.size() % slot_floats) stream.push_back(0.0f);
}

}  // namespace

int pack_fp32(const mi_nerf_net* net, const mi_nerf_params* p, void* blob, size_t blob_bytes) {
    // Wn: the network's width (the parameter tensors' shapes, model/NeRF.py:24-30); W: the width of the kernel that will run it (layout.h
    // kernel_width: hidden units Wn .. W-1 get zero weights and biases); Hn / W/2 likewise for linear_d's W // 2 outputs
    MN_CHECK_ARG(net->W >= 2 && net->W <= MAX_KERNEL_WIDTH, "unsupported width W=%d (the fp32 inference kernels run 2 <= W <= %d)", net->W, MAX_KERNEL_WIDTH);
    const int D = net->D, Wn = net->W, Hn = Wn / 2, W = kernel_width(Wn), NT = W / 32;
    const int in_x = 3 + 6 * net->L_x, in_d = 3 + 6 * net->L_d;
    const BlobLayout L = make_layout(D, Wn, net->skip, net->L_x, net->L_d);
    MN_CHECK_ARG(blob_bytes >= L.total_bytes, "blob too small: %zu < %u", blob_bytes, L.total_bytes);
    memset(blob, 0, L.total_bytes);

    std::vector<float> stream;
    stream.reserve(L.stream_bytes_full / 4);
    const bool w16 = wide_kernel_width(W);   // the wide kernels' stream order (same part sizes: a quad is 256 weights in either order)
    if (w16) {
        const int NT16 = W / 16;
        emit_part16(stream, p->linear_x_w[0], Wn, in_x, NT16, enc_ksteps16(KERNEL_LX, net->L_x, 0));
        for (int l = 1; l < D; ++l) {
            const bool cat = (net->skip >= 0 && l == net->skip + 1);
            const int n_in = cat ? Wn + in_x : Wn;
            if (cat) emit_part16(stream, p->linear_x_w[l], Wn, n_in, NT16, enc_ksteps16(KERNEL_LX, net->L_x, 0));
            emit_part16(stream, p->linear_x_w[l], Wn, n_in, NT16, act_ksteps16(W, cat ? in_x : 0, Wn));
        }
        emit_part16(stream, p->linear_feat_w, Wn, Wn, NT16, act_ksteps16(W, 0, Wn));
        emit_part16(stream, p->linear_d_w, Hn, Wn + in_d, NT16 / 2, act_ksteps16(W, 0, Wn));
        MN_CHECK_ARG(stream.size() * 4 == L.stream_bytes_hoist, "internal: hoisted stream %zu != %u", stream.size() * 4, L.stream_bytes_hoist);
        emit_part16(stream, p->linear_d_w, Hn, Wn + in_d, NT16 / 2, enc_ksteps16(KERNEL_LD, net->L_d, Wn));
    } else {
    emit_part(stream, p->linear_x_w[0], Wn, in_x, NT, enc_ksteps(KERNEL_LX, net->L_x, 0));
    for (int l = 1; l < D; ++l) {
        const bool cat = (net->skip >= 0 && l == net->skip + 1);
        const int n_in = cat ? Wn + in_x : Wn;
        if (cat) emit_part(stream, p->linear_x_w[l], Wn, n_in, NT, enc_ksteps(KERNEL_LX, net->L_x, 0));   // [gamma(x), h]
        emit_part(stream, p->linear_x_w[l], Wn, n_in, NT, act_ksteps(W, cat ? in_x : 0, Wn));
    }
    emit_part(stream, p->linear_feat_w, Wn, Wn, NT, act_ksteps(W, 0, Wn));
    emit_part(stream, p->linear_d_w, Hn, Wn + in_d, NT / 2, act_ksteps(W, 0, Wn));              // [feature, gamma(d)]
    MN_CHECK_ARG(stream.size() * 4 == L.stream_bytes_hoist, "internal: hoisted stream %zu != %u", stream.size() * 4, L.stream_bytes_hoist);
    emit_part(stream, p->linear_d_w, Hn, Wn + in_d, NT / 2, enc_ksteps(KERNEL_LD, net->L_d, Wn));
    }
    MN_CHECK_ARG(stream.size() * 4 == L.stream_bytes_full, "internal: full stream %zu != %u", stream.size() * 4, L.stream_bytes_full);

    uint32_t* hdr = (uint32_t*)blob;
    hdr[0] = BLOB_MAGIC; hdr[1] = w16 ? 6 : 1; hdr[2] = D; hdr[3] = W; hdr[4] = (uint32_t)net->skip; hdr[5] = KERNEL_LX; hdr[6] = KERNEL_LD;   // the LAYOUT's W and L
    hdr[13] = net->L_x; hdr[14] = net->L_d; hdr[15] = Wn;                                                                              // the network's
    hdr[7] = L.stream_off; hdr[8] = L.stream_bytes_hoist; hdr[9] = L.stream_bytes_full; hdr[10] = L.side_off; hdr[11] = L.side_floats;
    hdr[12] = 4;   // stream element bytes
    memcpy((char*)blob + L.stream_off, stream.data(), L.stream_bytes_full);

    float* side = (float*)((char*)blob + L.side_off);                  // zero-filled above: the padded units' entries stay zero
    for (int l = 0; l < D; ++l) memcpy(side + L.bias_trunk + (size_t)l * W, p->linear_x_b[l], Wn * 4);
    memcpy(side + L.bias_feat, p->linear_feat_b, Wn * 4);
    memcpy(side + L.bias_d, p->linear_d_b, Hn * 4);
    memcpy(side + L.dens_w, p->linear_density_w, Wn * 4);
    side[L.dens_b] = p->linear_density_b[0];
    for (int c = 0; c < 3; ++c) memcpy(side + L.color_w + (size_t)c * (W / 2), p->linear_color_w + (size_t)c * Hn, Hn * 4);
    memcpy(side + L.color_b, p->linear_color_b, 3 * 4);
    for (int f = 0; f < in_d; ++f)
        for (int n = 0; n < Hn; ++n) side[L.wdir_t + (size_t)f * (W / 2) + n] = p->linear_d_w[(size_t)n * (Wn + in_d) + Wn + f];
    return MI_NERF_OK;
}

size_t packed_bytes_bwd(const mi_nerf_net* net) {
    if (!native_width(net->W)) {
        set_error("the training kernels exist for W = 128 and 256 (got %d; inference pads narrower networks)", net->W);
        return 0;
    }
    return HEADER_BYTES + (size_t)bwd_stream_bytes(net->D, net->W);
}

int pack_bwd_fp32(const mi_nerf_net* net, const mi_nerf_params* p, void* blob, size_t blob_bytes) {
    MN_CHECK_ARG(native_width(net->W), "the training kernels exist for W = 128 and 256 (got %d; inference pads narrower networks)", net->W);
    const int D = net->D, W = net->W, NT = W / 32;
    const int in_x = 3 + 6 * net->L_x, in_d = 3 + 6 * net->L_d;
    const size_t total = packed_bytes_bwd(net);
    MN_CHECK_ARG(blob_bytes >= total, "blob too small: %zu < %zu", blob_bytes, total);
    memset(blob, 0, total);
    std::vector<float> stream;
    stream.reserve((total - HEADER_BYTES) / 4);
    emit_part_t(stream, p->linear_d_w, W / 2, W + in_d, 0, W, NT, act_ksteps(W / 2, 0));     // d feature = Wd[:, :W]^T d hidden
    emit_part_t(stream, p->linear_feat_w, W, W, 0, W, NT, act_ksteps(W, 0));
    for (int l = D - 1; l >= 1; --l) {
        const bool cat = (net->skip >= 0 && l == net->skip + 1);
        emit_part_t(stream, p->linear_x_w[l], W, cat ? W + in_x : W, cat ? in_x : 0, W, NT, act_ksteps(W, 0));
    }
    MN_CHECK_ARG(stream.size() * 4 + HEADER_BYTES == total, "internal: backward stream %zu != %zu", stream.size() * 4, total - HEADER_BYTES);
    uint32_t* hdr = (uint32_t*)blob;
    hdr[0] = BLOB_MAGIC; hdr[1] = 2; hdr[2] = D; hdr[3] = W; hdr[4] = (uint32_t)net->skip; hdr[5] = net->L_x; hdr[6] = net->L_d;
    hdr[7] = HEADER_BYTES; hdr[8] = (uint32_t)(total - HEADER_BYTES);
    memcpy((char*)blob + HEADER_BYTES, stream.data(), stream.size() * 4);
    return MI_NERF_OK;
}

// Gather map for packing ON THE DEVICE (training re-packs after every optimiser step): run the host packer over a
// parameter set whose values are their own flat indices + 1 (exact in fp32 below 2^24); every blob float then names
// its source (0 = constant zero / header).  kind 0: forward blob, 1: backward-data blob.
int pack_map(const mi_nerf_net* net, int kind, int32_t* map, size_t map_len) {
    const int D = net->D, W = net->W;
    const ParamOffsets po = make_param_offsets(D, W, net->skip, net->L_x, net->L_d);
    MN_CHECK_ARG(po.total < (1u << 24), "network too large for the index map (%u parameters)", po.total);
    MN_CHECK_ARG(kind == 0 || kind == 1, "kind must be 0 (forward) or 1 (backward)");
    MN_CHECK_ARG(kind == 0 || native_width(W), "the training kernels exist for W = 128 and 256 (got %d; inference pads narrower networks)", W);
    const size_t bytes = kind == 0 ? make_layout(D, W, net->skip, net->L_x, net->L_d).total_bytes : packed_bytes_bwd(net);
    MN_CHECK_ARG(map && map_len * 4 >= bytes, "map too small: %zu entries for %zu bytes", map_len, bytes);
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
    std::vector<float> blob(bytes / 4);
    if (int rc = kind == 0 ? pack_fp32(net, &p, blob.data(), bytes) : pack_bwd_fp32(net, &p, blob.data(), bytes)) return rc;
    for (size_t i = 0; i < HEADER_BYTES / 4; ++i) map[i] = 0;                   // header words are not floats
    for (size_t i = HEADER_BYTES / 4; i < bytes / 4; ++i) map[i] = (int32_t)blob[i];
    return MI_NERF_OK;
}

}  // namespace minerf
