// layout.h -- packed-weight blob layout shared by the host packer (pack.cpp) and the kernels.
//
// The MLP kernels keep a 32-point tile's activations in registers in MFMA accumulator layout and feed
// the accumulator of layer l straight back as the B operand of layer l+1.  That fixes, per k-step, WHICH
// input feature each half of the wave contributes, so the weights (the A operand) are stored pre-permuted
// in exactly the order the kernel consumes them ("stream"), as 1 KiB "quads":
//
//   quad(T, kq) = A fragments of output tile T (32 output features) for k-steps 4kq..4kq+3:
//                 float [64 lanes][4]  -> one ds_read_b128 per lane yields 4 MFMA A operands.
//   lane l: i = l & 31 (output feature 32T+i), hh = l >> 5 (which of the k-step's two input features)
//
// v_mfma_f32_32x32x2_f32: A[i][k] lane l -> i=l&31,k=l>>5;  B[k][j] lane l -> k=l>>5,j=l&31;
// D[i][j]: j = l&31, i = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// Hence accumulator register r of output tile t, on lane half hh, holds feature
//   feat(t, r, hh) = 32t + (r&3) + 8*(r>>2) + 4*hh        (for point j = l&31)
// and k-step s = 16t + r of the next layer multiplies features (feat(t,r,0), feat(t,r,1)).
//
// Encoded inputs (gamma(x), gamma(d)) are produced per lane as: k-step s < 3L -> level k=s/3, axis c=s%3,
// half 0 = sin(2^k p_c) (channel 3+6k+c), half 1 = cos(2^k p_c) (channel 3+6k+3+c);
// s = 3L -> (p_x, p_y); s = 3L+1 -> (p_z, zero pad); rest zero pad to a multiple of 4 k-steps.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace minerf {

constexpr int QUAD_BYTES = 1024;
constexpr int SLOT_QUADS = 16;
constexpr int SLOT_BYTES = QUAD_BYTES * SLOT_QUADS;   // 16 KiB ring slot
constexpr int HEADER_BYTES = 1024;
constexpr uint32_t BLOB_MAGIC = 0x4D494E46u;          // 'MINF'

__host__ __device__ constexpr int pe_ksteps(int L) { return ((3 * L + 2 + 3) / 4) * 4; }   // padded to 4

// The kernels are instantiated for L_x = 10, L_d = 4 (the reference's defaults, config.py:54-55).  A network with FEWER frequencies
// (--L_x / --L_d below the defaults) runs on the same kernels: gamma_L is a prefix of gamma_10 in the reference's channel order
// (PositionalEncoding.py:18-24), so the packer lays the stream out for 10 / 4 and gives the missing channels zero weights -- the
// kernel evaluates the full encoding, the extra products add exact zeros.  More frequencies than 10 / 4 are refused.
constexpr int KERNEL_LX = 10, KERNEL_LD = 4;

// The same for the WIDTH (opts.netWidth, config.py:57).  The fp32 inference kernels are instantiated for W = 128 and W = 256 (mlp_fp32.hip:
// 32 points per wave on v_mfma_f32_32x32x2_f32) and W = 384 and 512 (mlp_fp32_wide.hip: 16 points per wave on v_mfma_f32_16x16x4_f32, the shape whose
// accumulators and B operands of a 512-wide layer fit one wave's register file; its stream order is the "W16" one below); a network of any
// other width 2 <= W <= 512 runs on the next instantiated one: the packer lays the blob out for kernel_width(W) and gives the hidden units the
// network does not have zero weights and zero biases.  Such a unit is exactly 0 before and after its ReLU and multiplies zero weights in the
// next layer, so the real units' sums gain exact zeros in places where the k order leaves the real terms' order alone: the result is the
// W-wide network's, at the padded width's cost.  (W / 2, the width of linear_d, is W // 2 as in model/NeRF.py:28.)  The training kernels,
// the bf16 and the split-precision variants take their native widths only.
constexpr int MAX_KERNEL_WIDTH = 512;
__host__ __device__ constexpr int kernel_width(int W) { return W <= 128 ? 128 : (W <= 256 ? 256 : (W <= 384 ? 384 : 512)); }
__host__ __device__ constexpr bool wide_kernel_width(int Wk) { return Wk > 256; }      // 384, 512: mlp_fp32_wide.hip and the W16 stream order
// k-steps of the W16 stream order (mlp_fp32_wide.hip): a k-step of v_mfma_f32_16x16x4_f32 multiplies FOUR input features, one per lane quarter
// q = lane >> 4.  Encoded inputs: k-step s < ceil(3L / 2) carries the (sin, cos) pairs of two (level, axis) combinations m = 2s + (q >> 1)
// (level m / 3, axis m % 3; q & 1: 0 sin, 1 cos); then one k-step (p_x, p_y, p_z, 0); zero-padded to a multiple of 4 k-steps.
__host__ __device__ constexpr int pe_ksteps16(int L) { return (((3 * L + 1) / 2 + 1 + 3) / 4) * 4; }
__host__ __device__ constexpr bool native_width(int W) { return W == 128 || W == 256; }

struct BlobLayout {
    // all byte offsets from the blob start
    uint32_t stream_off;          // == HEADER_BYTES
    uint32_t stream_bytes_hoist;  // stream length when the view-direction part of linear_d is hoisted (fused mode)
    uint32_t stream_bytes_full;   // ... including the per-point direction k-steps (embedded mode)
    uint32_t side_off;            // fp32 side tables, natural feature order
    uint32_t side_floats;         // padded to a multiple of 4
    // side-table sub-offsets, in floats from side start
    uint32_t bias_trunk;          // D x W
    uint32_t bias_feat;           // W
    uint32_t bias_d;              // W/2
    uint32_t dens_w;              // W
    uint32_t dens_b;              // 1 (+3 pad)
    uint32_t color_w;             // 3 x W/2
    uint32_t color_b;             // 3 (+1 pad)
    uint32_t wdir_t;              // in_d x W/2  (transposed direction block of linear_d)
    uint32_t total_bytes;
};

__host__ __device__ inline uint32_t round_up_u32(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// elem_bytes: 4 (fp32 stream) or 2 (bf16 stream; quads keep 1 KiB, see mlp_bf16.hip)
inline BlobLayout make_layout(int D, int W_net, int skip, int /*L_x*/, int /*L_d*/) {
    constexpr int L_x = KERNEL_LX, L_d = KERNEL_LD;              // the layout is the kernels', whatever the network's own L and W (see above)
    const int W = kernel_width(W_net);
    BlobLayout b{};
    const int NT = W / 32;
    const int in_d = 3 + 6 * L_d;
    const uint32_t pe_quads = (uint32_t)(pe_ksteps(L_x) / 4) * NT;
    const uint32_t h_quads = (uint32_t)(W / 2 / 4) * NT;
    uint32_t quads = pe_quads;                                     // trunk layer 0
    for (int l = 1; l < D; ++l) quads += h_quads + ((skip >= 0 && l == skip + 1) ? pe_quads : 0);
    quads += h_quads;                                              // linear_feat
    quads += (uint32_t)(W / 2 / 4) * (NT / 2);                     // linear_d, feature part
    b.stream_off = HEADER_BYTES;
    b.stream_bytes_hoist = round_up_u32(quads, SLOT_QUADS) * QUAD_BYTES;
    uint32_t full = round_up_u32(quads, SLOT_QUADS) + round_up_u32((uint32_t)(pe_ksteps(L_d) / 4) * (NT / 2), SLOT_QUADS);
    b.stream_bytes_full = full * QUAD_BYTES;
    b.side_off = b.stream_off + b.stream_bytes_full;
    uint32_t f = 0;
    b.bias_trunk = f; f += (uint32_t)D * W;
    b.bias_feat = f;  f += W;
    b.bias_d = f;     f += W / 2;
    b.dens_w = f;     f += W;
    b.dens_b = f;     f += 4;
    b.color_w = f;    f += 3 * (W / 2);
    b.color_b = f;    f += 4;
    b.wdir_t = f;     f += (uint32_t)in_d * (W / 2);
    b.side_floats = round_up_u32(f, 4);
    b.total_bytes = b.side_off + b.side_floats * 4;
    return b;
}

// ---------------------------------------------------------------------------------------------
// training path
// ---------------------------------------------------------------------------------------------
// Flat parameter vector of one NeRFModule in module.parameters() order (model/NeRF.py:24-30):
// linear_x[0..D).{weight,bias}, linear_d, linear_feat, linear_density, linear_color.  Offsets in floats.
struct ParamOffsets {
    uint32_t w_x[16], b_x[16];
    int in_l[16];                 // input width of trunk layer l
    uint32_t w_d, b_d, w_feat, b_feat, w_dens, b_dens, w_color, b_color;
    uint32_t total;
};

inline ParamOffsets make_param_offsets(int D, int W, int skip, int L_x, int L_d) {
    ParamOffsets p{};
    const int in_x = 3 + 6 * L_x, in_d = 3 + 6 * L_d;
    uint32_t f = 0;
    for (int l = 0; l < D; ++l) {
        p.in_l[l] = (l == 0) ? in_x : ((skip >= 0 && l == skip + 1) ? W + in_x : W);
        p.w_x[l] = f; f += (uint32_t)W * p.in_l[l];
        p.b_x[l] = f; f += W;
    }
    p.w_d = f;     f += (uint32_t)(W / 2) * (W + in_d);
    p.b_d = f;     f += W / 2;
    p.w_feat = f;  f += (uint32_t)W * W;
    p.b_feat = f;  f += W;
    p.w_dens = f;  f += W;
    p.b_dens = f;  f += 1;
    p.w_color = f; f += 3 * (W / 2);
    p.b_color = f; f += 3;
    p.total = f;
    return p;
}

// Backward-data stream (mlp_train.hip): the TRANSPOSED weights as MFMA A operands, in the order the chain
// d_hidden -> linear_d^T (feature block) -> linear_feat^T -> linear_x[D-1]^T ... linear_x[1]^T consumes them.
// Blob = header | stream; the side tables of the forward blob are reused.
inline uint32_t bwd_stream_bytes(int D, int W) {
    const uint32_t NT = W / 32;
    const uint32_t quads = (uint32_t)(W / 16) * NT + (uint32_t)D * (uint32_t)(W / 8) * NT;
    return round_up_u32(quads, SLOT_QUADS) * QUAD_BYTES;
}

}  // namespace minerf
