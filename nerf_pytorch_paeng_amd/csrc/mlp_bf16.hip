// mlp_bf16.hip -- bf16-MFMA variant of the fused positional-encoding + NeRF MLP forward (BASELINE config #5).
//
// Same dataflow as mlp_fp32.hip (one wave owns 32 points, activations stay in registers, the accumulator of
// layer l becomes the B operand of layer l+1, weights stream L2 -> LDS in consumption order), on
// v_mfma_f32_32x32x16_bf16: bf16 weights and activations, fp32 accumulation, fp32 biases / heads / compositing.
//
//   A fragment: lane l (i = l&31, h = l>>5) holds A[i][k = 8h + j], j = 0..7  (8 bf16 = 16 B = one ds_read_b128)
//   B fragment: lane l holds B[k = 8h + j][col = l&31]
//   D: col = l&31, row = (r&3) + 8(r>>2) + 4h  (as the f32 MFMA)
// so converting accumulator registers 8s..8s+7 of output tile t to bf16 gives the B fragment of k-step 2t+s whose
// element j is feature 32t + 16s + 8(j>>2) + 4h + (j&3); the weights are packed in that order on the host.
// Encoded inputs: slot u = 16*ks + 8h + j is channel u of gamma(.) (zero weight beyond the last channel).
//
// One "quad" (1 KiB) is now the A fragment of ONE MFMA (tile T, k-step ks); a 32 KiB slot = 4 k-steps x 8 tiles;
// a 256-wide layer is 128 MFMAs = 4096 cycles per wave, 16x shorter than in fp32, so the exposed per-layer and
// per-tile VALU work (accumulator reads, packing, gamma(x)) and the weight stream weigh far more here.
#include <string.h>
#include <vector>
#include "common.h"
#include "layout.h"

namespace minerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BSLOT_QUADS = 32;
constexpr int BSLOT_BYTES = BSLOT_QUADS * QUAD_BYTES;      // 32 KiB
constexpr int BNSLOT = 3;
constexpr int BRING_BYTES = BNSLOT * BSLOT_BYTES;
constexpr int BDMA = BSLOT_QUADS / 4;                      // DMAs per wave per slot

__host__ __device__ constexpr int enc_ksteps16(int L) { return (3 + 6 * L + 15) / 16; }

static inline uint32_t bround(uint32_t quads) { return (quads + BSLOT_QUADS - 1) / BSLOT_QUADS * BSLOT_QUADS; }

struct BlobLayoutBf16 {
    uint32_t stream_off, stream_bytes, side_off, side_floats;
    uint32_t bias_trunk, bias_feat, bias_d, dens_w, dens_b, color_w, color_b, wdir_t, total_bytes;
};

static BlobLayoutBf16 make_layout_bf16(int D, int W, int skip, int L_x, int L_d) {
    BlobLayoutBf16 b{};
    const int NT = W / 32, in_d = 3 + 6 * L_d;
    const uint32_t pe_q = bround((uint32_t)enc_ksteps16(L_x) * NT), h_q = bround((uint32_t)(W / 16) * NT);
    uint32_t quads = pe_q;
    for (int l = 1; l < D; ++l) quads += h_q + ((skip >= 0 && l == skip + 1) ? pe_q : 0);
    quads += h_q + bround((uint32_t)(W / 16) * (NT / 2));
    b.stream_off = HEADER_BYTES;
    b.stream_bytes = quads * QUAD_BYTES;
    b.side_off = b.stream_off + b.stream_bytes;
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
// host: packer
// ---------------------------------------------------------------------------------------------
static inline uint16_t f32_to_bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// cols[ks][h][j]: input column of the weight row, -1 = zero
static void emit_part_bf16(std::vector<uint16_t>& st, const float* Wm, int n_out, int n_in, int NT, const std::vector<int>& cols) {
    const int KS = (int)cols.size() / 16;
    for (int ks = 0; ks < KS; ++ks)
        for (int T = 0; T < NT; ++T)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int col = cols[ks * 16 + (lane >> 5) * 8 + j];
                    const int n = 32 * T + (lane & 31);
                    st.push_back((col >= 0 && n < n_out) ? f32_to_bf16_rne(Wm[(size_t)n * n_in + col]) : (uint16_t)0);
                }
    const size_t slot_elems = BSLOT_BYTES / 2;
    while (st.size() % slot_elems) st.push_back(0);
}
static std::vector<int> enc_cols16(int L, int base) {
    const int nch = 3 + 6 * L, KS = enc_ksteps16(L);
    std::vector<int> c(KS * 16);
    for (int u = 0; u < KS * 16; ++u) c[u] = u < nch ? base + u : -1;
    return c;
}
static std::vector<int> act_cols16(int W, int base) {
    std::vector<int> c;
    for (int t = 0; t < W / 32; ++t)
        for (int s = 0; s < 2; ++s)
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 8; ++j) c.push_back(base + 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3));
    return c;
}

static int check_net_bf16(const mi_nerf_net* net) {
    MN_CHECK_ARG(net != nullptr, "net is NULL");
    MN_CHECK_ARG(net->W == 256, "the bf16 variant is built for W=256 only (got %d)", net->W);
    MN_CHECK_ARG(net->D >= 2 && net->D <= 16 && net->L_x == 10 && net->L_d == 4 && net->skip >= -1, "unsupported network for bf16");
    return MI_NERF_OK;
}

size_t packed_bytes_bf16(const mi_nerf_net* net) {
    if (check_net_bf16(net)) return 0;
    return make_layout_bf16(net->D, net->W, net->skip, net->L_x, net->L_d).total_bytes;
}

int pack_bf16(const mi_nerf_net* net, const mi_nerf_params* p, void* blob, size_t blob_bytes) {
    if (int rc = check_net_bf16(net)) return rc;
    const int D = net->D, W = net->W, NT = W / 32;
    const int in_x = 3 + 6 * net->L_x, in_d = 3 + 6 * net->L_d;
    const BlobLayoutBf16 L = make_layout_bf16(D, W, net->skip, net->L_x, net->L_d);
    MN_CHECK_ARG(blob_bytes >= L.total_bytes, "blob too small: %zu < %u", blob_bytes, L.total_bytes);
    memset(blob, 0, L.total_bytes);
    std::vector<uint16_t> st;
    st.reserve(L.stream_bytes / 2);
    emit_part_bf16(st, p->linear_x_w[0], W, in_x, NT, enc_cols16(net->L_x, 0));
    for (int l = 1; l < D; ++l) {
        const bool cat = (net->skip >= 0 && l == net->skip + 1);
        const int n_in = cat ? W + in_x : W;
        if (cat) emit_part_bf16(st, p->linear_x_w[l], W, n_in, NT, enc_cols16(net->L_x, 0));
        emit_part_bf16(st, p->linear_x_w[l], W, n_in, NT, act_cols16(W, cat ? in_x : 0));
    }
    emit_part_bf16(st, p->linear_feat_w, W, W, NT, act_cols16(W, 0));
    emit_part_bf16(st, p->linear_d_w, W / 2, W + in_d, NT / 2, act_cols16(W, 0));
    MN_CHECK_ARG(st.size() * 2 == L.stream_bytes, "internal: bf16 stream %zu != %u", st.size() * 2, L.stream_bytes);
    uint32_t* hdr = (uint32_t*)blob;
    hdr[0] = BLOB_MAGIC; hdr[1] = 1; hdr[2] = D; hdr[3] = W; hdr[4] = (uint32_t)net->skip; hdr[5] = net->L_x; hdr[6] = net->L_d;
    hdr[7] = L.stream_off; hdr[8] = L.stream_bytes; hdr[9] = L.stream_bytes; hdr[10] = L.side_off; hdr[11] = L.side_floats;
    hdr[12] = 2;   // stream element bytes
    memcpy((char*)blob + L.stream_off, st.data(), L.stream_bytes);
    float* side = (float*)((char*)blob + L.side_off);
    for (int l = 0; l < D; ++l) memcpy(side + L.bias_trunk + (size_t)l * W, p->linear_x_b[l], W * 4);
    memcpy(side + L.bias_feat, p->linear_feat_b, W * 4);
    memcpy(side + L.bias_d, p->linear_d_b, (W / 2) * 4);
    memcpy(side + L.dens_w, p->linear_density_w, W * 4);
    side[L.dens_b] = p->linear_density_b[0];
    memcpy(side + L.color_w, p->linear_color_w, 3 * (W / 2) * 4);
    memcpy(side + L.color_b, p->linear_color_b, 3 * 4);
    for (int f = 0; f < in_d; ++f)
        for (int n = 0; n < W / 2; ++n) side[L.wdir_t + (size_t)f * (W / 2) + n] = p->linear_d_w[(size_t)n * (W + in_d) + W + f];
    return MI_NERF_OK;
}

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
struct MlpArgsB {
    const char* stream;
    const float* side;
    const float* rays;
    const float* z;
    float* out;
    long long n_wtiles;
    int S, tpr, D, skip_layer;
    unsigned stream_bytes, side_floats;
    unsigned o_bias_trunk, o_bias_feat, o_bias_d, o_dens_w, o_dens_b, o_color_w, o_color_b, o_wdir_t;
};

struct BRing {
    const char* sbase;      // stream + wave's 8 KiB share
    unsigned voff;          // lane*16
    unsigned fetch_off, stream_bytes;
    unsigned fetch_lds, lds_lo, lds_hi;
    unsigned read_slot;
};

template <int IMM>
__device__ __forceinline__ void bdma16(const char* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr), "i"(IMM)
        : "memory");
}
// DMA number i of the slot being fetched: 8 KiB per wave; the 13-bit instruction offset reaches 4 KiB, so the
// second half goes through a +4 KiB base
__device__ __forceinline__ void bring_dma(const BRing& r, int i) {
    const char* g = r.sbase + r.fetch_off;
    if (i == 0) bdma16<0>(g, r.voff, r.fetch_lds);
    else if (i == 1) bdma16<1024>(g, r.voff, r.fetch_lds);
    else if (i == 2) bdma16<2048>(g, r.voff, r.fetch_lds);
    else if (i == 3) bdma16<3072>(g, r.voff, r.fetch_lds);
    else if (i == 4) bdma16<0>(g + 4096, r.voff, r.fetch_lds + 4096);
    else if (i == 5) bdma16<1024>(g + 4096, r.voff, r.fetch_lds + 4096);
    else if (i == 6) bdma16<2048>(g + 4096, r.voff, r.fetch_lds + 4096);
    else if (i == 7) bdma16<3072>(g + 4096, r.voff, r.fetch_lds + 4096);
}
__device__ __forceinline__ void bring_next_fetch(BRing& r) {
    r.fetch_off += BSLOT_BYTES;
    if (r.fetch_off >= r.stream_bytes) r.fetch_off = 0;
    r.fetch_lds += BSLOT_BYTES;
    if (r.fetch_lds >= r.lds_hi) r.fetch_lds = r.lds_lo;
}
// consume the next slot: everything but the 8 DMAs issued during the phase that ends here has landed (slot p+1 was
// issued two phases ago); barrier; slot p+2 streams into ring[(p+2)%3] == ring[(p-1)%3] during the new phase
__device__ __forceinline__ void bring_advance(BRing& r) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    bring_next_fetch(r);
    r.read_slot = (r.read_slot + 1 == BNSLOT) ? 0 : r.read_slot + 1;
}
__device__ __forceinline__ bf16x8 bring_read(const char* smem, const BRing& r, int lane, int qs) {
    if (qs >= 1 && qs <= BDMA) bring_dma(r, qs - 1);
    return *(const bf16x8*)(smem + r.read_slot * BSLOT_BYTES + lane * 16 + qs * QUAD_BYTES);
}

// acc[0..NT) += A(stream) x B over KS k-steps of 16; a[] is the A pipeline (one k-step ahead), as in mlp_fp32.hip
template <int NT, int KS, int NT_NEXT, int NB>
__device__ __forceinline__ void bgemm_part(f32x16 (&acc)[8], const bf16x8 (&b)[NB], bf16x8 (&a)[8], const char* smem, BRing& ring,
                                           int lane) {
    static_assert(KS <= NB, "B registers");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[ks], acc[t], 0, 0, 0);
            if (ks + 1 < KS) {
                const int q = (ks + 1) * NT + t;
                if (q % BSLOT_QUADS == 0) bring_advance(ring);
                a[t] = bring_read(smem, ring, lane, q % BSLOT_QUADS);
            } else if (t < NT_NEXT) {
                if (t == 0) bring_advance(ring);                   // the next part starts a fresh slot
                a[t] = bring_read(smem, ring, lane, t);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int t = NT; t < NT_NEXT; ++t) a[t] = bring_read(smem, ring, lane, t);
}

template <int NT>
__device__ __forceinline__ void bacc_init(f32x16 (&acc)[8], const float* vec_lds, int hh) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(vec_lds + 32 * t + 8 * g + 4 * hh);
            acc[t][4 * g + 0] = v[0]; acc[t][4 * g + 1] = v[1]; acc[t][4 * g + 2] = v[2]; acc[t][4 * g + 3] = v[3];
        }
}
// accumulators -> bf16 B fragments of the next layer: fragment 2t+s, element j = acc[t][8s + j]
template <int NT, bool RELU>
__device__ __forceinline__ void bacc_to_b(const f32x16 (&acc)[8], bf16x8 (&hb)[16]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = acc[t][8 * s + j];
                v[j] = (__bf16)(RELU ? __builtin_fmaxf(x, 0.0f) : x);
            }
            hb[2 * t + s] = v;
        }
}
__device__ __forceinline__ float bxhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

template <int W, int LX, int LD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_bf16_kernel(const MlpArgsB a) {
    static_assert(W == 256, "bf16 variant: W = 256");
    constexpr int NT = W / 32, KH = W / 16, KPE = enc_ksteps16(LX), IN_X = 3 + 6 * LX, IN_D = 3 + 6 * LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* side = (float*)(smem + BRING_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hh = lane >> 5;
    float* scratch = side + a.side_floats + wave * (W / 2);
    for (unsigned i = tid * 4; i < a.side_floats; i += 256 * 4) *(f32x4*)(side + i) = *(const f32x4*)(a.side + i);
    const long long n_wg_tiles = (a.n_wtiles + 3) >> 2;
    if ((long long)blockIdx.x >= n_wg_tiles) return;

    BRing ring;
    ring.sbase = a.stream + wave * (BDMA * QUAD_BYTES);
    ring.voff = lane * 16;
    ring.fetch_off = 0;
    ring.stream_bytes = a.stream_bytes;
    ring.lds_lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (BDMA * QUAD_BYTES);
    ring.lds_hi = ring.lds_lo + BRING_BYTES;
    ring.fetch_lds = ring.lds_lo;
    ring.read_slot = BNSLOT - 1;
#pragma unroll
    for (int i = 0; i < BDMA; ++i) bring_dma(ring, i);       // slot 0
    bring_next_fetch(ring);
#pragma unroll
    for (int i = 0; i < BDMA; ++i) bring_dma(ring, i);       // slot 1; slot p+2 streams in while slot p is consumed

    f32x16 acc[8];
    bf16x8 aq[8];
    bf16x8 hb[16];
    bf16x8 peb[KPE];
    bring_advance(ring);
#pragma unroll
    for (int t = 0; t < NT; ++t) aq[t] = bring_read(smem, ring, lane, t);

    for (long long wgt = blockIdx.x; wgt < n_wg_tiles; wgt += gridDim.x) {
        long long wt = wgt * 4 + wave;
        const bool wave_active = wt < a.n_wtiles;
        if (!wave_active) wt = a.n_wtiles - 1;
        const long long ray = wt / a.tpr;
        const int sample = (int)(wt - ray * a.tpr) * 32 + col;
        const bool valid = wave_active && sample < a.S;
        const int sc = sample < a.S ? sample : a.S - 1;
        const long long out_idx = ray * a.S + sc;
        const float* rp = a.rays + ray * 6;
        const float ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5];
        const float zv = a.z[out_idx];
        const float p[3] = {ox + dx * zv, oy + dy * zv, oz + dz * zv};
        // gamma(x) in fp32, rounded once to bf16: slot u = 16ks + 8hh + j is channel u
        const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p[0]), __builtin_fabsf(p[1])), __builtin_fabsf(p[2])) * (float)(1 << (LX - 1));
        const bool fast = amax < SINCOS_FAST_LIMIT;
#pragma unroll
        for (int ks = 0; ks < KPE; ++ks) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u0 = 16 * ks + j, u1 = u0 + 8;              // this element's channel on lane half 0 / 1
                float r0 = 0.f, r1 = 0.f;
                // both halves evaluate one sin-or-cos with per-half (argument, quadrant shift)
                auto chan = [&](int u, float& arg, int& qs, bool& ident, bool& pad) {
                    pad = u >= IN_X; ident = u < 3;
                    const int c = ident ? u : (pad ? 0 : (u - 3) % 3), k = (ident || pad) ? 0 : (u - 3) / 6;
                    qs = (!ident && !pad && ((u - 3) % 6) >= 3) ? 1 : 0;
                    arg = p[c] * (float)(1 << k);
                };
                float a0, a1; int q0, q1; bool i0, i1, p0, p1;
                chan(u0, a0, q0, i0, p0); chan(u1, a1, q1, i1, p1);
                const float arg = hh ? a1 : a0;
                const int qs = hh ? q1 : q0;
                const float sc_ = fast ? sin_cos_fast(arg, qs) : sin_cos_slow(arg, qs);
                r0 = p0 ? 0.f : (i0 ? a0 : sc_);
                r1 = p1 ? 0.f : (i1 ? a1 : sc_);
                v[j] = (__bf16)(hh ? r1 : r0);
            }
            peb[ks] = v;
        }
        // hoisted view-direction term of linear_d (fp32): scratch[n] = b_d[n] + sum_f Wd[n][W+f] * gamma(d/|d|)[f]
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
        {
            const float* wdt = side + a.o_wdir_t;
            const float* bd = side + a.o_bias_d;
#pragma unroll
            for (int n0 = 0; n0 < W / 2; n0 += 64) {
                const int n = n0 + lane;
                float s = bd[n];
#pragma unroll
                for (int f = 0; f < IN_D; ++f) s = __builtin_fmaf(wdt[f * (W / 2) + n], g[f], s);
                scratch[n] = s;
            }
        }
        // ---- trunk ----
        bacc_init<NT>(acc, side + a.o_bias_trunk, hh);
        bgemm_part<NT, KPE, NT>(acc, peb, aq, smem, ring, lane);
#pragma unroll 1
        for (int l = 1; l < a.D; ++l) {
            bacc_to_b<NT, true>(acc, hb);
            bacc_init<NT>(acc, side + a.o_bias_trunk + l * W, hh);
            if (l == a.skip_layer) bgemm_part<NT, KPE, NT>(acc, peb, aq, smem, ring, lane);
            bgemm_part<NT, KH, NT>(acc, hb, aq, smem, ring, lane);
        }
        // ---- density head in fp32 on the un-rounded trunk output ----
        float ds = 0.f;
        {
            const float* dw = side + a.o_dens_w + 4 * hh;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 w = *(const f32x4*)(dw + 32 * t + 8 * gq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ds = __builtin_fmaf(__builtin_fmaxf(acc[t][4 * gq + e], 0.f), w[e], ds);
                    if ((t * 4 + gq) % 8 == 7) asm volatile("" ::: "memory");
                }
        }
        const float dens = bxhalf_sum(ds) + side[a.o_dens_b];
        bacc_to_b<NT, true>(acc, hb);
        // ---- feature layer (no activation) and view-direction layer ----
        bacc_init<NT>(acc, side + a.o_bias_feat, hh);
        bgemm_part<NT, KH, NT / 2>(acc, hb, aq, smem, ring, lane);
        bacc_to_b<NT, false>(acc, hb);
        bacc_init<NT / 2>(acc, scratch, hh);
        bgemm_part<NT / 2, KH, NT>(acc, hb, aq, smem, ring, lane);
        // ---- colour head in fp32 ----
        asm volatile("" ::: "memory");
        float cs[3] = {0.f, 0.f, 0.f};
        {
            const float* cw = side + a.o_color_w + 4 * hh;
#pragma unroll
            for (int t = 0; t < NT / 2; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const f32x4 w = *(const f32x4*)(cw + c * (W / 2) + 32 * t + 8 * gq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) cs[c] = __builtin_fmaf(__builtin_fmaxf(acc[t][4 * gq + e], 0.f), w[e], cs[c]);
                    }
                    asm volatile("" ::: "memory");
                }
        }
        const float r0 = bxhalf_sum(cs[0]) + side[a.o_color_b + 0];
        const float r1 = bxhalf_sum(cs[1]) + side[a.o_color_b + 1];
        const float r2 = bxhalf_sum(cs[2]) + side[a.o_color_b + 2];
        if (valid && hh == 0) {
            f32x4 o; o[0] = r0; o[1] = r1; o[2] = r2; o[3] = dens;
            *(f32x4*)(a.out + out_idx * 4) = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

int mlp_rays_bf16(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev, int64_t n_rays, int S,
                  float* raw_dev, hipStream_t st) {
    if (int rc = check_net_bf16(net)) return rc;
    MN_CHECK_ARG(n_rays >= 0 && S >= 1, "bad sizes n_rays=%lld S=%d", (long long)n_rays, S);
    if (n_rays == 0) return MI_NERF_OK;
    MN_CHECK_ARG(packed_dev && rays_dev && z_dev && raw_dev, "NULL device pointer");
    const BlobLayoutBf16 L = make_layout_bf16(net->D, net->W, net->skip, net->L_x, net->L_d);
    MlpArgsB a{};
    a.stream = (const char*)packed_dev + L.stream_off;
    a.side = (const float*)((const char*)packed_dev + L.side_off);
    a.rays = rays_dev; a.z = z_dev; a.out = raw_dev;
    a.S = S; a.tpr = (S + 31) / 32; a.n_wtiles = (long long)n_rays * a.tpr;
    a.D = net->D;
    a.skip_layer = (net->skip >= 0 && net->skip + 1 < net->D) ? net->skip + 1 : -1;
    a.stream_bytes = L.stream_bytes; a.side_floats = L.side_floats;
    a.o_bias_trunk = L.bias_trunk; a.o_bias_feat = L.bias_feat; a.o_bias_d = L.bias_d; a.o_dens_w = L.dens_w; a.o_dens_b = L.dens_b;
    a.o_color_w = L.color_w; a.o_color_b = L.color_b; a.o_wdir_t = L.wdir_t;
    const size_t lds = BRING_BYTES + (size_t)a.side_floats * 4 + 4 * (256 / 2) * 4;
    MN_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded: %zu bytes", lds);
    auto kern = mlp_bf16_kernel<256, 10, 4>;
    static bool attr_set = false;
    if (!attr_set) {
        MN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const long long n_wg = (a.n_wtiles + 3) / 4;
    const int grid = (int)(n_wg < cus ? n_wg : cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    MN_LAUNCH_CHECK("mlp_bf16_kernel");
    return MI_NERF_OK;
}

}  // namespace minerf
