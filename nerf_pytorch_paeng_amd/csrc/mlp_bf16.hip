// mlp_bf16.hip -- bf16-MFMA variant of the fused positional-encoding + NeRF MLP forward (BASELINE config #5).
//
// Same idea as mlp_fp32.hip -- activations never leave registers, the accumulator of layer l becomes the B operand of
// layer l+1, weights stream L2 -> LDS in consumption order -- rebuilt around what v_mfma_f32_32x32x16_bf16 makes
// expensive: at 16x the fp32 MFMA rate a 256-wide layer is only 4096 matrix cycles per 32 points, so everything that
// is NOT an MFMA (accumulator -> bf16 packing, biases, gamma(x), heads) and the weight stream itself decide the speed.
// (Round 1's kernel, with the fp32 kernel's k-outer order: 5.8 VALU instructions per MFMA, all of them exposed at the
// layer boundaries; 7.5 GB of L2 -> LDS weight traffic per launch; 42 % MFMA busy -- profiles/r02_bf16_old_pmc*.json.)
//
//   * OUTPUT-TILE-MAJOR order ("t-outer").  A layer is 8 jobs; job t computes output features 32t..32t+31 over ALL
//     k-steps with ONE 16-register accumulator (a single dependent chain runs at full rate on this instruction).  Tile t
//     of layer l is exactly B fragments 2t, 2t+1 of layer l+1, which that layer does not touch before its k-steps 2t,
//     2t+1 -- so the packing of job j's accumulator (v_cvt_pk_bf16_f32 + ReLU as v_pk_max_i16 on the packed pair) is
//     dealt out over the first groups of job j+1, across layer boundaries as well: there IS no layer boundary.
//     The bias is the C operand of a job's first MFMA (16 registers from LDS, no accumulator initialisation).
//   * TWO POINT TILES PER WAVE (64 points, one wave per SIMD): every A fragment (one ds_read_b128 per lane) feeds two
//     MFMAs, so LDS reads and the L2 -> LDS stream are half of the one-tile design per FLOP (256 points per workgroup
//     per pass over the 1.2 MB stream).  Live registers: 2 x 16 input fragments + 2 x 16 output fragments (256), two
//     accumulators in flight + two being packed (64), gamma(x) fragments (32), A pipeline (16), biases (32).
//   * HEADS ON THE MATRIX PIPE.  Density (256 -> 1) is row 3 of an extra 32-row output tile over the trunk output (16
//     k-steps, right after the feature layer), colour (128 -> 3) rows 0..2 of another over the view-direction layer's output
//     (8 k-steps): (r, g, b) and the density land in registers 0..2 / 3 of lanes 0..31, one 16-byte store per point; no
//     VALU dot products, no cross-half shuffles.  (+2 % MFMAs, -500 VALU per tile.)
//   * gamma(x) by ANGLE DOUBLING: one accurate sin/cos per axis (Cody-Waite + Cephes, as the fp32 kernel), then
//     s' = 2sc, c' = 1 - 2s^2 for the nine higher octaves.  The recurrence doubles the error per octave (<= 2^9 * 1e-7 =
//     5e-5 at the top octave), two orders below the bf16 rounding (2^-9 relative) the values get next.  The fp32
//     kernel keeps one full-precision evaluation per channel; this is the bf16 variant's own accuracy contract
//     (PSNR against the fp32 path, tests/test_gpu_parity.py).
//
//   A fragment: lane l (i = l&31, h = l>>5) holds A[i][k = 8h + j], j = 0..7  (8 bf16 = 16 B = one ds_read_b128)
//   B fragment: lane l holds B[k = 8h + j][col = l&31]
//   D: col = l&31, row = (r&3) + 8(r>>2) + 4h  (as the f32 MFMA)
// so accumulator registers 8s..8s+7 of output tile t, packed pairwise, are the B fragment of k-step 2t+s whose element j
// is feature 32t + 16s + 8(j>>2) + 4h + (j&3); the weights are packed in that order on the host.
// Encoded inputs: slot u = 16*ks + 8h + j is channel u of gamma(x) (zero weight beyond the last channel).
//
// Stream (1 KiB quads = the A fragment of ONE MFMA pair; 32 KiB slots, 3-slot ring, LDS-DMA two slots ahead):
//   layer 0:   for T in 0..7: 4 gamma(x) k-steps                                   32 quads
//   layer l:   for T in 0..7: 16 activation k-steps [4 gamma(x) k-steps if skip]   128 | 160 quads
//   tail:      feature layer (8 x 16) | head tile over the trunk output (16) | view-direction layer (4 x 16)
//              | head tile over the view-direction output (8) | 8 quads of padding  224 quads
#include <string.h>
#include <type_traits>
#include <vector>
#include "common.h"
#include "layout.h"

namespace minerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

constexpr int NP = 2;                                      // point tiles per wave
#ifdef MN_BF16_DA
constexpr int DA = MN_BF16_DA;                             // A/B variant (tools/ab_probe.py)
#else
constexpr int DA = 4;                                      // A-operand pipeline depth (fragments in flight)
#endif
constexpr int BSLOT_QUADS = 32;
constexpr int BSLOT_BYTES = BSLOT_QUADS * QUAD_BYTES;      // 32 KiB
constexpr int BNSLOT = 3;
constexpr int BRING_BYTES = BNSLOT * BSLOT_BYTES;
constexpr int BDMA = BSLOT_QUADS / 4;                      // DMAs per wave per slot
constexpr int TAIL_USED = 128 + 16 + 64 + 8;               // quads of the tail body that carry weights
constexpr int TAIL_QUADS = 224;                            // ... padded to whole slots

__host__ __device__ constexpr int enc_ksteps16(int L) { return (3 + 6 * L + 15) / 16; }

struct BlobLayoutBf16 {
    uint32_t stream_off, stream_bytes, side_off, side_floats;
    uint32_t bias_trunk, bias_feat, bias_d, head_b, wdir_t, total_bytes;
};

static BlobLayoutBf16 make_layout_bf16(int D, int W, int skip, int L_x, int L_d) {
    BlobLayoutBf16 b{};
    const int NT = W / 32, in_d = 3 + 6 * L_d;
    const uint32_t pe_q = (uint32_t)enc_ksteps16(L_x) * NT, h_q = (uint32_t)(W / 16) * NT;
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

// One quad: the A fragment of output rows row0..row0+31 for the 16 input columns cols[h*8 + j] (-1: zero).
// rowmap (optional, 32 entries): weight-matrix row feeding output row i of the tile, -1: zero row.
static void emit_quad(std::vector<uint16_t>& st, const float* Wm, int n_out, int n_in, int row0, const int* rowmap, const int* cols) {
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const int col = cols[(lane >> 5) * 8 + j];
            const int n = rowmap ? rowmap[lane & 31] : row0 + (lane & 31);
            st.push_back((col >= 0 && n >= 0 && n < n_out) ? f32_to_bf16_rne(Wm[(size_t)n * n_in + col]) : (uint16_t)0);
        }
}
static std::vector<int> enc_cols16(int L, int base) {
    const int nch = 3 + 6 * L, KS = enc_ksteps16(L);
    std::vector<int> c(KS * 16);
    for (int u = 0; u < KS * 16; ++u) c[u] = u < nch ? base + u : -1;
    return c;
}
// input columns in the order the packed accumulators present them: fragment 2t+s, half h, element j
static std::vector<int> act_cols16(int W, int base) {
    std::vector<int> c;
    for (int t = 0; t < W / 32; ++t)
        for (int s = 0; s < 2; ++s)
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 8; ++j) c.push_back(base + 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3));
    return c;
}
// a layer in output-tile-major order: for every tile, all its k-steps
static void emit_layer(std::vector<uint16_t>& st, const float* Wm, int n_out, int n_in, int NT, const std::vector<int>& cols) {
    const int KS = (int)cols.size() / 16;
    for (int T = 0; T < NT; ++T)
        for (int ks = 0; ks < KS; ++ks) emit_quad(st, Wm, n_out, n_in, 32 * T, nullptr, cols.data() + 16 * ks);
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
    emit_layer(st, p->linear_x_w[0], W, in_x, NT, enc_cols16(net->L_x, 0));
    for (int l = 1; l < D; ++l) {
        const bool cat = (net->skip >= 0 && l == net->skip + 1);
        std::vector<int> cols = act_cols16(W, cat ? in_x : 0);          // input columns are cat([gamma(x), h]), NeRF.py:41 ...
        if (cat) {                                                      // ... consumed activations first, gamma(x) last
            const std::vector<int> enc = enc_cols16(net->L_x, 0);
            cols.insert(cols.end(), enc.begin(), enc.end());
        }
        emit_layer(st, p->linear_x_w[l], W, cat ? W + in_x : W, NT, cols);
    }
    // tail: feature layer | head tile, density row over the trunk output | view-direction layer | head tile, colour rows
    emit_layer(st, p->linear_feat_w, W, W, NT, act_cols16(W, 0));
    int rowmap[32];
    {
        const std::vector<int> act = act_cols16(W, 0);
        for (int i = 0; i < 32; ++i) rowmap[i] = (i == 3) ? 0 : -1;     // output row 3 <- linear_density row 0
        for (int ks = 0; ks < W / 16; ++ks) emit_quad(st, p->linear_density_w, 1, W, 0, rowmap, act.data() + 16 * ks);
    }
    emit_layer(st, p->linear_d_w, W / 2, W + in_d, NT / 2, act_cols16(W, 0));
    {
        const std::vector<int> act = act_cols16(W / 2, 0);
        for (int i = 0; i < 32; ++i) rowmap[i] = (i < 3) ? i : -1;      // output rows 0..2 <- linear_color rows 0..2
        for (int ks = 0; ks < W / 32; ++ks) emit_quad(st, p->linear_color_w, 3, W / 2, 0, rowmap, act.data() + 16 * ks);
    }
    st.resize(st.size() + (size_t)(TAIL_QUADS - TAIL_USED) * (QUAD_BYTES / 2), 0);
    MN_CHECK_ARG(st.size() * 2 == L.stream_bytes, "internal: bf16 stream %zu != %u", st.size() * 2, L.stream_bytes);
    uint32_t* hdr = (uint32_t*)blob;
    hdr[0] = BLOB_MAGIC; hdr[1] = 2; hdr[2] = D; hdr[3] = W; hdr[4] = (uint32_t)net->skip; hdr[5] = net->L_x; hdr[6] = net->L_d;
    hdr[7] = L.stream_off; hdr[8] = L.stream_bytes; hdr[9] = L.stream_bytes; hdr[10] = L.side_off; hdr[11] = L.side_floats;
    hdr[12] = 2;   // stream element bytes
    memcpy((char*)blob + L.stream_off, st.data(), L.stream_bytes);
    float* side = (float*)((char*)blob + L.side_off);
    for (int l = 0; l < D; ++l) memcpy(side + L.bias_trunk + (size_t)l * W, p->linear_x_b[l], W * 4);
    memcpy(side + L.bias_feat, p->linear_feat_b, W * 4);
    memcpy(side + L.bias_d, p->linear_d_b, (W / 2) * 4);
    memcpy(side + L.head_b, p->linear_color_b, 3 * 4);
    side[L.head_b + 3] = p->linear_density_b[0];
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
    unsigned n_wtiles;          // 32-point tiles (n_rays * tpr)
    unsigned n_rays;
    unsigned n_iter;            // tile pairs per wave (the same for every wave: the ring barriers are workgroup-wide)
    unsigned ppr;               // ray-major walk: pairs per ray (tpr / 2); 0: flat walk
    int S, tpr, D, skip_layer;
    unsigned stream_bytes, side_floats;
    unsigned o_bias_trunk, o_bias_feat, o_bias_d, o_head_b, o_wdir_t;
    unsigned long long* diag;   // MN_DIAG builds only: per-wave cycle sums of the kernel's segments
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
// here; tests/test_packing_cpu.py disassembles the object and checks that every M0 write is ours).  Earlier form: SGPR base +
// save/restore per DMA = ten instructions per quad, 8 % of the kernel (ablation build, profiles/r02_bf16_ablation.json).
__device__ __forceinline__ void bdma_set_m0(unsigned lds_in) {
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_in);      // wave-uniform by construction; pin to an SGPR
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_addr) : "memory");
}
template <int IMM>
__device__ __forceinline__ void bdma16(const char* gaddr_lane) {
    asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gaddr_lane), "i"(IMM) : "memory");
}
// DMA number i (0..7) of the slot being fetched
__device__ __forceinline__ void bring_dma(const BRing& r, int i) {
#ifdef MN_BF16_NODMA
    return;
#endif
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
// consume the next slot: everything but the 8 DMAs issued during the phase that ends here has landed (slot p+1 was
// issued two phases ago); barrier; slot p+2 streams into ring[(p+2)%3] == ring[(p-1)%3] during the new phase.
// Other vector-memory operations of the wave (input prefetches, result stores) share the counter and retire in order:
// they can only make this wait stricter.
__device__ __forceinline__ void bring_advance(BRing& r) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#ifndef MN_BF16_NOBARRIER                                    // ablation builds (timing experiments only, results are garbage)
    __syncthreads();
#endif
    bring_next_fetch(r);
    r.read_slot = (r.read_slot + 1 == BNSLOT) ? 0 : r.read_slot + 1;
}
// fragment at slot position qs; positions 1..8 also issue one of the slot's DMAs (never a burst: each is ~5 issue slots)
__device__ __forceinline__ u32x4b bring_read(const char* smem, const BRing& r, int lane, int qs) {
    if (qs >= 1 && qs <= BDMA) bring_dma(r, qs - 1);
    return *(const u32x4b*)(smem + r.read_slot * BSLOT_BYTES + lane * 16 + qs * QUAD_BYTES);
}

// two accumulator registers -> one dword of the next layer's B fragment (round to nearest even), optionally ReLU on the
// packed pair: as signed 16-bit integers every negative bf16 (and -0.0) is below zero.  Pinned where it is written.
template <bool RELU>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    unsigned d;
    if (RELU) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(d) : "v"(lo), "v"(hi));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
    return d;
}

// ---------------------------------------------------------------------------------------------
// THE FRAGMENT FILE: all 256 AGPRs, managed by hand.
//
// Two sets (ping-pong between consecutive layers) x NP point tiles x 16 B fragments x 4 registers = 256 = the whole
// accumulation-register file.  With 430 of the wave's 512 registers live, hipcc's allocator could not place these tuples
// (it treats MFMA operands as "either file" values and thrashed: fragments copied AGPR -> VGPR in front of every MFMA,
// 16-register accumulators spilled around the packing, up to 300 spilled registers) -- so the fragments never become
// compiler values at all: they are written with v_accvgpr_write_b32 a[N] and read as the MFMA's B operand a[N:N+3] with N a
// compile-time constant, and the compiler allocates only the VGPR side (accumulators, biases, A pipeline, gamma(x): ~200 of
// 256).  It must keep out of the AGPRs entirely: this file is built with -mllvm -amdgpu-spill-vgpr-to-agpr=0 and contains no
// MFMA builtin; one clobber of a255 makes the kernel descriptor reserve the whole file.
//
// Consequence: the MFMAs are asm statements and hipcc inserts NO hazard wait states around them.  They hold by construction:
//   * a dependent chain on one accumulator needs none;
//   * every other reader of an MFMA result (the packing of a finished tile, the final store) is at least two MFMA issues
//     (>= 64 cycles) behind the MFMA that wrote it -- the packing of point tile 0 starts after the NEXT job's first group, the
//     store is preceded by explicit s_nops;
//   * a fragment register is written at least one whole group (>= 64 cycles) before the MFMA that reads it and never while an
//     MFMA that reads it can be in flight (a layer writes the OTHER set; the two fragments packed across a layer boundary are
//     the last ones that layer reads);
//   * VGPR operands (A fragments, biases, gamma(x)) come from LDS reads the compiler tracks (s_waitcnt before the asm).
// ---------------------------------------------------------------------------------------------
template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}
__host__ __device__ constexpr int frag_reg(int set, int p, int f) { return ((set * NP + p) * 16 + f) * 4; }

template <int R>
__device__ __forceinline__ void agpr_write(unsigned d) { asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(d), "n"(R)); }

// first MFMA of a job (C operand = bias) / accumulate; B operand from the fragment file (IC<R>) or from a VGPR fragment
template <int R>
__device__ __forceinline__ void mfma_first(f32x16& acc, const u32x4b& afrag, IC<R>, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%3:%4], %2" : "=&v"(acc) : "v"(afrag), "v"(c), "n"(R), "n"(R + 3));
}
__device__ __forceinline__ void mfma_first(f32x16& acc, const u32x4b& afrag, const u32x4b& bfrag, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(afrag), "v"(bfrag), "v"(c));
}
template <int R>
__device__ __forceinline__ void mfma_acc(f32x16& acc, const u32x4b& afrag, IC<R>) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(afrag), "n"(R), "n"(R + 3));
}
__device__ __forceinline__ void mfma_acc(f32x16& acc, const u32x4b& afrag, const u32x4b& bfrag) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(afrag), "v"(bfrag));
}

// Pair J (0..3) of a finished 32x32 tile: accumulator registers 4J..4J+3 -> two packed dwords -> registers R0 + 2J, R0 + 2J + 1 of the
// fragment file, where R0 is the first register of fragment 2T (fragments 2T and 2T+1 are adjacent: dwords 0..3 and 4..7).
// Two dwords per statement, interleaved, so that no instruction reads the result of the one in front of it (a lone
// cvt -> max -> write chain costs a wait state per link).
template <bool RELU, int R0, int J>
__device__ __forceinline__ void pack_pair(const f32x16& acc) {
    unsigned t0, t1;
    if (RELU)
        asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\tv_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\t"
                     "v_accvgpr_write_b32 a[%6], %0\n\tv_accvgpr_write_b32 a[%7], %1"
                     : "=&v"(t0), "=&v"(t1) : "v"(acc[4 * J]), "v"(acc[4 * J + 1]), "v"(acc[4 * J + 2]), "v"(acc[4 * J + 3]), "n"(R0 + 2 * J), "n"(R0 + 2 * J + 1));
    else
        asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\t"
                     "v_accvgpr_write_b32 a[%6], %0\n\tv_accvgpr_write_b32 a[%7], %1"
                     : "=&v"(t0), "=&v"(t1) : "v"(acc[4 * J]), "v"(acc[4 * J + 1]), "v"(acc[4 * J + 2]), "v"(acc[4 * J + 3]), "n"(R0 + 2 * J), "n"(R0 + 2 * J + 1));
}
// The NP x 4 pairs of the previous job's packing (point tile 0 first) dealt over the first `groups` groups of a job
__host__ __device__ constexpr int pack_lo(int ks, int groups) { return ks >= groups ? NP * 4 : (NP * 4 * ks) / groups; }
template <bool RELU, int SET, int F0, int KS, int GROUPS>
__device__ __forceinline__ void pack_group(const f32x16 (&prev)[NP]) {
#ifdef MN_BF16_NOPACK
    return;
#endif
    static_for<pack_lo(KS, GROUPS), pack_lo(KS + 1, GROUPS)>([&](auto w_c) __attribute__((always_inline)) {
        constexpr int w = decltype(w_c)::value, p = w >> 2, j = w & 3;
        pack_pair<RELU, frag_reg(SET, p, F0), j>(prev[p]);
    });
}

// 16 floats of a natural-order vector in LDS in accumulator order: register r of lane half hh is feature
// 32t + (r&3) + 8(r>>2) + 4hh -> quarter g (registers 4g..4g+3) is one 16-byte read at 32t + 8g + 4hh
__device__ __forceinline__ void cin_quarter(f32x16& c, const float* vec_t_hh, int g) {
    const f32x4 v = *(const f32x4*)(vec_t_hh + 8 * g);
    c[4 * g + 0] = v[0]; c[4 * g + 1] = v[1]; c[4 * g + 2] = v[2]; c[4 * g + 3] = v[3];
}

// ---------------------------------------------------------------------------------------------
// One job: output tile of 32 features x NP point tiles over KS k-steps, stream quads Q0..Q0+KS-1 of the current body
// (bodies start on a slot boundary, so every ring position below is a compile-time constant).
// csel(p): C operand of point tile p's first MFMA (the bias).  bsrc(p_c, ks_c): B operand -- IC<register> (fragment file) or a
// VGPR fragment.  Group ks = [NP MFMAs on fragment a[(Q0+ks) % DA]] [refill that register with the quad DA positions further
// down the stream: ring advance / DMA issue / ds_read_b128] [hook(ks_c): the previous job's packing, bias reads, ...], pinned.
// QEND/QPAD: stream positions >= QEND skip QPAD quads (the padding at the end of the tail body).
// ---------------------------------------------------------------------------------------------
template <int Q0, int KS, int QEND, int QPAD, typename CSel, typename BSrc, typename Hook>
__device__ __forceinline__ void job(f32x16 (&acc)[NP], CSel csel, BSrc bsrc, u32x4b (&a)[DA], const char* smem, BRing& ring, int lane, Hook hook) {
    static_for<0, KS>([&](auto ks_c) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_c)::value;
        // In-order issue: the wave sits at the SECOND MFMA until the matrix pipe has taken the first (32 cycles), so whatever
        // follows both MFMAs has only the second one's 24 free cycles.  The work of a group is therefore split: the ring
        // bookkeeping (advance, one DMA, the A-pipeline refill of the register the PREVIOUS group consumed) rides behind the first
        // MFMA, the hook behind the second.
        constexpr int q0 = Q0 + ks + DA - 1;                              // stream position being read into register q0 % DA
        constexpr int qn = (q0 >= QEND) ? q0 + QPAD : q0;
        static_for<0, NP>([&](auto p_c) __attribute__((always_inline)) {
            constexpr int p = decltype(p_c)::value;
            if constexpr (ks == 0) mfma_first(acc[p], a[(Q0 + ks) % DA], bsrc(p_c, ks_c), csel(p));
            else mfma_acc(acc[p], a[(Q0 + ks) % DA], bsrc(p_c, ks_c));
            if constexpr (p == 0) {
                if constexpr (qn % BSLOT_QUADS == 0) bring_advance(ring);
                a[q0 % DA] = bring_read(smem, ring, lane, qn % BSLOT_QUADS);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        hook(ks_c);
        // the C operand of the first MFMAs is dead for the compiler once they are issued, but the matrix pipe reads it for a few more
        // cycles: keep its registers out of the allocator's hands until the next group
        if constexpr (ks == 0) { asm volatile("" ::"v"(csel(0))); asm volatile("" ::"v"(csel(NP - 1))); }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <int W, int LX, int LD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_bf16_kernel(const MlpArgsB a) {
    static_assert(W == 256, "bf16 variant: W = 256");
    constexpr int NT = W / 32, KH = W / 16, KPE = enc_ksteps16(LX), IN_X = 3 + 6 * LX, IN_D = 3 + 6 * LD;
    static_assert(KPE == 4 && NT == 8 && KH == 16 && NP == 2, "stream positions and the fragment file are laid out for 63 -> 64 encoded channels, W = 256, 2 point tiles");
    constexpr int BIG = 1 << 30;
    asm volatile("" ::: "a255");                             // reserve the whole accumulation-register file (see THE FRAGMENT FILE)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* side = (float*)(smem + BRING_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hh = lane >> 5;
    float* scratch = side + a.side_floats + wave * (NP * (W / 2));      // per wave, per point tile: hoisted direction bias
    char* pe_lds = (char*)(side + a.side_floats + 4 * NP * (W / 2)) + wave * (NP * enc_ksteps16(LX) * QUAD_BYTES) + lane * 16;   // parked gamma(x) fragments
    for (unsigned i = tid * 4; i < a.side_floats; i += 256 * 4) *(f32x4*)(side + i) = *(const f32x4*)(a.side + i);

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

    u32x4b aq[DA];
    bring_advance(ring);                                     // also publishes the side tables (barrier)
#pragma unroll
    for (int i = 0; i < DA - 1; ++i) aq[i] = bring_read(smem, ring, lane, i);      // position q is read while group q - (DA - 1) computes

    // ---- tile walk: a wave takes PAIRS of consecutive 32-sample tiles.  Ray-major (ppr > 0): a wave walks whole rays, so the
    // hoisted view-direction term is computed once per ray; flat otherwise.  Inputs of the next pair are loaded a pair ahead.
    const unsigned NW = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    auto pair_of = [&](unsigned it) -> unsigned {
        if (a.ppr) { const unsigned blk = it / a.ppr; return (blk * NW + wid) * a.ppr + (it - blk * a.ppr); }
        return it * NW + wid;
    };
    unsigned n_tile[NP];  float nx[NP][7];
    auto load_inputs = [&](unsigned it) __attribute__((always_inline)) {
        const unsigned pr = pair_of(it);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            unsigned t = 2u * pr + p;
            n_tile[p] = t;
            if (t >= a.n_wtiles) t = a.n_wtiles - 1;            // inactive: recompute the last tile, store nothing
            const unsigned ray = t / (unsigned)a.tpr, chunk = t - ray * (unsigned)a.tpr;
            const int sample = (int)chunk * 32 + col;
            const int sc = sample < a.S ? sample : a.S - 1;
            const float* rp = a.rays + (size_t)ray * 6;
#pragma unroll
            for (int e = 0; e < 6; ++e) nx[p][e] = rp[e];
            nx[p][6] = a.z[(size_t)ray * a.S + sc];
        }
    };
    load_inputs(0);
    unsigned bias_ray[NP] = {~0u, ~0u};

    f32x16 acc[NP], prev[NP];
    f32x16 cin, cnext;                                       // bias of the current / next job (shared by the point tiles)
    auto csel1 = [&](int) __attribute__((always_inline)) -> const f32x16& { return cin; };
    u32x4b peb[NP][KPE];

    // ---- bodies (straight-line code, everything static) ------------------------------------------------------------------------
    // A trunk layer reads fragment set SIN (fragments 14, 15 are still being packed from `prev` when it starts), writes set
    // 1 - SIN, leaves its last tile in `prev`; the bias of the NEXT body's first tile is read while the last tile computes.
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
            else return IC<frag_reg(SIN, p, ks)>{};
        };
        static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
            constexpr int t = decltype(t_c)::value;
            auto hook = [&](auto ks_c) __attribute__((always_inline)) {
                constexpr int ks = decltype(ks_c)::value;
                // previous tile -> fragments of the layer that follows it, over groups 0..13 (fragments 14, 15 feed k-steps 14, 15)
                if constexpr (t == 0) pack_group<true, SIN, 14, ks, 14>(prev);
                else pack_group<true, SOUT, 2 * (t - 1), ks, 14>(prev);
                if constexpr (ks >= KS - 4) {                   // bias of the next job (C operand of its first MFMA): read as late as possible
                    const float* v = (t + 1 < NT) ? bias + 32 * (t + 1) + 4 * hh : next_bias + 4 * hh;
                    cin_quarter(cnext, v, ks - (KS - 4));
                }
                if constexpr (SKIP && ks >= KH - 2 && ks < KH - 2 + KPE) {      // gamma(x) fragment of k-step ks + 2
#pragma unroll
                    for (int p = 0; p < NP; ++p) per[p][ks - (KH - 2)] = *(const u32x4b*)(pe_lds + (p * KPE + (ks - (KH - 2))) * QUAD_BYTES);
                }
            };
            job<t * KS, KS, BIG, 0>(acc, csel1, bsrc, aq, smem, ring, lane, hook);
#pragma unroll
            for (int p = 0; p < NP; ++p) prev[p] = acc[p];
            cin = cnext;
        });
    };

#ifdef MN_DIAG
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = bstamp();
#endif
    for (unsigned it = 0; it < a.n_iter; ++it) {
        // ---- prologue: this pair's points, gamma(x) fragments, hoisted view-direction bias ------------------------------------
        unsigned tile[NP]; bool valid[NP]; size_t out_idx[NP];
        float in_o[NP][3], in_d[NP][3], in_z[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            tile[p] = n_tile[p];
            const bool active = tile[p] < a.n_wtiles;
            const unsigned t = active ? tile[p] : a.n_wtiles - 1;
            const unsigned ray = t / (unsigned)a.tpr, chunk = t - ray * (unsigned)a.tpr;
            const int sample = (int)chunk * 32 + col;
            valid[p] = active && sample < a.S;
            out_idx[p] = (size_t)ray * a.S + (sample < a.S ? sample : a.S - 1);
            tile[p] = ray;                                       // from here on: the tile's ray
#pragma unroll
            for (int e = 0; e < 3; ++e) { in_o[p][e] = nx[p][e]; in_d[p][e] = nx[p][3 + e]; }
            in_z[p] = nx[p][6];
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // pts = rays_o + rays_d * z (nerf_process.py:69-70)
            const float pt[3] = {in_o[p][0] + in_d[p][0] * in_z[p], in_o[p][1] + in_d[p][1] * in_z[p], in_o[p][2] + in_d[p][2] * in_z[p]};
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
#pragma unroll
            for (int ks = 0; ks < KPE; ++ks) {
                u32x4b v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = hh ? chan(16 * ks + 8 + 2 * i) : chan(16 * ks + 2 * i);
                    const float hi = hh ? chan(16 * ks + 8 + 2 * i + 1) : chan(16 * ks + 2 * i + 1);
                    v[i] = pack2<false>(lo, hi);
                }
                peb[p][ks] = v;
                *(u32x4b*)(pe_lds + (p * KPE + ks) * QUAD_BYTES) = v;
            }
            // hoisted view-direction term of linear_d (fp32): scratch[n] = b_d[n] + sum_f Wd[n][W+f] * gamma(d/|d|)[f]
            float* sc_p = scratch + p * (W / 2);
            if (tile[p] != bias_ray[p]) {
                bias_ray[p] = tile[p];
                if (p == 1 && tile[1] == tile[0]) {             // both tiles on one ray: copy (same wave: no barrier needed)
#pragma unroll
                    for (int n0 = 0; n0 < W / 2; n0 += 64) sc_p[n0 + lane] = scratch[n0 + lane];
                } else {
                    const float dx = in_d[p][0], dy = in_d[p][1], dz = in_d[p][2];
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
                        sc_p[n] = s;
                    }
                }
            }
        }
        BSTAMP(0);   // prologue
        // ---- layer 0: 8 jobs of 4 k-steps over gamma(x) (VGPR fragments), output into set 0 -----------------------------------------
        {
            const float* b0 = side + a.o_bias_trunk + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) cin_quarter(cin, b0, g);
            auto bsrc = [&](auto p_c, auto ks_c) __attribute__((always_inline)) -> const u32x4b& { return peb[decltype(p_c)::value][decltype(ks_c)::value]; };
            static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                auto hook = [&](auto ks_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value;
                    if constexpr (t > 0) pack_group<true, 0, 2 * (t - 1), ks, KPE>(prev);
                    const float* v = (t + 1 < NT) ? b0 + 32 * (t + 1) : side + a.o_bias_trunk + W + 4 * hh;
                    cin_quarter(cnext, v, ks);
                };
                job<t * KPE, KPE, BIG, 0>(acc, csel1, bsrc, aq, smem, ring, lane, hook);
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
            // The head tiles are short-lived (the density tile is reduced to one register per point tile right after its job),
            // which keeps the VGPR side of the tail away from its limit.
            f32x16 hd[NP], hc[NP], cind[NP], cnextd[NP], cinh;          // density / colour head tiles; per-point-tile direction bias
            float dens[NP];
            auto cseld = [&](int p) __attribute__((always_inline)) -> const f32x16& { return cind[p]; };
            auto cselh = [&](int) __attribute__((always_inline)) -> const f32x16& { return cinh; };
            const float* bf = side + a.o_bias_feat + 4 * hh;
            auto bsrc_in = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(SIN, decltype(p_c)::value, decltype(ks_c)::value)>{}; };
            auto bsrc_out = [&](auto p_c, auto ks_c) __attribute__((always_inline)) { return IC<frag_reg(SOUT, decltype(p_c)::value, decltype(ks_c)::value)>{}; };
            // feature layer: no activation on its outputs; its first job still packs the trunk's last tile (ReLU)
            static_for<0, NT>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                auto hook = [&](auto ks_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value;
                    if constexpr (t == 0) pack_group<true, SIN, 14, ks, 14>(prev);
                    else pack_group<false, SOUT, 2 * (t - 1), ks, 14>(prev);
                    if constexpr (ks >= 12 && t + 1 < NT) {
                        cin_quarter(cnext, bf + 32 * (t + 1), ks - 12);
                    } else if constexpr (ks == 12 && t + 1 == NT) {       // density tile: row 3 = density bias (lane half 0 only)
#pragma unroll
                        for (int r = 0; r < 16; ++r) cnext[r] = 0.0f;
                        const float db = side[a.o_head_b + 3];
                        cnext[3] = hh == 0 ? db : 0.0f;
                    }
                };
                job<t * KH, KH, BIG, 0>(acc, csel1, bsrc_in, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) prev[p] = acc[p];
                cin = cnext;
            });
            // density tile over the trunk output (row 3); packs the feature layer's last tile; reads the direction bias of tile 0
            {
                auto hook = [&](auto ks_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value;
                    pack_group<false, SOUT, 14, ks, KH>(prev);
                    if constexpr (ks >= 12) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) cin_quarter(cnextd[p], scratch + p * (W / 2) + 4 * hh, ks - 12);
                    }
                };
                job<128, KH, BIG, 0>(hd, csel1, bsrc_in, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) cind[p] = cnextd[p];
            }
            // view-direction layer: 4 jobs over the feature layer's output; ReLU'd tiles go into fragments 0..7 of set SIN (the
            // trunk output is dead once the density tile has run)
            static_for<0, NT / 2>([&](auto t_c) __attribute__((always_inline)) {
                constexpr int t = decltype(t_c)::value;
                auto hook = [&](auto ks_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value;
                    if constexpr (t > 0) pack_group<true, SIN, 2 * (t - 1), ks, KH>(prev);
                    if constexpr (t == 0 && ks == 6) {          // the density tile finished >= 12 MFMAs ago: keep its one useful register
#pragma unroll
                        for (int p = 0; p < NP; ++p) asm volatile("v_mov_b32 %0, %1" : "=v"(dens[p]) : "v"(hd[p][3]));
                    }
                    if constexpr (t == 1 && ks == 8) load_inputs(it + 1 < a.n_iter ? it + 1 : it);      // next pair's rays and depths, ~3000 cycles ahead
                    if constexpr (ks >= 12 && t + 1 < NT / 2) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) cin_quarter(cnextd[p], scratch + p * (W / 2) + 32 * (t + 1) + 4 * hh, ks - 12);
                    } else if constexpr (ks == 12 && t + 1 == NT / 2) {   // colour tile: rows 0..2 = colour bias (lane half 0 only)
#pragma unroll
                        for (int r = 0; r < 16; ++r) cinh[r] = 0.0f;
                        const f32x4 hb4 = *(const f32x4*)(side + a.o_head_b);
                        cinh[0] = hh == 0 ? hb4[0] : 0.0f; cinh[1] = hh == 0 ? hb4[1] : 0.0f; cinh[2] = hh == 0 ? hb4[2] : 0.0f;
                    }
                };
                job<144 + t * KH, KH, BIG, 0>(acc, cseld, bsrc_out, aq, smem, ring, lane, hook);
#pragma unroll
                for (int p = 0; p < NP; ++p) { prev[p] = acc[p]; cind[p] = cnextd[p]; }
            });
            // colour tile over the view-direction output (rows 0..2)
            {
                auto hook = [&](auto ks_c) __attribute__((always_inline)) {
                    pack_group<true, SIN, 6, decltype(ks_c)::value, 6>(prev);      // the last direction tile is needed by k-steps 6, 7
                };
                job<208, KH / 2, TAIL_USED, TAIL_QUADS - TAIL_USED>(hc, cselh, bsrc_in, aq, smem, ring, lane, hook);
            }
            // the MFMAs are asm statements: hipcc does not know that `hc` is still in flight (XDL write -> vector-memory read)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (valid[p] && hh == 0) {                      // cat([rgb, density]) NeRF.py:51
                    f32x4 o; o[0] = hc[p][0]; o[1] = hc[p][1]; o[2] = hc[p][2]; o[3] = dens[p];
                    *(f32x4*)(a.out + out_idx[p] * 4) = o;
                }
        };
        if (l < a.D) { layer_01(l); BSTAMP(2); tail(IC<1>{}); }
        else { BSTAMP(2); tail(IC<0>{}); }
        BSTAMP(3);   // tail
    }
#ifdef MN_DIAG
    if (a.diag && lane == 0) {
        unsigned long long* d = a.diag + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = seg[i];
    }
#endif
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
    a.S = S; a.tpr = (S + 31) / 32;
    const long long n_wtiles = (long long)n_rays * a.tpr;
    MN_CHECK_ARG(n_wtiles < (1LL << 30), "too many points for one launch: %lld rays x %d samples", (long long)n_rays, S);
    a.n_wtiles = (unsigned)n_wtiles; a.n_rays = (unsigned)n_rays;
    a.D = net->D;
    a.skip_layer = (net->skip >= 0 && net->skip + 1 < net->D) ? net->skip + 1 : -1;
    a.stream_bytes = L.stream_bytes; a.side_floats = L.side_floats;
    a.o_bias_trunk = L.bias_trunk; a.o_bias_feat = L.bias_feat; a.o_bias_d = L.bias_d; a.o_head_b = L.head_b; a.o_wdir_t = L.wdir_t;
    const size_t lds = BRING_BYTES + (size_t)a.side_floats * 4 + 4 * NP * (256 / 2) * 4 + 4 * NP * enc_ksteps16(10) * QUAD_BYTES;
    MN_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded: %zu bytes", lds);
    auto kern = mlp_bf16_kernel<256, 10, 4>;
    static LdsOptIn opt_in = {};
    if (int rc = ensure_lds_opt_in(opt_in, (const void*)kern)) return rc;
    const int n_cus = device_cus();
    const long long n_pairs = (n_wtiles + 1) / 2;
    const long long n_wg = (n_pairs + 3) / 4;
    const int grid = (int)(n_wg < n_cus ? n_wg : n_cus);
    const long long NW = (long long)grid * 4;
    if (a.tpr % 2 == 0 && n_rays >= NW) {                    // ray-major: every wave gets whole rays
        a.ppr = (unsigned)(a.tpr / 2);
        a.n_iter = (unsigned)((n_rays + NW - 1) / NW) * a.ppr;
    } else {
        a.ppr = 0;
        a.n_iter = (unsigned)((n_pairs + NW - 1) / NW);
    }
#ifdef MN_DIAG
    {   // diagnostic build: run once with stamps and print the per-segment averages (cycles per tile PAIR per wave)
        unsigned long long* dbuf = nullptr;
        const size_t n = (size_t)grid * 4 * 8;
        MN_HIP(hipMalloc(&dbuf, n * 8));
        MN_HIP(hipMemsetAsync(dbuf, 0, n * 8, st));
        a.diag = dbuf;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
        MN_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> hbuf(n);
        MN_HIP(hipMemcpy(hbuf.data(), dbuf, n * 8, hipMemcpyDeviceToHost));
        (void)hipFree(dbuf);
        static const char* names[4] = {"prologue", "layer0", "trunk", "tail"};
        static const double ideal[4] = {0, 2048, 7 * 8192 + 2048, 216 * 64};
        double tot = 0;
        fprintf(stderr, "[mn_diag bf16] grid=%d pairs/wave=%u  cycles per pair (mean over waves; ideal MFMA cycles in brackets):\n", grid, a.n_iter);
        for (int sgi = 0; sgi < 4; ++sgi) {
            double sum = 0;
            for (size_t w = 0; w < (size_t)grid * 4; ++w) sum += (double)hbuf[w * 8 + sgi];
            const double per = sum / ((double)grid * 4) / (double)a.n_iter;
            tot += per;
            fprintf(stderr, "[mn_diag bf16]   %-10s %10.0f  [%6.0f]\n", names[sgi], per, ideal[sgi]);
        }
        fprintf(stderr, "[mn_diag bf16]   %-10s %10.0f  [%6.0f]\n", "total", tot, ideal[1] + ideal[2] + ideal[3]);
        return MI_NERF_OK;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    MN_LAUNCH_CHECK("mlp_bf16_kernel");
    return MI_NERF_OK;
}

}  // namespace minerf
