// common.h -- error plumbing and small device helpers shared by all translation units.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mi_nerf.h"

namespace minerf {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define MN_CHECK_ARG(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            ::minerf::set_error(__VA_ARGS__);    \
            return MI_NERF_EINVAL;               \
        }                                        \
    } while (0)

#define MN_HIP(call)                                                   \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return ::minerf::hip_fail(e__, #call);  \
    } while (0)

#define MN_LAUNCH_CHECK(name)                                                \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) return ::minerf::hip_fail(e__, "launch " name); \
    } while (0)

// Per-DEVICE launch state (a process may drive several GPUs: the 160 KB dynamic-LDS opt-in is an attribute of the function ON a
// device, and the CU count is a property of the device): small arrays indexed by the current device ordinal.
// The entry points are re-entrant per stream and callable from several host threads: the only state the library keeps between calls are
// these per-device words, atomics written with the same value by whoever gets there first (hipFuncSetAttribute is idempotent), and the
// thread-local error text.  tests/test_gpu_streams.py.
constexpr int MN_MAX_DEVICES = 64;
struct LdsOptIn { std::atomic<bool> done[MN_MAX_DEVICES]; };
int device_cus();                                             // multiProcessorCount of the current device (cached per device)
int ensure_lds_opt_in(LdsOptIn& state, const void* kernel);  // hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB), once per device

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// An MFMA written as an asm statement keeps WRITING its destination tuple until P + 4 wait states after issue (8-pass 16x16x32: 12), and hipcc
// does not know: an element of the tuple that no later code reads is free for the allocator the moment the statement ends, and whatever
// lands there first (an address, a shuffle result) is overwritten when the matrix result arrives.  That took down round 3's NOPACK ablation
// build and sat as a 1-2 wait-state near miss in the shipped STASH kernel (colour tile element 3 vs. a ds_bpermute of
// finish_masks).  Rule: every tuple with elements the code does not read (density tile: only [3]; colour tile: only [0..2]) is named whole by
// keep_tuple() at a point >= 12 wait states behind its last MFMA.  tools/mfma_hazard_check.py checks the objects (tests/test_packing_cpu.py).
__device__ __forceinline__ void keep_tuple(const f32x4& x) { asm volatile("" ::"v"(x)); }

// ---------------------------------------------------------------------------------------------
// counter-based uniform generator (numpy mirror: oracle/restate.py counter_uniform)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline float counter_uniform(uint32_t seed, uint32_t stream_id, uint32_t ray, uint32_t sample) {
    uint32_t h = fmix32(ray + 0x9E3779B9u * seed + 0x632BE5ABu);
    h = fmix32(h ^ (sample * 0x85EBCA6Bu + stream_id * 0xC2B2AE35u + 0x27D4EB2Fu));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------------------------
// stratified coarse depths and their jitter (stages.hip: stratified_kernel; mlp_bf16.hip draws them in its own prologue)
// ---------------------------------------------------------------------------------------------
// torch.linspace(0,1,S)[i] in fp32 (symmetric fill: ATen RangeFactories) and z = near*(1-t) + far*t
__device__ __forceinline__ float strat_edge(int i, int S, float step, float near_, float far_) {
    const float t = (S == 1) ? 0.0f : (i < S / 2) ? step * (float)i : 1.0f - step * (float)(S - 1 - i);   // steps=1 -> [start]
    return near_ * (1.0f - t) + far_ * t;               // nerf_process.py:53
}

// Where a stage's uniforms come from: an explicit [n, S] tensor (injected randomness: parity tests, the training path), or -- values ==
// NULL -- the counter-based generator evaluated in the consuming kernel itself, keyed on (seed, stream, ray0 + ray, sample): the
// values mi_nerf_fill_uniform would have written, without the tensor, its launch or its HBM round trip.
struct Jitter {
    const float* values;
    uint32_t seed, stream;
    long long ray0;
};
__device__ __forceinline__ float jitter_at(const Jitter& j, long long ray, int sample, int S) {
    return j.values ? j.values[ray * S + sample] : counter_uniform(j.seed, j.stream, (uint32_t)(j.ray0 + ray), (uint32_t)sample);
}

// what a kernel needs to draw the coarse depths itself (host side of mlp_rays_bf16's strat argument)
struct StratDraw { float near_, far_; const float* t_rand; uint32_t seed; int64_t ray0; float* z_out; };
// what the bf16 coarse launch needs to ALSO do render_rays' middle (composite the coarse pass, resample, merge: stage_composite_fine_z) for the
// rays its workgroups wholly own -- offered by mi_nerf_render_rays, taken by mlp_rays_bf16 only for small launches (`taken` says which)
struct FineDraw { int Nf, det; const float* u; uint32_t seed; int64_t ray0; float *rgb_c, *disp_c, *w_c, *z_f; bool taken; };
// one stratified coarse depth (nerf_process.py:42-60): sample i of S on ray `ray`
__device__ __forceinline__ float stratified_depth(long long ray, int i, int S, float step, float near_, float far_, const Jitter& t_rand) {
    const float zi = strat_edge(i, S, step, near_, far_);
    const float lower = (i == 0) ? zi : 0.5f * (zi + strat_edge(i - 1, S, step, near_, far_));       // :55,57
    const float upper = (i == S - 1) ? zi : 0.5f * (strat_edge(i + 1, S, step, near_, far_) + zi);   // :55,56
    return lower + (upper - lower) * jitter_at(t_rand, ray, i, S);     // :60
}

// ---------------------------------------------------------------------------------------------
// accurate sin / cos for positional encoding.  Arguments are 2^k * x (exact in fp32) and reach
// ~3e3 rad for lego, so the reduction matters: 3-term Cody-Waite with FMA (pi/2 = HI + MID + LO),
// then the Cephes single-precision minimax polynomials on [-pi/4, pi/4].  |error| <~ 1.5e-7.
// quad_shift 0 -> sin(y), 1 -> cos(y) (cos y = sin(y + pi/2): same reduction, next quadrant).
// Beyond 2^22 the float multiple count is no longer exact: fall back to the libm path.
// ---------------------------------------------------------------------------------------------
// The three stages are separate functions so that a caller can spread one evaluation over several instruction groups
// (mlp_fp32.hip computes the NEXT tile's gamma(x) under the current tile's MFMAs); sin_cos_fast is their composition.
__device__ __forceinline__ void sc_reduce(float y, int quad_shift, float& r, int& q) {      // Cody-Waite, 3 terms
    const float nf = __builtin_rintf(y * 0.636619772367581343f);
    r = __builtin_fmaf(-nf, 1.57079637050628662109375f, y);
    r = __builtin_fmaf(-nf, -4.371138828673793e-08f, r);
    r = __builtin_fmaf(-nf, -1.7151245100058819e-15f, r);
    q = (int)nf + quad_shift;
}
__device__ __forceinline__ void sc_poly_sin(float r, float& r2, float& sp) {               // Cephes sinf kernel
    r2 = r * r;
    sp = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, r2, -1.6666654611e-1f);
    sp = __builtin_fmaf(sp * r2, r, r);
}
__device__ __forceinline__ void sc_poly_cos(float r2, float& cp) {                          // Cephes cosf kernel
    cp = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, r2, 4.166664568298827e-2f);
    cp = __builtin_fmaf(cp * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
}
__device__ __forceinline__ void sc_poly(float r, float& sp, float& cp) {
    float r2;
    sc_poly_sin(r, r2, sp);
    sc_poly_cos(r2, cp);
}
__device__ __forceinline__ float sc_select(float sp, float cp, int q) {
    const float v = (q & 1) ? cp : sp;
    return (q & 2) ? -v : v;
}
__device__ __forceinline__ float sin_cos_fast(float y, int quad_shift) {
    float r, sp, cp;
    int q;
    sc_reduce(y, quad_shift, r, q);
    sc_poly(r, sp, cp);
    return sc_select(sp, cp, q);
}
// |y| below this bound keeps the float multiple count exact; callers branch ONCE per point on the largest
// argument and use the libm path (sinf/cosf, Payne-Hanek) for anything bigger or non-finite.
constexpr float SINCOS_FAST_LIMIT = 4.0e6f;
__device__ __forceinline__ float sin_cos_slow(float y, int quad_shift) { return quad_shift ? cosf(y) : sinf(y); }

}  // namespace minerf
