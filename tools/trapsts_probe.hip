// Probe: do the sticky exception bits of TRAPSTS (hwreg 3, EXCP = bits 8:0: invalid, input-denormal, div0, OVERFLOW (bit 3), underflow, inexact, int-div0)
// accumulate for v_cvt_pk_f16_f32 / v_fma_mixlo_f16 on gfx950 with traps disabled, and can a wave clear and read them (s_setreg / s_getreg)?
// If so the split-precision forward gets its range contract for two scalar instructions per 32-point unit.
//   hipcc --offload-arch=gfx950 -O2 tools/trapsts_probe.hip -o tools/trapsts_probe.bin && tools/trapsts_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void probe(const float* in, unsigned* out, int n_cases) {
    const int lane = threadIdx.x;
    for (int c = 0; c < n_cases; ++c) {
        float x = in[c * 64 + lane];
        unsigned before, after, after_clear, packed, lo;
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0\n\ts_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(before));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(packed) : "v"(x));
        asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(packed), "v"(-2048.0f), "v"(x * 2048.0f));
        asm volatile("s_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after) : "v"(packed), "v"(lo));
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0\n\ts_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after_clear));
        if (lane == 0) { out[c * 5 + 0] = before; out[c * 5 + 1] = after; out[c * 5 + 2] = after_clear; }
        if (lane == 5) { out[c * 5 + 3] = packed; out[c * 5 + 4] = lo; }
    }
}

int main() {
    // case c: every lane 1.0 except lane 5 = the probe value (a single lane must be enough to set the wave's bit)
    const float vals[] = {1.0f, 65504.0f, 65519.9f, 65520.0f, 70000.0f, -70000.0f, 1e30f, INFINITY, NAN, 1e-8f, 0.1f};
    const int n = sizeof(vals) / sizeof(vals[0]);
    float h[n * 64];
    for (int c = 0; c < n; ++c) for (int l = 0; l < 64; ++l) h[c * 64 + l] = (l == 5) ? vals[c] : 1.0f;
    float* din; unsigned* dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, n * 5 * 4);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, n);
    unsigned o[n * 5];
    if (hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
    printf("# TRAPSTS.EXCP bits: 0 invalid, 1 input denormal, 2 div0, 3 OVERFLOW, 4 underflow, 5 inexact, 6 int div0\n");
    for (int c = 0; c < n; ++c)
        printf("lane 5 = %-12g  cleared 0x%03x  after cvt_pk_f16 + fma_mixlo 0x%03x  cleared again 0x%03x   f16 pair 0x%08x  lo 0x%04x\n", vals[c], o[c * 5], o[c * 5 + 1], o[c * 5 + 2],
               o[c * 5 + 3], o[c * 5 + 4] & 0xffff);
    return 0;
}
