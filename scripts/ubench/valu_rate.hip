// Micro-benchmark: sustained issue rate of plain and packed f32 VALU ops on gfx950 at
// 1, 2, 4, 8 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters)
{
    float a[16];
    f32x2 p[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    const float c = 1.000001f, d = 0.999f;
    const f32x2 c2 = {c, c}, d2 = {d, d};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], c, d);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], c2, d2);
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = a[i] + d;
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = p[i] + d2;
        } else if (MODE == 4) {            // butterfly-like: 2 VGPR sources, fresh destination
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float x = a[i], y = a[i + 8];
                    a[i] = x + y;
                    a[i + 8] = x - y;
                }
            }
        } else if (MODE == 5) {            // fma with three VGPR operands
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[(i + 5) & 15], a[(i + 11) & 15], a[i]);
        } else if (MODE == 6) {            // butterfly shape of fft_regs.hpp: two VGPRs + a 32-bit literal (v_fmamk / v_fmaak)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[(i + 5) & 15], 0.923879504f + 0.001f * i, a[i]);
        } else if (MODE == 7) {            // v_permlane32_swap
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 1]), false, false);
                    a[i] = __uint_as_float(q[0]); a[i + 1] = __uint_as_float(q[1]);
                }
        } else if (MODE == 8) {            // v_permlane16_swap
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 1]), false, false);
                    a[i] = __uint_as_float(q[0]); a[i + 1] = __uint_as_float(q[1]);
                }
        } else if (MODE == 9) {            // v_mul_f32, two VGPRs
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = a[i] * a[(i + 5) & 15];
        } else {                           // the FFT's mix: 2 x (fmamk, fmamk, fma3) + 2 add
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    a[i] = __builtin_fmaf(a[i + 1], 0.41421f, a[i]);
                    a[i + 1] = __builtin_fmaf(a[i + 2], a[i + 3], a[i + 1]);
                    a[i + 2] = a[i + 2] + a[i];
                    a[i + 3] = __builtin_fmaf(a[i], -0.7071f, a[i + 3]);
                }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int waves_per_simd, float* d_out)
{
    const int iters = 2000;
    const int threads = 256;                       // 4 waves = 1 per SIMD
    const int blocks = 256 * waves_per_simd;       // one block per CU per wave/SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = (double)iters * 8 * ((MODE == 1 || MODE == 3 || MODE == 7 || MODE == 8) ? 8 : 16);
    const double waves_per_simd_d = waves_per_simd;
    const double cyc = ms * 1e-3 * 2.4e9;          // at nominal 2.4 GHz
    printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz), %.1f TFLOP/s\n",
           name, waves_per_simd, ms, cyc / (insts_per_wave * waves_per_simd_d),
           insts_per_wave * waves_per_simd_d * 1024 * 64 * (MODE % 2 ? 2 : 1) * (MODE < 2 ? 2 : 1) / (ms * 1e-3) / 1e12);
}
int main()
{
    float* d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w, d_out);
        run<1>("v_pk_fma_f32", w, d_out);
        run<2>("v_add_f32", w, d_out);
        run<3>("v_pk_add_f32", w, d_out);
        run<4>("add/sub 2vgpr", w, d_out);
        run<5>("fma 3vgpr", w, d_out);
        run<6>("fmamk literal", w, d_out);
        run<7>("permlane32_swap", w, d_out);
        run<8>("permlane16_swap", w, d_out);
        run<9>("v_mul 2vgpr", w, d_out);
        run<10>("fft mix", w, d_out);
    }
    return 0;
}
