// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with one wave per SIMD, (a) two alternating accumulator chains with
// operands in registers, (b) the same with the B operands read from LDS (two ds_read_b128 per 8 instructions), (c) four chains.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f32_rate.hip -o mfma_f32_rate.   Prints cycles per matrix instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters)
{
    __shared__ float lds[64 * 16 * 2 + 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 16 * 2; i += blockDim.x) lds[i] = 0.001f * i;
    __syncthreads();
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float4 fa = {1.0f + lane, 2.0f, 3.0f, 4.0f}, fb = {0.5f, 0.25f, 0.125f, 1.0f / (1 + lane)};
    const float4* bp = reinterpret_cast<const float4*>(lds) + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float4 b0 = fb, b1 = fb;
        if (MODE == 1) { b0 = bp[(it & 7) * 64]; b1 = bp[(8 + (it & 7)) * 64]; }
        if (MODE == 2) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, b0.x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, b1.x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, b0.y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, b1.y, a3, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, b0.z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, b1.z, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, b0.w, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, b1.w, a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, b0.x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, b1.x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, b0.y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, b1.y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, b0.z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, b1.z, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, b0.w, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, b1.w, a1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const f32x4 s = a0 + a1 + a2 + a3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main()
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 64); hipMemset(cyc, 0, 64);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char* names[3] = {"two chains, operands in registers", "two chains, B operands from LDS (2 ds_read_b128 / 8)", "four chains, operands in registers"};
    for (int m = 0; m < 3; ++m) printf("v_mfma_f32_16x16x4_f32, one wave per SIMD, %-55s %6.2f cycles per instruction\n", names[m], (double)h[m] / (8.0 * iters));
    return 0;
}
