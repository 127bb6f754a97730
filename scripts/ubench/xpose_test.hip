// Validates the in-register 32x32 transpose network (lane bit <-> register bit) used by the
// fused feature kernel: v_cndmask_b32_dpp for lane bits 0..3, v_permlane16_swap for bit 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "xpose32.hpp"

__global__ void k(const float* in, float* out)
{
    float r[32];
    const int lane = threadIdx.x;
    for (int i = 0; i < 32; ++i) r[i] = in[lane * 32 + i];
    af::feat::xpose32(r, lane);
    for (int i = 0; i < 32; ++i) out[lane * 32 + i] = r[i];
}

int main()
{
    std::vector<float> h(64 * 32), o(64 * 32);
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) h[l * 32 + i] = (float)(l * 100 + i);
    float *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, h.size() * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) {
            const int half = l >> 5, ll = l & 31;
            const float want = (float)((half * 32 + i) * 100 + ll);   // (reg i, lane ll) <- (reg ll, lane i)
            if (o[l * 32 + i] != want) {
                if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, i, o[l * 32 + i], want);
                ++bad;
            }
        }
    printf("xpose32: %d mismatches\n", bad);
    return bad != 0;
}
