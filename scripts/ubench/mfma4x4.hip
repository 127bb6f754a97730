// Layout probe of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4 outer products, K = 1):
// which lane supplies A_b[i] / B_b[j], which lane / VGPR receives D_b[i][j].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out)
{
    const int l = threadIdx.x;
    // A = 1 + i (i = l % 4) + 10 * block ; B = 100 * (1 + j) ; so D_b[i][j] = (1 + i + 10 b) * 100 (1 + j)
    const float a = 1.0f + (l & 3) + 10.0f * (l >> 2);
    const float b = 100.0f * (1.0f + (l & 3));
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main()
{
    float* d; hipMalloc(&d, 256 * sizeof(float));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok_rowmajor = 1, ok_colmajor = 1;   // rowmajor: VGPR r = i, lane%4 = j ; colmajor: VGPR r = j, lane%4 = i
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2, q = l & 3;
            const float vr = (1.0f + r + 10.0f * blk) * 100.0f * (1.0f + q);   // i = r, j = q
            const float vc = (1.0f + q + 10.0f * blk) * 100.0f * (1.0f + r);   // i = q, j = r
            if (h[l * 4 + r] != vr) ok_rowmajor = 0;
            if (h[l * 4 + r] != vc) ok_colmajor = 0;
        }
    printf("D[vgpr=i][lane%%4=j]: %d   D[vgpr=j][lane%%4=i]: %d   lane5: %g %g %g %g\n", ok_rowmajor, ok_colmajor,
           h[20], h[21], h[22], h[23]);
    return 0;
}
