// stream_sweep.hip -- which launch shape gets the most out of HBM for the library's streaming kernels?
// A 2:1 kernel of to_mono's shape (one float4 in, one float2 out per lane and step) and a 1:1 copy, swept over
// block size, grid size, loads in flight per lane, range-per-workgroup vs grid-stride walk, plain vs
// non-temporal accesses.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/stream_sweep.hip -o gpurun_out/stream_sweep && gpurun_out/stream_sweep
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));

template <int U, bool RANGE, bool NT, int COPY>
__global__ void k(const vf4* __restrict__ src, void* __restrict__ dstv, size_t n)
{
    const size_t B = blockDim.x;
    size_t i, hi, step;
    if (RANGE) {                                   // every workgroup walks its own contiguous range
        const size_t per = (n + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * per;
        hi = lo + per < n ? lo + per : n;
        i = lo + threadIdx.x;
        step = B;
    } else {                                       // grid-stride
        hi = n;
        i = (size_t)blockIdx.x * B + threadIdx.x;
        step = (size_t)gridDim.x * B;
    }
    float keep = 0.0f;
    auto emit = [&](size_t at, vf4 v) {
        if (COPY == 2) {                           // read-only: fold, store only on a value that never occurs
            keep += (v.x + v.y) + (v.z + v.w);
        } else if (COPY) {
            vf4* d = static_cast<vf4*>(dstv) + at;
            if (NT) __builtin_nontemporal_store(v, d); else *d = v;
        } else {
            vf2 o;
            o.x = (v.x + v.y) / 2.0f;
            o.y = (v.z + v.w) / 2.0f;
            vf2* d = static_cast<vf2*>(dstv) + at;
            if (NT) __builtin_nontemporal_store(o, d); else *d = o;
        }
    };
    for (; i + (U - 1) * step < hi; i += U * step) {
        vf4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * step) : src[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u) emit(i + u * step, v[u]);
    }
    for (; i < hi; i += step) emit(i, src[i]);
    if (COPY == 2 && keep == 12345.678f) static_cast<float*>(dstv)[threadIdx.x] = keep;
}

template <int U, bool RANGE, bool NT, int COPY>
float run(const vf4* src, void* dst, size_t n, int grid, int block)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<U, RANGE, NT, COPY>), dim3(grid), dim3(block), 0, 0, src, dst, n);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<U, RANGE, NT, COPY>), dim3(grid), dim3(block), 0, 0, src, dst, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / reps;
}

template <int U, bool RANGE, bool NT, int COPY>
void sweep(const vf4* src, void* dst, size_t n)
{
    const double bytes = (double)n * (COPY == 2 ? 16.0 : COPY ? 32.0 : 24.0);
    for (int block : {256, 512, 1024}) {
        const int full = (int)((n + (size_t)block * U - 1) / ((size_t)block * U));
        for (int grid : {512, 1024, 2048, 4096, 8192, 16384, full}) {
            if (grid > full) continue;
            const float ms = run<U, RANGE, NT, COPY>(src, dst, n, grid, block);
            printf("%s U=%d %s %s block=%4d grid=%8d  %.4f ms  %.0f GB/s\n", COPY == 2 ? "read " : COPY ? "copy " : "mono2", U,
                   RANGE ? "range " : "stride", NT ? "nt   " : "plain", block, grid, ms, bytes / ms / 1e6);
        }
    }
}

int main(int argc, char** argv)
{
    const bool only_read = argc > 1;
    const size_t n = (size_t)1 << 26;              // float4 elements: 1 GiB in
    vf4* src;
    void* dst;
    hipMalloc(&src, n * sizeof(vf4));
    hipMalloc(&dst, n * sizeof(vf4));
    hipMemset(src, 0, n * sizeof(vf4));
    hipMemset(dst, 0, n * sizeof(vf4));
    // warm the clocks
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<4, true, true, 1>), dim3(2048), dim3(256), 0, 0, src, dst, n);
    hipDeviceSynchronize();
#define SW(U) if (!only_read) {                                 \
    sweep<U, true, true, 0>(src, dst, n);  \
    sweep<U, false, true, 0>(src, dst, n); \
    sweep<U, true, false, 0>(src, dst, n); \
    sweep<U, false, false, 0>(src, dst, n); }
    SW(1) SW(2) SW(4) SW(8)
    sweep<4, true, true, 1>(src, dst, n);
    sweep<4, false, true, 1>(src, dst, n);
    sweep<4, false, false, 1>(src, dst, n);
    sweep<8, false, true, 1>(src, dst, n);
    sweep<1, false, true, 2>(src, dst, n);
    sweep<2, true, true, 2>(src, dst, n);
    sweep<4, true, true, 2>(src, dst, n);
    sweep<4, true, false, 2>(src, dst, n);
    sweep<8, true, true, 2>(src, dst, n);
    sweep<4, false, true, 2>(src, dst, n);
    hipDeviceSynchronize();
    printf("done %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
