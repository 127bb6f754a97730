// Micro-benchmark: LDS-array cycles of the LDS instructions the hop-256 feature kernel is made of,
// by base alignment.  Each mode is its own kernel (template instance) so that a
//   rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- ./lds_cost
// pass lists the counters per mode; stdout gives wall cycles per wave-instruction per CU.
// Build: hipcc -O3 --offload-arch=gfx950 lds_cost.hip -o lds_cost
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kWaves = 8;
constexpr int kWaveDw = 16 * 96 + 64;   // per-wave LDS region (dwords): 16 rows of up to 96 dwords

template <int OFF, int STRIDE>
__device__ __forceinline__ void addtid16(unsigned m0, const float (&v)[16])
{
#define ONE(i) "ds_write_addtid_b32 %" #i " offset:%c[o" #i "]\n\t"
    asm volatile("s_mov_b32 m0, %[m]\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %[v0] offset:%c[o0]\n\tds_write_addtid_b32 %[v1] offset:%c[o1]\n\t"
                 "ds_write_addtid_b32 %[v2] offset:%c[o2]\n\tds_write_addtid_b32 %[v3] offset:%c[o3]\n\t"
                 "ds_write_addtid_b32 %[v4] offset:%c[o4]\n\tds_write_addtid_b32 %[v5] offset:%c[o5]\n\t"
                 "ds_write_addtid_b32 %[v6] offset:%c[o6]\n\tds_write_addtid_b32 %[v7] offset:%c[o7]\n\t"
                 "ds_write_addtid_b32 %[v8] offset:%c[o8]\n\tds_write_addtid_b32 %[v9] offset:%c[o9]\n\t"
                 "ds_write_addtid_b32 %[v10] offset:%c[o10]\n\tds_write_addtid_b32 %[v11] offset:%c[o11]\n\t"
                 "ds_write_addtid_b32 %[v12] offset:%c[o12]\n\tds_write_addtid_b32 %[v13] offset:%c[o13]\n\t"
                 "ds_write_addtid_b32 %[v14] offset:%c[o14]\n\tds_write_addtid_b32 %[v15] offset:%c[o15]"
                 :
                 : [m] "s"(m0), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]),
                   [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]),
                   [v11] "v"(v[11]), [v12] "v"(v[12]), [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15]),
                   [o0] "n"(OFF), [o1] "n"(OFF + STRIDE), [o2] "n"(OFF + 2 * STRIDE), [o3] "n"(OFF + 3 * STRIDE),
                   [o4] "n"(OFF + 4 * STRIDE), [o5] "n"(OFF + 5 * STRIDE), [o6] "n"(OFF + 6 * STRIDE),
                   [o7] "n"(OFF + 7 * STRIDE), [o8] "n"(OFF + 8 * STRIDE), [o9] "n"(OFF + 9 * STRIDE),
                   [o10] "n"(OFF + 10 * STRIDE), [o11] "n"(OFF + 11 * STRIDE), [o12] "n"(OFF + 12 * STRIDE),
                   [o13] "n"(OFF + 13 * STRIDE), [o14] "n"(OFF + 14 * STRIDE), [o15] "n"(OFF + 15 * STRIDE)
                 : "memory", "m0");
#undef ONE
}

// MODE: 0 addtid rows 256 B apart, base 256-B aligned      1 same, base + 128 B     2 same, base + 16 B
//       3 addtid rows 272 B apart (the kernel's T1 today)  4 addtid rows 384 B apart (128-B aligned, alternating halves)
//       5 ds_write_b32 lane-consecutive, aligned           6 ds_write_b32 lane-consecutive, + 16 B
//       7 ds_read_b32 lane-consecutive                     8 ds_read_b64 lane-consecutive
//       9 ds_read_b128 lane-consecutive                   10 ds_write_b64 lane-consecutive   11 ds_write_b128
template <int MODE>
__global__ __launch_bounds__(64 * kWaves) void k(float* out, int iters)
{
    __shared__ __attribute__((aligned(256))) float lds[kWaves * kWaveDw + 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* wb = lds + w * kWaveDw;             // kWaveDw * 4 = 6400 B = 25 * 256: aligned
    const unsigned m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uintptr_t>(wb));
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane * 0.5f + i;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) addtid16<0, 256>(m0, v);
        else if constexpr (MODE == 1) addtid16<128, 256>(m0, v);
        else if constexpr (MODE == 2) addtid16<16, 256>(m0, v);
        else if constexpr (MODE == 3) addtid16<0, 272>(m0, v);
        else if constexpr (MODE == 4) addtid16<0, 384>(m0, v);
        else if constexpr (MODE == 5 || MODE == 6) {
            float* d = wb + lane + (MODE == 6 ? 4 : 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) d[64 * i] = v[i];
            asm volatile("" ::: "memory");
        } else if constexpr (MODE == 7) {
            const float* s = wb + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) { float x; asm volatile("ds_read_b32 %0, %1 offset:%c2" : "=v"(x) : "v"((unsigned)(uintptr_t)s), "n"(256 * i)); acc += x; }
        } else if constexpr (MODE == 8) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const unsigned a = (unsigned)(uintptr_t)(wb + 2 * lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) { f2 x; asm volatile("ds_read_b64 %0, %1 offset:%c2" : "=v"(x) : "v"(a), "n"(512 * (i & 7))); acc += x.x + x.y; }
        } else if constexpr (MODE == 9) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const unsigned a = (unsigned)(uintptr_t)(wb + 4 * lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) { f4 x; asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(x) : "v"(a), "n"(1024 * (i & 3))); acc += x.x + x.w; }
        } else if constexpr (MODE == 10) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const unsigned a = (unsigned)(uintptr_t)(wb + 2 * lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) { f2 x = {v[i], v[(i + 1) & 15]}; asm volatile("ds_write_b64 %0, %1 offset:%c2" : : "v"(a), "v"(x), "n"(512 * (i & 7)) : "memory"); }
        } else {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const unsigned a = (unsigned)(uintptr_t)(wb + 4 * lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) { f4 x = {v[i], v[(i + 1) & 15], v[(i + 2) & 15], v[(i + 3) & 15]}; asm volatile("ds_write_b128 %0, %1 offset:%c2" : : "v"(a), "v"(x), "n"(1024 * (i & 3)) : "memory"); }
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE>
void run(const char* name, float* d_out)
{
    const int iters = 4000, blocks = 512;          // two 8-wave workgroups per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * kWaves), 0, 0, d_out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * kWaves), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_cu = (double)iters * 16 * kWaves * 2;
    printf("mode %2d %-44s %.3f ms -> %.2f cycles per wave-instruction per CU (at 2.4 GHz)\n", MODE, name, ms,
           ms * 1e-3 * 2.4e9 / insts_per_cu);
}

int main()
{
    float* d_out; hipMalloc(&d_out, 512 * 512 * sizeof(float));
    run<0>("addtid rows 256 B, base aligned 256", d_out);
    run<1>("addtid rows 256 B, base + 128 B", d_out);
    run<2>("addtid rows 256 B, base + 16 B", d_out);
    run<3>("addtid rows 272 B (T1 today)", d_out);
    run<4>("addtid rows 384 B", d_out);
    run<5>("ds_write_b32 consecutive, aligned", d_out);
    run<6>("ds_write_b32 consecutive, + 16 B", d_out);
    run<7>("ds_read_b32 consecutive", d_out);
    run<8>("ds_read_b64 consecutive", d_out);
    run<9>("ds_read_b128 consecutive", d_out);
    run<10>("ds_write_b64 consecutive", d_out);
    run<11>("ds_write_b128 consecutive", d_out);
    return 0;
}
