// features_dft.hip -- STFT power + mel for frame lengths the register FFT does not factor (n_fft outside
// {1024, 512, 256}; features.hpp, namespace dft): the DFT as a matrix product on v_mfma_f32_16x16x4_f32.
//
// Per wave and 16-frame unit:  for every tile of 16 bins
//     D_re, D_im [16 bins x 16 frames] = sum over K groups of 16 samples:  T_re/im^T (A, fragments from L2)  x  frames^T (B, LDS)
//     P = D_re^2 + D_im^2                       (lane (frame n, kq) holds bins 16 bt + 4 kq + 0..3 of frame n)
//     power output: P goes to HBM;  mel output: mel^T[16 mt .. + 16][frame] += W fragment (A) x P (B, as it sits)
// The K index of MFMA step j of a group is 16 g + 4 kq + j: a lane's four B operands are one 16-byte LDS read of
// consecutive samples, its four A operands one 16-byte fragment load (the host lays the fragments out in that order).
// Bounded by the f32 matrix pipe: 2 x (n_fft / 2 + 1) x n_fft multiply-adds per frame (docs/OPS.md "STFT").
#include "af_common.hpp"
#include "features_dev.hpp"

namespace af {
namespace feat {
namespace dft {

using v2::UnitDesc;

__device__ __forceinline__ UnitDesc load_unit(const UnitDesc* units, int64_t u)
{
    // one scalar 32-byte load (wave-uniform index)
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) u32x8* cptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const u32x8 r = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(units))[lo];
    UnitDesc d;
    d.pcm0 = (int64_t)(((unsigned long long)r[1] << 32) | r[0]);
    d.out0 = (int64_t)(((unsigned long long)r[3] << 32) | r[2]);
    d.n_avail = (int)r[4];
    d.n_rows = (int)r[5];
    d.flags = (int)r[6];
    d.sel = (int)r[7];
    return d;
}

template <int CH, bool I16, bool SEP>
__global__ __launch_bounds__(256) void af_features_dft_kernel(const Params p)
{
    using St = Stage<CH, I16>;
    constexpr int MS = St::MS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nw = (int)blockDim.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    float* tile = smem + (size_t)w * p.tile_stride;
    const int tile_len = 15 * p.hop + 16 * p.kg;                       // samples a unit's 16 frames reach
    const float* xrow = tile + n * p.hop + 4 * kq;
    const bool vec_mel = (p.n_mels % 4 == 0) && (reinterpret_cast<uintptr_t>(p.out) % 16 == 0);
    const int64_t n_sets = (p.total_units + nw - 1) / nw;

    for (int64_t set = blockIdx.x; set < n_sets; set += gridDim.x) {
        const int64_t u = set * nw + w;
        const bool valid = u < p.total_units;                          // wave-uniform; idle waves keep the barriers company
        UnitDesc ud{};
        if (valid) ud = load_unit(p.units, u);
        const int sel = SEP ? ud.sel : -1;
        __syncthreads();                                               // (nobody still reads the previous tiles)
        if (valid) {
            // signal tile: to_mono (capture.rs:30-42) / i16 scaling on the way in, zeros behind the end of the clip
            const bool al = ((reinterpret_cast<uintptr_t>(p.pcm) + (uintptr_t)(ud.pcm0 * St::kBytesPerFrame)) % 16) == 0;
            for (int v = lane; v * MS < tile_len; v += 64) {
                float m[MS];
                St::load(p.pcm, ud.pcm0, (int64_t)v * MS, (int64_t)ud.n_avail, al, m, sel);
#pragma unroll
                for (int e = 0; e < MS; ++e) tile[v * MS + e] = m[e];
            }
        }
        wave_sync();
        f32x4 macc[kMaxMelTiles];
#pragma unroll
        for (int mt = 0; mt < kMaxMelTiles; ++mt) macc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int bt = 0; bt < p.nb; ++bt) {
            __syncthreads();                                           // the waves stream the same fragments: keep them together
            if (!valid) continue;
            f32x4 are = {0.f, 0.f, 0.f, 0.f}, aim = {0.f, 0.f, 0.f, 0.f};
            const float4* fr = reinterpret_cast<const float4*>(p.tfrag) + (size_t)bt * p.kg * 2 * 64 + lane;
            float4 tr = fr[0], ti = fr[64];
            for (int g = 0; g < p.kg; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(xrow + 16 * g);
                const float4 cr = tr, ci = ti;
                if (g + 1 < p.kg) {                                    // the next group's fragments while this one multiplies
                    tr = fr[(size_t)(2 * g + 2) * 64];
                    ti = fr[(size_t)(2 * g + 3) * 64];
                }
                are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.x, b.x, are, 0, 0, 0);
                aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.x, b.x, aim, 0, 0, 0);
                are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.y, b.y, are, 0, 0, 0);
                aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.y, b.y, aim, 0, 0, 0);
                are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.z, b.z, are, 0, 0, 0);
                aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.z, b.z, aim, 0, 0, 0);
                are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.w, b.w, are, 0, 0, 0);
                aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.w, b.w, aim, 0, 0, 0);
            }
            float pw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[i] = __builtin_fmaf(are[i], are[i], aim[i] * aim[i]);
            if (p.out_power) {
                if (n < ud.n_rows) {
                    float* o = p.out + (ud.out0 + n) * p.n_bins + 16 * bt + 4 * kq;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (16 * bt + 4 * kq + i < p.n_bins) o[i] = pw[i];
                }
            } else {
                const uint32_t mask = p.wmask[bt];                     // wave-uniform
                const float4* wf = reinterpret_cast<const float4*>(p.wfrag) + (size_t)bt * p.nmt * 64 + lane;
#pragma unroll
                for (int mt = 0; mt < kMaxMelTiles; ++mt) {
                    if (mt < p.nmt && ((mask >> mt) & 1u)) {
                        const float4 a = wf[mt * 64];
                        macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, pw[0], macc[mt], 0, 0, 0);
                        macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, pw[1], macc[mt], 0, 0, 0);
                        macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, pw[2], macc[mt], 0, 0, 0);
                        macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, pw[3], macc[mt], 0, 0, 0);
                    }
                }
            }
        }
        if (valid && !p.out_power && n < ud.n_rows) {
            // mel^T tiles: rows = filters 16 mt + 4 kq + i, column = frame n: 16 bytes of one output row per lane
            float* orow = p.out + (ud.out0 + n) * p.n_mels + 4 * kq;
#pragma unroll
            for (int mt = 0; mt < kMaxMelTiles; ++mt) {
                if (mt < p.nmt) {
                    f32x4 r = macc[mt];
                    if (p.log_mel) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) r[i] = __log10f(fmaxf(r[i], 1e-10f));
                    }
                    const int m0 = 16 * mt + 4 * kq;
                    if (vec_mel) {
                        if (m0 < p.n_mels) *reinterpret_cast<float4*>(orow + 16 * mt) = float4{r[0], r[1], r[2], r[3]};
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (m0 + i < p.n_mels) orow[16 * mt + i] = r[i];
                    }
                }
            }
        }
    }
}

template <int CH, bool I16, bool SEP>
static af_status launch_k(const Params& p, int waves, int grid, hipStream_t stream)
{
    auto kern = af_features_dft_kernel<CH, I16, SEP>;
    const size_t smem = sizeof(float) * (size_t)p.tile_stride * (size_t)waves;
    if (smem > 160 * 1024) return fail(AF_ERR_UNSUPPORTED, "features (DFT kernel): %zu bytes of LDS for one workgroup", smem);
    static LdsAttrFlags attr_set;
    AF_HIP(allow_big_lds(kern, 160 * 1024, attr_set));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), smem, stream, p);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status launch(const Params& p, int channels, bool i16, bool separate, int waves, int grid, hipStream_t stream)
{
    if (grid <= 0) return AF_OK;
    if (!(waves == 1 || waves == 2 || waves == 4)) return fail(AF_ERR_INVALID_ARG, "features (DFT kernel): %d waves per workgroup", waves);
    if (p.nmt > kMaxMelTiles) return fail(AF_ERR_UNSUPPORTED, "features (DFT kernel): %d mel tiles (limit %d)", p.nmt, kMaxMelTiles);
    if (channels == 2 && !i16) return separate ? launch_k<2, false, true>(p, waves, grid, stream) : launch_k<2, false, false>(p, waves, grid, stream);
    if (channels == 1 && !i16) return launch_k<1, false, false>(p, waves, grid, stream);
    if (channels == 2 && i16) return separate ? launch_k<2, true, true>(p, waves, grid, stream) : launch_k<2, true, false>(p, waves, grid, stream);
    if (channels == 1 && i16) return launch_k<1, true, false>(p, waves, grid, stream);
    return fail(AF_ERR_UNSUPPORTED, "features: channels must be 1 or 2 (got %d)", channels);
}

}  // namespace dft
}  // namespace feat
}  // namespace af
