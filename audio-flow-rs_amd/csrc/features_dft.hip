// features_dft.hip -- STFT power + mel for frame lengths the register FFT does not factor (n_fft outside
// {1024, 512, 256}; features.hpp, namespace dft): the DFT as a matrix product on v_mfma_f32_16x16x4_f32.
//
// The periodic Hann window is symmetric (w[n] = w[N - n], w[0] = 0), so with e[n] = x[n] + x[N - n] and
// o[n] = x[n] - x[N - n] (n = 1 .. N / 2; the middle sample of an even N alone in e)
//     Re X[k] =  sum_n w[n] cos(2 pi k n / N) e[n]          Im X[k] = -sum_n w[n] sin(2 pi k n / N) o[n]
// -- two products of HALF the length each.  Per wave and 16-frame unit: the folded rows E, O [16 frames][N / 2] are built
// straight from the PCM (to_mono on the way) into wave-private LDS, then for every tile of 16 bins
//     D_re [16 bins x 16 frames] = Te^T (A, fragments from L2) x E^T (B, LDS),   D_im likewise from To, O
//     P = D_re^2 + D_im^2                       (lane (frame n, kq) holds bins 16 bt + 4 kq + 0..3 of frame n)
//     power output: P goes to HBM;  mel output: mel^T[16 mt .. + 16][frame] += W fragment (A) x P (B, as it sits)
// The K index of MFMA step j of a group is 16 g + 4 kq + j: a lane's four B operands are one 16-byte LDS read, its
// four A operands one 16-byte fragment load (the host lays the fragments out in that order).
// Bounded by the f32 matrix pipe: (n_fft / 2 + 1) x n_fft multiply-adds per frame (docs/OPS.md "STFT").
#include <type_traits>
#include "af_common.hpp"
#include "features_dev.hpp"

// (ablation builds, scripts/ablate: 1 no fragment loads inside the product loop, 2 no LDS reads of the folded rows there)
#ifndef AF_DFT_ABL
#define AF_DFT_ABL 0
#endif
// (experiment) where the next group's LDS reads and the bookkeeping go: 0 a full scheduling barrier behind the reads,
// 1 the compiler's choice, 2 a sched_group_barrier pattern
// (experiment) 0: a tile's first matrix instructions take a zero C operand (a branch per group); 1: the accumulators are zeroed
// where a tile ends
#ifndef AF_DFT_ZERO
#define AF_DFT_ZERO 1
#endif
#ifndef AF_DFT_SCHED
#define AF_DFT_SCHED 1
#endif

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

template <int CH, bool I16, bool SEP, int NMT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(NMT <= 5 ? 3 : 2))) void af_features_dft_kernel(const Params p)
{
    using St = Stage<CH, I16>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nw = (int)blockDim.x >> 6;                                // 2: the waves of a workgroup share one unit (1: diagnostics)
    const int n = lane & 15, kq = lane >> 4;
    // folded rows, one per frame: E [16 kg] | O [16 kg] | padding.  At a row stride of 0 (mod 16) dwords every ds_read_b128
    // is four-way conflicted; 8 (mod 16) is conflict-free (its 16-lane groups, {n 0-3, 12-15 of kq} + {n 4-11 of kq + 1},
    // features.hpp, then cover all 16 sixteen-byte slots); 4 (mod 16), the host's choice, leaves one two-way slot per read
    // and fits one more workgroup on the CU (features_api.hip, AF_DFT_PAD)
    float* rows = smem;
    const int rs = p.row_stride, nh = p.nh, N = p.n_fft;
    const int kpad = 16 * p.kg;                                        // entries of E (and of O) the products read (zero behind nh)
    const float* erow = rows + n * rs + 4 * kq;
    const float* orow_l = erow + kpad;
    const bool vec_mel = (p.n_mels % 4 == 0) && (reinterpret_cast<uintptr_t>(p.out) % 16 == 0);
    const int64_t n_sets = p.total_units;

    unsigned long long c_stage = 0, c_prod = 0, c_units = 0;
    const unsigned long long k_t0 = p.stamps ? __builtin_amdgcn_s_memtime() : 0, k_r0 = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0;

    // Units: the first one by workgroup index, the rest from a device-wide counter (p.queue; null: a static stride) -- the
    // workgroups of a launch do not all run at the same pace (three waves share a SIMD with whoever is there), and with 31.2
    // units per workgroup a static deal leaves 3 % of the launch to its stragglers.  Thread 0 takes the next ticket while the
    // current unit's rows are requested and leaves it in a padding word of the last row (two words, alternating: nothing
    // else reads or writes them) once those loads are back.  It must be consumed there: an atomic whose result stays pending
    // into the product loops makes the compiler wait with vmcnt(0) at every loop head (a first version: 6.8 ms instead of 1.45).
    volatile unsigned* mailbox = reinterpret_cast<volatile unsigned*>(smem + 16 * rs - 2);
    int64_t set = blockIdx.x;
    unsigned ticket = 0;
    for (unsigned iter = 0;; ++iter) {
        __syncthreads();                                               // the partner has left the previous unit's rows (and its ticket is visible)
        if (iter) set = p.queue ? (int64_t)gridDim.x + (int64_t)__builtin_amdgcn_readfirstlane(mailbox[(iter - 1) & 1]) : set + gridDim.x;
        if (set >= n_sets) break;
        if (p.queue && threadIdx.x == 0) ticket = atomicAdd(p.queue, 1u);
        const UnitDesc ud = load_unit(p.units, set);
        const int sel = SEP ? ud.sel : -1;
        const unsigned long long s_t0 = p.stamps ? __builtin_amdgcn_s_memtime() : 0;
        {
            // folded rows: entry k of frame f pairs sample n = k + 1 with sample N - n of that frame; to_mono (capture.rs:30-42)
            // / i16 scaling on the way in, zeros behind the end of the clip and behind the last pair
            // (a valid unit holds at least one whole frame: avail >= n_fft; indices are clamped and the value selected, so
            // the 32 loads of one entry of all 16 frames are in flight together)
            const int64_t avail = ud.n_avail;
            const auto mono = [&](int64_t i) -> float {
                const int64_t ic = i < avail ? i : avail - 1;
                float v;
                if constexpr (CH == 1) {
                    v = St::sample(p.pcm, ud.pcm0 + ic);
                } else {
                    const float l = St::sample(p.pcm, 2 * (ud.pcm0 + ic)), r = St::sample(p.pcm, 2 * (ud.pcm0 + ic) + 1);
                    v = sel < 0 ? (l + r) * 0.5f : sel ? r : l;
                }
                return i < avail ? v : 0.0f;
            };
            // (each wave builds the rows of half the frames: 16 loads in flight per entry with two waves)
            const int f_lo = nw == 2 ? 8 * w : 0;
            bool done_wide = false;
            if constexpr (!(AF_DFT_ABL & 4)) {
                // A lane takes four consecutive entries of a frame -- samples n .. n + 3 and their partners N - n - 3 .. N - n as
                // vectors (f32: 16-byte loads, two each for stereo, global loads only need dword alignment; i16: four or eight
                // shorts declared 2-byte aligned, whatever the compiler makes of them) -- and every load of the wave's half of
                // the rows is in flight at once: the rows take one trip to HBM instead of four.  Frames that reach behind
                // the end of the clip (wave-uniform test) take the element-wise path below.
                const int J = kpad >> 2;                                // 16-byte pieces per row half
                const int n_f = nw == 2 ? 8 : 16;
                const int64_t last = (int64_t)(f_lo + n_f - 1) * p.hop + N;      // one behind the last sample these frames touch
                if (last <= avail) {
                    done_wide = true;
                    typedef typename std::conditional<I16, int16_t, float>::type Elem;
                    const Elem* x0 = static_cast<const Elem*>(p.pcm) + ud.pcm0 * CH;
                    const int j_valid = (nh + 3) >> 2;                  // pieces that hold a pair; the rest of a row half is zero
                    for (int j = lane; j < J; j += 64) {               // (52 pieces per row half at 400: one pass, 16 loads per lane)
                        // (a piece behind the last pair loads nothing: with the row padded to four K groups its samples would lie
                        // behind the frame -- and its partners in front of it: n_fft 36 faulted here in the first version)
                        const int jl = j < j_valid ? j : 0;
                        for (int f0 = 0; f0 < n_f; f0 += 8) {
                            float4 a4[8], b4[8];
#pragma unroll
                            for (int f = 0; f < 8; ++f) {
                                const Elem* xf = x0 + (int64_t)(f_lo + f0 + f) * p.hop * CH;
                                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                                typedef short s4u __attribute__((ext_vector_type(4), aligned(2)));
                                if constexpr (I16) {
                                    constexpr float sc = 1.0f / 32768.0f;      // docs/OPS.md "PCM formats"
                                    const short* xs = reinterpret_cast<const short*>(xf);
                                    if constexpr (CH == 1) {
                                        const s4u av = *reinterpret_cast<const s4u*>(xs + 4 * jl + 1);
                                        const s4u bv = *reinterpret_cast<const s4u*>(xs + N - 4 * jl - 4);
                                        a4[f] = float4{(float)av[0] * sc, (float)av[1] * sc, (float)av[2] * sc, (float)av[3] * sc};
                                        b4[f] = float4{(float)bv[3] * sc, (float)bv[2] * sc, (float)bv[1] * sc, (float)bv[0] * sc};
                                    } else {
                                        const s4u a0 = *reinterpret_cast<const s4u*>(xs + 2 * (4 * jl + 1)), a1 = *reinterpret_cast<const s4u*>(xs + 2 * (4 * jl + 3));
                                        const s4u b0 = *reinterpret_cast<const s4u*>(xs + 2 * (N - 4 * jl - 4)), b1 = *reinterpret_cast<const s4u*>(xs + 2 * (N - 4 * jl - 2));
                                        const auto mix = [&](short l, short r) { const float fl = (float)l * sc, fr = (float)r * sc; return sel < 0 ? (fl + fr) * 0.5f : sel ? fr : fl; };
                                        a4[f] = float4{mix(a0[0], a0[1]), mix(a0[2], a0[3]), mix(a1[0], a1[1]), mix(a1[2], a1[3])};
                                        b4[f] = float4{mix(b1[2], b1[3]), mix(b1[0], b1[1]), mix(b0[2], b0[3]), mix(b0[0], b0[1])};
                                    }
                                } else if constexpr (CH == 1) {
                                    const f4u av = *reinterpret_cast<const f4u*>(xf + 4 * jl + 1);
                                    const f4u bv = *reinterpret_cast<const f4u*>(xf + N - 4 * jl - 4);
                                    a4[f] = float4{av[0], av[1], av[2], av[3]};
                                    b4[f] = float4{bv[3], bv[2], bv[1], bv[0]};   // partner of sample n + i is N - n - i
                                } else {
                                    const f4u a0 = *reinterpret_cast<const f4u*>(xf + 2 * (4 * jl + 1)), a1 = *reinterpret_cast<const f4u*>(xf + 2 * (4 * jl + 3));
                                    const f4u b0 = *reinterpret_cast<const f4u*>(xf + 2 * (N - 4 * jl - 4)), b1 = *reinterpret_cast<const f4u*>(xf + 2 * (N - 4 * jl - 2));
                                    const auto mix = [&](float l, float r) { return sel < 0 ? (l + r) * 0.5f : sel ? r : l; };   // to_mono (capture.rs:30-42) / one channel
                                    a4[f] = float4{mix(a0[0], a0[1]), mix(a0[2], a0[3]), mix(a1[0], a1[1]), mix(a1[2], a1[3])};
                                    b4[f] = float4{mix(b1[2], b1[3]), mix(b1[0], b1[1]), mix(b0[2], b0[3]), mix(b0[0], b0[1])};
                                }
                            }
#pragma unroll
                            for (int f = 0; f < 8; ++f) {
                                const float av[4] = {a4[f].x, a4[f].y, a4[f].z, a4[f].w}, bv[4] = {b4[f].x, b4[f].y, b4[f].z, b4[f].w};
                                float e[4], o[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int k = 4 * j + i, nn = k + 1;
                                    const bool pair = k < nh, self = (N - nn == nn);
                                    e[i] = !pair ? 0.0f : self ? av[i] : av[i] + bv[i];
                                    o[i] = (!pair || self) ? 0.0f : av[i] - bv[i];
                                }
                                float* er = rows + (f_lo + f0 + f) * rs + 4 * j;
                                *reinterpret_cast<float4*>(er) = float4{e[0], e[1], e[2], e[3]};
                                *reinterpret_cast<float4*>(er + kpad) = float4{o[0], o[1], o[2], o[3]};
                            }
                        }
                    }
                }
            }
            // (two entries per lane and pass = 32 loads in flight: the rows wait for HBM, not for arithmetic)
            for (int k0 = lane; k0 < kpad && !done_wide; k0 += 128) {
                for (int fh = 0; fh < (nw == 2 ? 1 : 2); ++fh) {
                    float a[2][8], b[2][8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = k0 + 64 * h;
                        const int nn = k + 1;
                        const bool pair = k < nh;
                        const int na = pair ? nn : 1, nb2 = pair ? N - nn : 1;   // (entries behind the last pair: any valid sample, zeroed below)
#pragma unroll
                        for (int f = 0; f < 8; ++f) {
                            a[h][f] = mono((int64_t)(f_lo + 8 * fh + f) * p.hop + na);
                            b[h][f] = mono((int64_t)(f_lo + 8 * fh + f) * p.hop + nb2);
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = k0 + 64 * h;
                        const int nn = k + 1;
                        const bool pair = k < nh, self = (N - nn == nn);
                        if (k < kpad) {
#pragma unroll
                            for (int f = 0; f < 8; ++f) {
                                const float e = !pair ? 0.0f : self ? a[h][f] : a[h][f] + b[h][f];
                                const float o = (!pair || self) ? 0.0f : a[h][f] - b[h][f];
                                rows[(f_lo + 8 * fh + f) * rs + k] = e;
                                rows[(f_lo + 8 * fh + f) * rs + kpad + k] = o;
                            }
                        }
                    }
                }
            }
        }
        if (p.queue && threadIdx.x == 0) mailbox[iter & 1] = ticket;
        __syncthreads();                                               // all 16 rows are there
        unsigned long long s_t1 = 0;
        if (p.stamps) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            s_t1 = __builtin_amdgcn_s_memtime();
        }
        f32x4 macc[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) macc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

        // One flat walk over (bin tile, K group) in chunks of kChunk groups; the fragment stream is contiguous in that order.
        // Fragments are double-buffered by CHUNK: all loads of chunk c + 1 go out, unconditionally (clamped index), at the
        // top of chunk c.  The compiler's wait at the top of the next chunk is then vmcnt(0) on loads that are a whole chunk
        // (32 matrix instructions) old.  Finer rings lose to its bookkeeping: behind a conditional load, or across the back
        // edge of the loop, it cannot count how many loads follow the one it needs and waits for all of them -- a ring of
        // depth one, 590 cycles per group instead of 256 with one wave per SIMD.  The mel fragments of a bin tile that ends in
        // a chunk are requested at that chunk's top BEFORE the fragment loads (so the wait for them counts those 2 kChunk
        // behind them and nothing else); what ends a tile inside the chunk only saves the power values, the mel products
        // follow once, behind the chunk (kg >= kChunk: at most one tile ends per chunk).
        constexpr int kChunk = 4;
        // this wave's bin tiles: first, first + nw, ... (the odd tile of 13 alternates between the two waves from unit to unit)
        const int first = nw == 2 ? (int)((w + set) & 1) : 0;
        const int nt_w = (p.nb - first + nw - 1) / nw;
        const int total = (nt_w * p.kg + kChunk - 1) / kChunk * kChunk;         // (whole chunks: the groups behind the last tile multiply into accumulators nobody reads)
        const float4* fr = reinterpret_cast<const float4*>(p.tfrag) + lane;
        const float4* wfb = reinterpret_cast<const float4*>(p.wfrag) + lane;
        float4 bufA[2 * kChunk], bufB[2 * kChunk];
        // request cursor: a running group index into the table (no multiplications per load), stepping over the partner's
        // tiles at a tile end and standing still on this wave's last group (the loads stay unconditional)
        int rq = first * p.kg < p.nb * p.kg ? first * p.kg : p.nb * p.kg - 1, rg = 0;
        int r_left = nt_w * p.kg;                                      // groups of this wave not yet requested
        const int r_skip = (nw - 1) * p.kg;
        const auto request = [&](float4 (&buf)[2 * kChunk]) {
#pragma unroll
            for (int d = 0; d < kChunk; ++d) {
                const float4* g0 = fr + (size_t)rq * 128;              // [group][cos, -sin][64 lanes]
                buf[2 * d] = g0[0];
                buf[2 * d + 1] = g0[64];
                if (r_left > 1) {
                    --r_left;
                    ++rq;
                    if (++rg == p.kg) { rg = 0; rq += r_skip; }
                }
            }
        };
        request(bufA);
        float4 wfm[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) wfm[mt] = float4{0.f, 0.f, 0.f, 0.f};
        uint32_t mask = (p.out_power || nt_w == 0) ? 0u : p.wmask[first];   // of the bin tile in progress (scalar, a tile ahead)
        f32x4 are = {0.f, 0.f, 0.f, 0.f}, aim = {0.f, 0.f, 0.f, 0.f};
        int g = 0, bt = first;
        // B operands one group ahead: a wave alone on its SIMD otherwise waits for its two LDS reads in every group
        float4 be_n = *reinterpret_cast<const float4*>(erow);
        float4 bo_n = *reinterpret_cast<const float4*>(orow_l);
        float pw[4] = {0.f, 0.f, 0.f, 0.f};
        const auto chunk = [&](const float4 (&cur)[2 * kChunk], float4 (&nxt)[2 * kChunk], int q0) {
            const bool ends = g + kChunk >= p.kg;                      // a bin tile ends inside this chunk
            if (ends && !p.out_power) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
                    if (mt < p.nmt && ((mask >> mt) & 1u)) wfm[mt] = wfb[((size_t)bt * p.nmt + mt) * 64];
            }
            if constexpr (!(AF_DFT_ABL & 1)) request(nxt);
            bool done = false;
#pragma unroll
            for (int d = 0; d < kChunk; ++d) {
                const float4 cr = (AF_DFT_ABL & 1) ? bufA[2 * d] : cur[2 * d], ci = (AF_DFT_ABL & 1) ? bufA[2 * d + 1] : cur[2 * d + 1];
                const float4 be = be_n, bo = bo_n;
                if constexpr (!(AF_DFT_ABL & 2)) {
                    const int gn = (g + 1 == p.kg) ? 0 : g + 1;        // (behind the last group of a bin tile: group 0 of the next)
                    be_n = *reinterpret_cast<const float4*>(erow + 16 * gn);
                    bo_n = *reinterpret_cast<const float4*>(orow_l + 16 * gn);
#if AF_DFT_SCHED == 0
                    __builtin_amdgcn_sched_barrier(0);                 // (the reads stay here, a whole group ahead of their use)
#endif
                }
                // a tile's first products start from zero (no write to the accumulators outside the matrix pipe: a
                // v_mov between two of its instructions waits out the first one's latency)
#if AF_DFT_ZERO == 0
                if (g == 0) {                                          // wave-uniform
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.x, be.x, z, 0, 0, 0);
                    aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.x, bo.x, z, 0, 0, 0);
                } else
#endif
                {
                    are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.x, be.x, are, 0, 0, 0);
                    aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.x, bo.x, aim, 0, 0, 0);
                }
                are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.y, be.y, are, 0, 0, 0);
                aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.y, bo.y, aim, 0, 0, 0);
                are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.z, be.z, are, 0, 0, 0);
                aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.z, bo.z, aim, 0, 0, 0);
                are = __builtin_amdgcn_mfma_f32_16x16x4f32(cr.w, be.w, are, 0, 0, 0);
                aim = __builtin_amdgcn_mfma_f32_16x16x4f32(ci.w, bo.w, aim, 0, 0, 0);
#if AF_DFT_SCHED == 2
                // one matrix instruction, the two LDS reads, one more, then the scalar / vector bookkeeping in the shadow of the rest
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
                if (++g == p.kg) {                                     // the bin tile is complete (the padding groups never get here)
#pragma unroll
                    for (int i = 0; i < 4; ++i) pw[i] = __builtin_fmaf(are[i], are[i], aim[i] * aim[i]);
#if AF_DFT_ZERO == 1
                    are = f32x4{0.f, 0.f, 0.f, 0.f};
                    aim = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
                    g = 0;
                    done = true;
                }
            }
            if (done) {
                if (p.out_power) {
                    if (n < ud.n_rows) {
                        float* o = p.out + (ud.out0 + n) * p.n_bins + 16 * bt + 4 * kq;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (16 * bt + 4 * kq + i < p.n_bins) o[i] = pw[i];
                    }
                } else {
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) {
                        if (mt < p.nmt && ((mask >> mt) & 1u)) {
                            const float4 a = wfm[mt];
                            macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, pw[0], macc[mt], 0, 0, 0);
                            macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, pw[1], macc[mt], 0, 0, 0);
                            macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, pw[2], macc[mt], 0, 0, 0);
                            macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, pw[3], macc[mt], 0, 0, 0);
                        }
                    }
                    mask = bt + nw < p.nb ? p.wmask[bt + nw] : 0u;
                }
                bt += nw;
            }
        };
        for (int q0 = 0; q0 < total; q0 += 2 * kChunk) {
            chunk(bufA, bufB, q0);
            if (q0 + kChunk < total) chunk(bufB, bufA, q0 + kChunk);
        }
        if (!p.out_power && nw == 2) {
            // the two waves' partial mel sums: wave 1 hands its tiles over through the rows' LDS, wave 0 adds and stores
            __syncthreads();                                           // both are done with the rows
            float* red = smem + lane;
            if (w == 1) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
                    if (mt < p.nmt) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) red[(mt * 4 + i) * 64] = macc[mt][i];
                    }
            }
            __syncthreads();
            if (w == 0) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
                    if (mt < p.nmt) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) macc[mt][i] += red[(mt * 4 + i) * 64];
                    }
            }
        }
        if (!p.out_power && n < ud.n_rows && w == 0) {
            // mel^T tiles: rows = filters 16 mt + 4 kq + i, column = frame n: 16 bytes of one output row per lane
            float* orow = p.out + (ud.out0 + n) * p.n_mels + 4 * kq;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
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
        if (p.stamps) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long s_t2 = __builtin_amdgcn_s_memtime();
            c_stage += s_t1 - s_t0;
            c_prod += s_t2 - s_t1;
            c_units += 1;
        }
    }
    if (p.stamps && lane == 0) {
        atomicAdd(p.stamps, c_stage);
        atomicAdd(p.stamps + 1, c_prod);
        atomicAdd(p.stamps + 2, c_units);
        atomicAdd(p.stamps + 3, __builtin_amdgcn_s_memtime() - k_t0);
        atomicAdd(p.stamps + 4, __builtin_amdgcn_s_memrealtime() - k_r0);
        atomicAdd(p.stamps + 5, 1ull);
    }
}

template <int CH, bool I16, bool SEP, int NMT>
static af_status launch_kn(const Params& p, int waves, int grid, hipStream_t stream)
{
    auto kern = af_features_dft_kernel<CH, I16, SEP, NMT>;
    const size_t smem = sizeof(float) * (size_t)p.tile_stride;          // one unit's rows per workgroup, whatever its waves
    if (smem > 160 * 1024) return fail(AF_ERR_UNSUPPORTED, "features (DFT kernel): %zu bytes of LDS for one workgroup", smem);
    static LdsAttrFlags attr_set;
    AF_HIP(allow_big_lds(kern, 160 * 1024, attr_set));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), smem, stream, p);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

// (two instances by mel tiles: 80 mels = 5 tiles of 16 leave 24 registers to the rest of the kernel)
template <int CH, bool I16, bool SEP>
static af_status launch_k(const Params& p, int waves, int grid, hipStream_t stream)
{
    if (p.nmt <= 5) return launch_kn<CH, I16, SEP, 5>(p, waves, grid, stream);
    return launch_kn<CH, I16, SEP, kMaxMelTiles>(p, waves, grid, stream);
}

af_status launch(const Params& p, int channels, bool i16, bool separate, int waves, int grid, hipStream_t stream)
{
    if (grid <= 0) return AF_OK;
    if (!(waves == 1 || waves == 2)) return fail(AF_ERR_INVALID_ARG, "features (DFT kernel): %d waves per workgroup", waves);
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
