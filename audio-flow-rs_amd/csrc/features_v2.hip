// features_v2.hip -- the fused  interleaved PCM -> to_mono -> Hann STFT(1024) -> |X|^2 -> mel
// kernel for hop 256 (the BASELINE configuration), built for four waves per SIMD.
//
// Same contract as features_kernel.hip (to_mono: reference capture.rs:30-42; STFT / mel:
// docs/OPS.md, build-defined); what differs is how a workgroup is cut:
//   * 512 threads = 8 wave64, one 16-frame unit per iteration; wave w owns the frame pair
//     (2w, 2w+1), packed as the real / imaginary part of ONE 1024-point complex FFT.
//   * the FFT is 16 x 16 x 4 with 16 complex points per lane (32 data VGPRs instead of the 64 of
//     the 32 x 32 scheme), so the kernel fits 128 VGPRs and two workgroups (16 waves) share a CU:
//       stage 1   lane 16 c + m takes n2 = c + 4 m:  radix-16 over n1  (n = 64 n1 + n2), twiddle W1024^(n2 k1)
//       T1        through LDS: row k1, position = writing lane;  lane (k1, c) reads its columns c + 4m as
//                 positions 16 c .. 16 c + 15 (four ds_read_b128)
//       stage 2a  radix-16 over m -> b
//       T2        in registers (v_permlane32_swap / v_permlane16_swap): lane (k1, j) gets b = j + 4b' of its four c lanes
//       stage 2b  twiddle W64^(c b) from four per-lane register constants, folded into four radix-4 over c -> a;
//                 lane L = k1 + 16 j then holds Z[L + 64 t], t = b' + 4a
//     All butterflies are FMA-shaped (fft_regs.hpp).  The transposition stores are
//     ds_write_addtid_b32 (row / plane offset in the instruction, no address register, half the
//     LDS cycles of ds_write_b32).
//   * split of the packed spectrum: Z[1024 - k] sits in lane (64 - L) % 64, register 15 - t
//     (lane 0: 16 - t), exchanged through the same buffer; power rows [frame][kProw2] in LDS.
//   * mel: the banded filterbank groups are dealt to the 8 waves (features_api.hip build_plan),
//     v_mfma_f32_16x16x4_f32; a wave's A fragments (L2-resident) are fetched once per unit, after
//     stage 1, where the register pressure has dropped.
//   * every unit is described by one 32-byte record (UnitDesc, built on the host): no clip search,
//     no 64-bit index arithmetic in the loop.
#include "features_dev.hpp"
#include "fft_regs.hpp"

// AF_ABL (diagnostic builds only, scripts/ablate/): bit mask of LDS access groups compiled OUT of the kernel, to
// attribute SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT to them.  Every guard that bounds a global access stays in.
//   1 tile reads (stage 1)   2 window rows   4 first twiddle rows   8 T1 stores   16 T1 reads   32 second twiddle rows
//   64 exchange stores   128 exchange reads   256 power-row stores   512 mel operand reads   1024 tile update
//   2048 (not LDS) mel stores go to the wrong place inside the unit's 5 KB, 1 KB contiguous per instruction
#ifndef AF_ABL
#define AF_ABL 0
#endif
#ifndef AF_V2_ALL_SLOTS
#define AF_V2_ALL_SLOTS 0
#endif
// AF_V2_PH (diagnostic builds only, scripts/ablate/v2_phases.sh): bit mask of PHASES compiled out, for the steady-clock
// phase table of profiles/r05_v2_ablations.txt.  Results are garbage; every guard that bounds a global access stays in.
//   1 arithmetic of FFT part 1 (window, radix-16, first twiddle)   2 arithmetic of FFT part 2 (radix-16, T2, radix-4, split)
//   4 mel MFMAs   8 workgroup barriers   16 global stores   32 PCM prefetch loads   64 filterbank fragment loads
//   128 the whole mel block (operand reads, MFMAs, stores)   256 every mel MFMA replaced by a v_mfma_f32_4x4x1_16b_f32 (a quarter of the MACs)
#ifndef AF_V2_PH
#define AF_V2_PH 0
#endif
// AF_V2_CLK (diagnostic builds only): with stamps enabled the PRODUCT instance runs (not the stamping one) and every
// workgroup adds its lifetime in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime) to stamps[0], [1] and
// 1 to stamps[2]: the clock the chip holds inside this kernel = stamps[0] / stamps[1] x 100 MHz.
#ifndef AF_V2_CLK
#define AF_V2_CLK 0
#endif
// (experiment) cache policy nt on: 1 the PCM prefetch loads, 2 the mel row stores
#ifndef AF_V2_NT
#define AF_V2_NT 0
#endif
// staging loads of the next unit requested at the top of a unit; the others go out behind FFT part 1 (99: all at the top)
#ifndef AF_V2_PF_EARLY
#define AF_V2_PF_EARLY 99
#endif
#ifndef AF_V2_TW_EARLY
#define AF_V2_TW_EARLY 0
#endif
// 1: the 16 new hops of unit u + 2 are requested behind the tile update of unit u (they fly through that unit's mel
// phase and the whole of unit u + 1) instead of at the top of unit u + 1; 0: round 2-4's placement
#ifndef AF_V2_PF_AHEAD
#define AF_V2_PF_AHEAD 0
#endif
// 1: the split's partner-lane addresses start from lane_now() (0: from the hoisted lane number; experiment)
#ifndef AF_V2_LANE_NOW
#define AF_V2_LANE_NOW 1
#endif
// (ablation, MEL4 instances) 1: no mel MFMAs, 2: no mel stores, 4: no piece combine
#ifndef AF_V2_M4
#define AF_V2_M4 0
#endif
// (experiment, 4 x 4 x 1 mel instances) 1: the mel rows of a unit go through an LDS image [16][n_mels] and leave as one contiguous
// run of 16-byte stores per thread (one more barrier per unit) instead of 320 separate 16-byte pieces
#ifndef AF_V2_M4_STAGE
#define AF_V2_M4_STAGE 0
#endif
// work distribution with p.queue: 1 = the pool's chunks dealt by one counter (every chunk starts with a whole-tile load);
// 2 = every range's tail is a deque of chunks: its owner takes them from the front WITHOUT a break (claimed two units ahead, the
// tile keeps shifting), the workgroups that have run out steal from the back of the others' (one packed word per range: low half
// front claims, high half back claims; a claim is valid while front + back < chunks)
#ifndef AF_V2_POOL_MODE
#define AF_V2_POOL_MODE 1
#endif
#ifndef AF_V2_POOL_GUIDED
#define AF_V2_POOL_GUIDED 0
#endif
// (experiment) start the workgroups staggered (s_sleep units of 64 cycles per step of blockIdx % 16); 0: together
#ifndef AF_V2_STAGGER
#define AF_V2_STAGGER 0
#endif
// issue priority of the kernel's waves outside their raised sections (experiment: 1 when it shares CUs with another kernel)
#ifndef AF_V2_BASE_PRIO
#define AF_V2_BASE_PRIO 0
#endif

namespace af {
namespace feat {
namespace v2 {

// (a stand-in for a value an ablated access would have delivered: lane-dependent, and opaque per use so that the
// arithmetic behind it is neither folded nor hoisted out of the unit loop)
__device__ __forceinline__ float abl_val(int lane, int i)
{
    float z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return (float)(lane + i) * 0.001f + z;
}
// (phase ablations) a value the compiler must have in a register here / a register whose content it cannot know
__device__ __forceinline__ void abl_use(float v) { asm volatile("" : : "v"(v)); }
__device__ __forceinline__ float abl_any() { float v; asm volatile("" : "=v"(v)); return v; }

__device__ __forceinline__ constexpr int brev4(int x)
{
    return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3);
}

// v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b;  v_permlane16_swap: rows 1, 3 of a <->
// rows 0, 2 of b (rows of 16 lanes)
__device__ __forceinline__ void swap_rows32(float& a, float& b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_rows16(float& a, float& b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// compiler-only ordering point between this wave's LDS accesses (a wave's DS operations execute
// in order, nothing has to be waited for)
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// Lane-consecutive LDS stores without an address register: ds_write_addtid_b32 writes lane l's
// value to  M0 + offset + 4 l  (half the LDS cycles of ds_write_b32: no address VGPR to move).
// m0 = LDS byte address of the wave's buffer (wave-uniform); value i goes to row i, rows STRIDE
// bytes apart from BASE.  M0 is set inside every statement and listed as clobbered (the compiler may keep
// a value of its own there), followed by the wait state the ISA asks for between an M0 write and an add-TID access.
// AF_V2_NO_ADDTID (diagnostics): the same stores as plain ds_write_b32.
struct WaveBuf {
    float* wb;
    unsigned m0;
    int lane;
};
template <int STRIDE, int BASE>
__device__ __forceinline__ void addtid_store8(const WaveBuf& b, float v0, float v1, float v2, float v3,
                                              float v4, float v5, float v6, float v7)
{
    static_assert(BASE >= 0 && BASE + 7 * STRIDE < 65536 && STRIDE % 4 == 0 && BASE % 4 == 0, "16-bit offset field");
#ifdef AF_V2_NO_ADDTID
    float* d = b.wb + BASE / 4 + b.lane;
    d[0] = v0; d[STRIDE / 4] = v1; d[2 * (STRIDE / 4)] = v2; d[3 * (STRIDE / 4)] = v3;
    d[4 * (STRIDE / 4)] = v4; d[5 * (STRIDE / 4)] = v5; d[6 * (STRIDE / 4)] = v6; d[7 * (STRIDE / 4)] = v7;
    asm volatile("" ::: "memory");
#else
    asm volatile("s_mov_b32 m0, %8\n\t"
                 "s_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:%9\n\t"
                 "ds_write_addtid_b32 %1 offset:%10\n\t"
                 "ds_write_addtid_b32 %2 offset:%11\n\t"
                 "ds_write_addtid_b32 %3 offset:%12\n\t"
                 "ds_write_addtid_b32 %4 offset:%13\n\t"
                 "ds_write_addtid_b32 %5 offset:%14\n\t"
                 "ds_write_addtid_b32 %6 offset:%15\n\t"
                 "ds_write_addtid_b32 %7 offset:%16"
                 :
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "s"(b.m0),
                   "n"(BASE), "n"(BASE + STRIDE), "n"(BASE + 2 * STRIDE), "n"(BASE + 3 * STRIDE),
                   "n"(BASE + 4 * STRIDE), "n"(BASE + 5 * STRIDE), "n"(BASE + 6 * STRIDE),
                   "n"(BASE + 7 * STRIDE)
                 : "memory", "m0");
#endif
}
template <int BASE>
__device__ __forceinline__ void addtid_store1(const WaveBuf& b, float v0)
{
#ifdef AF_V2_NO_ADDTID
    b.wb[BASE / 4 + b.lane] = v0;
    asm volatile("" ::: "memory");
#else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2" : : "v"(v0), "s"(b.m0), "n"(BASE)
                 : "memory", "m0");
#endif
}
template <int STRIDE, int BASE>
__device__ __forceinline__ void addtid_store16(const WaveBuf& b, const float (&v)[16])
{
    addtid_store8<STRIDE, BASE>(b, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    addtid_store8<STRIDE, BASE + 8 * STRIDE>(b, v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
}

// The signal tile is stored with the dword pairs of every odd 32-sample block swapped (index ^ 2).  Stage 1
// reads it with lane 16 c + m on sample c + 4 m (+ 64 j): within a 32-lane half of a ds_read_b32 the samples
// n2 and n2 + 32 would share a bank; with the swap the odd block's pairs land on the banks the even block
// leaves free (brute-forced: one lane per bank).
__device__ __forceinline__ int tile_swz(int idx) { return idx ^ (((idx >> 5) & 1) << 1); }

// tid through a register the compiler cannot see through (see the prefetch in the unit loop)
__device__ __forceinline__ int opaque_tid(int tid)
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return tid + z;
}

// The lane number, recomputed where it is asked for (two VALU instructions, no memory): lane-derived addresses built
// from it cannot be hoisted out of the unit loop, where they would sit in registers -- or in scratch, whose reload
// in the middle of the FFT waits for the PCM prefetch requested before it (vector memory returns in order).
__device__ __forceinline__ int lane_now()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l));
    return l;
}

// MS consecutive samples (idx a multiple of MS = 2, 4 or 8: they stay inside one 32-sample block)
template <int MS>
__device__ __forceinline__ void tile_store(float* tile, int idx, const float (&m)[MS])
{
    const bool odd = (idx >> 5) & 1;
    if constexpr (MS == 2) {
        float* d = tile + (idx ^ (odd ? 2 : 0));
        d[0] = m[0];
        d[1] = m[1];
    } else {
        float* d = tile + idx;
#pragma unroll
        for (int g = 0; g < MS; g += 4) {
            d[g] = odd ? m[g + 2] : m[g];
            d[g + 1] = odd ? m[g + 3] : m[g + 1];
            d[g + 2] = odd ? m[g] : m[g + 2];
            d[g + 3] = odd ? m[g + 1] : m[g + 3];
        }
    }
}
// where those MS samples physically start (for raw copies between blocks of equal parity)
template <int MS>
__device__ __forceinline__ int tile_phys(int idx)
{
    return MS == 2 ? tile_swz(idx) : idx;
}

// ---------------------------------------------------------------------------
// One frame pair: signal tile -> two power rows in this wave's buffer wb.
//   tile   mono samples of the unit, frame f starts at tile[f * kHop]
//   wb     this wave's buffer (kWb dwords), m0 its byte offset in LDS
// Part 1a: this lane's 20 tile samples and its 16 window values (LDS reads only; the tile may be rewritten as soon
// as every wave of the workgroup has these in registers).
__device__ __forceinline__ void stage1_load(const float* __restrict__ tile, int w, int lane,
                                            const float* __restrict__ wtab, float (&s)[20], float (&win)[16])
{
    // lane l = 16 c + m runs stage 1 for column n2 = c + 4 m, so that the 16 columns c + 4 m' a lane (k1, c)
    // reads after the transposition sit next to each other (positions 16 c .. 16 c + 15 of a row: four
    // ds_read_b128 instead of sixteen ds_read_b32).  The window / twiddle table rows are stored per lane.
    const int n2 = (lane >> 4) + 4 * (lane & 15);
    const float* src = tile + (2 * w) * kHop + tile_swz(n2);
#pragma unroll
    for (int j = 0; j < 20; ++j) s[j] = (AF_ABL & 1) ? abl_val(lane, j) : src[64 * j];
    const float4* wrow = reinterpret_cast<const float4*>(wtab + lane * kWinRow2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 v = (AF_ABL & 2) ? float4{abl_val(lane, j), 0.5f, 0.25f, 0.125f} : wrow[j];
        win[4 * j] = v.x; win[4 * j + 1] = v.y; win[4 * j + 2] = v.z; win[4 * j + 3] = v.w;
    }
}

// Part 1b: stage 1 and the first transposition; ur / ui hold stage 2a's inputs (element brev4(m)).
__device__ __forceinline__ void fft_pair_1(const float (&s)[20], const float (&win)[16], float* wb, const WaveBuf& wv,
                                           int lane, const float* __restrict__ tw1, float (&ur)[16], float (&ui)[16])
{
    const int k1 = lane & 15, hi = lane >> 4;
    float re[16], im[16];
    // (experiment, AF_V2_TW_EARLY = n > 0) the first n pieces of the twiddle row requested by hand in front of the
    // transform: hipcc sinks plain loads below it
    constexpr int kTwEarly = AF_V2_TW_EARLY;
    f32x4 twr[kTwEarly > 0 ? kTwEarly : 1];
    if constexpr (kTwEarly > 0) {
        const unsigned ta = (unsigned)reinterpret_cast<uintptr_t>(tw1 + lane * kTw1Row);
#pragma unroll
        for (int jj = 0; jj < kTwEarly; ++jj)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(twr[jj]) : "v"(ta), "n"(16 * jj));
    }

    // ---- stage 1: z[64 n1 + n2] = w * (a + i b); frame b = frame a shifted by
    // kHop samples = 4 registers.  The window multiply is fused into the twiddle-free first DIT
    // stage (element e holds input brev4(e): elements (e, e + 1) are inputs (na, na + 8)).
    if constexpr (AF_V2_PH & 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { re[e] = s[e]; im[e] = win[e]; }
#pragma unroll
        for (int e = 16; e < 20; ++e) abl_use(s[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const int na = brev4(e), nb = na + 8;
            const float mr = s[nb] * win[nb], mi = s[nb + 4] * win[nb];
            re[e] = __builtin_fmaf(s[na], win[na], mr);
            re[e + 1] = __builtin_fmaf(s[na], win[na], -mr);
            im[e] = __builtin_fmaf(s[na + 4], win[na], mi);
            im[e + 1] = __builtin_fmaf(s[na + 4], win[na], -mi);
        }
    }
    if constexpr (!(AF_V2_PH & 1)) fftr::dit_fft<16, 2>(re, im);                  // Y[k1] at element k1

    // twiddle W1024^(n2 k1): this lane's row of the table holds (cos, sin) for k1 = 1..15 (k1 = 0 needs none), so
    // that every 16-byte piece is used whole: with k1 = 0 in the row hipcc dropped the unused first pair, re-cut the
    // rest into 8-byte-aligned ds_read2_b64 (8 LDS cycles each, banked modulo 32: lanes l and l + 8 of a row stride
    // of 36 dwords collide) and the eight reads cost 116 LDS cycles per frame pair instead of 32
    // (scripts/ablate/lds_groups.sh: all of SQ_LDS_BANK_CONFLICT came from here).
    {
        const float4* trow = reinterpret_cast<const float4*>(tw1 + lane * kTw1Row);
        float4 late[8];
#pragma unroll
        for (int jj = kTwEarly; jj < 8; ++jj) late[jj] = (AF_ABL & 4) ? float4{abl_val(lane, jj), 0.5f, 0.25f, 0.125f} : trow[jj];
        if constexpr (kTwEarly > 0) {
            __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the hand-issued reads (and the ones above)
#pragma unroll
            for (int jj = 0; jj < kTwEarly; ++jj) asm volatile("" : "+v"(twr[jj]));
        }
        if constexpr (AF_V2_PH & 1) {
#pragma unroll
            for (int jj = kTwEarly; jj < 8; ++jj) { abl_use(late[jj].x); abl_use(late[jj].y); abl_use(late[jj].z); abl_use(late[jj].w); }
        } else
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float4 t = jj < kTwEarly ? float4{twr[jj < kTwEarly ? jj : 0][0], twr[jj < kTwEarly ? jj : 0][1],
                                                    twr[jj < kTwEarly ? jj : 0][2], twr[jj < kTwEarly ? jj : 0][3]}
                                           : late[jj];
            {
                const float xr = re[2 * jj + 1], xi = im[2 * jj + 1];
                re[2 * jj + 1] = __builtin_fmaf(xr, t.x, xi * t.y);
                im[2 * jj + 1] = __builtin_fmaf(xi, t.x, -(xr * t.y));
            }
            if (jj < 7) {
                const float xr = re[2 * jj + 2], xi = im[2 * jj + 2];
                re[2 * jj + 2] = __builtin_fmaf(xr, t.z, xi * t.w);
                im[2 * jj + 2] = __builtin_fmaf(xi, t.z, -(xr * t.w));
            }
        }
    }

    // ---- T1: stage-1 element k1 goes to row rho(k1), position = writing lane; lane (k1, c = hi) reads positions
    // 16 c + m (= columns c + 4m) of row rho(k1) into element brev4(m).  rho makes the ds_read_b128 conflict-free:
    // its 16-lane groups are {0-3, 12-15 of c} + {4-11 of c + 1} (and the complement), the row stride is 68
    // dwords = 17 slots of 16 bytes, so slot = rho(k1) + 4 c (mod 16); rho maps k1 = 4..11 onto
    // {0, 4, 8, 12, 1, 5, 9, 13}, a set invariant under + 4, and the other k1 onto its complement.
    {
        constexpr unsigned long long kRho = 0xfb73d951c840ea62ull;   // rho(k1) = nibble k1
        const int row = (int)((kRho >> (4 * k1)) & 15);
        const float4* rd = reinterpret_cast<const float4*>(wb + row * kT1Row + 16 * hi);
        // row r holds element inv(r): 4, 8, 0, 12, 5, 9, 1, 13, 6, 10, 2, 14, 7, 11, 3, 15
        if constexpr (!(AF_ABL & 8)) {
        addtid_store8<4 * kT1Row, 0>(wv, re[4], re[8], re[0], re[12], re[5], re[9], re[1], re[13]);
        addtid_store8<4 * kT1Row, 8 * 4 * kT1Row>(wv, re[6], re[10], re[2], re[14], re[7], re[11], re[3], re[15]);
        }
        lds_order();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = (AF_ABL & 16) ? float4{re[4 * q], re[4 * q + 1], re[4 * q + 2], re[4 * q + 3]} : rd[q];
            ur[brev4(4 * q)] = v.x; ur[brev4(4 * q + 1)] = v.y; ur[brev4(4 * q + 2)] = v.z; ur[brev4(4 * q + 3)] = v.w;
        }
        lds_order();
        if constexpr (!(AF_ABL & 8)) {
        addtid_store8<4 * kT1Row, 0>(wv, im[4], im[8], im[0], im[12], im[5], im[9], im[1], im[13]);
        addtid_store8<4 * kT1Row, 8 * 4 * kT1Row>(wv, im[6], im[10], im[2], im[14], im[7], im[11], im[3], im[15]);
        }
        lds_order();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = (AF_ABL & 16) ? float4{im[4 * q], im[4 * q + 1], im[4 * q + 2], im[4 * q + 3]} : rd[q];
            ui[brev4(4 * q)] = v.x; ui[brev4(4 * q + 1)] = v.y; ui[brev4(4 * q + 2)] = v.z; ui[brev4(4 * q + 3)] = v.w;
        }
        lds_order();
    }

}

// Part 2: stage 2, spectrum split, power rows.
//   ltw    this lane's stage-2 twiddles (registers for the whole launch, behind Params::tabs), j = lane / 16:
//          cos, sin of W32^j, then for b' = 0..3 tan, cos of w = W64^(j + 4 b')
__device__ __forceinline__ void fft_pair_2(float* wb, const WaveBuf& wv, int lane, const float (&ltw)[kLaneTw],
                                           float (&ur)[16], float (&ui)[16])
{
    // ---- stage 2a: radix-16 over m -> b.  Its twiddle W64^(c b) is applied AFTER the transposition below, where
    // c has become the register index and b = j + 4 b' has one lane-dependent part only: every lane keeps its
    // four w = W64^b (and w^2) in registers and no table row is read (was 8 ds_read_b128 per frame pair).
    if constexpr (!(AF_V2_PH & 2)) fftr::dit_fft<16>(ur, ui);

    // ---- T2: lane (k1, j) needs U[k1][c][j + 4b'] for c = 0..3 from the four lanes (k1, c): per b' a
    // 4 x 4 transpose (lane row c, register j) -> (lane row j, register c) of registers 4b'..4b'+3,
    // done in registers: v_permlane32_swap exchanges rows {2, 3} of one register with rows {0, 1} of
    // another, v_permlane16_swap rows {1, 3} with rows {0, 2}.  Afterwards element 4b' + c holds U_c.
    if constexpr (!(AF_V2_PH & 2))
#pragma unroll
    for (int bp = 0; bp < 4; ++bp) {
        swap_rows32(ur[4 * bp], ur[4 * bp + 2]);
        swap_rows32(ur[4 * bp + 1], ur[4 * bp + 3]);
        swap_rows16(ur[4 * bp], ur[4 * bp + 1]);
        swap_rows16(ur[4 * bp + 2], ur[4 * bp + 3]);
        swap_rows32(ui[4 * bp], ui[4 * bp + 2]);
        swap_rows32(ui[4 * bp + 1], ui[4 * bp + 3]);
        swap_rows16(ui[4 * bp], ui[4 * bp + 1]);
        swap_rows16(ui[4 * bp + 2], ui[4 * bp + 3]);
    }

    // ---- stage 2b: Z_a = sum_c W4^(c a) w^c U_c, in place (Z[lane + 64 t], t = b' + 4a, is element 4b' + a):
    //   E+- = U0 +- w^2 U2,  O+- = U1 +- w^2 U3,  Z0 / Z2 = E+ +- w O+,  Z1 / Z3 = E- +- (-i w) O-
    // w^2 = W32^j W8^b': the lane's W32^j as a plain complex product (registers ltw[0..1]), W8^b' folded into the
    // E / O butterflies at compile time (fft_regs.hpp); w itself in the FMA form w = cos (1 - i tan) from
    // ltw[2 + 2b'], ltw[3 + 2b'] (angle <= 84.4 degrees).
    const float c2 = ltw[0], s2 = ltw[1];
    if constexpr (!(AF_V2_PH & 2))
#pragma unroll
    for (int bp = 0; bp < 4; ++bp) {
        const float t1 = ltw[2 + 2 * bp], c1 = ltw[3 + 2 * bp];
        float epr = ur[4 * bp], epi = ui[4 * bp], opr = ur[4 * bp + 1], opi = ui[4 * bp + 1];
        const float u2r = ur[4 * bp + 2], u3r = ur[4 * bp + 3], u2i = ui[4 * bp + 2], u3i = ui[4 * bp + 3];
        float emr = __builtin_fmaf(c2, u2r, s2 * u2i), emi = __builtin_fmaf(c2, u2i, -(s2 * u2r));    // W32^j U2
        float omr = __builtin_fmaf(c2, u3r, s2 * u3i), omi = __builtin_fmaf(c2, u3i, -(s2 * u3r));    // W32^j U3
        fftr::butterfly(4 * bp, epr, epi, emr, emi);          // (E+, E-) = U0 +- W8^b' (W32^j U2)
        fftr::butterfly(4 * bp, opr, opi, omr, omi);          // (O+, O-) = U1 +- W8^b' (W32^j U3)
        // w O / cos:  (or + oi t) + i (oi - or t)
        const float vpr = __builtin_fmaf(opi, t1, opr), vpi = __builtin_fmaf(-opr, t1, opi);
        const float vmr = __builtin_fmaf(omi, t1, omr), vmi = __builtin_fmaf(-omr, t1, omi);
        ur[4 * bp] = __builtin_fmaf(vpr, c1, epr);      ui[4 * bp] = __builtin_fmaf(vpi, c1, epi);        // a = 0
        ur[4 * bp + 2] = __builtin_fmaf(-vpr, c1, epr); ui[4 * bp + 2] = __builtin_fmaf(-vpi, c1, epi);   // a = 2
        // -i w O- = cos (vmi - i vmr)
        ur[4 * bp + 1] = __builtin_fmaf(vmi, c1, emr);  ui[4 * bp + 1] = __builtin_fmaf(-vmr, c1, emi);   // a = 1
        ur[4 * bp + 3] = __builtin_fmaf(-vmi, c1, emr); ui[4 * bp + 3] = __builtin_fmaf(vmr, c1, emi);    // a = 3
    }
#define ZR(t) ur[4 * ((t) & 3) + ((t) >> 2)]
#define ZI(t) ui[4 * ((t) & 3) + ((t) >> 2)]

    // ---- split of the packed spectrum.  A[k] = (Z[k] + conj Z[N-k]) / 2 (frame a),
    // B[k] = (Z[k] - conj Z[N-k]) / 2i (frame b); the 1/2 sits in the window table.  Z[N-k] for
    // k = lane + 64 t is register 15 - t of lane (64 - lane) % 64 (lane 0: register 16 - t).
    // Published: registers 8..15 as planes 0..7, register 0 as plane 8 (lane 0's "register 16");
    // real parts at plane * 64, imaginary parts 576 dwords further.
    {
        if constexpr (!(AF_ABL & 64)) {
        addtid_store8<256, 0>(wv, ZR(8), ZR(9), ZR(10), ZR(11), ZR(12), ZR(13), ZR(14), ZR(15));
        addtid_store1<8 * 256>(wv, ZR(0));
        addtid_store8<256, 576 * 4>(wv, ZI(8), ZI(9), ZI(10), ZI(11), ZI(12), ZI(13), ZI(14), ZI(15));
        addtid_store1<576 * 4 + 8 * 256>(wv, ZI(0));
        }
        lds_order();
        const int ln = AF_V2_LANE_NOW ? lane_now() : lane;
        const int pl = (64 - ln) & 63;
        const int d = ln == 0 ? 1 : 0;
        const float* ex = wb + pl + 64 * d;
        float pa[9], pb[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // t = 8 only matters for lane 0 (bin 512, its own register 8 = plane 0)
            const float pr = (AF_ABL & 128) ? ZR(15 - t) : t < 8 ? ex[(7 - t) * 64] : wb[pl];
            const float pi = (AF_ABL & 128) ? ZI(15 - t) : t < 8 ? ex[576 + (7 - t) * 64] : wb[576 + pl];
            if constexpr (AF_V2_PH & 2) {
                pa[t] = pr; pb[t] = pi;
                if (t < 7) { abl_use(ZR(t)); abl_use(ZI(t)); }
                continue;
            }
            const float a1 = ZR(t) + pr, a2 = ZI(t) - pi;
            const float b1 = ZI(t) + pi, b2 = ZR(t) - pr;
            pa[t] = __builtin_fmaf(a1, a1, a2 * a2);
            pb[t] = __builtin_fmaf(b1, b1, b2 * b2);
        }
        lds_order();
        // power rows [frame][514] (they alias the exchange planes: every lane has read its partners)
        if constexpr (!(AF_ABL & 256)) {
        addtid_store8<256, 0>(wv, pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], pa[6], pa[7]);
        addtid_store8<256, 4 * kProw2>(wv, pb[0], pb[1], pb[2], pb[3], pb[4], pb[5], pb[6], pb[7]);
        if (lane == 0) {
            if constexpr (kBinMajor) {
                // bin 512 and the three zero bins behind it (the last mel slot may reach bin 515; the words alias T1)
                *reinterpret_cast<float4*>(wb + 512) = float4{pa[8], 0.f, 0.f, 0.f};
                *reinterpret_cast<float4*>(wb + kProw2 + 512) = float4{pb[8], 0.f, 0.f, 0.f};
            } else {
                wb[512] = pa[8];
                wb[kProw2 + 512] = pb[8];
            }
        }
        } else {
            float sink = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) sink += pa[t] + pb[t];
            if (sink == 1.2345e-30f) wb[lane] = sink;      // keeps the arithmetic alive, never taken in practice
        }
    }
#undef ZR
#undef ZI
}

typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ UnitDesc load_desc(const UnitDesc* units, unsigned u)
{
    // Through the scalar cache (a vector load would make every unit wait for vmcnt(0), i.e. for the previous
    // unit's mel stores), and as a load the compiler tracks: the record is read through a constant-address-space
    // pointer with a wave-uniform index, which hipcc turns into s_load_dwordx8 and waits for where the fields are
    // first used -- so the next unit's record can be requested a whole unit ahead.
    typedef const __attribute__((address_space(4))) u32x8* cptr;
    const u32x8 r = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(units))[__builtin_amdgcn_readfirstlane(u)];
    UnitDesc d;
    d.pcm0 = (int64_t)(((unsigned long long)r[1] << 32) | r[0]);
    d.out0 = (int64_t)(((unsigned long long)r[3] << 32) | r[2]);
    d.n_avail = (int)r[4];
    d.n_rows = (int)r[5];
    d.flags = (int)r[6];
    d.sel = (int)r[7];
    return d;
}

// Diagnostics (Params::stamps != null only): per-phase cycle sums of waves 0..3.
#define AF_STAMP2(i)                                                        \
    do {                                                                    \
        if constexpr (STAMPS) {                                             \
            __builtin_amdgcn_sched_barrier(0);                              \
            const unsigned long long _t = __builtin_amdgcn_s_memtime();     \
            __builtin_amdgcn_s_waitcnt(0xC07F);                             \
            acc_t[i] += _t - t_prev;                                        \
            t_prev = _t;                                                    \
            __builtin_amdgcn_sched_barrier(0);                              \
        }                                                                   \
    } while (0)

// POWER (output = |X|^2 rows instead of mel rows) is a template parameter on purpose: as a runtime switch the
// fragment loads sat in one `if (!out_power)` and the MFMAs that consume them in another; hipcc's wait-count
// pass joins the paths, kept the six fragment loads "pending" around the loop, and made the next unit's first
// LDS reads -- which reuse their registers -- wait for vmcnt(5..0): for the PCM prefetch requested just before.
// MEL4 (round 5): the mel phase on 16 blocks of 4 x 4 x 1 (v_mfma_f32_4x4x1_16b_f32; features.hpp Plan4) instead of
// 16 x 16 x 4 tiles: a third of the multiply-adds for the banded filterbank, and the launch is power-limited
// (DESIGN.md section 9, round 5).  NSLOT then counts 16-byte operand quads (4 steps each) per wave.
template <int CH, bool I16, bool SEP, int NSLOT, bool STAMPS, bool POWER, bool MEL4 = false>
__global__ __launch_bounds__(kT, 4) void af_features_v2_kernel(const Params p)
{
    using St = Stage<CH, I16>;
    constexpr int MS = St::MS;                       // mono samples per 16-byte staging unit
    constexpr int PFU = 16 * kHop / MS / kT;         // staging units per thread and unit of work
    static_assert(PFU >= 1 && PFU * MS * kT == 16 * kHop, "prefetch shape");
    static_assert((16 * kHop) % 64 == 0 && kTail % 64 == 0, "tile moves keep the pair order");
    constexpr int kFreshIters = (kTile / MS + kT - 1) / kT;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wbufs = smem;                             // first: ds_write_addtid reaches M0 < 64 KB
    float* tile = wbufs + kW * kWb;
    float* wtab = tile + kTile;
    float* tw1 = wtab + 64 * kWinRow2;
    int* lmeta = reinterpret_cast<int*>(tw1 + 64 * kTw1Row);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* wb = wbufs + w * kWb;
    // LDS byte address of this wave's buffer = the low half of its flat address
    const WaveBuf wv{wb, (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uintptr_t>(wb)), lane};

    // Work: workgroup b owns the contiguous units [r0, r1) (consecutive units of a clip share 15/16 of their signal tile).  The
    // workgroups do not run at one pace -- the average lifetime differs by 8 % between the XCDs and by 10 % inside one
    // (profiles/r05_v2_ablations.txt section 8) -- so with p.queue set every range gives its last `pool` units to a common
    // pool of chunks of (unsigned)kPoolChunk units, dealt by a device-wide counter to whoever has finished its own part.
    const int64_t G = gridDim.x, bid = blockIdx.x;
    unsigned u = (unsigned)(p.total_units * bid / G);
    unsigned u_end = (unsigned)(p.total_units * (bid + 1) / G);
    if (u >= u_end) return;
    const unsigned pool = p.queue ? (unsigned)p.pool_units : 0u;       // the same for every range (<= the shortest range - 1)
    u_end -= pool;
#if AF_V2_CLK
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // tables -> LDS (visible after the first barrier)
    for (int i = tid; i < kTabFloats / 4; i += kT)
        reinterpret_cast<float4*>(wtab)[i] = reinterpret_cast<const float4*>(p.tabs)[i];
    for (int i = tid; i < kW * kMaxSlotsW; i += kT) lmeta[i] = p.meta[i];

    // this wave's share of the mel plan
    int slot_end_w[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) slot_end_w[q] = p.plan.slot_end[w][q];
    const int n_slots_w = p.plan.wave_nslots[w];
    const int n_mels = p.plan.n_mels;
    const bool vec_ok = (n_mels % 4 == 0) && (reinterpret_cast<uintptr_t>(p.out) % 16 == 0);
    const float* pbase = wbufs + ((lane & 15) >> 1) * kWb + (lane & 1) * kProw2 + (kBinMajor ? 4 : 1) * (lane >> 4);
    const unsigned row_off = (unsigned)(lane & 15) * (unsigned)n_mels + 4u * (unsigned)(lane >> 4);

    // MEL4: this lane's block slot (first bin of its piece, filter quad, combine flags) and the group's step count
    // (the lane word is fetched with the fragments in every unit: a register, or a scratch slot, less across the FFT)
    const int m4q = MEL4 ? p.m4_quads[w >> 2] : 0;                    // wave-uniform
    // this lane's stage-2 twiddles, in registers for the whole launch
    float ltw[kLaneTw];
    {
        const float* lt = p.tabs + kTabFloats + kLaneTw * lane;
#pragma unroll
        for (int q = 0; q < kLaneTw; ++q) ltw[q] = lt[q];
        // Wait for them HERE, with a wait the compiler sees: otherwise its wait-count pass carries these loads as
        // possibly pending around the unit loop and puts an s_waitcnt vmcnt(1) in front of their first use in every
        // iteration -- which waits for the PCM prefetch requested at the top of the iteration, in the middle of the FFT.
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
#pragma unroll
        for (int q = 0; q < kLaneTw; ++q) asm volatile("" : "+v"(ltw[q]));
    }

#if AF_V2_STAGGER
    // (experiment) the workgroups start AF_V2_STAGGER x 64 x (blockIdx % 16) cycles apart: every workgroup walks the same phases
    // in the same time, so the whole chip does its FFT arithmetic, then its mel phase, ... together
    for (unsigned i = 0; i < (blockIdx.x & 15u); ++i) __builtin_amdgcn_s_sleep(AF_V2_STAGGER);
#endif
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = STAMPS ? __builtin_amdgcn_s_memtime() : 0;
    bool fresh = true;
    bool pf_ready = false;                           // pf[] already holds / awaits the new hops of the next unit
    uint4 pf[PFU];
    UnitDesc ud_next = load_desc(p.units, u);
    volatile unsigned* ticket_box = reinterpret_cast<volatile unsigned*>(lmeta + kW * kMaxSlotsW);
    [[maybe_unused]] bool owner = true;              // (pool mode 2) still inside the own range
    for (;;) {                                       // segments: the workgroup's own part, then chunks of the pool
    while (u < u_end) {
        const UnitDesc ud = ud_next;
        AF_STAMP2(0);
        const int sel = SEP ? ud.sel : -1;
        const int64_t n_avail = ud.n_avail;

        if (fresh) {
            if constexpr (!(AF_V2_PH & 8)) __syncthreads();     // nobody still reads the old tile / power rows
#pragma unroll
            for (int i = 0; i < kFreshIters; ++i) {
                const int v = tid + i * kT;
                if (v * MS < kTile) {
                    float m[MS];
                    St::load(p.pcm, ud.pcm0, (int64_t)v * MS, n_avail, true, m, sel);
                    tile_store<MS>(tile, v * MS, m);
                }
            }
        }
        if constexpr (!(AF_V2_PH & 8)) __syncthreads();         // B1: tile ready, previous mel phase done
        AF_STAMP2(1);
        // the next unit's record, a unit ahead (the table ends with one spare record)
        ud_next = load_desc(p.units, u + 1);
        __builtin_amdgcn_sched_barrier(0);

        // the ticket of the next pool chunk is taken a whole unit before it is needed (the counter is one address for 512
        // workgroups that come for it at about the same time) and put into LDS behind this unit's vmcnt(0) below
#if AF_V2_POOL_MODE == 2
        // (after B1: the claim this workgroup's thread 0 stored two units ago is visible) extend the segment by the next chunk of
        // the own range while the deque has one
        if (pool && owner && u + 1 == u_end) {
            const unsigned old_w = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket_box[0]);
            if ((old_w & 0xFFFFu) + (old_w >> 16) < pool / (unsigned)kPoolChunk) u_end += (unsigned)kPoolChunk;
            else owner = false;
        }
        const bool take_ticket = pool && owner && (u + 2 == u_end) && tid == 0;
        unsigned tk = 0;
        if (take_ticket) tk = atomicAdd(p.queue + 64 + (unsigned)bid, 1u);
#else
        const bool take_ticket = pool && (u + 1 == u_end) && tid == 0;
        unsigned tk = 0;
        if (take_ticket) tk = atomicAdd(p.queue, 1u);
#endif
        const bool has_next = (u + 1 < u_end) && (ud.flags & 1);
        // requests the 16 new hops of the unit BEHIND unit `d` (whose tile starts at d.pcm0) into pf[LO..HI)
        const auto issue_prefetch = [&](const UnitDesc& d, bool wanted, auto lo_c, auto hi_c) {
            constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
            if constexpr (AF_V2_PH & 32) {
#pragma unroll
                for (int i = LO; i < HI; ++i) pf[i] = uint4{0u, 0u, 0u, 0u};
            } else
            if (wanted && LO < HI) {
                // (lane offsets from an opaque copy of tid: hoisted out of the unit loop they end up in scratch,
                // and a scratch reload waits for every outstanding vector memory operation)
                const int t0 = (w << 6) + lane_now();
                if (d.n_avail >= kTile + 16 * kHop) {  // wave-uniform: the new region lies inside the clip
                    const char* src = static_cast<const char*>(p.pcm) +
                                      (d.pcm0 + kTile) * St::kBytesPerFrame;
#pragma unroll
                    for (int i = LO; i < HI; ++i) {
                        const uint4* a = reinterpret_cast<const uint4*>(src + (unsigned)(t0 + i * kT) * 16u);
                        if constexpr (AF_V2_NT & 1) {
                            typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                            const u32x4v q = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(a));
                            pf[i] = uint4{q[0], q[1], q[2], q[3]};
                        } else {
                            pf[i] = *a;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = LO; i < HI; ++i)
                        pf[i] = St::load_raw(p.pcm, d.pcm0, kTile + (int64_t)(t0 + i * kT) * MS, d.n_avail, true);
                }
            }
        };
        // AF_V2_PF_EARLY of the PFU staging loads go out here, the rest behind part 1 (experiment: default all here)
        constexpr int kPfEarly = AF_V2_PF_EARLY < PFU ? AF_V2_PF_EARLY : PFU;
        issue_prefetch(ud, has_next && !pf_ready, std::integral_constant<int, 0>{}, std::integral_constant<int, kPfEarly>{});

        float wf[NSLOT * kSlotSteps];
        int l4 = 0;
        {
            float ur[16], ui[16];
            // The issue arbiter prefers the older wave: waves 4..7 of a workgroup share their SIMDs with waves
            // 0..3 and would run the two FFT parts ~25 % slower (10.5 k against 8.5 k cycles), everyone then
            // waiting for them at B2.  One priority step for the younger half evens that out (-2.3 %).
            if (w >= 4) __builtin_amdgcn_s_setprio(1 + AF_V2_BASE_PRIO);
            else if (AF_V2_BASE_PRIO) __builtin_amdgcn_s_setprio(AF_V2_BASE_PRIO);
            float s1[20], win[16];
            stage1_load(tile, w, lane, wtab, s1, win);
            fft_pair_1(s1, win, wb, wv, lane, tw1, ur, ui);
            issue_prefetch(ud, has_next && !pf_ready, std::integral_constant<int, kPfEarly>{}, std::integral_constant<int, PFU>{});

            AF_STAMP2(2);
            // This wave's filterbank fragments (L2-resident, one float4 per slot), issued where the
            // register pressure has dropped and early enough to be back for the mel phase.
            if constexpr (!POWER && (AF_V2_PH & (64 | 128))) {
#pragma unroll
                for (int q = 0; q < NSLOT * kSlotSteps; ++q) wf[q] = abl_val(lane, q);
            } else if constexpr (MEL4) {
                int opaque_zero;
                asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
                const float4* wbase = reinterpret_cast<const float4*>(p.wfrag4) + (size_t)(w >> 2) * kM4Quads * 64;
                const unsigned wo = (unsigned)lane_now() + (unsigned)opaque_zero;
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    const float4 v = wbase[wo + q * 64];
                    wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
                }
                l4 = (p.lane4 + (w >> 2) * 64)[wo];
            } else if constexpr (!POWER) {
                int opaque_zero;                     // pins the loads here (and defeats hoisting)
                asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
                // (scalar base + 32-bit lane offset: a hoisted 64-bit lane pointer is one more loop-carried register pair)
                const float4* wbase = reinterpret_cast<const float4*>(p.wfrag) + (size_t)w * kMaxSlotsW * 64;
                const unsigned wo = (unsigned)lane_now() + (unsigned)opaque_zero;
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    const float4 v = wbase[wo + q * 64];
                    wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < NSLOT * kSlotSteps; ++q) wf[q] = 0.0f;
            }
            fft_pair_2(wb, wv, lane, ltw, ur, ui);
            __builtin_amdgcn_s_setprio(AF_V2_BASE_PRIO);
        }
        AF_STAMP2(3);

        if constexpr (!(AF_V2_PH & 8)) __syncthreads();         // B2: power rows ready, tile reads done
        // the tile update and the mel phase hold up the next unit's FFT and issue few instructions:
        // let them go ahead of the other workgroup's FFT stream
        __builtin_amdgcn_s_setprio(AF_V2_BASE_PRIO ? 3 : 2);
        AF_STAMP2(4);
        // Everything this wave has in flight is old by now (prefetch: a whole FFT ago; fragments: half of one; the
        // previous unit's mel stores: a unit ago) and all of it is needed right below.  One wait the compiler sees,
        // instead of its own: it counts the fragment loads against the mel stores issued between the slots (vector
        // memory operations retire in order), so a slot behind a flush waited for that flush's HBM stores.
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
        if (take_ticket) *ticket_box = tk;

        if (has_next && !(AF_ABL & 1024)) {
            // The next tile is this one shifted by 16 hops: the last kTail samples move to the front,
            // the new samples follow.  A staging unit that lands on part of the old tail is handled by
            // one thread: read the old samples, store the new ones there, store the old ones at the front.
            const int t0 = (w << 6) + lane_now();
#pragma unroll
            for (int i = 0; i < PFU; ++i) {
                const int v = t0 + i * kT;
                float m[MS];
                St::convert(pf[i], m, sel);
                if ((i + 1) * kT * MS > 16 * kHop - kTail) {        // compile time: this round reaches the tail
                    if (v * MS >= 16 * kHop - kTail) {
                        // (the old samples move down by 16 kHop, an even number of 32-sample blocks: source and
                        // destination have the same pair order, the move is a raw copy)
                        float* dst = tile + tile_phys<MS>(kTail + v * MS);
                        float old[MS];
#pragma unroll
                        for (int e = 0; e < MS; ++e) old[e] = dst[e];
                        tile_store<MS>(tile, kTail + v * MS, m);
                        St::store(dst - 16 * kHop, old);
                        continue;
                    }
                }
                tile_store<MS>(tile, kTail + v * MS, m);
            }
        }
        if constexpr (AF_V2_PF_AHEAD) {
            // The staging registers are free again: request the new hops of the unit AFTER next (they lie behind the
            // next unit's tile; its record arrived a phase ago).  They are in flight through this mel phase and the whole
            // next unit -- requested at the top of the next unit (rounds 2-4) they had one FFT's time, which the loaded
            // HBM latency of a chip whose workgroups all request at the same moment exceeds (profiles/r05_v2_ablations.txt).
            const bool has_next2 = has_next && (u + 2 < u_end) && (ud_next.flags & 1);
            issue_prefetch(ud_next, has_next2, std::integral_constant<int, 0>{}, std::integral_constant<int, PFU>{});
            pf_ready = has_next2;
        }

        if constexpr (!POWER && (AF_V2_PH & 128)) {
            abl_use(wf[0]);
        } else if constexpr (MEL4) {
            // mel on 4 x 4 x 1 blocks: lane (block b, j) supplies frame 4 (w & 3) + j of the piece's bins (B) and filter
            // 4 q + j of them (A, registers); the accumulator of block b is its piece's 4 filters x 4 frames.  Operands in two
            // halves of at most 16 steps (the whole set would be 32 more live registers next to the fragments).
            // (everything derived from the lane word starts from an opaque copy: hoisted out of the unit loop the row
            // pointer, the frame number and the output offset end up in scratch, reloaded here in every unit)
            const int l4o = l4;
            const int fr = 4 * (w & 3) + ((l4o >> 20) & 3);
            const float4* prow = reinterpret_cast<const float4*>(wbufs + (fr >> 1) * kWb + (fr & 1) * kProw2 + (l4o & 1023));
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh < NSLOT; hh += 4) {
                float4 bq[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) bq[k] = hh + k < NSLOT ? prow[hh + k] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (hh + k < NSLOT && hh + k < m4q) {                 // wave-uniform
                        const int st = 4 * (hh + k);
                        if constexpr (AF_V2_M4 & 1) {
                            abl_use(wf[st]); abl_use(wf[st + 1]); abl_use(wf[st + 2]); abl_use(wf[st + 3]);
                            abl_use(bq[k].x); abl_use(bq[k].y); abl_use(bq[k].z); abl_use(bq[k].w);
                            continue;
                        }
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[st], bq[k].x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[st + 1], bq[k].y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[st + 2], bq[k].z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[st + 3], bq[k].w, acc1, 0, 0, 0);
                    }
                }
            }
            f32x4 r = acc0 + acc1;
            // the pieces of a band lie side by side in one row of four blocks: piece 0 += piece 1 and piece 2 += piece 3
            // (row_shl:4 = the value of the lane four to the right), then piece 0 += piece 2 (row_shl:8)
            const bool add1 = (l4o >> 18) & 1, add2 = (l4o >> 19) & 1;
            if constexpr (!(AF_V2_M4 & 4)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (a copy first: __builtin_bit_cast of the vector ELEMENT r[i] reads element 0 with this compiler)
                const float ri = r[i];
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ri), 0x104, 0xF, 0xF, true));
                r[i] = add1 ? ri + v : ri;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ri = r[i];
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ri), 0x108, 0xF, 0xF, true));
                r[i] = add2 ? ri + v : ri;
            }
            }
            if (p.log_mel) {                                            // wave-uniform: log10(max(mel, 1e-10))
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = __log10f(fmaxf(r[i], 1e-10f));
            }
#if AF_V2_M4_STAGE
            {
                float* stage = reinterpret_cast<float*>(lmeta + kW * kMaxSlotsW) + 4;
                if ((l4o >> 17) & 1)
                    *reinterpret_cast<float4*>(stage + fr * n_mels + 4 * ((l4o >> 10) & 127)) = float4{r[0], r[1], r[2], r[3]};
                __syncthreads();
                const int t0 = (w << 6) + lane_now();
                if (t0 * 4 < ud.n_rows * n_mels)
                    reinterpret_cast<float4*>(p.out + ud.out0 * n_mels)[t0] = reinterpret_cast<const float4*>(stage)[t0];
            }
            if (false) {
                float* o = nullptr;
#else
            if (((l4o >> 17) & 1) && fr < ud.n_rows) {                  // the band's first piece holds its sums
                // uniform base + 32-bit lane offset
                float* o = p.out + ud.out0 * n_mels + (unsigned)(fr * n_mels + 4 * ((l4o >> 10) & 127));
#endif
                if constexpr (AF_V2_M4 & 2) {
                    abl_use(r[0]); abl_use(r[1]); abl_use(r[2]); abl_use(r[3]);
                } else
                if (vec_ok) {
                    *reinterpret_cast<float4*>(o) = float4{r[0], r[1], r[2], r[3]};
                } else {
                    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
                }
            }
        } else if constexpr (!POWER) {
            // mel: banded W (A, registers) x P^T (B, bins x 16 frames from the power rows) on
            // v_mfma_f32_16x16x4_f32; all B operands first, then the chain; a slot that ends a group
            // sends the 16 frames x <= 16 filters accumulator to HBM (one 16-byte store per lane)
            float bop[NSLOT * kSlotSteps];
            {
                const int* lm = lmeta + w * kMaxSlotsW;
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    // one ds_read_b128: step i of the slot takes bins bin0 + 4 kq + i (the fragments are cut to match)
                    const float* bq = pbase + lm[q];
                    const float4 bv = (AF_ABL & 512) ? float4{abl_val(lane, q), 0.5f, 0.25f, 0.125f}
                                      : kBinMajor    ? *reinterpret_cast<const float4*>(bq)
                                                     : float4{bq[0], bq[4], bq[8], bq[12]};
                    bop[q * kSlotSteps] = bv.x; bop[q * kSlotSteps + 1] = bv.y;
                    bop[q * kSlotSteps + 2] = bv.z; bop[q * kSlotSteps + 3] = bv.w;
                }
            }
            if constexpr (STAMPS) {                                  // diagnostics: operands and fragments landed
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                AF_STAMP2(6);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                AF_STAMP2(7);
            }
            float* obase = p.out + ud.out0 * n_mels;                // wave-uniform
            const bool fvalid = (lane & 15) < ud.n_rows;
            const int i0 = 4 * (lane >> 4);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            // Slots this wave does not use are skipped (wave-uniform branch).  Round 2 had to run every compiled-in
            // slot: skipped slots left their fragment loads unconsumed and hipcc made the next unit's first LDS
            // reads -- which overwrite those registers -- wait for vmcnt(5..0), i.e. for the PCM prefetch just
            // requested.  The explicit vmcnt(0) above retires all of them for the compiler, so the branch is free
            // of that; AF_V2_ALL_SLOTS = 1 (ablation) restores the branch-free form.
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                if (AF_V2_ALL_SLOTS || q < n_slots_w) {
#pragma unroll
                    for (int st = q * kSlotSteps; st < (q + 1) * kSlotSteps; st += 2) {
                        if constexpr (AF_V2_PH & 4) {
                            acc0[st & 3] += wf[st] + bop[st];
                            acc1[st & 3] += wf[st + 1] + bop[st + 1];
                            continue;
                        }
                        if constexpr (AF_V2_PH & 256) {      // proxy: a quarter of the matrix work per step (16 blocks of 4 x 4 x 1)
                            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[st], bop[st], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[st + 1], bop[st + 1], acc1, 0, 0, 0);
                            continue;
                        }
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st], bop[st], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st + 1], bop[st + 1], acc1, 0, 0, 0);
                    }
                    const int e = slot_end_w[q];                    // wave-uniform (kernarg)
                    if (e != 0) {
                        f32x4 r = acc0 + acc1;
                        if (p.log_mel) {                            // wave-uniform: log10(max(mel, 1e-10))
#pragma unroll
                            for (int i = 0; i < 4; ++i) r[i] = __log10f(fmaxf(r[i], 1e-10f));
                        }
                        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                        const int mm = e & 255, cnt = (e >> 8) & 31;
                        float* o = obase + mm + row_off;            // uniform base + 32-bit lane offset
                        if constexpr (AF_ABL & 2048)                // (diagnostic: wrong place, 1 KB contiguous per instruction)
                            o = obase + (mm >> 4) * 256 + 4 * lane;
                        if constexpr (AF_V2_PH & 16) {
                            abl_use(r[0]); abl_use(r[1]); abl_use(r[2]); abl_use(r[3]);
                        } else
                        if (vec_ok && ((cnt | mm) & 3) == 0) {      // wave-uniform choice
                            if constexpr (AF_V2_NT & 2) {
                                if (fvalid && i0 < cnt) __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(o));
                            } else
                            if (fvalid && i0 < cnt) *reinterpret_cast<float4*>(o) = float4{r[0], r[1], r[2], r[3]};
                        } else if (fvalid) {
                            if (i0 < cnt) o[0] = r[0];
                            if (i0 + 1 < cnt) o[1] = r[1];
                            if (i0 + 2 < cnt) o[2] = r[2];
                            if (i0 + 3 < cnt) o[3] = r[3];
                        }
                    }
                }
            }
        } else {
            // power output: this wave's two rows [frame][513], consecutive lanes on consecutive bins
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = 2 * w + r;
                if (f < ud.n_rows) {
                    float* o = p.out + (ud.out0 + f) * kBins;
                    for (int k = lane; k < kBins; k += 64) o[k] = wb[r * kProw2 + k];
                }
            }
        }
        __builtin_amdgcn_s_setprio(AF_V2_BASE_PRIO);
        AF_STAMP2(5);
        ++u;
        fresh = !has_next;
    }
        if (!pool) break;
#if AF_V2_POOL_MODE == 2
        // out of own work: steal a chunk from the back of another range's deque.  Thread 0 walks the global ticket list
        // (ticket t probes range t % G; K rounds over all ranges) until a claim is valid or the list ends.
        owner = false;
        __syncthreads();                             // everybody is done with this segment (and with the box)
        if (tid == 0) {
            const unsigned K = pool / (unsigned)kPoolChunk;
            unsigned found = 0xFFFFFFFFu;
            for (;;) {
                const unsigned t = atomicAdd(p.queue, 1u);
                if (t >= K * (unsigned)G) break;
                const unsigned v = t % (unsigned)G;
                const unsigned old_w = atomicAdd(p.queue + 64 + v, 0x10000u);
                const unsigned f = old_w & 0xFFFFu, r = old_w >> 16;
                if (f + r < K) { found = (unsigned)(p.total_units * (v + 1) / G) - pool + (K - 1u - r) * (unsigned)kPoolChunk; break; }
            }
            ticket_box[1] = found;
        }
        __syncthreads();
        const unsigned found = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket_box[1]);
        if (found == 0xFFFFFFFFu) break;
        u = found;
        u_end = u + (unsigned)kPoolChunk;
        fresh = true;
        pf_ready = false;
        ud_next = load_desc(p.units, u);
        continue;
#endif
        // the next chunk of the pool: ticket t is chunk t / G of range t % G (the chunks of one range are dealt far apart in time)
        __syncthreads();                             // (the ticket was stored a mel phase ago; everybody is done with this segment)
        const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)*ticket_box);
#if AF_V2_POOL_GUIDED
        // (experiment) chunks that shrink: every range's pool is cut into 1/2, 1/4, 1/8 and the rest, the large ones dealt first
        if (t >= 4u * (unsigned)G) break;
        const unsigned rb = t % (unsigned)G, rk = t / (unsigned)G;
        const unsigned c0 = pool >> 1, c1 = pool >> 2, c2 = pool >> 3;
        const unsigned off = rk == 0 ? 0u : rk == 1 ? c0 : rk == 2 ? c0 + c1 : c0 + c1 + c2;
        const unsigned len = rk == 0 ? c0 : rk == 1 ? c1 : rk == 2 ? c2 : pool - (c0 + c1 + c2);
        u = (unsigned)(p.total_units * (rb + 1) / G) - pool + off;
        u_end = u + len;
#else
        const unsigned per_range = pool / (unsigned)kPoolChunk;
        if (t >= per_range * (unsigned)G) break;
        const unsigned rb = t % (unsigned)G, rk = t / (unsigned)G;
        u = (unsigned)(p.total_units * (rb + 1) / G) - pool + rk * (unsigned)kPoolChunk;
        u_end = u + (unsigned)kPoolChunk;
#endif
        fresh = true;
        pf_ready = false;
        ud_next = load_desc(p.units, u);
    }
    if constexpr (STAMPS) {
        if (lane == 0 && w < 4)
            for (int i = 0; i < 8; ++i) atomicAdd(p.stamps + w * 8 + i, acc_t[i]);
    }
#if AF_V2_CLK
    if (p.stamps && tid == 0) {
        atomicAdd(p.stamps, __builtin_amdgcn_s_memtime() - clk_t0);
        atomicAdd(p.stamps + 1, __builtin_amdgcn_s_memrealtime() - clk_r0);
        atomicAdd(p.stamps + 2, 1ull);
        // the spread of the workgroups' lifetimes (100 MHz ticks): sum and maximum by blockIdx % 8, the latest end (ticks since the
        // first workgroup's start is not knowable here: s_memrealtime is a free-running counter, so ends are compared directly)
        const unsigned long long life = __builtin_amdgcn_s_memrealtime() - clk_r0;
        atomicAdd(p.stamps + 8 + (blockIdx.x & 7), life);
        atomicMax(p.stamps + 16 + (blockIdx.x & 7), life);
        atomicMax(p.stamps + 24, __builtin_amdgcn_s_memrealtime());
        atomicMax(p.stamps + 25, ~clk_r0);                             // (~earliest start: the buffer starts zeroed)
        // dispatch ramp and tail of ONE launch (meaningful when the stamps hold a single launch): sum and latest of the starts,
        // sum and ~earliest of the ends
        const unsigned long long clk_r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(p.stamps + 26, clk_r0);
        atomicMax(p.stamps + 27, clk_r0);
        atomicAdd(p.stamps + 28, clk_r1);
        atomicMax(p.stamps + 29, ~clk_r1);
    }
#endif
}

template <int CH, bool I16, bool SEP, int NSLOT, bool STAMPS, bool POWER>
static af_status launch_kp(const Params& p, int grid, hipStream_t stream)
{
    auto kern = af_features_v2_kernel<CH, I16, SEP, NSLOT, STAMPS, POWER>;
    constexpr size_t smem = sizeof(float) * (kLdsFloats + 4);        // (+ the ticket word of the pool)
    static_assert(2 * smem <= 160 * 1024, "two workgroups per CU");
    static LdsAttrFlags attr_set;
    AF_HIP(allow_big_lds(kern, (int)smem, attr_set));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kT), smem, stream, p);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

template <int CH, bool I16, bool SEP, int NSLOT, bool STAMPS = false>
static af_status launch_k(const Params& p, int grid, hipStream_t stream)
{
    // (the power-output rows need no mel slots: one instance per format there)
    if (p.out_power) return launch_kp<CH, I16, SEP, 6, false, true>(p, grid, stream);
    return launch_kp<CH, I16, SEP, NSLOT, STAMPS, false>(p, grid, stream);
}

// the mel phase on 4 x 4 x 1 blocks: instances by operand quads per wave (7 or 8 = 28 / 32 steps)
template <int CH, bool I16, bool SEP>
static af_status launch_m4(const Params& p, int grid, hipStream_t stream)
{
    const int most = p.m4_quads[0] > p.m4_quads[1] ? p.m4_quads[0] : p.m4_quads[1];
    auto go = [&](auto kern) -> af_status {
        constexpr size_t smem = sizeof(float) * (kLdsFloats + 4 + (AF_V2_M4_STAGE ? 16 * 128 : 0));
        static LdsAttrFlags attr_set;
        AF_HIP(allow_big_lds(kern, (int)smem, attr_set));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kT), smem, stream, p);
        AF_HIP(hipGetLastError());
        return AF_OK;
    };
    if (most <= 6) return go(af_features_v2_kernel<CH, I16, SEP, 6, false, false, true>);
    if (most <= 7) return go(af_features_v2_kernel<CH, I16, SEP, 7, false, false, true>);
    return go(af_features_v2_kernel<CH, I16, SEP, 8, false, false, true>);
}

template <int CH, bool I16>
static af_status launch_fmt(const Params& p, bool separate, int grid, hipStream_t stream)
{
    int most = 0;
    for (int w = 0; w < kW; ++w) most = p.plan.wave_nslots[w] > most ? p.plan.wave_nslots[w] : most;
    if (most > 8) return fail(AF_ERR_UNSUPPORTED, "features v2: %d mel slots on one wave (limit 8)", most);
    if (p.wfrag4 && !p.out_power && (!p.stamps || AF_V2_CLK)) {
        if constexpr (CH == 2) {
            if (separate) return launch_m4<CH, I16, true>(p, grid, stream);
        }
        return launch_m4<CH, I16, false>(p, grid, stream);
    }
    if constexpr (CH == 2) {
        if (separate) return launch_k<CH, I16, true, 8>(p, grid, stream);
    }
#if !AF_V2_CLK
    if constexpr (CH == 2 && !I16) {
        if (p.stamps) return launch_k<CH, I16, false, 8, true>(p, grid, stream);
    }
#endif
    if (most <= 6) return launch_k<CH, I16, false, 6>(p, grid, stream);
    return launch_k<CH, I16, false, 8>(p, grid, stream);
}

af_status launch(const Params& p, int channels, bool i16, bool separate, int grid, hipStream_t stream)
{
    if (grid <= 0) return AF_OK;
    if (channels == 2 && !i16) return launch_fmt<2, false>(p, separate, grid, stream);
    if (channels == 1 && !i16) return launch_fmt<1, false>(p, false, grid, stream);
    if (channels == 2 && i16) return launch_fmt<2, true>(p, separate, grid, stream);
    if (channels == 1 && i16) return launch_fmt<1, true>(p, false, grid, stream);
    return fail(AF_ERR_UNSUPPORTED, "features: channels must be 1 or 2 (got %d)", channels);
}

}  // namespace v2
}  // namespace feat
}  // namespace af
