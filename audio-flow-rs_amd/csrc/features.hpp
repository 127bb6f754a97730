// features.hpp -- shared declarations for the fused STFT+mel path
// (FeaturePipeline of include/audioflow_dsp.h; docs/OPS.md "STFT", "Mel").
#pragma once

#include <cstdint>

#include "af_common.hpp"

namespace af {
namespace feat {

constexpr int kNfft = 1024;
constexpr int kBins = kNfft / 2 + 1;      // 513
constexpr int kFramesPerIter = 16;        // spectrogram frames per workgroup iteration
constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;
constexpr int kProw = 514;                // dwords per power row (513 + 1; == 2 mod 32)
constexpr int kXStride = 33;              // transpose row stride (32 + 1)
constexpr int kPStride = 35;              // partner-exchange row stride (17 re + 17 im + 1)
constexpr int kHalfDw = 32 * kPStride;    // exchange region of one half-wave (1120 dwords)
constexpr int kWbuf = 2 * kHalfDw + 8;    // per-wave LDS buffer (2248 dwords; == 8 mod 32)
constexpr int kMaxSteps = 64;             // MFMA K-steps (4 bins each) per wave
constexpr int kWinRow = 36;               // LDS window table row: 32 values + 4 pad
// LDS twiddle table row: 32 (cos,sin) + 2 pad.  The kernel reads a row as 8-byte pairs
// (ds_read2_b64, banked modulo 32 dwords): a stride = 2 mod 4 dwords spreads 16 lanes x 2
// dwords over all 32 banks (68 = 4 mod 32 made lanes n2 and n2 + 8 collide).
constexpr int kTwRow = 66;

// Mel-phase plan (built on the host): the filterbank is cut into groups of at
// most 16 consecutive filters (one MFMA row tile each); a group is multiplied
// only over the band of bins where it is non-zero, in K-steps of 4 bins.  Whole
// groups are dealt to the 4 waves so that the busiest wave has as few steps as
// possible; no group is split, so every wave writes its own mel columns.
// A wave's steps are laid out in slots of 4 (a group starts on a slot boundary
// and is padded with zero-weight steps), so the kernel's MFMA sequence is
// branch-free inside a slot and the "group finished" code exists once per slot.
constexpr int kSlotSteps = 4;
constexpr int kMaxSlotsW = kMaxSteps / kSlotSteps;       // slots per wave
constexpr int kPlanWaves = 8;            // the general kernel deals the groups to 4 waves, the hop-256 kernel to 8
struct Plan {
    int n_mels, n_groups, n_steps;        // n_steps = real (unpadded) steps over all waves
    int n_waves;                          // waves the groups are dealt to
    int wave_nslots[kPlanWaves];          // slots used by each wave
    // per slot: 0 = group continues; else 0x10000 | filters << 8 | first filter
    // (the group ends with this slot and its accumulator is written out)
    int slot_end[kPlanWaves][kMaxSlotsW];
};

// ---- hop-256 kernel (features_v2.hip) ---------------------------------------------------------
// One wave64 owns one frame pair (f, f+1) packed as one 1024-point complex FFT done as
// 16 x 16 x 4 with 16 register-resident points per lane, one transposition through a wave-private
// LDS buffer and one in registers; 8 waves = one 16-frame unit, two workgroups per CU = 4 waves per SIMD.
namespace v2 {
constexpr int kW = 8;                     // waves per workgroup
constexpr int kT = 64 * kW;               // threads per workgroup
constexpr int kHop = 256;
constexpr int kTile = 15 * kHop + kNfft;  // mono samples of one unit (4864)
constexpr int kTail = kNfft - kHop;       // samples shared with the next unit (768)
// Power rows of this kernel: frame f = 2w + r lives in wave w's buffer at r * kProw2.  The mel phase reads them with
// ds_read_b128 (lane (frame n, k-group kq) takes bins bin0 + 4 kq .. + 3 of a slot): its 16-lane groups are
// {n 0-3, 12-15 of kq} + {n 4-11 of kq + 1}, so with kWb = 16 and kProw2 = 8 (mod 64 dwords) the 16-byte slot of a
// lane is 2 n + kq (mod 16) and every group covers all 16 slots = 64 banks.
#ifndef AF_V2_MEL_B32
constexpr bool kBinMajor = true;
constexpr int kProw2 = 520;               // dwords per power row: 513 bins + 3 zeros (a slot may end at bin 515) + pad
constexpr int kWb = 1168;                 // per-wave LDS buffer in dwords: exchange planes (2 x 576), T1 (16 x 68), 2 power rows
#else                                     // ablation (scripts/ablate): round 2's four ds_read_b32 per slot
constexpr bool kBinMajor = false;
constexpr int kProw2 = 514;               // == 2 (mod 32)
constexpr int kWb = 1284;                 // == 4 (mod 32): 16 frames x 2 k-groups of a ds_read_b32 on 32 different banks
#endif
constexpr int kT1Row = 68;                // first transposition: row k1 = 64 columns + 4: the 16 lanes (k1, c) of a
                                          // ds_read_b128 phase start 4 k1 banks apart and cover all 64
constexpr int kWinRow2 = 20;              // window table row (one per lane): 16 values + 4 pad
constexpr int kTw1Row = 36;               // twiddle table row (one per lane): 16 (cos, sin) + 4 pad
constexpr int kTabFloats = 64 * kWinRow2 + 64 * kTw1Row;    // the tables every workgroup copies into LDS
constexpr int kLaneTw = 10;               // per-lane stage-2 twiddles (registers), behind the tables in global memory
constexpr int kLaneTwFloats = 64 * kLaneTw + 2;
constexpr int kLdsFloats = kW * kWb + kTile + kTabFloats + kW * kMaxSlotsW;
static_assert((!kBinMajor || (kWb % 64 == 16 && kProw2 % 64 == 8)) && kWb >= 2 * 576 && kWb >= 16 * kT1Row && kWb >= 2 * kProw2, "wave buffer layout");

// One 16-frame unit, 32 bytes, read with one scalar load.
struct UnitDesc {
    int64_t pcm0;        // first PCM frame of the unit's signal tile (frames from the buffer start)
    int64_t out0;        // first output row
    int32_t n_avail;     // PCM frames from pcm0 to the end of the clip (clamped to 2^30)
    int32_t n_rows;      // valid output rows (1..16)
    int32_t flags;       // bit 0: the next unit continues this clip (its tile = this one shifted by 16 hops);
                         // bits 8..15: how many units follow in this clip (clamped to 255; features_v3 prefetches by it)
    int32_t sel;         // -1: downmix; 0 / 1: that channel alone (AF_CH_SEPARATE)
};

struct Params {
    const void* pcm;
    float* out;               // mel rows [total_frames][n_mels] or power rows [total_frames][513]
    const float* tabs;        // wintab [64][kWinRow2] | tw1 [64][kTw1Row] | lanetw [64][kLaneTw] (not copied to LDS)
    const float* wfrag;       // [8][kMaxSlotsW][64][4]
    const int* meta;          // [8][kMaxSlotsW]
    const UnitDesc* units;
    Plan plan;
    int64_t total_units;
    int out_power, log_mel;
    unsigned long long* stamps;   // diagnostics only: [4 waves][8] summed phase cycles, or null
    // mel phase on v_mfma_f32_4x4x1_16b_f32 (round 5; Plan4 below): null = the 16 x 16 x 4 plan above
    const float* wfrag4;      // [2 groups][kM4Quads][64][4]
    const int* lane4;         // [2 groups][64]: k_start | quad << 10 | owner << 17 | add1 << 18 | add2 << 19 | (lane & 3) << 20
    int m4_quads[2];          // 16-byte operand quads (= 4 steps) of each group
    // work distribution: with `queue` (a device counter, zeroed before the launch) every workgroup's range gives its last
    // `pool_units` units (a multiple of the kernel's chunk, the same for all ranges) to a pool dealt by that counter
    unsigned* queue;
    int pool_units;
};

// The mel phase on 16 blocks of 4 x 4 x 1 (round 5).  The banded filterbank touches two filters per bin; a 16-filter
// tile multiplies every bin of its band with all 16 rows, a 4-filter tile wastes a quarter of that.  A tile = one filter
// quad x one frame quad, accumulating over the quad's band; wave w takes frame quad w & 3 and the filter quads of group
// w >> 2; its 16 block slots hold the group's tiles, long bands cut into up to four pieces that sit side by side in
// one row of four blocks (summed at the end by two DPP row shifts).  All pieces of a group run `steps` bins.
constexpr int kM4Quads = 8;               // at most 32 steps per wave
// units per chunk of the common pool, and the share of every workgroup's range that goes there: 1 / 2^kPoolShift
#ifndef AF_V2_POOL_CHUNK
#define AF_V2_POOL_CHUNK 11
#endif
#ifndef AF_V2_POOL_SHIFT
#define AF_V2_POOL_SHIFT 3
#endif
constexpr int kPoolChunk = AF_V2_POOL_CHUNK;
constexpr int kPoolShift = AF_V2_POOL_SHIFT;
struct Plan4 {
    bool ok = false;
    int quads[2] = {0, 0};                // steps / 4 per group
    int n_pieces[2] = {0, 0};
    int read_conflict_score[2] = {0, 0};  // 16-byte LDS slots two lanes of a read group share, after the arrangement search (0: conflict-free)
};

af_status launch(const Params& p, int channels, bool i16, bool separate, int grid, hipStream_t stream);
}  // namespace v2

// ---- any other frame length (features_dft.hip, round 5) ------------------------------------------
// n_fft outside {1024, 512, 256} (400 = 25 ms at 16 kHz, the usual speech frame; websocket.rs:274 names the 16 kHz
// format) has no power-of-two factorisation to hand to the register FFT.  The frame's DFT is taken as what it is, a
// matrix product, on v_mfma_f32_16x16x4_f32: D[bin][frame] = sum_k T[k][bin] x[frame * hop + k], T = window x
// (cos | -sin), 16 frames x 16 bins per tile, f32 throughout -- at half the length: the window is symmetric, so the real
// part is a product with the folded samples x[n] + x[N - n] and the imaginary part one with x[n] - x[N - n].  The
// accumulator layout (rows = bins 4 kq + i, column = frame) is the B-operand layout of the mel product that follows
// (mel^T[filter][frame] += W[filter][bin] P[bin][frame]), so the power values never leave their registers.  One
// two-wave workgroup = one 16-frame unit (the unit records of the hop-256 kernel, with the handle's hop): each wave
// builds half of the folded rows in LDS and multiplies every other bin tile, four workgroups per CU = two waves per
// SIMD; the fragment stream is double-buffered by chunks of four K groups; the waves' partial mel sums meet in LDS.
namespace dft {
constexpr int kMaxMelTiles = 8;           // n_mels <= 128
struct Params {
    const void* pcm;
    float* out;               // mel rows [total_frames][n_mels] or power rows [total_frames][n_fft / 2 + 1]
    const float* tfrag;       // [bin tile][k group][cos, -sin][64 lanes][4]: w[n] cos / -sin (2 pi n bin / N), n = 16 g + 4 kq + j + 1, bin = 16 bt + (lane & 15)
    const float* wfrag;       // [bin tile][mel tile][64 lanes][4]: W[16 mt + (lane & 15)][16 bt + 4 kq + i]
    const uint32_t* wmask;    // [bin tile]: bit mt set = that tile of W has a non-zero weight
    const v2::UnitDesc* units;
    unsigned* queue;          // device counter, zeroed before the launch: units behind the first gridDim.x are dealt by it (null: static stride)
    int64_t total_units;
    int hop, kg, nb, n_bins, n_mels, nmt;   // kg = K groups of 16 folded samples, nb = bin tiles of 16
    int n_fft, nh;            // nh = n_fft / 2 folded samples per frame (n = 1 .. nh paired with n_fft - n)
    int row_stride;           // floats per frame row: E [16 kg] | O [16 kg] | 8 pad (8 mod 16: conflict-free 16-byte reads)
    int tile_stride;          // floats of LDS per wave: 16 frame rows
    int out_power, log_mel;
    unsigned long long* stamps;   // diagnostics (af_features_debug_stamps), or null: summed over waves -- [0] cycles building the
                                  // folded rows, [1] cycles in the products, [2] units, [3] kernel lifetime in cycles, [4] the same
                                  // in 100 MHz ticks, [5] waves
};
// `waves` per workgroup: 2 (they share one 16-frame unit, each taking every other bin tile) or 1 (diagnostics); `grid` workgroups
af_status launch(const Params& p, int channels, bool i16, bool separate, int waves, int grid, hipStream_t stream);
}  // namespace dft

// ---- hop-256 kernel, DFT on the matrix cores (features_v3.hip) ---------------------------------
// The 1024-point transform of 16 frames at once as two stages of 32-point DFTs written as matrix products
// (n = 32 n1 + n2, k = k1 + 32 k2): stage 1 contracts n1 with a per-n2 matrix that carries the Hann window AND the
// twiddle W1024^(n2 k1); stage 2 contracts n2 with the plain 32-point DFT matrix.  Operands are f16 pairs
// (x = hi + 2^-11 lo'), three v_mfma_f32_16x16x32_f16 per product (hi hi, hi lo', lo' hi), f32 accumulation.
namespace v3 {
constexpr int kW = 8;                     // waves per workgroup, ONE workgroup per CU (256 VGPRs per lane)
constexpr int kT = 64 * kW;
constexpr int kHop = v2::kHop;
constexpr int kTile = v2::kTile;          // 4864 mono samples = 152 blocks of 32
constexpr int kTail = v2::kTail;          // 768
constexpr int kBlk = kTile / 32;          // 152
constexpr int kQuads = kBlk / 4;          // 38 groups of four blocks
constexpr int kTRow = 2 * kBlk;           // bytes per row of the f16 signal image T[n2][blk] (304: 16-lane b64 stores conflict-free)
// Y: bytes per (k1 slot, frame): Yr hi | Yi hi | Yr lo' | Yi lo' (64 B each) + 32 pad; inside a part the 16-byte piece g
// sits at position g ^ (f >> 2).  With 288 = 18 x 16 the ds_read_b128 of stage 2 are conflict-free and the ds_write_b64
// of stage 1 two-way (scripts/proto/lds_layout_search.py; without the swizzle the stores were four-way: 16 LDS cycles each)
constexpr int kYRow = 288;
constexpr int kYSlot = 16 * kYRow;        // one k1 slot: 16 frames
// power rows, BIN-major: PT[bin][frame], 16 dwords per bin.  Stage 2 stores one bin of 16 frames per lane group
// (two-way at most: free for ds_write_b32; frame-major rows of 520 dwords were eight-way), the mel phase reads its B
// operands as ds_read_b32 (lanes (frame, kq) on bins b + kq: consecutive bins alternate between the bank halves)
constexpr int kPBins = 520;
constexpr int kHaloRow = 112;             // bytes per row of the f32 copy of the tile's last 768 samples [n2][24]
// LDS map (bytes).  Two signal images (hi and lo' planes each): the tile update for the next unit writes one while the
// slower waves' stage 1 still reads the other.
constexpr int kOffY = 0;
constexpr int kTBuf = 32 * kTRow;                           // one plane of one image
constexpr int kOffThi = kOffY + 16 * kYSlot;                // 73 728: hi planes of image 0, 1
constexpr int kOffTlo = kOffThi + 2 * kTBuf;                // lo' planes of image 0, 1
constexpr int kOffHalo = kOffTlo + 2 * kTBuf;               // two buffers
constexpr int kOffP = kOffHalo + 2 * 32 * kHaloRow;
constexpr int kOffWmax = kOffP + 16 * kPBins * 4;           // per wave (max, max of the last 768 samples) of the next tile's piece
constexpr int kOffWfresh = kOffWmax + 64;                   // the same of a tile read whole at a clip start
constexpr int kOffA2s = kOffWfresh + 64;                    // the k1 = 16 matrix (one wave reads it): [2 row blocks][hi, lo'][64] uint4
constexpr int kLdsBytes = kOffA2s + 2 * 2 * 64 * 16;
static_assert(kLdsBytes <= 160 * 1024 && kOffThi % 16 == 0 && kOffTlo % 16 == 0 && kOffHalo % 16 == 0 && kOffP % 16 == 0 &&
                  kOffWmax % 16 == 0 && kOffWfresh % 16 == 0 && kOffA2s % 16 == 0,
              "LDS map");
// fragment tables (uint4 = 8 f16 per lane): a1 [32 n2][2 row blocks][hi, lo'][64]; a2 [4 row blocks][hi, lo'][64];
// a2s (k1 = 16, its twiddle W64^n2 folded in) [2 row blocks][hi, lo'][64]
constexpr int kA1Vec = 32 * 2 * 2 * 64;
constexpr int kA2Vec = 4 * 2 * 64;
constexpr int kA2sVec = 2 * 2 * 64;
constexpr int kTabVec = kA1Vec + kA2Vec + kA2sVec;

struct Params {
    const void* pcm;
    float* out;               // mel rows [total_frames][n_mels]
    const uint4* tabs;        // a1 | a2 | a2s
    const float* wfrag;       // mel plan fragments [8][kMaxSlotsW][64][4] (f32; step i of a slot covers bins bin0 + 4 i + kq)
    const int* meta;          // [8][kMaxSlotsW]: first bin of every slot
    const v2::UnitDesc* units;
    Plan plan;
    int64_t total_units;
    int log_mel;
    unsigned long long* stamps;   // diagnostics only: [4 waves][8] summed phase cycles, or null
    int stamp_wave0;              // ... of waves stamp_wave0 .. stamp_wave0 + 3
};
// host: the three fragment tables for a window (values of the 1024-point frame, zero behind n_fft)
void build_tables(const double* window1024, uint4* tabs);
af_status launch(const Params& p, int channels, bool i16, bool separate, int grid, hipStream_t stream);
}  // namespace v3

struct Params {
    const void* pcm;          // interleaved PCM (f32 or i16)
    float* mel;               // [total_frames][n_mels]  (or [total_frames][513] power rows)
    const float* wintab;      // [32][kWinRow]: row n2 holds 0.5 * hann[32 n1 + n2], n1 = 0..31
    const float* twtab;       // [32][kTwRow]: row n2 holds (cos, sin)(2 pi k1 n2 / 1024), k1 = 0..31
    const float* wfrag;       // [4][kMaxSlotsW][64][4] MFMA A fragments: one float4 per lane and slot
    const int* meta;          // [4][kMaxSlotsW] first bin of every slot; its 4 steps are 4 bins apart
    Plan plan;                // by value: lives in the kernarg segment -> scalar (SMEM) loads
    const int64_t* pcm_off;   // [n_clips] first PCM frame of each clip (spans may overlap)
    const int64_t* pcm_len;   // [n_clips] PCM frames in each clip
    const int64_t* out_off;   // [n_clips+1] first output frame of each clip
    const int64_t* unit_off;  // [n_clips+1] first 16-frame work unit of each clip
    int64_t total_units;      // units this launch covers ...
    int64_t unit_begin;       // ... starting at this one (a sub-range = a chunk of the host-buffer pipeline)
    int n_clips;
    int hop;
    int aligned16;            // every clip base and every staging unit is 16-B aligned
    int out_power;            // 1: write |X|^2 rows [frame][513] instead of mel rows
    int log_mel;              // 1: mel rows hold log10(max(mel, 1e-10))
    int separate;             // 1 (stereo only): table entry e is channel e & 1 of clip e / 2
    unsigned long long* stamps;  // diagnostics only: [4 waves][8] summed phase cycles, or null
};

// launches the fused kernel; smem_bytes computed by the caller
af_status launch_features(const Params& p, int channels, bool i16, int grid, size_t smem_bytes,
                          hipStream_t stream);
size_t features_smem_bytes(int hop);

// stand-alone mel contraction: power rows [n_frames][513] -> mel rows [n_frames][n_mels]
struct MelApplyParams {
    const float* power;
    float* mel;
    const float* wfrag;
    const int* meta;
    Plan plan;
    int64_t n_frames;
    int log_mel;
};
af_status launch_mel_apply(const MelApplyParams& p, int grid, hipStream_t stream);

}  // namespace feat
}  // namespace af
