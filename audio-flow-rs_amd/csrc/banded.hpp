// banded.hpp -- the polyphase resampler and the FIR as one banded matrix product on MFMA.
//
// Both filters are  y[m] = sum_{k<T} proto[r(m) + k L] * x[q(m) + c - k],
//   q(m) = floor(m M / L), r(m) = m M mod L
// (docs/OPS.md "Polyphase resampler": c = T/2; "FIR": L = M = 1, c = 0, proto = taps).
//
// 16 consecutive outputs m = 16 b + i read one window of x that is only
// W = T + floor((r(16b) + 15 M) / L) samples wide, so they are a [16 x W] banded matrix
// A_p (p = b mod nb, nb = L / gcd(16, L) distinct matrices) times the window.  16 blocks of
// the same p, one "block period" (S = 16 M / gcd(16, L) input samples) apart, are the 16
// columns of one v_mfma_f32_16x16x4_f32 chain of NK = ceil(W / 4) steps.  A_p stays in the
// wave's registers for the whole launch, the windows come from LDS, one ds_read_b32 per MFMA.
#pragma once
#include <cstddef>
#include <cstdint>

#include "af_common.hpp"

namespace af {
namespace banded {

constexpr int kResamplerTaps = 64;   // taps per phase of the polyphase resampler (docs/OPS.md)

struct Plan {
    bool ok = false;          // false: shape not covered, the caller runs its VALU kernel
    uint32_t L = 1, M = 1;
    int T = 0;                // taps per phase
    int c = 0;                // x offset of tap 0
    int nb = 0;               // distinct block matrices
    int S = 0;                // input samples per block period
    int NK = 0;               // MFMA steps per chain (template instance)
    int wpp = 0;              // waves sharing one block matrix
    int nwaves = 0;           // nb * wpp
    int G = 0;                // block periods per column and tile
    int ch = 1;               // 2: x is interleaved stereo, downmixed while it is staged
    bool pairs = false;       // mono x staged in sample pairs: segments start with a lead-in word
    int seg_len = 0;          // floats of x per column segment
    int seg_stride = 0;       // odd, >= seg_len
    int64_t out_per_tile = 0; // 256 * G * nb
    size_t smem = 0;
    // Streamed form (ratios with more than 16 block matrices, e.g. 48 k -> 44.1 k: nb = 147): the matrices do
    // not stay in registers, a tile is a RANGE of P consecutive matrices times 16 C periods and every wave
    // fetches the fragments of its next matrix (L2-resident table) while it multiplies with the current one.
    bool streamed = false;
    int P = 0, C = 0;         // matrices per tile, column sets (of 16 periods) per tile
    int ob_off = 0, ob_stride = 0;   // streamed, deferred stores: LDS image of a tile's outputs (floats), 0: direct stores
    // fragments [nb][NK][64]; streamed plans keep q(16 p), p = 0..nb, as ints behind them
    size_t afrag_floats() const { return (size_t)nb * NK * 64 + (streamed ? (size_t)nb + 1 : 0); }
};

// Decide whether (L, M, T) runs on the MFMA kernel and with which tile shape.
Plan make_plan(uint32_t L, uint32_t M, int T, int c, int ch = 1);

// Clips the MFMA kernel can address (32-bit buffer offsets); longer ones go to the caller's
// thread-per-output kernel, which indexes with 64 bits.
bool fits(const Plan& p, int64_t x_len);

// Gather the A fragments [nb][NK][64] from the prototype (device, prototype order, L*T floats).
af_status build_afrag(const Plan& p, const float* proto_dev, float* afrag_dev, hipStream_t stream);

// Outputs m in [m0, m0 + n_out) of every clip; x[0] of a clip has absolute index x_abs0
// (x_abs0, x_len and x_stride count frames: with p.ch == 2 a frame is two floats),
// samples outside [x_abs0, x_abs0 + x_len) read as zero.  The tile grid is anchored at
// absolute output 0, so a sample's summation order does not depend on m0 / chunking.
af_status launch(const Plan& p, const float* afrag_dev, const float* x_dev, int64_t x_abs0,
                 int64_t x_len, int64_t x_stride, float* y_dev, int64_t y_stride, int64_t m0,
                 int64_t n_out, int64_t n_clips, hipStream_t stream, bool accumulate = false);

// A FIR as a sum of banded passes: K <= 177 taps is one pass; a longer filter is cut into parts of
// 176 taps, part j being the FIR of taps [176 j, 176 j + 176) applied to x delayed by 176 j samples;
// pass 0 writes y, the others accumulate into it (each output's summation order stays independent of
// how a stream was chunked: every pass anchors its tiles at absolute output 0).
constexpr int kFirPartTaps = 176;
constexpr int kFirMaxParts = 6;               // 1024 taps
struct FirParts {
    int n = 0;                                // 0: not on the MFMA path
    Plan plan[kFirMaxParts];
    size_t afrag_off[kFirMaxParts] = {};      // floats
    int tap0[kFirMaxParts] = {};
    size_t afrag_floats = 0;
    bool ok() const { return n > 0; }
};
FirParts make_fir_parts(int K);
bool fits(const FirParts& p, int64_t x_len);
af_status build_fir_afrag(const FirParts& p, const float* taps_dev, float* afrag_dev, hipStream_t stream);
af_status launch_fir(const FirParts& p, const float* afrag_dev, const float* x_dev, int64_t x_abs0, int64_t x_len,
                     int64_t x_stride, float* y_dev, int64_t y_stride, int64_t m0, int64_t n_out,
                     int64_t n_clips, hipStream_t stream);

// The resampler followed by a FIR (BASELINE config 4's middle stages) as ONE launch: the resampled
// signal only exists in LDS.  Covered: ratios with a single block matrix (L divides 16: 48 k / 32 k ->
// 16 k, 48 k -> 24 k ...), FIRs of at most 129 taps, mono / stereo, f32 / i16, uniform or ragged batches,
// whole clips or a window of a stream; everything else runs the two kernels above.
// Input formats of the fused launch (interleaved PCM; stereo is downmixed while it is staged, capture.rs:30-42).
enum FusedFmt { kFmtMonoF32 = 0, kFmtStereoF32 = 1, kFmtStereoI16 = 2, kFmtMonoI16 = 3 };
inline int fused_fmt(int channels, bool i16) { return channels == 2 ? (i16 ? kFmtStereoI16 : kFmtStereoF32) : (i16 ? kFmtMonoI16 : kFmtMonoF32); }
inline int fused_bytes_per_frame(int fmt) { return fmt == kFmtMonoF32 ? 4 : fmt == kFmtStereoF32 ? 8 : fmt == kFmtStereoI16 ? 4 : 2; }

struct FusedPlan {
    bool ok = false;
    Plan a;                    // the resampler's plan
    int fmt = kFmtMonoF32;     // how x is staged
    int K = 0;                 // FIR taps
    int NKB = 0;               // MFMA steps of the FIR chain (20 or 36)
    int H2 = 0, pad = 0;       // resampled samples recomputed per tile (multiple of 16), H2 - (K - 1)
    int P2 = 0;                // FIR block periods per column and tile
    int seg_len_b = 0, seg_stride_b = 0;
    int shift = 0, seg_stride_a = 0;   // x segments in LDS (pair formats: one frame early when the origin is odd)
    int64_t z_per_tile = 0;    // 256 * P2
    size_t smem = 0;
    // ten block matrices (44.1 k -> 16 k; resfir10_kernel): a tile is 5 120 resampled samples AND 5 120 FIR outputs, the
    // K - 1 samples in front of it come from the previous tile through LDS (H2 = 0: nothing is recomputed inside a
    // workgroup's contiguous range of tiles); a range that starts inside a clip runs phase A of `lead_tiles` = 1 tile
    // in front of it first, so a window of a stream must have that tile's input in its buffer
    int nbk = 1;               // block matrices of the resampler: 1 (resfir_kernel) or 10 (resfir10_kernel)
    int lead_tiles = 0;
    int carry_off = 0;         // floats: the two carry buffers behind the segments
    size_t afrag_b_floats() const { return (size_t)NKB * 64; }
};
FusedPlan make_fused_plan(const Plan& resampler, int K, int fmt);
// x index (frames from sample 0 of the stream) of the first sample tile t of the fused launch stages
int64_t fused_tile_base(const FusedPlan& f, int64_t t);
af_status build_fused_fir_afrag(const FusedPlan& f, const float* taps_dev, float* afrag_dev, hipStream_t stream);

// One tile of a ragged batch (32 bytes, one scalar load): which clip, which tile of it.
struct FusedTile {
    int64_t x_off;             // first frame of the clip, in frames from the batch's first frame
    int64_t z_off;             // first FIR output of the clip, in floats from z
    int32_t x_len;             // frames of the clip
    int32_t n_res;             // resampled samples (= FIR outputs) of the clip
    int32_t t;                 // tile of the clip (tiles are anchored at output 0)
    int32_t pad;
};
struct FusedArgs {
    const void* x = nullptr;   // device; uniform: clip i starts x_stride frames behind clip i - 1
    float* z = nullptr;        // device; uniform: row i = outputs [m0, m0 + n_out) of clip i, z_stride floats apart
    int64_t x_abs0 = 0;        // absolute stream index of x[0] (streaming; even in the pair formats); 0 for whole clips
    int64_t x_len = 0, x_stride = 0, z_stride = 0;
    int64_t n_res = 0;         // resampled samples that exist, [0, n_res): zeros before, nothing behind
    int64_t m0 = 0, n_out = 0; // FIR outputs wanted per clip (absolute)
    int64_t n_clips = 0;
    const FusedTile* tiles = nullptr;   // device; ragged batch (then x_len .. n_clips above are ignored)
    int64_t n_tiles = 0;
    int wg_per_cu = 2;         // 1: leave half of every CU to another kernel (the chain's pipelined form)
};
// z[clip][m - m0] = FIR(resample(to_mono(x[clip])))[m]
af_status launch_fused(const FusedPlan& f, const float* afrag_a, const float* afrag_b, const FusedArgs& args,
                       hipStream_t stream);
// longest clip (frames) the fused kernel's 32-bit byte offsets reach
bool fits(const FusedPlan& f, int64_t x_len);

// ---- the reference's wire path as ONE launch -------------------------------------------------------------------
// interleaved PCM (f32 | i16, mono | stereo) -> to_mono (capture.rs:30-42) -> polyphase resampler (any ratio with a
// single block matrix: 48 k / 32 k -> 16 k, 48 -> 24 k ...; resampler.rs:60-62 names 48 -> 16 k) -> (x.clamp(-1, 1) * 32767) as
// i16, little-endian (websocket.rs:246-251) -> padded base64 (websocket.rs:254).  The 16 kHz f32 signal and the i16 signal
// exist in LDS only.  The banded resampler of resfir_kernel's phase A with 15 of its 16 block periods per column in use: a
// tile is 3 840 resampled samples = 640 base64 groups of 6 samples = 12 messages of 320 samples (the reference's send batch,
// specs/0002-design.md:1543-1546) -- no group and no message straddles two tiles, so no workgroup needs another's samples.
// Block alignment, k-slot mapping and chain order are the stand-alone resampler's: the samples are its bits.
struct WirePlan {
    bool ok = false;
    Plan a;                    // the resampler's plan (mono; nb = 1, eight waves, G = 16)
    int fmt = kFmtMonoF32;     // how x is staged (FusedFmt)
    int Gv = 0;                // block periods per column whose samples are used (15)
    int frame = 0;             // samples per message; 0: every clip is ONE base64 string
    int row_len = 0;           // characters per message: 4 ceil(2 frame / 3)
    int shift = 0, seg_stride_a = 0;
    int img_off = 0;           // floats from the start of LDS: the tile's samples as i16
    int fimg_off = 0;          // ... and as f32 (only the instances that also emit the messages' energy / peak use it)
    int64_t y_per_tile = 0;    // 256 Gv
    size_t smem = 0, smem_level = 0;   // without / with the f32 image
};
WirePlan make_wire_plan(const Plan& resampler, int fmt, int frame_samples);
// characters of one clip's text: frame == 0: af_base64_len(2 n_res); else (n_res / frame) messages of row_len (a trailing
// partial batch is never sent: specs/0002-design.md:1543-1546 `if self.batch_buffer.len() < 320 { return None; }`)
inline int64_t wire_text_len(const WirePlan& w, int64_t n_res)
{
    return w.frame ? (n_res / w.frame) * (int64_t)w.row_len : 4 * ((2 * n_res + 2) / 3);
}
// One tile of a ragged batch (64 bytes, two scalar loads).
struct WireTile {
    int64_t x_off;             // first frame of the clip, in frames from x
    int64_t text_off;          // first character of the clip's text, in bytes from text
    int64_t pcm_off;           // first sample of the clip's PCM16 row, in samples from pcm16
    int32_t x_len;             // frames of the clip
    int32_t n_res;             // resampled samples of the clip
    int32_t t;                 // tile of the clip
    uint32_t msg_lo, msg_hi;   // first message of the clip in the batch's energy / peak rows (64 bits)
    int32_t pad[5];
};
static_assert(sizeof(WireTile) == 64, "WireTile is read with two s_load_dwordx8");
struct WireArgs {
    const void* x = nullptr;   // device PCM; uniform: clip i starts x_stride frames behind clip i - 1
    char* text = nullptr;      // device; uniform: row i starts text_stride bytes behind row i - 1
    int16_t* pcm16 = nullptr;  // device, optional: the quantised samples beside the text (row i at pcm_stride samples)
    // device, optional (messages only): per message the mean-square energy of VoiceActivityDetector::calculate_energy
    // (vad.rs:157-168: the f32 left fold over the message's resampled samples, bit for bit) and max |x| (the `peak` of the
    // audio_level event, events/mod.rs:73-75); clip i's messages start at msg_stride * i
    float* energy = nullptr;
    float* peak = nullptr;
    int64_t msg_stride = 0;
    int64_t x_len = 0, x_stride = 0, text_stride = 0, pcm_stride = 0;
    int64_t n_res = 0;         // resampled samples per clip
    int64_t n_clips = 0;
    const WireTile* tiles = nullptr;    // device; ragged batch (then x_len .. n_clips are ignored)
    int64_t n_tiles = 0;
    int text_align = 16;       // largest of 16 / 8 / 4 / 1 that divides the text pointer and every row / message offset
};
af_status launch_wire(const WirePlan& w, const float* afrag_a, const WireArgs& args, hipStream_t stream);
bool fits(const WirePlan& w, int64_t x_len);

}  // namespace banded

// resample.hip: MFMA plan (ch = 1 | 2) and device fragments of the cached design for a rate pair.
af_status resampler_fragments(uint32_t in_hz, uint32_t out_hz, int ch, banded::Plan* plan, const float** afrag_dev);
}  // namespace af
