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
constexpr int kXStride = 33;              // exchange row stride (32 + 1)
constexpr int kHalfDw = 32 * kXStride;    // exchange region of one half-wave (1056 dwords)
constexpr int kWbuf = 2 * kHalfDw + 8;    // per-wave LDS buffer (2120 dwords; == 8 mod 32)
constexpr int kSlotDw = 16 * 17;          // one 16x16 partial tile, rows padded to 17
constexpr int kMaxSeg = 8;                // K-segments per wave in the mel phase (<= tiles)
constexpr int kMaxTiles = 8;              // n_mels <= 128
constexpr int kMaxSlots = kMaxTiles + kWaves - 1;

// Mel-phase plan: the banded filterbank as a list of 16x16x4 MFMA steps,
// split into contiguous ranges over the 4 waves (built on the host).
struct Plan {
    int n_mels, n_tiles, n_slots, n_steps;
    int wave_nseg[kWaves];
    int seg_slot[kWaves][kMaxSeg];
    int seg_bin0[kWaves][kMaxSeg];
    int seg_step0[kWaves][kMaxSeg];
    int seg_nsteps[kWaves][kMaxSeg];
    int tile_slot0[kMaxTiles];
    int tile_nslots[kMaxTiles];
};

struct Params {
    const void* pcm;          // interleaved PCM (f32 or i16)
    float* mel;               // [total_frames][n_mels]
    const float* wintab;      // [1024] 0.5 * periodic Hann
    const float2* twtab;      // [32][32] (cos, sin)(2 pi k1 n2 / 1024)
    const float* wfrag;       // [n_steps][64] MFMA A fragments of the filterbank
    const Plan* plan;
    const int64_t* pcm_off;   // [n_clips+1] first PCM frame of each clip
    const int64_t* out_off;   // [n_clips+1] first output frame of each clip
    const int64_t* unit_off;  // [n_clips+1] first 16-frame work unit of each clip
    int64_t total_units;
    int n_clips;
    int hop;
    int aligned16;            // every clip base and every staging unit is 16-B aligned
};

// launches the fused kernel; smem_bytes computed by the caller
af_status launch_features(const Params& p, int channels, bool i16, int grid, size_t smem_bytes,
                          hipStream_t stream);
size_t features_smem_bytes(int hop, int n_slots);

}  // namespace feat
}  // namespace af
