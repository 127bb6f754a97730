// chain.hip -- the full feature chain of BASELINE config 4 as one handle:
//   interleaved PCM (f32 | i16) -> to_mono [R capture.rs:30-42] -> polyphase resample [B]
//   -> FIR [B] -> Hann STFT -> |X|^2 -> mel [B]
// Two launches where the shapes allow it (banded::FusedPlan: downmix + resampler + FIR in one kernel,
// the 16 kHz signal before the FIR only exists in LDS; then the fused STFT / mel kernel), otherwise
// every stage is the corresponding device entry point of this library, enqueued back to back on the
// caller's stream.  All intermediate buffers are allocated when the batch is described, so
// af_chain_run_device() only launches kernels (it can be captured into a hipGraph and replayed per
// streaming step).  Batches are equal-length clips (af_chain_set_batch_uniform) or ragged ones
// (af_chain_set_batch: AudioFrame is variable-length, capture.rs:13-28; stage by stage they run as ONE
// strip with zero gaps, see strip_layout); PCM is f32 or the wire format's i16 (websocket.rs:246-251).
//
// af_chain_stream_* (BASELINE config 5 on the whole chain): one handle owns the input history the
// resampler, the FIR (K - 1 samples) and the STFT (n_fft - hop samples) need, and replays ONE captured
// hipGraph per step: H2D of the new frames, the fused resampler + FIR launch, the STFT / mel launch, the
// history move, D2H of the rows.  Steps are whole tiles of both kernels, tiles are anchored at absolute
// sample 0, so the rows equal one-shot af_chain_run on the whole stream bit for bit.  Reference analogue
// of the carry / flush: BatchResampler, resampler.rs:132-166; intended pipeline: specs/0002-design.md:1102-1125.
#include <chrono>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "af_common.hpp"
#include "banded.hpp"

using namespace af;

struct af_chain {
    af_chain_config cfg{};
    int fmt = banded::kFmtMonoF32;   // banded::FusedFmt of (channels, sample_format)
    size_t bpf = 4;                  // bytes per PCM frame
    af_features* feat = nullptr;
    af_fir* fir = nullptr;           // taps + MFMA fragments, built once
    banded::FusedPlan fused;         // ok: resampler + FIR are one launch
    const float* afrag_res = nullptr;
    DevBuf d_afrag_fir;
    int opt_unfused = -1;            // af_chain_set_option("unfused"); -1: the process-wide af_debug_set switch
    // the pipelined form of the two-launch chain (uniform batches): see run_pipelined
    int opt_pipeline = 0;            // af_chain_set_option("pipeline", chunks); 0: off (the default)
    hipStream_t s_a = nullptr, s_b = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<hipEvent_t> ev_chunk;
    ~af_chain()
    {
        for (hipEvent_t e : ev_chunk) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (s_a) (void)hipStreamDestroy(s_a);
        if (s_b) (void)hipStreamDestroy(s_b);
    }
    // the batch
    bool ragged = false;
    size_t n_clips = 0, n_in = 0, n_res = 0;         // uniform: frames per clip in / resampled
    std::vector<int64_t> off, len, res_len, z_off;   // ragged: per clip (frames in, frames in, resampled, floats)
    int64_t total_in = 0, total_z = 0;               // frames of PCM / floats of the 16 kHz buffers described
    size_t total_frames = 0;
    DevBuf d_tiles;                                  // ragged + fused: banded::FusedTile per tile
    int64_t n_tiles = 0;
    // ragged, stage by stage: the clips as ONE strip with zero gaps (see strip_layout below)
    struct StripGroup { int64_t p0, p_len, r0, r_len; };      // one launch of resampler / FIR each
    std::vector<int64_t> p_off;                      // mono f32 sample at which clip i starts in the strip
    std::vector<StripGroup> groups;
    int64_t total_p = 0, total_r = 0;
    bool strip_ready = false, strip_ok = true;
    DevBuf d_strip, d_clips;                         // the strip; device table {src frame, frames, strip sample, gap begin, gap end}
    DevBuf d_f32, d_mono, d_res, d_fir, d_pcm, d_mel;
};

namespace {

// Resampler and FIR as one launch when the shapes are covered (banded.hpp, FusedPlan).
af_status plan_fused(af_chain* c, const float* taps)
{
    banded::Plan rp;
    AF_TRY(resampler_fragments(c->cfg.input_rate, c->cfg.output_rate, 1, &rp, &c->afrag_res));
    c->fused = banded::make_fused_plan(rp, (int)c->cfg.n_fir_taps, c->fmt);
    if (!c->fused.ok) return AF_OK;
    DevBuf d_taps;
    AF_TRY(d_taps.reserve(c->cfg.n_fir_taps * sizeof(float)));
    AF_TRY(c->d_afrag_fir.reserve(c->fused.afrag_b_floats() * sizeof(float)));
    AF_HIP(hipMemcpy(d_taps.p, taps, c->cfg.n_fir_taps * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(banded::build_fused_fir_afrag(c->fused, d_taps.as<float>(), c->d_afrag_fir.as<float>(), nullptr));
    AF_HIP(hipStreamSynchronize(nullptr));
    return AF_OK;
}

bool want_unfused(const af_chain* c)
{
    return c->opt_unfused >= 0 ? c->opt_unfused != 0 : debug_knob(kKnobChainUnfused) != 0;
}

// the wire format's i16 (websocket.rs:246-251) back to f32 for the stage-by-stage path: x / 32768, the scale of
// the fused kernels' staging (docs/OPS.md "PCM formats")
__global__ void pcm16_to_f32_kernel(const int16_t* __restrict__ in, float* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i] * (1.0f / 32768.0f);
}

// ---- ragged batches, stage by stage: one strip ---------------------------------------------------------------
// The stand-alone resampler and FIR take equal-length clips.  A ragged batch is laid out as ONE signal instead: clip i
// starts at mono sample p_off[i], a multiple of the input span of one full round of block matrices (lcm(16, L) outputs
// = lcm(16, L) M / L input samples), and is followed by zeros that reach over the resampler's window and over K - 1
// resampled samples.  One launch per stage then computes every clip with exactly the zeros a clip edge has, and -- the
// tiles of the banded kernels being anchored at the strip's output 0, a multiple of the block round away from the
// clip's own -- with the bits of a launch on that clip alone.  The resampler's tail past a clip's end (what it makes of
// the clip's last samples beyond res_len) is zeroed before the FIR reads the strip.  Strips beyond the kernels' 32-bit
// offsets are cut into groups at clip starts.
struct ClipRow { int64_t src, len, dst, gap0, gap1; };            // frames, frames, strip sample, resampled gap [gap0, gap1)

template <int CH, bool I16>
__global__ void strip_gather_kernel(const void* __restrict__ pcm, float* __restrict__ strip, const ClipRow* __restrict__ rows)
{
    const ClipRow r = rows[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < r.len; j += stride) {
        const int64_t f = r.src + j;
        float v;
        // the arithmetic of the stand-alone stages: x / 32768 (docs/OPS.md "PCM formats"), then (L + R) * 0.5 (capture.rs:30-42)
        if constexpr (I16) {
            const int16_t* q = static_cast<const int16_t*>(pcm) + f * CH;
            const float a = (float)q[0] * (1.0f / 32768.0f);
            v = CH == 2 ? (a + (float)q[CH - 1] * (1.0f / 32768.0f)) * 0.5f : a;
        } else {
            const float* q = static_cast<const float*>(pcm) + f * CH;
            v = CH == 2 ? (q[0] + q[CH - 1]) * 0.5f : q[0];
        }
        strip[r.dst + j] = v;
    }
}

__global__ void strip_zero_gaps_kernel(float* __restrict__ res, const ClipRow* __restrict__ rows)
{
    const ClipRow r = rows[blockIdx.x];
    for (int64_t j = r.gap0 + threadIdx.x; j < r.gap1; j += blockDim.x) res[j] = 0.0f;
}

constexpr int64_t kStripGroupMax = (1ll << 30) - (1ll << 22);     // samples per launch (the banded kernels' 32-bit offsets)

af_status strip_layout(af_chain* c)
{
    const size_t n = c->n_clips;
    const bool resample = c->cfg.input_rate != c->cfg.output_rate;
    const uint32_t g = std::gcd(c->cfg.input_rate, c->cfg.output_rate);
    const int64_t L = c->cfg.output_rate / g, M = c->cfg.input_rate / g;
    const int64_t round_out = std::lcm<int64_t>(16, L);             // outputs of one round of block matrices
    const int64_t a_in = round_out / L * M;                         // ... and the input samples under them
    const int64_t K = (int64_t)c->cfg.n_fir_taps;
    // zeros behind a clip: the resampler's window (T taps each way is more than it reaches) + what K - 1 resampled
    // samples need + one more resampled sample for out_len's rounding
    const int64_t gap_in = (resample ? 2 * banded::kResamplerTaps : 0) + ((K > 0 ? K - 1 : 0) + 2) * M / L + M / L + 2;
    c->p_off.assign(n, 0);
    c->groups.clear();
    c->strip_ok = true;
    std::vector<ClipRow> rows(n);
    int64_t pos = 0, g0 = 0;
    for (size_t i = 0; i < n; ++i) {
        int64_t end = pos + c->len[i] + gap_in;
        end = (end + a_in - 1) / a_in * a_in;
        if (end - g0 > kStripGroupMax && pos > g0) {                // close the group in front of this clip
            c->groups.push_back({g0, pos - g0, g0 / M * L, (pos - g0) / M * L});
            g0 = pos;
        }
        if (end - g0 > kStripGroupMax) c->strip_ok = false;         // (a single clip of hours: the fused launch only)
        c->p_off[i] = pos;
        c->z_off[i] = pos / M * L;                                  // (pos is a multiple of a_in = round_out M / L)
        rows[i] = ClipRow{c->off[i], c->len[i], pos, c->z_off[i] + c->res_len[i], end / M * L};
        AF_REQUIRE(rows[i].gap1 - rows[i].gap0 >= (K > 0 ? K - 1 : 0), "strip layout: gap of %lld resampled samples behind "
                   "clip %zu, the FIR needs %lld", (long long)(rows[i].gap1 - rows[i].gap0), i, (long long)(K - 1));
        pos = end;
    }
    if (pos > g0) c->groups.push_back({g0, pos - g0, g0 / M * L, (pos - g0) / M * L});
    c->total_p = pos;
    c->total_r = pos / M * L;
    c->strip_ready = false;
    if (K) AF_TRY(c->d_fir.reserve((size_t)(c->total_r ? c->total_r : 1) * sizeof(float)));
    if (n) {
        AF_TRY(c->d_clips.reserve(n * sizeof(ClipRow)));
        AF_HIP(hipMemcpy(c->d_clips.p, rows.data(), n * sizeof(ClipRow), hipMemcpyHostToDevice));
    }
    return AF_OK;
}

// the strip itself and the resampled strip: at af_chain_set_batch where the stage-by-stage path is the one that runs,
// on first use where it is only the fallback of a batch that has the fused launch (its mono strip would double the
// memory of a batch that never needs it)
af_status strip_alloc(af_chain* c)
{
    if (c->strip_ready) return AF_OK;
    const bool resample = c->cfg.input_rate != c->cfg.output_rate;
    const size_t np = (size_t)(c->total_p ? c->total_p : 1);
    AF_TRY(c->d_strip.reserve(np * sizeof(float)));
    AF_HIP(hipMemset(c->d_strip.p, 0, np * sizeof(float)));         // the gaps are never written again
    if (resample) AF_TRY(c->d_res.reserve((size_t)(c->total_r ? c->total_r : 1) * sizeof(float)));
    c->strip_ready = true;
    return AF_OK;
}

// gather -> (resampler, zero its tails) -> FIR over the strip; *out = what the feature kernel reads
af_status strip_run(af_chain* c, const void* pcm_dev, hipStream_t s, const float** out)
{
    const bool resample = c->cfg.input_rate != c->cfg.output_rate;
    const bool i16 = c->cfg.sample_format == AF_PCM_I16;
    if (!c->strip_ok)
        return fail(AF_ERR_UNSUPPORTED, "af_chain: a clip of this ragged batch is beyond the stage-by-stage chain's 32-bit offsets");
    AF_TRY(strip_alloc(c));
    const ClipRow* rows = c->d_clips.as<ClipRow>();
    int64_t longest = 1;
    for (size_t i = 0; i < c->n_clips; ++i) longest = std::max(longest, c->len[i]);
    const unsigned gx = (unsigned)std::min<int64_t>((longest + 256 * 8 - 1) / (256 * 8), 4096);
    for (size_t y0 = 0; y0 < c->n_clips; y0 += 65535) {             // (grid.y is 16 bits)
        const dim3 grid(gx, (unsigned)std::min<size_t>(65535, c->n_clips - y0));
        float* strip = c->d_strip.as<float>();
        if (c->cfg.channels == 2 && i16) hipLaunchKernelGGL((strip_gather_kernel<2, true>), grid, dim3(256), 0, s, pcm_dev, strip, rows + y0);
        else if (c->cfg.channels == 2) hipLaunchKernelGGL((strip_gather_kernel<2, false>), grid, dim3(256), 0, s, pcm_dev, strip, rows + y0);
        else if (i16) hipLaunchKernelGGL((strip_gather_kernel<1, true>), grid, dim3(256), 0, s, pcm_dev, strip, rows + y0);
        else hipLaunchKernelGGL((strip_gather_kernel<1, false>), grid, dim3(256), 0, s, pcm_dev, strip, rows + y0);
        AF_HIP(hipGetLastError());
    }
    const float* cur = c->d_strip.as<float>();
    if (resample) {
        for (const auto& gr : c->groups)
            AF_TRY(af_resample_batch_device(c->cfg.input_rate, c->cfg.output_rate, cur + gr.p0, 1, (size_t)gr.p_len,
                                            c->d_res.as<float>() + gr.r0, s));
        if (c->cfg.n_fir_taps && c->n_clips) {
            hipLaunchKernelGGL(strip_zero_gaps_kernel, dim3((unsigned)c->n_clips), dim3(256), 0, s, c->d_res.as<float>(), rows);
            AF_HIP(hipGetLastError());
        }
        cur = c->d_res.as<float>();
    }
    if (c->cfg.n_fir_taps) {
        for (const auto& gr : c->groups)
            AF_TRY(af_fir_apply_batch_device(c->fir, cur + gr.r0, 1, (size_t)gr.r_len, c->d_fir.as<float>() + gr.r0, s));
        cur = c->d_fir.as<float>();
    }
    *out = cur;
    return AF_OK;
}

// ---- the two launches side by side (af_chain_set_option("pipeline", chunks); off by default) --------------------
// The fused resampler + FIR launch is at its memory side: with ONE workgroup per CU it is as fast as with two (0.77 ms
// either way on config 4's shard).  The feature kernel is at its arithmetic side and needs little memory traffic.  Each
// needs 128 registers per lane and 72-78 KB of LDS per workgroup, so one workgroup of EACH fits a CU together.  The batch
// is cut into chunks of clips; stream A runs the first launch chunk after chunk at one workgroup per CU, stream B follows
// one chunk behind with the feature kernel, also at one workgroup per CU, behind an event per chunk.  Same kernels, same
// data, same bits; the caller's stream forks into the two and joins them again (event record / wait: capturable).
// MEASURED (scripts/gpu_bench_chain_pipeline.py, kernel trace gpurun_out/r03/pipe_kt): the launches do run side by side,
// but a chunk of the first launch then takes 152 instead of 126 us and a chunk of the feature kernel 150-165 instead of
// 72 us -- sharing the SIMDs' issue slots and the matrix pipe costs what the overlap buys: 1.16-1.19 ms against 1.13 one
// after the other, and with the feature kernel's waves at a higher issue priority 1.21-1.24.  Kept as an option only.
int pipeline_chunks(const af_chain* c)
{
    // (the pipelined form runs the feature kernel clip range by clip range: one-launch feature configurations only --
    // filterbanks on the dense fallback take two launches over whole batches)
    if (c->ragged || c->opt_pipeline == 0 || !features_one_launch(c->feat)) return 0;
    if (c->fused.nbk != 1) return 0;                                // (the ten-matrix launch needs its CU to itself)
    if (c->opt_pipeline > 0) return (int)std::min<size_t>((size_t)c->opt_pipeline, c->n_clips);
    return 0;                                                       // (see MEASURED above)
}

af_status pipeline_resources(af_chain* c, int chunks)
{
    if (!c->s_a) AF_HIP(hipStreamCreateWithFlags(&c->s_a, hipStreamNonBlocking));
    if (!c->s_b) AF_HIP(hipStreamCreateWithFlags(&c->s_b, hipStreamNonBlocking));
    if (!c->ev_fork) AF_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    if (!c->ev_join) AF_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    while ((int)c->ev_chunk.size() < chunks) {
        hipEvent_t e = nullptr;
        AF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_chunk.push_back(e);
    }
    return AF_OK;
}

af_status run_pipelined(af_chain* c, const void* pcm_dev, float* mel_dev, hipStream_t stream, int chunks)
{
    AF_REQUIRE(c->s_a && c->s_b && (int)c->ev_chunk.size() >= chunks, "af_chain: pipeline resources missing (set_batch builds them)");
    AF_HIP(hipEventRecord(c->ev_fork, stream));
    AF_HIP(hipStreamWaitEvent(c->s_a, c->ev_fork, 0));
    AF_HIP(hipStreamWaitEvent(c->s_b, c->ev_fork, 0));
    const size_t n = c->n_clips;
    // The caller's stream has been forked: whatever happens below, it is joined again before this returns (work the
    // caller enqueues next must stay ordered behind both side streams, and a stream capture in progress must see the
    // fork closed) -- the first error is kept and returned after the join.
    af_status st = AF_OK;
    const auto chunk = [&](int k) -> af_status {
        const size_t c0 = n * (size_t)k / (size_t)chunks, c1 = n * (size_t)(k + 1) / (size_t)chunks;
        if (c1 == c0) return AF_OK;
        banded::FusedArgs q;
        q.x = static_cast<const char*>(pcm_dev) + c0 * c->n_in * c->bpf;
        q.z = c->d_fir.as<float>() + c0 * c->n_res;
        q.x_len = q.x_stride = (int64_t)c->n_in;
        q.z_stride = q.n_res = q.n_out = (int64_t)c->n_res;
        q.n_clips = (int64_t)(c1 - c0);
        q.wg_per_cu = 1;
        AF_TRY(banded::launch_fused(c->fused, c->afrag_res, c->d_afrag_fir.as<float>(), q, c->s_a));
        AF_HIP(hipEventRecord(c->ev_chunk[k], c->s_a));
        AF_HIP(hipStreamWaitEvent(c->s_b, c->ev_chunk[k], 0));
        return features_run_clips(c->feat, c->d_fir.as<float>(), mel_dev, c->s_b, c0, c1, 1);
    };
    for (int k = 0; k < chunks && st == AF_OK; ++k) st = chunk(k);
    // stream B waits for everything stream A has been given (after an error its last chunk event may not exist), the
    // caller's stream for stream B
    hipError_t je = hipEventRecord(c->ev_join, c->s_a);
    if (je == hipSuccess) je = hipStreamWaitEvent(c->s_b, c->ev_join, 0);
    if (je == hipSuccess) je = hipEventRecord(c->ev_join, c->s_b);
    if (je == hipSuccess) je = hipStreamWaitEvent(stream, c->ev_join, 0);
    if (st != AF_OK) return st;
    if (je != hipSuccess) return fail(AF_ERR_HIP, "af_chain: joining the pipeline streams failed: %s", hipGetErrorString(je));
    return AF_OK;
}

bool fused_applies(const af_chain* c)
{
    if (!c->fused.ok || want_unfused(c)) return false;
    // (mono i16 is staged in dwords from the clip base: clips of an odd frame count put every second base on an odd frame)
    if (!c->ragged && c->fmt == banded::kFmtMonoI16 && (c->n_in & 1) && c->n_clips > 1) return false;
    if (!c->ragged) return banded::fits(c->fused, (int64_t)c->n_in);
    return c->n_tiles > 0 || c->n_clips == 0;        // the tile table exists: every clip fits and is aligned
}

}  // namespace

extern "C" {

void af_chain_default_config(af_chain_config* c)
{
    if (!c) return;
    c->input_rate = 48000;
    c->channels = 2;
    c->output_rate = 16000;
    c->fir_taps = nullptr;
    c->n_fir_taps = 0;
    c->hop = 256;
    c->n_mels = 80;
    c->f_min = 0.0f;
    c->f_max = 0.0f;
    c->device = -1;
    c->sample_format = AF_PCM_F32;
}

af_status af_chain_create(const af_chain_config* cfg, af_chain** out)
{
    AF_REQUIRE(cfg && out, "af_chain_create: null argument");
    *out = nullptr;
    AF_REQUIRE(cfg->input_rate > 0 && cfg->output_rate > 0, "sample rates must be positive");
    AF_REQUIRE(cfg->channels == 1 || cfg->channels == 2, "channels must be 1 or 2 (got %u)", cfg->channels);
    AF_REQUIRE(cfg->sample_format == AF_PCM_F32 || cfg->sample_format == AF_PCM_I16,
               "sample_format must be AF_PCM_F32 or AF_PCM_I16 (got %u)", cfg->sample_format);
    AF_REQUIRE(cfg->n_fir_taps == 0 || cfg->fir_taps, "fir_taps is null");
    AF_REQUIRE(cfg->n_fir_taps <= 1024, "FIR needs at most 1024 taps (got %zu)", cfg->n_fir_taps);
    AF_TRY(ensure_device(cfg->device));
    af_chain* c = new (std::nothrow) af_chain();
    if (!c) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    c->cfg = *cfg;
    c->cfg.fir_taps = nullptr;            // the taps are copied to the device below
    c->fmt = banded::fused_fmt((int)cfg->channels, cfg->sample_format == AF_PCM_I16);
    c->bpf = (size_t)banded::fused_bytes_per_frame(c->fmt);
    af_features_config fc;
    af_features_default_config(&fc);
    fc.sample_rate = cfg->output_rate;
    fc.channels = 1;
    fc.hop = cfg->hop;
    fc.n_mels = cfg->n_mels;
    fc.f_min = cfg->f_min;
    fc.f_max = cfg->f_max;
    fc.device = -1;
    af_status st = af_features_create(&fc, &c->feat);
    // everything run_device needs is built here, so that it only launches kernels
    if (st == AF_OK && cfg->n_fir_taps) st = af_fir_create(cfg->fir_taps, cfg->n_fir_taps, &c->fir);
    if (st == AF_OK) st = af_resample_prepare(cfg->input_rate, cfg->output_rate);
    if (st == AF_OK && cfg->n_fir_taps && cfg->input_rate != cfg->output_rate) st = plan_fused(c, cfg->fir_taps);
    if (st != AF_OK) { af_chain_destroy(c); return st; }
    *out = c;
    return AF_OK;
}

void af_chain_destroy(af_chain* c)
{
    if (!c) return;
    af_features_destroy(c->feat);
    af_fir_destroy(c->fir);
    delete c;
}

af_status af_chain_set_option(af_chain* c, const char* key, int value)
{
    AF_REQUIRE(c && key, "af_chain_set_option: null argument");
    if (std::strcmp(key, "unfused") == 0) {
        c->opt_unfused = value < 0 ? -1 : value != 0;
        return AF_OK;
    }
    if (std::strcmp(key, "pipeline") == 0) {             // chunks of the pipelined two-launch form; <= 0: off
        AF_REQUIRE(value <= 256, "af_chain_set_option: at most 256 pipeline chunks");
        c->opt_pipeline = value < 0 ? 0 : value;
        if (value > 0) AF_TRY(pipeline_resources(c, value));
        return AF_OK;
    }
    if (std::strcmp(key, "n_fft") == 0) {
        // the frame length of the STFT / mel stage (the config struct has no field for it: 1024 there); e.g. 400 with hop 160 for
        // 25 ms / 10 ms speech frames behind a 48 -> 16 kHz chain.  The stage's handle is rebuilt and the batch forgotten:
        // af_chain_set_batch* must follow.
        af_features_config fc;
        af_features_default_config(&fc);
        fc.sample_rate = c->cfg.output_rate;
        fc.channels = 1;
        fc.n_fft = (uint32_t)value;
        fc.hop = c->cfg.hop;
        fc.n_mels = c->cfg.n_mels;
        fc.f_min = c->cfg.f_min;
        fc.f_max = c->cfg.f_max;
        fc.device = -1;
        af_features* nf = nullptr;
        AF_TRY(af_features_create(&fc, &nf));
        af_features_destroy(c->feat);
        c->feat = nf;
        c->n_clips = 0;
        c->total_in = 0;
        c->total_frames = 0;
        return AF_OK;
    }
    // the STFT / mel stage is a feature handle of its own: its kernel choice follows af_debug_set("features.kernel") unless
    // the chain pins it here (tests/test_gpu_concurrency.py flips the process-wide switch under a running chain)
    if (std::strcmp(key, "kernel") == 0 || std::strcmp(key, "wg_per_cu") == 0 || std::strcmp(key, "mel4") == 0) return af_features_set_option(c->feat, key, value);
    return fail(AF_ERR_INVALID_ARG, "af_chain_set_option: unknown option '%s'", key);
}

size_t af_chain_n_frames(const af_chain* c, size_t n_pcm_frames)
{
    if (!c) return 0;
    const size_t n_res = af_resample_out_len(c->cfg.input_rate, c->cfg.output_rate, n_pcm_frames);
    return af_features_n_frames(c->feat, n_res);
}

af_status af_chain_set_batch_uniform(af_chain* c, size_t n_clips, size_t frames_per_clip,
                                     size_t* total_frames_out)
{
    AF_REQUIRE(c, "af_chain_set_batch_uniform: null handle");
    c->ragged = false;
    c->n_clips = n_clips;
    c->n_in = frames_per_clip;
    c->n_res = af_resample_out_len(c->cfg.input_rate, c->cfg.output_rate, frames_per_clip);
    c->total_in = (int64_t)(n_clips * frames_per_clip);
    c->total_z = (int64_t)(n_clips * c->n_res);
    c->n_tiles = 0;
    const bool resample = c->cfg.input_rate != c->cfg.output_rate;
    if (c->cfg.sample_format == AF_PCM_I16) AF_TRY(c->d_f32.reserve((size_t)c->total_in * c->cfg.channels * sizeof(float)));
    if (c->cfg.channels == 2) AF_TRY(c->d_mono.reserve((size_t)c->total_in * sizeof(float)));
    if (resample) AF_TRY(c->d_res.reserve((size_t)c->total_z * sizeof(float)));
    if (c->cfg.n_fir_taps) AF_TRY(c->d_fir.reserve((size_t)c->total_z * sizeof(float)));
    size_t total = 0;
    AF_TRY(af_features_set_batch_uniform(c->feat, n_clips, c->n_res, &total));
    if (const int chunks = pipeline_chunks(c)) AF_TRY(pipeline_resources(c, chunks));
    c->total_frames = total;
    if (total_frames_out) *total_frames_out = total;
    return AF_OK;
}

af_status af_chain_set_batch(af_chain* c, const int64_t* clip_offsets, size_t n_clips, size_t* total_frames_out)
{
    AF_REQUIRE(c, "af_chain_set_batch: null handle");
    AF_REQUIRE(clip_offsets || n_clips == 0, "af_chain_set_batch: null offsets");
    c->ragged = true;
    c->n_clips = n_clips;
    c->n_in = c->n_res = 0;
    c->off.assign(n_clips, 0);
    c->len.assign(n_clips, 0);
    c->res_len.assign(n_clips, 0);
    c->z_off.assign(n_clips, 0);
    bool fused_ok = c->fused.ok;
    for (size_t i = 0; i < n_clips; ++i) {
        const int64_t b = clip_offsets[i], e = clip_offsets[i + 1];
        AF_REQUIRE(b >= 0 && e >= b, "af_chain_set_batch: offsets must be monotone (clip %zu: %lld .. %lld)", i,
                   (long long)b, (long long)e);
        c->off[i] = b;
        c->len[i] = e - b;
        c->res_len[i] = (int64_t)af_resample_out_len(c->cfg.input_rate, c->cfg.output_rate, (size_t)(e - b));
        // the fused launch: 32-bit offsets inside a clip; mono i16 frames are 2 bytes, its loads are dwords
        if (!banded::fits(c->fused, e - b) || c->res_len[i] >= (1ll << 31)) fused_ok = false;
        if (c->fmt == banded::kFmtMonoI16 && (b & 1)) fused_ok = false;
    }
    c->total_in = n_clips ? clip_offsets[n_clips] : 0;
    // where the clips' 16 kHz signals live: the strip layout (shared by the fused launch and the stage-by-stage path, so
    // that the feature kernel's spans do not depend on which of them runs)
    AF_TRY(strip_layout(c));
    c->total_z = c->total_r;
    // one record per tile of the fused launch
    c->n_tiles = 0;
    if (fused_ok && n_clips) {
        std::vector<banded::FusedTile> tiles;
        for (size_t i = 0; i < n_clips; ++i) {
            const int64_t nt = (c->res_len[i] + c->fused.z_per_tile - 1) / c->fused.z_per_tile;
            for (int64_t t = 0; t < nt; ++t)
                tiles.push_back(banded::FusedTile{c->off[i], c->z_off[i], (int32_t)c->len[i], (int32_t)c->res_len[i],
                                                  (int32_t)t, 0});
        }
        if (tiles.size() >= ((size_t)1 << 31)) return fail(AF_ERR_UNSUPPORTED, "af_chain_set_batch: too many tiles");
        if (!tiles.empty()) {
            AF_TRY(c->d_tiles.reserve(tiles.size() * sizeof(banded::FusedTile)));
            AF_HIP(hipMemcpy(c->d_tiles.p, tiles.data(), tiles.size() * sizeof(banded::FusedTile), hipMemcpyHostToDevice));
        }
        c->n_tiles = (int64_t)tiles.size();
    }
    if (!fused_applies(c) && c->strip_ok) AF_TRY(strip_alloc(c));
    size_t total = 0;
    AF_TRY(af_features_set_batch_spans(c->feat, c->z_off.data(), c->res_len.data(), n_clips, &total));
    c->total_frames = total;
    if (total_frames_out) *total_frames_out = total;
    return AF_OK;
}

af_status af_chain_run_device(af_chain* c, const void* pcm_dev, float* mel_dev, void* stream)
{
    AF_REQUIRE(c, "af_chain_run_device: null handle");
    if (c->n_clips == 0 || c->total_in == 0) return AF_OK;
    AF_REQUIRE(pcm_dev && (mel_dev || c->total_frames == 0), "af_chain_run_device: null buffer");
    const bool resample = c->cfg.input_rate != c->cfg.output_rate;
    const bool i16 = c->cfg.sample_format == AF_PCM_I16;
    if (fused_applies(c)) {
        // mono i16 is staged as one dword per frame pair, and the buffer range of a clip is rounded up to whole dwords (an
        // odd frame count: banded_mfma.hip, resfir_kernel): from a base that is only 2-byte aligned the last dword would
        // reach 2 bytes into the NEXT aligned dword behind the caller's buffer.  Such a pointer (a slice at an odd
        // element) is refused rather than read from.
        if (c->fmt == banded::kFmtMonoI16 && reinterpret_cast<uintptr_t>(pcm_dev) % 4 != 0)
            return fail(AF_ERR_INVALID_ARG, "af_chain_run_device: mono i16 PCM must start on a 4-byte boundary (got %p)", pcm_dev);
        if (const int chunks = pipeline_chunks(c)) return run_pipelined(c, pcm_dev, mel_dev, as_stream(stream), chunks);
        banded::FusedArgs q;
        q.x = pcm_dev;
        q.z = c->d_fir.as<float>();
        if (c->ragged) {
            q.tiles = c->d_tiles.as<banded::FusedTile>();
            q.n_tiles = c->n_tiles;
        } else {
            q.x_len = q.x_stride = (int64_t)c->n_in;
            q.z_stride = q.n_res = q.n_out = (int64_t)c->n_res;
            q.n_clips = (int64_t)c->n_clips;
        }
        AF_TRY(banded::launch_fused(c->fused, c->afrag_res, c->d_afrag_fir.as<float>(), q, as_stream(stream)));
        return af_features_run_device(c->feat, c->d_fir.as<float>(), mel_dev, stream);
    }
    // ---- stage by stage ----
    if (c->ragged) {
        const float* z = nullptr;
        AF_TRY(strip_run(c, pcm_dev, as_stream(stream), &z));
        return af_features_run_device(c->feat, z, mel_dev, stream);
    }
    // equal-length clips: every stage is the corresponding batch entry point
    const float* cur = static_cast<const float*>(pcm_dev);
    if (i16) {
        const size_t n = (size_t)c->total_in * c->cfg.channels;
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3(grid), dim3(256), 0, as_stream(stream),
                           static_cast<const int16_t*>(pcm_dev), c->d_f32.as<float>(), n);
        AF_HIP(hipGetLastError());
        cur = c->d_f32.as<float>();
    }
    if (c->cfg.channels == 2 && resample && af_resample_has_stereo_path(c->cfg.input_rate, c->cfg.output_rate) &&
        2 * (int64_t)c->n_in < ((1ll << 30) - (1ll << 20))) {
        // downmix fused into the resampler's loads: the mono signal never reaches HBM
        AF_TRY(af_resample_batch_stereo_device(c->cfg.input_rate, c->cfg.output_rate, cur, c->n_clips,
                                               c->n_in, c->d_res.as<float>(), stream));
        cur = c->d_res.as<float>();
    } else {
        if (c->cfg.channels == 2) {
            // (frames are consecutive across clip boundaries: one launch for the whole buffer)
            AF_TRY(af_to_mono_device(cur, (size_t)c->total_in * 2, 2, c->d_mono.as<float>(), stream));
            cur = c->d_mono.as<float>();
        }
        if (resample) {
            AF_TRY(af_resample_batch_device(c->cfg.input_rate, c->cfg.output_rate, cur, c->n_clips, c->n_in,
                                            c->d_res.as<float>(), stream));
            cur = c->d_res.as<float>();
        }
    }
    if (c->cfg.n_fir_taps) {
        AF_TRY(af_fir_apply_batch_device(c->fir, cur, c->n_clips, c->n_res, c->d_fir.as<float>(), stream));
        cur = c->d_fir.as<float>();
    }
    return af_features_run_device(c->feat, cur, mel_dev, stream);
}

int af_chain_n_launches(const af_chain* c)
{
    if (!c) return 0;
    const bool resample = c->cfg.input_rate != c->cfg.output_rate;
    if (fused_applies(c)) return pipeline_chunks(c) ? 2 * pipeline_chunks(c) : 2;
    int fir_parts = 0;
    if (c->cfg.n_fir_taps) {
        const banded::FirParts fp = banded::make_fir_parts((int)c->cfg.n_fir_taps);
        fir_parts = fp.ok() ? fp.n : 1;
    }
    if (c->ragged) {
        // the strip: gather, then one resampler / FIR launch per group (one group unless the batch is beyond 2^30 samples)
        const int groups = (int)std::max<size_t>(1, c->groups.size());
        int n = 1 + 1 + (int)((c->n_clips + 65534) / 65535) - 1;    // features + gather
        if (resample) n += groups + (c->cfg.n_fir_taps ? 1 : 0);    // + the zeroing of the resampler's tails
        return n + fir_parts * groups;
    }
    int n = 1 + fir_parts;                                      // the STFT / mel kernel, the FIR passes
    if (c->cfg.sample_format == AF_PCM_I16) n += 1;             // i16 -> f32
    if (resample) n += 1;
    if (c->cfg.channels == 2 && !(resample && af_resample_has_stereo_path(c->cfg.input_rate, c->cfg.output_rate)))
        n += 1;                                                 // stand-alone to_mono
    return n;
}

int af_chain_fused_plan_describe(uint32_t input_rate, uint32_t output_rate, uint32_t channels, size_t n_fir_taps,
                                 int32_t* info)
{
    if (!info || !input_rate || !output_rate || (channels != 1 && channels != 2)) return 0;
    for (int i = 0; i < 10; ++i) info[i] = 0;
    if (input_rate == output_rate || n_fir_taps == 0 || n_fir_taps > 1024) return 0;
    const uint32_t g = std::gcd(input_rate, output_rate);
    const banded::Plan rp = banded::make_plan(output_rate / g, input_rate / g, banded::kResamplerTaps,
                                              banded::kResamplerTaps / 2, 1);
    const banded::FusedPlan f = banded::make_fused_plan(rp, (int)n_fir_taps, banded::fused_fmt((int)channels, false));
    const int32_t v[10] = {f.ok ? 1 : 0, f.a.NK, f.NKB, f.H2, f.pad, f.P2, f.seg_len_b, f.seg_stride_b,
                           (int32_t)f.z_per_tile, (int32_t)f.smem};
    if (f.ok)
        for (int i = 0; i < 10; ++i) info[i] = v[i];
    return f.ok ? 1 : 0;
}

af_status af_chain_run(af_chain* c, const void* pcm, size_t pcm_len, float* mel, size_t mel_cap)
{
    AF_REQUIRE(c, "af_chain_run: null handle");
    const size_t need = (size_t)c->total_in * c->cfg.channels;
    if (pcm_len < need)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "pcm buffer holds %zu samples, the batch describes %zu", pcm_len, need);
    const size_t n_out = c->total_frames * c->cfg.n_mels;
    if (mel_cap < n_out) return fail(AF_ERR_BUFFER_TOO_SMALL, "mel buffer holds %zu floats, need %zu", mel_cap, n_out);
    if (c->n_clips == 0 || c->total_in == 0) return AF_OK;
    AF_REQUIRE(pcm && (mel || n_out == 0), "af_chain_run: null buffer");
    const size_t in_bytes = (size_t)c->total_in * c->bpf;
    AF_TRY(c->d_pcm.reserve(in_bytes));
    AF_TRY(c->d_mel.reserve((n_out ? n_out : 1) * sizeof(float)));
    AF_HIP(hipMemcpy(c->d_pcm.p, pcm, in_bytes, hipMemcpyHostToDevice));
    AF_TRY(af_chain_run_device(c, c->d_pcm.p, c->d_mel.as<float>(), nullptr));
    if (n_out) AF_HIP(hipMemcpy(mel, c->d_mel.p, n_out * sizeof(float), hipMemcpyDeviceToHost));
    else AF_HIP(hipDeviceSynchronize());
    return AF_OK;
}

}  // extern "C"

// ================================================================================================
// af_chain_stream_*
//
// Notation (nb = 1 resamplers only: q(16 b) = b S): resampled sample m reads x[q(m) + lo_off .. q(m) + c], FIR
// output n reads resampled samples (n - K, n], STFT frame f reads FIR outputs [f hop, f hop + n_fft).  A step is
// Cz FIR outputs, a multiple of the fused launch's tile (z_per_tile) and of the feature kernel's unit (16 hop): step
// k emits the Cz / hop frames that start in [k Cz, (k + 1) Cz) and needs FIR outputs [k Cz, (k + 1) Cz + n_fft - hop).
// The tiles that cover them stage x from base(t0) = (t0 16 P2 - H2 / 16) S + lo_off - shift on, t0 = k Cz / z_per_tile:
// the device buffer holds x[base(t0_k), Xneed_k), Xneed_k = q((k + 1) Cz + n_fft - hop - 1) + c + 1, and every step
// moves both ends by Cx = Cz M / L frames: the last Lx - Cx frames are carried to the front, Cx new ones follow.
// Step 0 is run eagerly with the stream's true origin (zeros in front of sample 0: the clip edge of one-shot
// processing); every later step looks the same relative to its first tile, so it is launched with FIXED arguments
// ("the stream starts one tile before this step") and captured once.
struct af_chain_stream {
    af_chain* chain = nullptr;        // plans, fragments (its feature handle serves the replayed step)
    af_features* tail = nullptr;      // the final, shorter span (flush)
    int64_t Cz = 0, Cx = 0, H = 0;    // FIR outputs / input frames per step; n_fft - hop
    int64_t need0 = 0;                // input frames step 0 needs: Xneed_0
    int64_t base1 = 0;                // base(1): x index of the first staged sample of tile 1 (even)
    int64_t carry = 0;                // frames carried between steps: need0 - base(t0_1)
    int64_t Lx = 0;                   // frames of a steady step's buffer: carry + Cx
    int64_t z_cap = 0;                // floats of d_z
    size_t bpf = 0, rows_per_step = 0, row_len = 0;
    void* h_in = nullptr;             // page-locked: max(need0, Cx) frames
    float* h_out = nullptr;           // page-locked: rows of one step
    DevBuf d_x, d_z, d_rows;
    int64_t fill = 0;                 // frames waiting in h_in
    uint64_t steps = 0, frames_in = 0, rows_out = 0;
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_device_ms = 0.0f, last_wall_ms = 0.0f;
    bool flushed = false;
    bool priming = false;             // after a seek: the next `carry` frames are history only
};

namespace {

constexpr int64_t kEndless = (int64_t)1 << 60;

int64_t q_of(const af_chain* c, int64_t m) { return m * (int64_t)c->fused.a.M / (int64_t)c->fused.a.L; }

// first x frame a launch whose first tile is t reads: that tile's staging, or -- the ten-matrix launch, whose tiles take the
// K - 1 samples in front of them from the tile before -- the staging of the warm pass in front of it (lead_tiles = 1)
int64_t tile_base(const af_chain* c, int64_t t)
{
    return banded::fused_tile_base(c->fused, t - c->fused.lead_tiles);
}

void cs_free(af_chain_stream* s)
{
    if (!s) return;
    if (s->exec) (void)hipGraphExecDestroy(s->exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_out) (void)hipHostFree(s->h_out);
    af_features_destroy(s->tail);
    af_chain_destroy(s->chain);
    delete s;
}

// kernels of one step on s->stream: FIR outputs [m0, m0 + n_out) of the stream whose frame x_abs0 is d_x[0], then
// the spectrogram rows of `feat`'s span over them
af_status cs_kernels(af_chain_stream* s, af_features* feat, int64_t x_abs0, int64_t x_len, int64_t n_res, int64_t m0,
                     int64_t n_out)
{
    af_chain* c = s->chain;
    banded::FusedArgs q;
    q.x = s->d_x.p;
    q.z = s->d_z.as<float>();
    q.x_abs0 = x_abs0;
    q.x_len = q.x_stride = x_len;
    q.z_stride = n_out;
    q.n_res = n_res;
    q.m0 = m0;
    q.n_out = n_out;
    q.n_clips = 1;
    AF_TRY(banded::launch_fused(c->fused, c->afrag_res, c->d_afrag_fir.as<float>(), q, s->stream));
    return af_features_run_device(feat, s->d_z.as<float>(), s->d_rows.as<float>(), s->stream);
}

// the replayed step: new frames in, both kernels with the fixed "one tile in" arguments, history move, rows out
af_status cs_enqueue_step(af_chain_stream* s)
{
    char* x = s->d_x.as<char>();
    AF_HIP(hipMemcpyAsync(x + s->carry * s->bpf, s->h_in, (size_t)s->Cx * s->bpf, hipMemcpyHostToDevice, s->stream));
    // ("the stream starts 1 + lead tiles before this step": the warm pass of the ten-matrix launch included)
    AF_TRY(cs_kernels(s, s->chain->feat, s->base1, s->Lx, kEndless,
                      (1 + s->chain->fused.lead_tiles) * s->chain->fused.z_per_tile, s->Cz + s->H));
    // the last `carry` frames are the next step's history (carry <= Cx: the regions do not overlap)
    AF_HIP(hipMemcpyAsync(x, x + (size_t)s->Cx * s->bpf, (size_t)s->carry * s->bpf, hipMemcpyDeviceToDevice, s->stream));
    AF_HIP(hipMemcpyAsync(s->h_out, s->d_rows.p, s->rows_per_step * s->row_len * sizeof(float), hipMemcpyDeviceToHost,
                          s->stream));
    return AF_OK;
}

af_status cs_timed(af_chain_stream* s, bool replay)
{
    const auto t0 = std::chrono::steady_clock::now();
    AF_HIP(hipEventRecord(s->ev0, s->stream));
    if (replay) AF_HIP(hipGraphLaunch(s->exec, s->stream));
    else AF_TRY(cs_enqueue_step(s));
    AF_HIP(hipEventRecord(s->ev1, s->stream));
    AF_HIP(hipStreamSynchronize(s->stream));
    AF_HIP(hipEventElapsedTime(&s->last_device_ms, s->ev0, s->ev1));
    s->last_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return AF_OK;
}

// step 0: the stream's first need0 frames, true origin
af_status cs_first_step(af_chain_stream* s)
{
    char* x = s->d_x.as<char>();
    AF_HIP(hipMemcpyAsync(x, s->h_in, (size_t)s->need0 * s->bpf, hipMemcpyHostToDevice, s->stream));
    AF_TRY(cs_kernels(s, s->chain->feat, 0, s->need0, kEndless, 0, s->Cz + s->H));
    // history of step 1: frames [base(t0_1), need0) to the front
    const int64_t from = s->need0 - s->carry;
    AF_HIP(hipMemcpyAsync(s->d_z.p, x + (size_t)from * s->bpf, (size_t)s->carry * s->bpf, hipMemcpyDeviceToDevice,
                          s->stream));       // (through d_z's head: source and destination may overlap in d_x)
    AF_HIP(hipMemcpyAsync(s->h_out, s->d_rows.p, s->rows_per_step * s->row_len * sizeof(float), hipMemcpyDeviceToHost,
                          s->stream));
    AF_HIP(hipMemcpyAsync(x, s->d_z.p, (size_t)s->carry * s->bpf, hipMemcpyDeviceToDevice, s->stream));
    AF_HIP(hipStreamSynchronize(s->stream));
    return AF_OK;
}

}  // namespace

extern "C" {

af_status af_chain_stream_create(const af_chain_config* cfg, size_t step_frames, af_chain_stream** out)
{
    AF_REQUIRE(cfg && out, "af_chain_stream_create: null argument");
    *out = nullptr;
    af_chain_stream* s = new (std::nothrow) af_chain_stream();
    if (!s) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    af_status st = af_chain_create(cfg, &s->chain);
    if (st != AF_OK) { cs_free(s); return st; }
    af_chain* c = s->chain;
    const banded::FusedPlan& f = c->fused;
    if (!f.ok) {
        cs_free(s);
        return fail(AF_ERR_UNSUPPORTED,
                    "af_chain_stream: %u -> %u Hz with %zu FIR taps has no fused resampler + FIR launch (needs a ratio "
                    "with one block matrix, e.g. 48 / 32 k -> 16 k, or 44.1 k -> 16 k, and 1..129 taps); stream the stages with "
                    "af_resampler_*, af_fir_* and af_stream_* instead", cfg->input_rate, cfg->output_rate, cfg->n_fir_taps);
    }
    const int64_t hop = cfg->hop, n_fft = 1024;
    s->H = n_fft - hop;
    // a step = whole tiles of both kernels and a whole number of input frames
    int64_t quantum = std::lcm<int64_t>(f.z_per_tile, 16 * hop);
    quantum = std::lcm<int64_t>(quantum, (int64_t)f.a.L);
    const int64_t qx = quantum * f.a.M / f.a.L;                     // input frames per quantum
    int64_t nq = ((int64_t)step_frames + qx / 2) / qx;
    if (nq < 1) nq = 1;
    s->Cz = nq * quantum;
    s->Cx = nq * qx;
    s->bpf = c->bpf;
    s->need0 = q_of(c, s->Cz + s->H - 1) + f.a.c + 1;
    s->base1 = tile_base(c, 1 + f.lead_tiles);                      // x index of d_x[0] in a steady step's fixed coordinates
    const int64_t t01 = s->Cz / f.z_per_tile;
    s->carry = s->need0 - tile_base(c, t01);
    s->Lx = s->carry + s->Cx;
    if (s->base1 < 0 || (c->fmt != banded::kFmtMonoF32 && (s->base1 & 1)) || s->carry <= 0 || s->carry > s->Cx || tile_base(c, t01) < 0) {
        cs_free(s);
        return fail(AF_ERR_UNSUPPORTED, "af_chain_stream: step geometry not covered (base %lld, carry %lld, step %lld)",
                    (long long)s->base1, (long long)s->carry, (long long)s->Cx);
    }
    af_features_config fc;
    af_features_default_config(&fc);
    fc.sample_rate = cfg->output_rate;
    fc.channels = 1;
    fc.hop = cfg->hop;
    fc.n_mels = cfg->n_mels;
    fc.f_min = cfg->f_min;
    fc.f_max = cfg->f_max;
    fc.device = -1;
    st = af_features_create(&fc, &s->tail);
    size_t rows = 0;
    const int64_t start = 0, len = s->Cz + s->H;
    if (st == AF_OK) st = af_features_set_batch_spans(c->feat, &start, &len, 1, &rows);
    if (st != AF_OK) { cs_free(s); return st; }
    s->rows_per_step = rows;                                        // Cz / hop
    s->row_len = af_features_row_len(c->feat);
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        st = fail(e == hipErrorOutOfMemory ? AF_ERR_OUT_OF_MEMORY : AF_ERR_HIP, "%s failed: %s", what,
                  hipGetErrorString(e));
        return false;
    };
    const int64_t in_cap = std::max(s->need0, s->Cx);
    if (!hip_ok(hipHostMalloc(&s->h_in, (size_t)in_cap * s->bpf, hipHostMallocDefault), "hipHostMalloc") ||
        !hip_ok(hipHostMalloc(reinterpret_cast<void**>(&s->h_out), rows * s->row_len * sizeof(float),
                              hipHostMallocDefault), "hipHostMalloc") ||
        !hip_ok(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), "hipStreamCreate") ||
        !hip_ok(hipEventCreate(&s->ev0), "hipEventCreate") || !hip_ok(hipEventCreate(&s->ev1), "hipEventCreate")) {
        cs_free(s);
        return st;
    }
    const int64_t x_cap = std::max(s->need0, s->Lx);
    st = s->d_x.reserve((size_t)x_cap * s->bpf);
    // FIR outputs of a step, or of the flush (a little longer: what the resampler makes of the frames short of a
    // step); never grown later -- the captured graph holds this pointer
    s->z_cap = s->Cz + s->H + 64 + 64 * (int64_t)f.a.L / (int64_t)f.a.M;
    if (st == AF_OK) st = s->d_z.reserve((size_t)std::max<int64_t>(s->z_cap * 4, s->carry * (int64_t)s->bpf));
    if (st == AF_OK) st = s->d_rows.reserve(rows * s->row_len * sizeof(float));
    // one eager steady step on zeros (sets the kernels' LDS attributes, which a capture cannot do), then the
    // capture of the very same enqueue
    if (st == AF_OK && hip_ok(hipMemsetAsync(s->d_x.p, 0, (size_t)x_cap * s->bpf, s->stream), "hipMemsetAsync")) {
        std::memset(s->h_in, 0, (size_t)in_cap * s->bpf);
        st = cs_enqueue_step(s);
        if (st == AF_OK) hip_ok(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
    }
    if (st == AF_OK && hip_ok(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture")) {
        const af_status se = cs_enqueue_step(s);
        hipGraph_t g = nullptr;
        const hipError_t ee = hipStreamEndCapture(s->stream, &g);
        s->graph = g;
        if (se != AF_OK) st = se;
        else if (hip_ok(ee, "hipStreamEndCapture"))
            hip_ok(hipGraphInstantiate(&s->exec, s->graph, nullptr, nullptr, 0), "hipGraphInstantiate");
    }
    if (st != AF_OK) { cs_free(s); return st; }
    *out = s;
    return AF_OK;
}

void af_chain_stream_destroy(af_chain_stream* s) { cs_free(s); }

size_t af_chain_stream_step_frames(const af_chain_stream* s) { return s ? (size_t)s->Cx : 0; }
size_t af_chain_stream_rows_per_step(const af_chain_stream* s) { return s ? s->rows_per_step : 0; }
size_t af_chain_stream_row_len(const af_chain_stream* s) { return s ? s->row_len : 0; }

/* rows the stream will have produced in total once n_frames have been pushed and flushed (= af_chain_n_frames) */
size_t af_chain_stream_total_rows(const af_chain_stream* s, uint64_t n_frames)
{
    return s ? af_chain_n_frames(s->chain, (size_t)n_frames) : 0;
}

/* upper bound of the rows one af_chain_stream_push of n_frames can return */
size_t af_chain_stream_max_rows(const af_chain_stream* s, size_t n_frames)
{
    if (!s) return 0;
    int64_t pending = s->fill + (int64_t)n_frames;
    size_t steps = 0;
    if (s->priming) {
        if (pending < s->carry) return 0;
        pending -= s->carry;
    } else if (s->steps == 0) {
        if (pending < s->need0) return 0;
        pending -= s->need0;
        steps = 1;
    }
    steps += (size_t)(pending / s->Cx);
    return steps * s->rows_per_step;
}

af_status af_chain_stream_push(af_chain_stream* s, const void* pcm, size_t n_frames, float* rows_out, size_t rows_cap,
                               size_t* n_rows)
{
    AF_REQUIRE(s && n_rows, "af_chain_stream_push: null argument");
    *n_rows = 0;
    AF_REQUIRE(!s->flushed, "af_chain_stream_push after af_chain_stream_flush (af_chain_stream_reset starts a new stream)");
    AF_REQUIRE(pcm || n_frames == 0, "af_chain_stream_push: null buffer");
    const size_t will = af_chain_stream_max_rows(s, n_frames);
    if (rows_cap < will * s->row_len)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "row buffer holds %zu floats, this push can produce %zu", rows_cap,
                    will * s->row_len);
    AF_REQUIRE(rows_out || will == 0, "af_chain_stream_push: null row buffer");
    const char* src = static_cast<const char*>(pcm);
    int64_t left = (int64_t)n_frames;
    size_t wrote = 0;
    while (left > 0) {
        const int64_t target = s->priming ? s->carry : s->steps == 0 ? s->need0 : s->Cx;
        const int64_t take = std::min(left, target - s->fill);
        std::memcpy(static_cast<char*>(s->h_in) + (size_t)s->fill * s->bpf, src, (size_t)take * s->bpf);
        s->fill += take;
        src += (size_t)take * s->bpf;
        left -= take;
        s->frames_in += (uint64_t)take;
        if (s->fill < target) break;
        if (s->priming) {       // the history of the step the stream was moved to: to the front of the buffer, no rows
            AF_HIP(hipMemcpyAsync(s->d_x.p, s->h_in, (size_t)s->carry * s->bpf, hipMemcpyHostToDevice, s->stream));
            AF_HIP(hipStreamSynchronize(s->stream));
            s->priming = false;
            s->fill = 0;
            continue;
        }
        if (s->steps == 0) AF_TRY(cs_first_step(s));
        else AF_TRY(cs_timed(s, true));
        std::memcpy(rows_out + wrote * s->row_len, s->h_out, s->rows_per_step * s->row_len * sizeof(float));
        wrote += s->rows_per_step;
        s->steps += 1;
        s->fill = 0;
    }
    s->rows_out += wrote;
    *n_rows = wrote;
    return AF_OK;
}

size_t af_chain_stream_flush_rows(const af_chain_stream* s)
{
    if (!s || s->flushed) return 0;
    const size_t total = af_chain_stream_total_rows(s, s->frames_in);
    return total > (size_t)s->rows_out ? total - (size_t)s->rows_out : 0;
}

af_status af_chain_stream_flush(af_chain_stream* s, float* rows_out, size_t rows_cap, size_t* n_rows)
{
    AF_REQUIRE(s && n_rows, "af_chain_stream_flush: null argument");
    *n_rows = 0;
    if (s->flushed) return AF_OK;
    AF_REQUIRE(!s->priming, "af_chain_stream_flush: the stream was moved to step %llu and has not received that step's "
               "%lld history frames yet", (unsigned long long)s->steps, (long long)s->carry);
    af_chain* c = s->chain;
    const size_t total = af_chain_stream_total_rows(s, s->frames_in);
    const size_t rows = total - (size_t)s->rows_out;               // what the partial step still owes
    if (rows_cap < rows * s->row_len)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "row buffer holds %zu floats, the flush produces %zu", rows_cap,
                    rows * s->row_len);
    if (rows > 0) {
        AF_REQUIRE(rows_out, "af_chain_stream_flush: null row buffer");
        // the rest of the stream with its true coordinates, eagerly (its shape differs per stream): the buffer
        // holds [history | the fill frames], the stream ends behind them (zeros beyond: the clip edge of one-shot)
        const int64_t k = (int64_t)s->steps;
        const int64_t hist = k == 0 ? 0 : s->carry;
        const int64_t x_abs0 = k == 0 ? 0 : tile_base(c, k * s->Cz / c->fused.z_per_tile);
        const int64_t n_res = (int64_t)af_resample_out_len(c->cfg.input_rate, c->cfg.output_rate, (size_t)s->frames_in);
        const int64_t m0 = k * s->Cz, n_out = n_res - m0;
        AF_REQUIRE(n_out > 0 && x_abs0 + hist + s->fill == (int64_t)s->frames_in,
                   "stream bookkeeping: history %lld + %lld + %lld frames, %llu pushed", (long long)x_abs0,
                   (long long)hist, (long long)s->fill, (unsigned long long)s->frames_in);
        char* x = s->d_x.as<char>();
        AF_HIP(hipMemcpyAsync(x + (size_t)hist * s->bpf, s->h_in, (size_t)s->fill * s->bpf, hipMemcpyHostToDevice, s->stream));
        const int64_t start = 0;
        size_t got = 0;
        AF_TRY(af_features_set_batch_spans(s->tail, &start, &n_out, 1, &got));
        AF_REQUIRE(got == rows, "stream bookkeeping: %zu rows in the tail span, expected %zu", got, rows);
        AF_REQUIRE(n_out <= s->z_cap, "stream bookkeeping: %lld FIR outputs in the flush, room for %lld",
                   (long long)n_out, (long long)s->z_cap);
        AF_TRY(cs_kernels(s, s->tail, x_abs0, hist + s->fill, n_res, m0, n_out));
        AF_HIP(hipMemcpyAsync(s->h_out, s->d_rows.p, rows * s->row_len * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        AF_HIP(hipStreamSynchronize(s->stream));
        std::memcpy(rows_out, s->h_out, rows * s->row_len * sizeof(float));
    }
    s->rows_out += rows;
    s->flushed = true;
    *n_rows = rows;
    return AF_OK;
}

af_status af_chain_stream_reset(af_chain_stream* s)
{
    AF_REQUIRE(s, "af_chain_stream_reset: null handle");
    AF_HIP(hipStreamSynchronize(s->stream));
    s->fill = 0;
    s->steps = 0;
    s->frames_in = s->rows_out = 0;
    s->flushed = false;
    s->priming = false;
    return AF_OK;
}

/* frames step 0 consumes (every later step: af_chain_stream_step_frames) and frames of history a step reads in front
 * of its own -- what a time-sharding caller needs to cut a stream into per-GPU step ranges */
size_t af_chain_stream_first_step_frames(const af_chain_stream* s) { return s ? (size_t)s->need0 : 0; }
size_t af_chain_stream_history_frames(const af_chain_stream* s) { return s ? (size_t)s->carry : 0; }

/* Time sharding (BASELINE config 5 across GPUs): continue as if `step` steps had already run.  The next frames pushed
 * must be the stream's frames from *first_frame on: first the step's history (af_chain_stream_history_frames, no rows),
 * then the steps' own frames; rows come out from row step * rows_per_step on and are the rows an unbroken stream (or
 * af_chain_run on the whole recording) has there, bit for bit -- a steady step's launch does not depend on its index. */
af_status af_chain_stream_seek(af_chain_stream* s, uint64_t step, uint64_t* first_frame)
{
    AF_REQUIRE(s, "af_chain_stream_seek: null handle");
    AF_TRY(af_chain_stream_reset(s));
    uint64_t first = 0;
    if (step > 0) {
        AF_REQUIRE(step < ((uint64_t)1 << 40) / (uint64_t)s->Cx, "af_chain_stream_seek: step %llu out of range",
                   (unsigned long long)step);
        first = (uint64_t)(s->need0 - s->carry) + (step - 1) * (uint64_t)s->Cx;   // = base of the step's first tile
        s->steps = step;
        s->frames_in = first;
        s->rows_out = step * s->rows_per_step;
        s->priming = true;
    }
    if (first_frame) *first_frame = first;
    return AF_OK;
}

/* timing of the last replayed step: device time between the events around the graph launch and host wall time of
 * launch + wait; stats[4] = {device_ms, wall_ms, steps so far, rows so far} */
af_status af_chain_stream_stats(const af_chain_stream* s, double* stats)
{
    AF_REQUIRE(s && stats, "af_chain_stream_stats: null argument");
    stats[0] = s->last_device_ms;
    stats[1] = s->last_wall_ms;
    stats[2] = (double)s->steps;
    stats[3] = (double)s->rows_out;
    return AF_OK;
}

}  // extern "C"
