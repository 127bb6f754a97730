// wire.hip -- af_wire_*: the reference's intended capture -> wire path on a batch of clips, as ONE launch:
//   interleaved PCM (f32 | i16, 1 | 2 channels) -> to_mono [R capture.rs:30-42] -> polyphase resample [B]
//   (resampler.rs:60-62 names 48 -> 16 kHz) -> (x.clamp(-1, 1) * 32767) as i16, little-endian [R websocket.rs:246-251]
//   -> padded base64 [R websocket.rs:254], either one string per clip or one string per message of `frame_samples`
//   samples (20 ms = 320 samples: specs/0002-design.md:1543-1546; pipeline: specs/0002-design.md:1102-1125).
// The fused launch (banded.hpp WirePlan; banded_mfma.hip wire_kernel) covers every ratio with a single block matrix; other
// configurations run the same stages as separate launches (stereo-staging resampler, then wire_encode_kernel below) on
// equal-length batches.  The product path never touches the host oracle; without a gfx950 device every entry point fails.
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "af_common.hpp"
#include "banded.hpp"
#include "pcm_dev.hpp"

using namespace af;

struct af_wire {
    af_wire_config cfg{};
    int fmt = banded::kFmtMonoF32;
    size_t bpf = 4;
    banded::WirePlan plan;            // ok: one launch
    const float* afrag_res = nullptr;
    int row_len = 0;                  // characters per message (frame_samples > 0)
    int opt_unfused = 0;              // af_wire_set_option("unfused", 1): the stages as separate launches
    // the batch
    bool ragged = false, described = false;
    size_t n_clips = 0, n_in = 0, n_res = 0;
    std::vector<int64_t> off, res_len, text_off, pcm_off;      // ragged: per clip (+ 1)
    int64_t total_in = 0, total_text = 0, total_samples = 0, n_tiles = 0;
    DevBuf d_tiles;
    DevBuf d_mono, d_res;             // unfused: intermediates
    DevBuf d_pcm, d_text, d_pcm16, d_level;    // host-buffer entry point
};

namespace {

int64_t text_len_of(const af_wire* w, int64_t n_res)
{
    const int F = (int)w->cfg.frame_samples;
    return F ? (n_res / F) * (int64_t)w->row_len : 4 * ((2 * n_res + 2) / 3);
}

bool fused(const af_wire* w) { return w->plan.ok && !w->opt_unfused; }

// The stand-alone form of wire_kernel's second half: resampled f32 samples y[clip][n_res] -> PCM16 -> base64 text (and the
// samples themselves, if wanted), one group of <= 6 samples per thread.  Equal-length clips.
__global__ void wire_encode_kernel(const float* __restrict__ y, int64_t y_stride, int64_t n_res, int64_t n_clips, int frame,
                                   int gpf, int row_len, char* __restrict__ text, int64_t text_stride, int16_t* __restrict__ pcm16,
                                   int64_t pcm_stride)
{
    // (messages: the trailing partial batch is walked too -- its samples go to the PCM16 row, its text is not sent)
    const int64_t groups = frame ? ((n_res + frame - 1) / frame) * gpf : (n_res + 5) / 6;
    const int64_t total = groups * n_clips;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t clip = it / groups, gi = it - clip * groups;
        int64_t s0, toff;
        int cnt;
        bool send = true;
        if (frame) {
            const int64_t f = gi / gpf;
            const int j = (int)(gi - f * gpf);
            s0 = f * frame + 6 * j;
            const int left = frame - 6 * j;
            cnt = left < 6 ? left : 6;
            const int64_t have = n_res - s0;
            cnt = have < cnt ? (have < 0 ? 0 : (int)have) : cnt;
            send = (f + 1) * frame <= n_res;
            toff = f * row_len + 16 * j;
        } else {
            s0 = 6 * gi;
            const int64_t left = n_res - s0;
            cnt = left < 6 ? (int)left : 6;
            toff = 16 * gi;
        }
        const float* src = y + clip * y_stride + s0;
        unsigned q[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) q[e] = e < cnt ? (unsigned)pcmdev::quant1(src[e]) & 0xffffu : 0u;
        const uint4 o = pcmdev::b64_encode_words(q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16));
        char* dst = text + clip * text_stride + toff;
        if (cnt <= 0) continue;
        if (!send) {
        } else if (cnt == 6) {
            const unsigned ow[4] = {o.x, o.y, o.z, o.w};
            if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
                unsigned* d4 = reinterpret_cast<unsigned*>(dst);
                d4[0] = o.x; d4[1] = o.y; d4[2] = o.z; d4[3] = o.w;
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) dst[c] = (char)((ow[c >> 2] >> (8 * (c & 3))) & 0xffu);
            }
        } else {
            pcmdev::b64_store_tail(dst, o, 2u * (unsigned)cnt);
        }
        if (pcm16) {
            int16_t* d16 = pcm16 + clip * pcm_stride + s0;
            for (int e = 0; e < cnt; ++e) d16[e] = (int16_t)q[e];
        }
    }
}

// The stand-alone form of wire_kernel's LEVEL part: per message the reference's mean-square energy (vad.rs:157-168: f32 left
// fold, no FMA) and max |x|, one lane per message, from the resampled f32 rows.  Equal-length clips.
__global__ void wire_level_kernel(const float* __restrict__ y, int64_t y_stride, int64_t msgs_per_clip, int64_t n_clips, int frame,
                                  float* __restrict__ energy, float* __restrict__ peak)
{
#pragma clang fp contract(off)
    const int64_t total = msgs_per_clip * n_clips;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t clip = it / msgs_per_clip, f = it - clip * msgs_per_clip;
        const float* src = y + clip * y_stride + f * frame;
        float e = 0.0f, m = 0.0f;
        for (int i = 0; i < frame; ++i) {
            const float v = src[i];
            const float p = v * v;
            e = e + p;
            m = fmaxf(m, fabsf(v));
        }
        if (energy) energy[it] = e / (float)frame;
        if (peak) peak[it] = m;
    }
}

af_status describe(af_wire* w, const std::vector<int64_t>& off)
{
    const size_t n = off.size() - 1;
    w->n_clips = n;
    w->off = off;
    w->res_len.assign(n, 0);
    w->text_off.assign(n + 1, 0);
    w->pcm_off.assign(n + 1, 0);
    w->total_in = off[n] - off[0];
    bool uniform = true;
    for (size_t i = 0; i < n; ++i) {
        const int64_t len = off[i + 1] - off[i];
        if (len < 0) return fail(AF_ERR_INVALID_ARG, "clip offsets must be monotone");
        if (len != off[1] - off[0]) uniform = false;
        w->res_len[i] = (int64_t)af_resample_out_len(w->cfg.input_rate, w->cfg.output_rate, (size_t)len);
        w->text_off[i + 1] = w->text_off[i] + text_len_of(w, w->res_len[i]);
        w->pcm_off[i + 1] = w->pcm_off[i] + w->res_len[i];
    }
    w->total_text = w->text_off[n];
    w->total_samples = w->pcm_off[n];
    w->ragged = !uniform;
    w->n_in = n ? (size_t)(off[1] - off[0]) : 0;
    w->n_res = n ? (size_t)w->res_len[0] : 0;
    w->n_tiles = 0;
    if (!w->ragged && w->fmt == banded::kFmtMonoI16 && w->plan.ok && (w->n_in & 1) && n > 1)
        return fail(AF_ERR_UNSUPPORTED, "af_wire: mono i16 clips must start on even frames (equal-length clips of an odd frame count, %zu, do not)", w->n_in);
    if (w->ragged) {
        if (!w->plan.ok)
            return fail(AF_ERR_UNSUPPORTED, "af_wire: ragged batches need the fused launch (a ratio with one block matrix, "
                                            "1 or 2 channels, frame_samples 0 or a divisor of 3840)");
        std::vector<banded::WireTile> tiles;
        for (size_t i = 0; i < n; ++i) {
            const int64_t len = off[i + 1] - off[i];
            if (!banded::fits(w->plan, len)) return fail(AF_ERR_UNSUPPORTED, "a clip of %lld frames is beyond the wire kernel's offsets", (long long)len);
            // mono i16 is staged as one dword per frame pair from the clip's first frame: a clip that starts on an odd frame
            // of the buffer would put every dword across two aligned ones (and the last one 2 bytes behind the buffer)
            if (w->fmt == banded::kFmtMonoI16 && ((off[i] - off[0]) & 1) && len > 0)
                return fail(AF_ERR_UNSUPPORTED, "af_wire: mono i16 clips must start on even frames (clip %zu starts at frame %lld)",
                            i, (long long)(off[i] - off[0]));
            const int64_t nt = (w->res_len[i] + w->plan.y_per_tile - 1) / w->plan.y_per_tile;
            for (int64_t t = 0; t < nt; ++t) {
                banded::WireTile r{};
                r.x_off = off[i] - off[0];
                r.text_off = w->text_off[i];
                r.pcm_off = w->pcm_off[i];
                r.x_len = (int32_t)len;
                r.n_res = (int32_t)w->res_len[i];
                r.t = (int32_t)t;
                const uint64_t msg0 = w->row_len ? (uint64_t)(w->text_off[i] / w->row_len) : 0;
                r.msg_lo = (uint32_t)msg0;
                r.msg_hi = (uint32_t)(msg0 >> 32);
                tiles.push_back(r);
            }
        }
        w->n_tiles = (int64_t)tiles.size();
        if (!tiles.empty()) {
            AF_TRY(w->d_tiles.reserve(tiles.size() * sizeof(banded::WireTile)));
            AF_HIP(hipMemcpy(w->d_tiles.p, tiles.data(), tiles.size() * sizeof(banded::WireTile), hipMemcpyHostToDevice));
        }
    }
    w->described = true;
    return AF_OK;
}

int align_of(uintptr_t v)
{
    return v % 16 == 0 ? 16 : v % 8 == 0 ? 8 : v % 4 == 0 ? 4 : 1;
}

}  // namespace

extern "C" {

void af_wire_default_config(af_wire_config* cfg)
{
    if (!cfg) return;
    cfg->input_rate = 48000;           // resampler.rs:60-62 create_48k_to_16k
    cfg->output_rate = 16000;          // websocket.rs:274 "pcm_16000"
    cfg->channels = 2;
    cfg->sample_format = AF_PCM_F32;
    cfg->frame_samples = 0;
    cfg->device = -1;
}

af_status af_wire_create(const af_wire_config* cfg, af_wire** out)
{
    AF_REQUIRE(cfg && out, "af_wire_create: null argument");
    *out = nullptr;
    AF_REQUIRE(cfg->input_rate > 0 && cfg->output_rate > 0, "sample rates must be positive");
    AF_REQUIRE(cfg->channels >= 1 && cfg->channels <= 16, "channels must be 1..16 (got %u)", cfg->channels);
    AF_REQUIRE(cfg->sample_format == AF_PCM_F32 || cfg->sample_format == AF_PCM_I16, "sample_format must be AF_PCM_F32 or AF_PCM_I16");
    AF_REQUIRE(cfg->frame_samples % 2 == 0, "frame_samples must be even (got %u)", cfg->frame_samples);
    AF_TRY(ensure_device(cfg->device));
    af_wire* w = new (std::nothrow) af_wire;
    if (!w) return fail(AF_ERR_OUT_OF_MEMORY, "af_wire_create: out of memory");
    w->cfg = *cfg;
    const bool i16 = cfg->sample_format == AF_PCM_I16;
    w->bpf = (size_t)cfg->channels * (i16 ? 2 : 4);
    w->row_len = (int)(4 * ((2 * (int64_t)cfg->frame_samples + 2) / 3));
    if (cfg->channels <= 2 && cfg->input_rate != cfg->output_rate) {
        w->fmt = banded::fused_fmt((int)cfg->channels, i16);
        banded::Plan rp;
        af_status st = resampler_fragments(cfg->input_rate, cfg->output_rate, 1, &rp, &w->afrag_res);
        if (st != AF_OK) { delete w; return st; }
        w->plan = banded::make_wire_plan(rp, w->fmt, (int)cfg->frame_samples);
    }
    if (!w->plan.ok && (i16 || cfg->channels > 2)) {
        delete w;
        return fail(AF_ERR_UNSUPPORTED, "af_wire: %u-channel %s input at %u -> %u Hz with frame_samples %u is outside the fused launch "
                    "(ratios with one block matrix, 1 or 2 channels, frame_samples 0 or an even divisor of 3840), and the "
                    "stage-by-stage form takes 1- or 2-channel f32", cfg->channels, i16 ? "i16" : "f32", cfg->input_rate,
                    cfg->output_rate, cfg->frame_samples);
    }
    *out = w;
    return AF_OK;
}

void af_wire_destroy(af_wire* w) { delete w; }

size_t af_wire_out_samples(const af_wire* w, size_t n_frames)
{
    return w ? af_resample_out_len(w->cfg.input_rate, w->cfg.output_rate, n_frames) : 0;
}

size_t af_wire_text_len(const af_wire* w, size_t n_frames)
{
    return w ? (size_t)text_len_of(w, (int64_t)af_wire_out_samples(w, n_frames)) : 0;
}

size_t af_wire_message_len(const af_wire* w) { return w && w->cfg.frame_samples ? (size_t)w->row_len : 0; }

size_t af_wire_n_messages(const af_wire* w, size_t n_frames)
{
    return w && w->cfg.frame_samples ? af_wire_out_samples(w, n_frames) / w->cfg.frame_samples : 0;
}

af_status af_wire_set_batch_uniform(af_wire* w, size_t n_clips, size_t frames_per_clip, size_t* total_text, size_t* total_samples)
{
    AF_REQUIRE(w, "af_wire_set_batch_uniform: null handle");
    std::vector<int64_t> off(n_clips + 1);
    for (size_t i = 0; i <= n_clips; ++i) off[i] = (int64_t)(i * frames_per_clip);
    AF_TRY(describe(w, off));
    if (total_text) *total_text = (size_t)w->total_text;
    if (total_samples) *total_samples = (size_t)w->total_samples;
    return AF_OK;
}

af_status af_wire_set_batch(af_wire* w, const int64_t* clip_offsets, size_t n_clips, size_t* total_text, size_t* total_samples)
{
    AF_REQUIRE(w && (clip_offsets || n_clips == 0), "af_wire_set_batch: null argument");
    std::vector<int64_t> off(n_clips + 1, 0);
    for (size_t i = 0; i <= n_clips && clip_offsets; ++i) off[i] = clip_offsets[i];
    AF_TRY(describe(w, off));
    if (total_text) *total_text = (size_t)w->total_text;
    if (total_samples) *total_samples = (size_t)w->total_samples;
    return AF_OK;
}

af_status af_wire_text_offsets(const af_wire* w, int64_t* text_off, size_t cap)
{
    AF_REQUIRE(w && text_off, "af_wire_text_offsets: null argument");
    if (cap < w->text_off.size()) return fail(AF_ERR_BUFFER_TOO_SMALL, "text_off holds %zu entries, need %zu", cap, w->text_off.size());
    std::memcpy(text_off, w->text_off.data(), w->text_off.size() * sizeof(int64_t));
    return AF_OK;
}

af_status af_wire_set_option(af_wire* w, const char* key, int value)
{
    AF_REQUIRE(w && key, "af_wire_set_option: null argument");
    if (std::strcmp(key, "unfused") == 0) {
        if (value && w->ragged) return fail(AF_ERR_UNSUPPORTED, "af_wire: ragged batches run the fused launch only");
        if (value && (w->cfg.sample_format == AF_PCM_I16 || w->cfg.channels > 2))
            return fail(AF_ERR_UNSUPPORTED, "af_wire: the stage-by-stage form takes 1- or 2-channel f32");
        w->opt_unfused = value ? 1 : 0;
        return AF_OK;
    }
    return fail(AF_ERR_INVALID_ARG, "af_wire_set_option: unknown key '%s'", key);
}

int af_wire_n_launches(const af_wire* w)
{
    if (!w) return 0;
    return fused(w) ? 1 : 2;
}

size_t af_wire_total_messages(const af_wire* w)
{
    return w && w->row_len ? (size_t)(w->total_text / w->row_len) : 0;
}

af_status af_wire_run_device(af_wire* w, const void* pcm_dev, char* text_dev, int16_t* pcm16_dev, float* energy_dev,
                             float* peak_dev, void* stream)
{
    AF_REQUIRE(w, "af_wire_run_device: null handle");
    AF_REQUIRE(w->described, "af_wire_run_device: describe the batch first (af_wire_set_batch*)");
    if (w->n_clips == 0 || w->total_in == 0 || w->total_samples == 0) return AF_OK;
    AF_REQUIRE(pcm_dev && text_dev, "af_wire_run_device: null buffer");
    AF_REQUIRE(!(energy_dev || peak_dev) || w->cfg.frame_samples, "af_wire_run_device: energy / peak rows are per message (frame_samples > 0)");
    hipStream_t s = as_stream(stream);
    // the fused launch folds a tile's messages on the 64 lanes of one wave, 16-byte reads: messages of a multiple of 16
    // samples, at most 64 per tile; other message lengths take the stage-by-stage form for this call where that exists
    const bool want_level = energy_dev || peak_dev;
    const bool level_fused = !want_level || (w->cfg.frame_samples % 80 == 0 && w->plan.y_per_tile / (int64_t)w->cfg.frame_samples <= 64);
    const bool stages_possible = !w->ragged && w->cfg.sample_format == AF_PCM_F32 && w->cfg.channels <= 2;
    if (fused(w) && !level_fused && !stages_possible)
        return fail(AF_ERR_UNSUPPORTED, "af_wire: energy / peak rows in the fused launch need messages of a multiple of 80 samples, at "
                    "most 64 per 3840 (frame_samples %u)", w->cfg.frame_samples);
    if (fused(w) && level_fused) {
        if (w->fmt == banded::kFmtMonoI16 && reinterpret_cast<uintptr_t>(pcm_dev) % 4 != 0)
            return fail(AF_ERR_INVALID_ARG, "af_wire_run_device: mono i16 PCM must start on a 4-byte boundary (got %p)", pcm_dev);
        banded::WireArgs q;
        q.x = pcm_dev;
        q.text = text_dev;
        q.pcm16 = pcm16_dev;
        q.energy = energy_dev;
        q.peak = peak_dev;
        q.msg_stride = w->cfg.frame_samples ? (int64_t)(w->n_res / w->cfg.frame_samples) : 0;
        int al = align_of(reinterpret_cast<uintptr_t>(text_dev));
        if (w->cfg.frame_samples) al = std::min(al, align_of((uintptr_t)w->row_len));
        if (w->ragged) {
            q.tiles = w->d_tiles.as<banded::WireTile>();
            q.n_tiles = w->n_tiles;
            for (size_t i = 0; i < w->n_clips; ++i) al = std::min(al, align_of((uintptr_t)w->text_off[i]));
        } else {
            q.x_len = (int64_t)w->n_in;
            q.x_stride = (int64_t)w->n_in;
            q.n_res = (int64_t)w->n_res;
            q.n_clips = (int64_t)w->n_clips;
            q.text_stride = text_len_of(w, (int64_t)w->n_res);
            q.pcm_stride = (int64_t)w->n_res;
            if (w->n_clips > 1) al = std::min(al, align_of((uintptr_t)q.text_stride));
        }
        q.text_align = al;
        return banded::launch_wire(w->plan, w->afrag_res, q, s);
    }
    // stage by stage (equal-length clips, f32): to_mono + resampler, then quantiser + base64
    AF_REQUIRE(!w->ragged, "af_wire: ragged batches run the fused launch only");
    const float* y = nullptr;
    if (w->cfg.input_rate == w->cfg.output_rate && w->cfg.channels == 1) {
        y = static_cast<const float*>(pcm_dev);
    } else {
        AF_TRY(w->d_res.reserve(w->n_clips * w->n_res * sizeof(float)));
        if (w->cfg.channels == 2)
            AF_TRY(af_resample_batch_stereo_device(w->cfg.input_rate, w->cfg.output_rate, static_cast<const float*>(pcm_dev),
                                                   w->n_clips, w->n_in, w->d_res.as<float>(), stream));
        else
            AF_TRY(af_resample_batch_device(w->cfg.input_rate, w->cfg.output_rate, static_cast<const float*>(pcm_dev), w->n_clips,
                                            w->n_in, w->d_res.as<float>(), stream));
        y = w->d_res.as<float>();
    }
    const int F = (int)w->cfg.frame_samples, gpf = (F + 5) / 6;
    const int64_t groups = (F ? (((int64_t)w->n_res + F - 1) / F) * gpf : ((int64_t)w->n_res + 5) / 6) * (int64_t)w->n_clips;
    if (groups <= 0) return AF_OK;
    int64_t grid = (groups + 255) / 256;
    if (grid > (1ll << 30)) grid = 1ll << 30;
    hipLaunchKernelGGL(wire_encode_kernel, dim3((unsigned)grid), dim3(256), 0, s, y, (int64_t)w->n_res, (int64_t)w->n_res,
                       (int64_t)w->n_clips, F, gpf, w->row_len, text_dev, text_len_of(w, (int64_t)w->n_res), pcm16_dev,
                       (int64_t)w->n_res);
    AF_HIP(hipGetLastError());
    if ((energy_dev || peak_dev) && F && w->n_res / F > 0) {
        const int64_t msgs = (int64_t)(w->n_res / F);
        int64_t gl = (msgs * (int64_t)w->n_clips + 63) / 64;
        if (gl > (1ll << 30)) gl = 1ll << 30;
        hipLaunchKernelGGL(wire_level_kernel, dim3((unsigned)gl), dim3(64), 0, s, y, (int64_t)w->n_res, msgs, (int64_t)w->n_clips, F,
                           energy_dev, peak_dev);
        AF_HIP(hipGetLastError());
    }
    return AF_OK;
}

af_status af_wire_run(af_wire* w, const void* pcm, size_t pcm_len, char* text, size_t text_cap, int16_t* pcm16, size_t pcm16_cap,
                      float* energy, float* peak, size_t level_cap)
{
    AF_REQUIRE(w, "af_wire_run: null handle");
    AF_REQUIRE(w->described, "af_wire_run: describe the batch first (af_wire_set_batch*)");
    const size_t need = (size_t)w->total_in * w->cfg.channels;
    if (pcm_len < need) return fail(AF_ERR_BUFFER_TOO_SMALL, "pcm buffer holds %zu samples, the batch describes %zu", pcm_len, need);
    if (text_cap < (size_t)w->total_text) return fail(AF_ERR_BUFFER_TOO_SMALL, "text buffer holds %zu characters, need %zu", text_cap, (size_t)w->total_text);
    if (pcm16 && pcm16_cap < (size_t)w->total_samples)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "pcm16 buffer holds %zu samples, need %zu", pcm16_cap, (size_t)w->total_samples);
    const size_t n_msg = af_wire_total_messages(w);
    if ((energy || peak) && level_cap < n_msg)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "energy / peak buffers hold %zu messages, need %zu", level_cap, n_msg);
    if (w->n_clips == 0 || w->total_in == 0 || w->total_samples == 0) return AF_OK;
    AF_REQUIRE(pcm && text, "af_wire_run: null buffer");
    const size_t in_bytes = (size_t)w->total_in * w->bpf;
    AF_TRY(w->d_pcm.reserve(in_bytes));
    AF_TRY(w->d_text.reserve((size_t)w->total_text + 16));
    if (pcm16) AF_TRY(w->d_pcm16.reserve((size_t)w->total_samples * sizeof(int16_t) + 16));
    if (energy || peak) AF_TRY(w->d_level.reserve(2 * (n_msg + 4) * sizeof(float)));
    float* d_en = energy ? w->d_level.as<float>() : nullptr;
    float* d_pk = peak ? w->d_level.as<float>() + n_msg + 4 : nullptr;
    AF_HIP(hipMemcpy(w->d_pcm.p, pcm, in_bytes, hipMemcpyHostToDevice));
    AF_TRY(af_wire_run_device(w, w->d_pcm.p, w->d_text.as<char>(), pcm16 ? w->d_pcm16.as<int16_t>() : nullptr, d_en, d_pk, nullptr));
    if (energy && n_msg) AF_HIP(hipMemcpy(energy, d_en, n_msg * sizeof(float), hipMemcpyDeviceToHost));
    if (peak && n_msg) AF_HIP(hipMemcpy(peak, d_pk, n_msg * sizeof(float), hipMemcpyDeviceToHost));
    if (w->total_text) AF_HIP(hipMemcpy(text, w->d_text.p, (size_t)w->total_text, hipMemcpyDeviceToHost));
    if (pcm16) AF_HIP(hipMemcpy(pcm16, w->d_pcm16.p, (size_t)w->total_samples * sizeof(int16_t), hipMemcpyDeviceToHost));
    return AF_OK;
}

}  // extern "C"
