// stream.hip -- the feeder side of BASELINE config 5 ("streaming ... with hipGraph-captured steps"):
//   af_ring_*    RingBuffer of the reference (src-tauri/src/modules/audio/capture.rs:84-161): a
//                mutex-guarded f32 ring with capacity - 1 usable slots, writes truncated when full.
//                Same integer bookkeeping, two memcpy segments instead of a modulo per sample.
//   af_stream_*  a long stream cut into fixed steps of C frames: the handle owns the page-locked
//                staging buffers, the device buffer [H = n_fft - hop carried frames | C new frames], the
//                carry and one hipGraph (H2D copy, fused kernel, carry, D2H copy) that is captured once
//                and replayed for every step.  Step k runs when samples [kC, kC + C + H) have arrived
//                and produces the C / hop frames of the one-shot framing that START in [kC, kC + C)
//                (frame f covers samples [f hop, f hop + n_fft)): the span begins on a 16-frame unit
//                boundary, every frame meets the same lanes as in one-shot processing, and the
//                concatenated rows are bit-identical to it.  (The first H frames of a stream only
//                prime the carry.)  The reference's carry / flush analogue is
//                BatchResampler (resampler.rs:132-166), its feeder the capture loop (capture.rs:247-325).
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "af_common.hpp"

using namespace af;

// ---------------------------------------------------------------------------- ring buffer
struct af_ring {
    std::mutex mu;
    std::vector<float> buf;
    size_t capacity = 0;
    size_t write_pos = 0, read_pos = 0;
    size_t used() const { return capacity ? (write_pos + capacity - read_pos) % capacity : 0; }
};

// ---------------------------------------------------------------------------- stream
struct af_stream {
    af_stream_config cfg{};
    af_features* feat = nullptr;      // span [0, H + C): the replayed step
    af_features* tail = nullptr;      // the final, shorter span (flush)
    size_t C = 0, H = 0;              // frames per step, halo frames (n_fft - hop)
    size_t bpf = 0;                   // bytes per PCM frame
    size_t rows_per_step = 0, row_len = 0;
    void* h_in = nullptr;             // page-locked: C frames of PCM
    float* h_out = nullptr;           // page-locked: rows of one step
    DevBuf d_buf, d_rows;             // [H + C] frames | rows of one step (+ the halo's rows)
    size_t fill = 0;                  // frames waiting in h_in
    bool primed = false;              // the first H frames of the stream sit in the carry region
    uint64_t steps = 0;               // full steps done
    uint64_t frames_in = 0, rows_out = 0;
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_device_ms = 0.0f, last_wall_ms = 0.0f;
    bool flushed = false;
};

namespace {

void stream_free(af_stream* s)
{
    if (!s) return;
    if (s->exec) (void)hipGraphExecDestroy(s->exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_out) (void)hipHostFree(s->h_out);
    af_features_destroy(s->feat);
    af_features_destroy(s->tail);
    delete s;
}

// the work of one step on s->stream: new frames in, kernel, halo carry, rows out
af_status enqueue_step(af_stream* s)
{
    char* buf = s->d_buf.as<char>();
    AF_HIP(hipMemcpyAsync(buf + s->H * s->bpf, s->h_in, s->C * s->bpf, hipMemcpyHostToDevice, s->stream));
    AF_TRY(af_features_run_device(s->feat, buf, s->d_rows.as<float>(), s->stream));
    // the last H frames are the next step's halo (regions overlap only when C < H: excluded at create)
    AF_HIP(hipMemcpyAsync(buf, buf + s->C * s->bpf, s->H * s->bpf, hipMemcpyDeviceToDevice, s->stream));
    AF_HIP(hipMemcpyAsync(s->h_out, s->d_rows.p, s->rows_per_step * s->row_len * sizeof(float),
                          hipMemcpyDeviceToHost, s->stream));
    return AF_OK;
}

// run the captured step; its C / hop rows are in h_out
af_status run_step(af_stream* s)
{
    const auto t0 = std::chrono::steady_clock::now();
    AF_HIP(hipEventRecord(s->ev0, s->stream));
    AF_HIP(hipGraphLaunch(s->exec, s->stream));
    AF_HIP(hipEventRecord(s->ev1, s->stream));
    AF_HIP(hipStreamSynchronize(s->stream));
    AF_HIP(hipEventElapsedTime(&s->last_device_ms, s->ev0, s->ev1));
    s->last_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return AF_OK;
}

}  // namespace

extern "C" {

// ---- RingBuffer (capture.rs:84-161) -----------------------------------------------------------
af_status af_ring_create(size_t capacity_samples, af_ring** out)
{
    AF_REQUIRE(out, "af_ring_create: null argument");
    *out = nullptr;
    af_ring* r = new (std::nothrow) af_ring();
    if (!r) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    try {
        r->buf.assign(capacity_samples, 0.0f);                      // capture.rs:93-100
    } catch (...) {
        delete r;
        return fail(AF_ERR_OUT_OF_MEMORY, "ring of %zu samples does not fit host memory", capacity_samples);
    }
    r->capacity = capacity_samples;
    *out = r;
    return AF_OK;
}

void af_ring_destroy(af_ring* r) { delete r; }

size_t af_ring_capacity(const af_ring* r) { return r ? r->capacity : 0; }

size_t af_ring_write(af_ring* r, const float* data, size_t n)
{
    if (!r || !data || n == 0) return 0;
    std::lock_guard<std::mutex> lock(r->mu);
    if (r->capacity == 0) return 0;                                 // (the reference would divide by zero)
    const size_t available = r->capacity - r->used();               // capture.rs:108-110
    const size_t room = available ? available - 1 : 0;              // :111 one slot stays empty
    const size_t to_write = n < room ? n : room;
    const size_t first = std::min(to_write, r->capacity - r->write_pos);
    std::memcpy(r->buf.data() + r->write_pos, data, first * sizeof(float));
    std::memcpy(r->buf.data(), data + first, (to_write - first) * sizeof(float));
    if (to_write > 0) r->write_pos = (r->write_pos + to_write) % r->capacity;   // :118-120
    return to_write;
}

af_status af_ring_read(af_ring* r, size_t size, float* out, size_t out_cap, size_t* n_read, int* is_none)
{
    AF_REQUIRE(r && n_read, "af_ring_read: null argument");
    *n_read = 0;
    if (is_none) *is_none = 0;
    std::lock_guard<std::mutex> lock(r->mu);
    const size_t available = r->used();                             // capture.rs:130
    if (available == 0) {                                           // :131-133 `None`
        if (is_none) *is_none = 1;
        return AF_OK;
    }
    const size_t to_read = size < available ? size : available;     // :135
    if (to_read > out_cap)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "ring read of %zu samples into a buffer of %zu", to_read, out_cap);
    AF_REQUIRE(out || to_read == 0, "af_ring_read: null buffer");
    const size_t first = std::min(to_read, r->capacity - r->read_pos);
    std::memcpy(out, r->buf.data() + r->read_pos, first * sizeof(float));
    std::memcpy(out + first, r->buf.data(), (to_read - first) * sizeof(float));
    r->read_pos = (r->read_pos + to_read) % r->capacity;            // :143
    *n_read = to_read;
    return AF_OK;
}

size_t af_ring_available(af_ring* r)
{
    if (!r) return 0;
    std::lock_guard<std::mutex> lock(r->mu);
    return r->used();                                               // capture.rs:148-152
}

void af_ring_clear(af_ring* r)
{
    if (!r) return;
    std::lock_guard<std::mutex> lock(r->mu);                        // capture.rs:155-160
    r->write_pos = r->read_pos = 0;
    std::fill(r->buf.begin(), r->buf.end(), 0.0f);
}

// ---- FeatureStream ----------------------------------------------------------------------------
void af_stream_default_config(af_stream_config* cfg)
{
    if (!cfg) return;
    af_features_default_config(&cfg->features);
    cfg->step_frames = (size_t)1 << 20;
}

af_status af_stream_create(const af_stream_config* cfg, af_stream** out)
{
    AF_REQUIRE(cfg && out, "af_stream_create: null argument");
    *out = nullptr;
    const af_features_config& fc = cfg->features;
    // (the carry is the n_fft - hop frames a step's last frame shares with the next step's first; rounds 1-4 also asked
    // for n_fft % hop == 0, which nothing below needs: 400 / 160 streams like 1024 / 256)
    AF_REQUIRE(fc.hop > 0 && fc.n_fft >= fc.hop, "streaming needs hop <= n_fft");
    AF_REQUIRE(cfg->step_frames % ((size_t)16 * fc.hop) == 0 && cfg->step_frames >= fc.n_fft,
               "step_frames must be a multiple of 16 * hop (%u) and at least n_fft", 16 * fc.hop);
    AF_REQUIRE(fc.channel_mode == AF_CH_DOWNMIX, "streaming emits one spectrogram per stream");
    af_stream* s = new (std::nothrow) af_stream();
    if (!s) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    s->cfg = *cfg;
    s->C = cfg->step_frames;
    s->H = fc.n_fft - fc.hop;
    s->bpf = (size_t)fc.channels * (fc.sample_format == AF_PCM_I16 ? 2 : 4);
    af_status st = af_features_create(&fc, &s->feat);
    if (st == AF_OK) st = af_features_create(&fc, &s->tail);
    if (st != AF_OK) { stream_free(s); return st; }
    s->row_len = af_features_row_len(s->feat);
    const int64_t start = 0, len = (int64_t)(s->H + s->C);
    size_t rows = 0;
    st = af_features_set_batch_spans(s->feat, &start, &len, 1, &rows);
    if (st != AF_OK) { stream_free(s); return st; }
    s->rows_per_step = rows;                                        // = C / hop
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        st = fail(e == hipErrorOutOfMemory ? AF_ERR_OUT_OF_MEMORY : AF_ERR_HIP, "%s failed: %s", what,
                  hipGetErrorString(e));
        return false;
    };
    if (!hip_ok(hipHostMalloc(&s->h_in, s->C * s->bpf, hipHostMallocDefault), "hipHostMalloc") ||
        !hip_ok(hipHostMalloc(reinterpret_cast<void**>(&s->h_out), rows * s->row_len * sizeof(float),
                              hipHostMallocDefault), "hipHostMalloc") ||
        !hip_ok(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), "hipStreamCreate") ||
        !hip_ok(hipEventCreate(&s->ev0), "hipEventCreate") || !hip_ok(hipEventCreate(&s->ev1), "hipEventCreate")) {
        stream_free(s);
        return st;
    }
    st = s->d_buf.reserve((s->H + s->C) * s->bpf);
    if (st == AF_OK) st = s->d_rows.reserve(rows * s->row_len * sizeof(float));
    if (st == AF_OK && !hip_ok(hipMemsetAsync(s->d_buf.p, 0, (s->H + s->C) * s->bpf, s->stream), "hipMemsetAsync"))
        ;
    // one eager step on zeros (sets the kernel's LDS attribute, which a capture cannot do), then
    // the capture of the very same enqueue
    if (st == AF_OK) {
        std::memset(s->h_in, 0, s->C * s->bpf);
        st = enqueue_step(s);
        if (st == AF_OK && !hip_ok(hipStreamSynchronize(s->stream), "hipStreamSynchronize")) {}
    }
    if (st == AF_OK && hip_ok(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture")) {
        const af_status se = enqueue_step(s);
        hipGraph_t g = nullptr;
        const hipError_t ee = hipStreamEndCapture(s->stream, &g);
        s->graph = g;
        if (se != AF_OK) st = se;
        else if (hip_ok(ee, "hipStreamEndCapture"))
            hip_ok(hipGraphInstantiate(&s->exec, s->graph, nullptr, nullptr, 0), "hipGraphInstantiate");
    }
    if (st != AF_OK) { stream_free(s); return st; }
    *out = s;
    return AF_OK;
}

void af_stream_destroy(af_stream* s) { stream_free(s); }

size_t af_stream_rows_per_step(const af_stream* s) { return s ? s->rows_per_step : 0; }
size_t af_stream_row_len(const af_stream* s) { return s ? s->row_len : 0; }

/* rows the stream will have produced in total once n_frames have been pushed and flushed */
size_t af_stream_total_rows(const af_stream* s, uint64_t n_frames)
{
    if (!s || n_frames < s->cfg.features.n_fft) return 0;
    return (size_t)(1 + (n_frames - s->cfg.features.n_fft) / s->cfg.features.hop);
}

/* upper bound of the rows one af_stream_push of n_frames can return */
size_t af_stream_max_rows(const af_stream* s, size_t n_frames)
{
    if (!s) return 0;
    size_t pending = s->fill + n_frames;
    if (!s->primed) pending = pending > s->H ? pending - s->H : 0;
    return (pending / s->C) * s->rows_per_step;
}

af_status af_stream_push(af_stream* s, const void* pcm, size_t n_frames, float* rows_out, size_t rows_cap,
                         size_t* n_rows)
{
    AF_REQUIRE(s && n_rows, "af_stream_push: null argument");
    *n_rows = 0;
    AF_REQUIRE(!s->flushed, "af_stream_push after af_stream_flush (af_stream_reset starts a new stream)");
    AF_REQUIRE(pcm || n_frames == 0, "af_stream_push: null buffer");
    const size_t will = af_stream_max_rows(s, n_frames);
    if (rows_cap < will * s->row_len)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "row buffer holds %zu floats, this push can produce %zu", rows_cap,
                    will * s->row_len);
    AF_REQUIRE(rows_out || will == 0, "af_stream_push: null row buffer");
    const char* src = static_cast<const char*>(pcm);
    size_t left = n_frames, wrote = 0;
    while (left > 0) {
        const size_t target = s->primed ? s->C : s->H;
        const size_t take = std::min(left, target - s->fill);
        std::memcpy(static_cast<char*>(s->h_in) + s->fill * s->bpf, src, take * s->bpf);
        s->fill += take;
        src += take * s->bpf;
        left -= take;
        s->frames_in += take;
        if (s->fill < target) break;
        if (!s->primed) {                           // the stream's first H frames: the carry of step 0
            AF_HIP(hipMemcpyAsync(s->d_buf.p, s->h_in, s->H * s->bpf, hipMemcpyHostToDevice, s->stream));
            AF_HIP(hipStreamSynchronize(s->stream));
            s->primed = true;
        } else {
            AF_TRY(run_step(s));
            std::memcpy(rows_out + wrote * s->row_len, s->h_out, s->rows_per_step * s->row_len * sizeof(float));
            wrote += s->rows_per_step;
            s->steps += 1;
        }
        s->fill = 0;
    }
    s->rows_out += wrote;
    *n_rows = wrote;
    return AF_OK;
}

af_status af_stream_flush(af_stream* s, float* rows_out, size_t rows_cap, size_t* n_rows)
{
    AF_REQUIRE(s && n_rows, "af_stream_flush: null argument");
    *n_rows = 0;
    if (s->flushed) return AF_OK;
    const size_t total = af_stream_total_rows(s, s->frames_in);
    const size_t rows = total - (size_t)s->rows_out;               // what the partial step still owes
    if (rows_cap < rows * s->row_len)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "row buffer holds %zu floats, the flush produces %zu", rows_cap,
                    rows * s->row_len);
    if (rows > 0) {
        AF_REQUIRE(rows_out, "af_stream_flush: null row buffer");
        AF_REQUIRE(s->primed, "stream bookkeeping: rows owed before the carry was primed");
        // span [carry | the fill frames] through the same kernel, eagerly (its shape differs per stream)
        char* buf = s->d_buf.as<char>();
        AF_HIP(hipMemcpyAsync(buf + s->H * s->bpf, s->h_in, s->fill * s->bpf, hipMemcpyHostToDevice, s->stream));
        const int64_t start = 0, len = (int64_t)(s->H + s->fill);
        size_t got = 0;
        AF_TRY(af_features_set_batch_spans(s->tail, &start, &len, 1, &got));
        AF_REQUIRE(got == rows, "stream bookkeeping: %zu rows in the tail span, expected %zu", got, rows);
        AF_TRY(af_features_run_device(s->tail, buf, s->d_rows.as<float>(), s->stream));
        AF_HIP(hipMemcpyAsync(s->h_out, s->d_rows.p, rows * s->row_len * sizeof(float), hipMemcpyDeviceToHost,
                              s->stream));
        AF_HIP(hipStreamSynchronize(s->stream));
        std::memcpy(rows_out, s->h_out, rows * s->row_len * sizeof(float));
    }
    s->rows_out += rows;
    s->flushed = true;
    *n_rows = rows;
    return AF_OK;
}

af_status af_stream_reset(af_stream* s)
{
    AF_REQUIRE(s, "af_stream_reset: null handle");
    AF_HIP(hipStreamSynchronize(s->stream));
    s->fill = 0;
    s->primed = false;
    s->steps = 0;
    s->frames_in = s->rows_out = 0;
    s->flushed = false;
    return AF_OK;
}

/* timing of the last replayed step: device time between the events around the graph launch and
 * host wall time of launch + wait; stats[4] = {device_ms, wall_ms, steps so far, rows so far} */
af_status af_stream_stats(const af_stream* s, double* stats)
{
    AF_REQUIRE(s && stats, "af_stream_stats: null argument");
    stats[0] = s->last_device_ms;
    stats[1] = s->last_wall_ms;
    stats[2] = (double)s->steps;
    stats[3] = (double)s->rows_out;
    return AF_OK;
}

}  // extern "C"
