// harness.cpp -- C++ twin of the reference's own unit tests, driven through the C++ host
// mirror (include/audioflow.hpp) and therefore through the exact C-ABI entry points the Rust
// shim of INTEGRATION.md binds.  Each check cites the #[test] it ports
// (forfd8960/audio-flow-rs, src-tauri/src/...).  Needs an MI355X; exit code 0 = all passed.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "audioflow.hpp"

using namespace audioflow;

static int failures = 0;
#define CHECK(cond)                                                               \
    do {                                                                          \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

int main()
{
    // capture.rs:371-382 test_audio_frame_to_mono_single_channel
    {
        auto f = AudioFrame::new_({0.5f, -0.5f}, 16000, 1, 1000);
        auto m = f.to_mono();
        CHECK(m.channels == 1 && m.samples == std::vector<float>({0.5f, -0.5f}));
    }
    // capture.rs:384-400 test_audio_frame_to_mono_stereo
    {
        auto f = AudioFrame::new_({0.5f, 0.25f, -0.5f, -0.25f}, 16000, 2, 1000);
        auto m = f.to_mono();
        CHECK(m.channels == 1 && m.samples.size() == 2);
        CHECK(std::fabs(m.samples[0] - 0.375f) < 0.001f && std::fabs(m.samples[1] + 0.375f) < 0.001f);
    }
    // resampler.rs:184-190 test_no_resample_needed
    {
        auto r = AudioResampler::new_(16000, 16000);
        std::vector<float> in = {0.1f, 0.2f, 0.3f, 0.4f};
        CHECK(r.process(in) == in);
    }
    // resampler.rs:192-198 test_resample_rates, :200-204 test_same_rates_no_resampling
    {
        auto r = AudioResampler::new_(48000, 16000);
        CHECK(r.input_rate() == 48000 && r.output_rate() == 16000 && r.needs_resampling());
        CHECK(!AudioResampler::new_(48000, 48000).needs_resampling());
    }
    // resampler.rs:132-166 chunk/carry/flush bookkeeping + the 480 -> 160 framing of specs/0001-spec.md:270-272
    {
        auto b = BatchResampler::new_(48000, 16000);
        std::vector<float> x(480, 0.25f);
        size_t total = b.process(x).size();
        CHECK(b.buffered() == 480 % 128);
        total += b.flush().size();
        CHECK(b.buffered() == 0);
        CHECK(total == (512 - 32 + 2) / 3);          // ceil((4*128 - T/2) / 3) outputs released
        auto one = AudioResampler::new_(48000, 16000);
        auto y = one.process(x);
        auto tail = one.flush();
        CHECK(y.size() + tail.size() == 160);         // ceil(480 / 3)
        CHECK(std::fabs(y[60] - 0.25f) < 1e-6f);      // unit DC gain
    }
    // error mapping, resampler.rs:49 -> error.rs:109
    {
        bool threw = false;
        try { (void)AudioResampler::new_(0, 16000); }
        catch (const AudioError& e) { threw = std::string(e.what()).rfind("Resampling failed: ", 0) == 0; }
        CHECK(threw);
    }
    // vad.rs:283-298 test_energy_calculation (through the batched energy entry point)
    {
        std::vector<float> z(480, 0.0f), h(480, 0.5f);
        float e = 1.0f;
        CHECK(af_vad_energy(z.data(), 1, 480, &e) == AF_OK && e == 0.0f);
        CHECK(af_vad_energy(h.data(), 1, 480, &e) == AF_OK && std::fabs(e - 0.25f) < 0.0001f);
    }
    // vad.rs:211-237 silence / speech detection, :239-265 transitions, :267-281 reset
    {
        std::vector<float> silence(480, 0.0001f), speech(480, 0.5f);
        CHECK(VoiceActivityDetector(VadConfig{-50.0f}).detect(silence) == VadState::Silence);
        CHECK(VoiceActivityDetector(VadConfig{-50.0f}).detect(speech) == VadState::Speech);
        VoiceActivityDetector v(VadConfig{-50.0f, 0.0f, 2, 1});
        CHECK(v.state() == VadState::Silence);
        CHECK(v.detect(speech) == VadState::Speech);
        CHECK(v.detect(silence) == VadState::Speech);
        CHECK(v.detect(silence) == VadState::Ending);
        CHECK(v.detect(silence) == VadState::Silence);
        VoiceActivityDetector d;
        d.detect(speech);
        CHECK(d.is_speaking());
        d.reset();
        CHECK(d.state() == VadState::Silence && !d.is_speaking());
        // the same vectors through the batched device scan: clip 0 = vad.rs:239-265, clip 1 = four silent frames
        std::vector<float> clips;
        for (const auto* f : {&speech, &silence, &silence, &silence}) clips.insert(clips.end(), f->begin(), f->end());
        for (int i = 0; i < 4; ++i) clips.insert(clips.end(), silence.begin(), silence.end());
        auto st = VoiceActivityDetector::detect_batch(VadConfig{-50.0f, 0.0f, 2, 1}, clips, 2, 480);
        CHECK(st == std::vector<uint8_t>({1, 1, 2, 0, 0, 0, 0, 0}));
    }
    // websocket.rs:376-382 test_message_builder_audio (values per :342)
    {
        auto q = quantize_pcm16({0.5f, -0.5f, 0.25f, -0.25f});
        CHECK(q == std::vector<int16_t>({16383, -16383, 8191, -8191}));
    }
    // build-defined siblings: shapes and a sanity value
    {
        FeaturePipeline p;
        std::vector<float> pcm(2 * 6000, 0.0f);
        for (size_t i = 0; i < 6000; ++i) pcm[2 * i] = pcm[2 * i + 1] = 0.5f * std::sin(2.0 * M_PI * 1000.0 * i / 48000.0);
        auto out = p.process_batch({AudioFrame::new_(pcm, 48000, 2, 0), AudioFrame::new_({}, 48000, 2, 0)});
        CHECK(out.size() == 2 && out[0].n_frames == 20 && out[0].n_mels == 80 && out[1].n_frames == 0);
        size_t arg = 0;                                // the 1 kHz tone peaks in the filter around 1 kHz
        for (size_t m = 1; m < 80; ++m) if (out[0].data[m] > out[0].data[arg]) arg = m;
        CHECK(arg >= 18 && arg <= 22);
        auto fir = FirFilter::new_({0.5f, 0.5f});
        auto y = fir.process({1.0f, 1.0f, 1.0f});
        CHECK(y == std::vector<float>({0.5f, 1.0f, 1.0f}));
        y = fir.process({1.0f});
        CHECK(y == std::vector<float>({1.0f}));       // history carried
    }
    // Stft / MelFilterbank / level meter (docs/OPS.md): shapes, a known tone, Parseval
    {
        Stft st;
        std::vector<float> x(16000);
        for (size_t i = 0; i < x.size(); ++i) x[i] = 0.5f * std::sin(2.0 * M_PI * 1000.0 * i / 16000.0);
        CHECK(st.n_frames(x.size()) == 59 && st.n_bins() == 513);
        auto P = st.power(x);
        CHECK(P.size() == 59 * 513);
        size_t arg = 0;                                // 1 kHz at 16 kHz / 1024 = bin 64
        for (size_t k = 1; k < 513; ++k) if (P[k] > P[arg]) arg = k;
        CHECK(arg == 64);
        // periodic Hann: sum w^2 = 3N/8; a bin-centred tone of amplitude A has |X|^2 = (A N / 4)^2
        CHECK(std::fabs(P[64] / (0.5 * 1024 / 4) / (0.5 * 1024 / 4) - 1.0) < 1e-3);
        // n_fft 512: 1 kHz = bin 32 of 257, same amplitude law with N = 512
        Stft st512(128, 1, 512);
        CHECK(st512.n_frames(x.size()) == 1 + (16000 - 512) / 128 && st512.n_bins() == 257);
        auto P5 = st512.power(x);
        size_t arg5 = 0;
        for (size_t k = 1; k < 257; ++k) if (P5[k] > P5[arg5]) arg5 = k;
        CHECK(arg5 == 32 && std::fabs(P5[32] / (0.5 * 512 / 4) / (0.5 * 512 / 4) - 1.0) < 1e-3);
        // n_fft 400 / hop 160 (the matrix-product DFT kernel): 1 kHz = bin 25 of 201, the amplitude law with N = 400, and one
        // frame against the DFT sum taken here in double
        Stft st400(160, 1, 400);
        CHECK(st400.n_frames(x.size()) == 1 + (16000 - 400) / 160 && st400.n_bins() == 201);
        auto P4 = st400.power(x);
        CHECK(P4.size() == st400.n_frames(x.size()) * 201);
        size_t arg4 = 0;
        for (size_t k = 1; k < 201; ++k) if (P4[k] > P4[arg4]) arg4 = k;
        CHECK(arg4 == 25 && std::fabs(P4[25] / (0.5 * 400 / 4) / (0.5 * 400 / 4) - 1.0) < 1e-3);
        {
            const size_t f = 7;
            double num = 0.0, den = 0.0;
            for (size_t k = 0; k < 201; ++k) {
                double re = 0.0, im = 0.0;
                for (size_t n = 0; n < 400; ++n) {
                    const double w = 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / 400.0);
                    const double a = -2.0 * M_PI * (double)((k * n) % 400) / 400.0;
                    re += w * (double)x[f * 160 + n] * std::cos(a);
                    im += w * (double)x[f * 160 + n] * std::sin(a);
                }
                const double ref = re * re + im * im, d = (double)P4[f * 201 + k] - ref;
                num += d * d;
                den += ref * ref;
            }
            CHECK(std::sqrt(num / den) < 1e-5);
        }
        MelFilterbank fb(16000, 80);
        auto W = fb.weights();
        CHECK(W.size() == 80 * 513);
        auto mel = fb.apply(P);
        CHECK(mel.size() == 59 * 80);
        double direct = 0.0;                           // filter 40 of frame 0 by hand
        for (size_t k = 0; k < 513; ++k) direct += (double)W[40 * 513 + k] * P[k];
        CHECK(std::fabs(mel[40] - direct) <= 1e-5 * std::fabs(direct) + 1e-12);
        auto lv = level_meter(std::vector<float>(1600, 0.5f), 800);
        CHECK(lv.size() == 2 && std::fabs(lv[0].rms - 0.5f) < 1e-6f && lv[1].peak == 0.5f);
        CHECK(std::fabs(lv[0].dbfs - 20.0f * std::log10(0.5f)) < 1e-4f);
    }
    // capture.rs:426-561 ring buffer suite through the C++ mirror
    {
        auto r = RingBuffer::new_(1024);                                   // :426-441 write / read
        CHECK(r.write({1.0f, 2.0f, 3.0f, 4.0f, 5.0f}) == 5 && r.available() == 5);
        auto got = r.read(5);
        CHECK(got && *got == std::vector<float>({1.0f, 2.0f, 3.0f, 4.0f, 5.0f}) && r.available() == 0);
        CHECK(!r.read(5));                                                 // :487-492 empty -> None
        auto w = RingBuffer::new_(10);                                     // :459-471 wrap-around
        CHECK(w.write({1, 2, 3, 4, 5, 6, 7}) == 7);
        CHECK(w.read(5).value() == std::vector<float>({1, 2, 3, 4, 5}));
        CHECK(w.write({8, 9, 10}) == 3);
        CHECK(w.read(5).value() == std::vector<float>({6, 7, 8, 9, 10}));
        auto o = RingBuffer::new_(10);                                     // :473-485 overflow: capacity - 1
        CHECK(o.write(std::vector<float>(15, 1.0f)) == 9 && o.available() == 9);
        o.clear();                                                         // :516-529
        CHECK(o.available() == 0 && !o.read(1));
    }
    // FeatureStream (BASELINE config 5): stepped == one-shot, bit for bit
    {
        const size_t n = 16 * 256 * 5 + 999;
        std::vector<float> x(2 * n);
        for (size_t i = 0; i < x.size(); ++i) x[i] = 0.3f * std::sin(0.01f * (float)i) + 0.1f * std::sin(0.37f * (float)i);
        FeatureStream st(16 * 256 * 2);
        std::vector<float> rows;
        for (size_t pos = 0; pos < n; pos += 3000) {
            const size_t k = std::min<size_t>(3000, n - pos);
            auto r = st.push(std::vector<float>(x.begin() + 2 * pos, x.begin() + 2 * (pos + k)));
            rows.insert(rows.end(), r.begin(), r.end());
        }
        auto tail = st.flush();
        rows.insert(rows.end(), tail.begin(), tail.end());
        FeaturePipeline one;
        auto full = one.process_batch({AudioFrame::new_(x, 48000, 2, 0)});
        CHECK(full.size() == 1 && rows.size() == st.total_rows(n) * 80 && rows == full[0].data);
    }
    // ChainPipeline (BASELINE config 4's stages): two launches, and the mel rows of a 1 kHz tone peak where
    // they should after 48 -> 16 kHz and a low-pass FIR
    {
        const size_t n = 48000;
        std::vector<float> x(2 * n), taps(129);
        for (size_t i = 0; i < n; ++i) x[2 * i] = x[2 * i + 1] = 0.5f * std::sin(2.0f * 3.14159265f * 1000.0f * (float)i / 48000.0f);
        double sum = 0.0;
        for (int k = 0; k < 129; ++k) {
            const double t = (k - 64) * 0.9, sinc = k == 64 ? 1.0 : std::sin(3.14159265358979 * t) / (3.14159265358979 * t);
            taps[k] = (float)(sinc * (0.54 - 0.46 * std::cos(2.0 * 3.14159265358979 * k / 128.0)));
            sum += taps[k];
        }
        for (auto& t : taps) t = (float)(t / sum);
        ChainPipeline chain(48000, 2, 16000, taps);
        auto mel = chain.run(x, 1, n);
        const size_t nf = chain.n_frames(n);
        CHECK(nf == 1 + (16000 - 1024) / 256 && mel.size() == nf * 80 && chain.n_launches() == 2);
        size_t arg = 0;
        for (size_t m = 1; m < 80; ++m)
            if (mel[10 * 80 + m] > mel[10 * 80 + arg]) arg = m;
        // HTK mel of 1 kHz is 1000.0; 82 edges over mel(0..8 kHz) = 2840.0 -> filter centre index ~ 27.5
        CHECK(arg >= 26 && arg <= 29);
        // the same clip twice in a ragged batch (with an empty clip between) gives the same rows twice
        std::vector<float> xx(x);
        xx.insert(xx.end(), x.begin(), x.end());
        auto rag = chain.run_ragged(xx, {0, (int64_t)n, (int64_t)n, (int64_t)(2 * n)});
        CHECK(rag.size() == 2 * mel.size() && std::equal(mel.begin(), mel.end(), rag.begin()) &&
              std::equal(mel.begin(), mel.end(), rag.begin() + mel.size()));
        // the wire format (websocket.rs:246-251) through the same chain: i16 in, rows close to the f32 ones
        std::vector<int16_t> xi(x.size());
        for (size_t i = 0; i < x.size(); ++i) xi[i] = (int16_t)(x[i] * 32767.0f);
        ChainPipeline chain16(48000, 2, 16000, taps, 80, AF_PCM_I16);
        auto mel16 = chain16.run_ragged(xi, {0, (int64_t)n});
        CHECK(mel16.size() == mel.size() && std::fabs(mel16[10 * 80 + arg] / mel[10 * 80 + arg] - 1.0f) < 1e-3f);
        // ChainStream (config 5 on the whole chain): two pushes + flush == the one-shot rows, bit for bit
        ChainStream cs(1, 48000, 2, 16000, taps);
        std::vector<float> long_x;
        for (int r = 0; r < 5; ++r) long_x.insert(long_x.end(), x.begin(), x.end());      // 5 s: one step + a rest
        std::vector<float> rows = cs.push(std::vector<float>(long_x.begin(), long_x.begin() + 2 * 100000));
        auto r2 = cs.push(std::vector<float>(long_x.begin() + 2 * 100000, long_x.end()));
        rows.insert(rows.end(), r2.begin(), r2.end());
        auto r3 = cs.flush();
        rows.insert(rows.end(), r3.begin(), r3.end());
        auto whole = chain.run(long_x, 1, 5 * n);
        CHECK(cs.step_frames() == 184320 && rows.size() == cs.total_rows(5 * n) * 80 && rows == whole);
        // time sharding: a second "rank" starts at step 1 (history first) and produces the rest of the rows
        const uint64_t first = cs.seek(1);
        CHECK(first == cs.first_step_frames() - cs.history_frames());
        auto tail_rows = cs.push(std::vector<float>(long_x.begin() + 2 * first, long_x.end()));
        auto tail_flush = cs.flush();
        tail_rows.insert(tail_rows.end(), tail_flush.begin(), tail_flush.end());
        const size_t rows0 = 184320 / 3 / 256 * 80;
        CHECK(tail_rows.size() == whole.size() - rows0 && std::equal(tail_rows.begin(), tail_rows.end(), whole.begin() + rows0));
    }
    // WireEncoder: the capture -> wire path in one launch (capture.rs:30-42, resampler.rs:60-93, websocket.rs:244-254);
    // 20 ms messages of 320 samples (specs/0002-design.md:1543-1546): each equals MessageBuilder::audio_message's
    // payload for the same samples, a trailing partial batch is not sent
    {
        const size_t n = 48000 + 601;                         // 16 201 resampled samples: 50 messages + 201 left over
        std::vector<float> x(2 * n);
        for (size_t i = 0; i < n; ++i) {
            x[2 * i] = 2.6f * std::sin(0.013f * (float)i);    // the downmix goes beyond +-1: the quantiser clamps
            x[2 * i + 1] = 0.3f * std::cos(0.0071f * (float)i);
        }
        WireEncoder w(48000, 16000, 2, AF_PCM_F32, 320);
        std::vector<int16_t> q;
        const std::string text = w.run(x, 1, n, &q);
        CHECK(w.n_launches() == 1 && w.message_len() == 856 && w.n_messages(n) == 50);
        CHECK(text.size() == 50 * 856 && q.size() == w.out_samples(n) && w.out_samples(n) == 16201);
        // the same samples through the stand-alone stages: to_mono, AudioResampler, quantiser + base64
        auto mono = AudioFrame::new_(x, 48000, 2, 0).to_mono();
        auto r = AudioResampler::new_(48000, 16000);
        auto y = r.process(mono.samples);
        auto tail = r.flush();
        y.insert(y.end(), tail.begin(), tail.end());
        CHECK(y.size() == 16201);
        const auto q_ref = quantize_pcm16(y);
        CHECK(std::equal(q.begin(), q.end(), q_ref.begin()));
        CHECK(*std::max_element(q.begin(), q.end()) == 32767);
        for (size_t k : {size_t(0), size_t(17), size_t(49)}) {
            const std::vector<float> frame(y.begin() + 320 * k, y.begin() + 320 * (k + 1));
            CHECK(text.substr(856 * k, 856) == pcm16_base64(frame));
        }
        // one string per clip
        WireEncoder w0;
        const std::string whole = w0.run(x, 1, n);
        CHECK(whole == pcm16_base64(y));
    }
    std::printf(failures ? "%d check(s) failed\n" : "harness ok\n", failures);
    return failures ? 1 : 0;
}
