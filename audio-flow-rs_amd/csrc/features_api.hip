// features_api.hip -- host side of FeaturePipeline (include/audioflow_dsp.h):
// builds the constant tables (window, twiddles, banded mel plan), the per-batch
// clip tables, and launches the fused kernel of features_kernel.hip.
#include <algorithm>
#include <numeric>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "features.hpp"

using namespace af;
using namespace af::feat;

struct af_features {
    af_features_config cfg{};
    int device = 0;
    int n_cus = 256;
    std::vector<float> W;            // [n_mels][513] f32, as used on the device (n_fft < 1024: on every (1024 / n_fft)-th bin)
    std::vector<float> Wn;           // [n_mels][n_fft / 2 + 1]: the filterbank as the API reports it
    Plan plan{};                     // groups dealt to the 4 waves of the general kernel
    Plan plan8{};                    // ... and to the 8 waves of the hop-256 kernel (n_waves == 0: not usable)
    DevBuf d_const;                  // wintab | twtab | wfrag | meta | v2 tables | wfrag8 | meta8
    const float* d_wintab = nullptr;
    const float* d_twtab = nullptr;
    const float* d_wfrag = nullptr;
    const int* d_meta = nullptr;
    const float* d_tabs2 = nullptr;
    const float* d_wfrag8 = nullptr;
    const int* d_meta8 = nullptr;
    DevBuf d_mono;                   // channels > 2: the downmixed batch the fused kernel reads
    DevBuf d_units;                  // v2::UnitDesc per 16-frame unit (hop 256 only)
    bool units_ok = false;
    DevBuf d_tabs3;                  // features_v3.hip: f16-pair fragment tables of the two DFT stages (hop 256, mel output),
    v2::Plan4 plan4;                  // hop-256 kernel: the mel phase on 4 x 4 x 1 blocks (ok: used instead of plan8)
    DevBuf d_mel4;                   // its fragments [2][kM4Quads][64][4] | lane words [2][64]
    int opt_mel4 = -1;               // af_features_set_option("mel4"): 0 forces the 16 x 16 x 4 plan, -1 the default
    bool v3_ok = false;              // then its mel plan's fragments and slot table (steps of a slot 4 bins apart)
    bool v3_tried = false;           // build_v3 ran (on af_features_set_option(h, "kernel", 3))
    std::vector<double> win1024;     // the window of the 1024-point frame (zero behind n_fft): what build_v3 needs later
    Plan plan8b{};
    const float* d_wfrag8b = nullptr;
    const int* d_meta8b = nullptr;
    // filterbanks the banded MFMA plan does not cover (few, wide filters: 20 mels at 48 kHz ...): the kernels write
    // |X|^2 rows into d_power and dense_mel_kernel contracts them with W (two launches instead of one)
    bool dense_mel = false;
    DevBuf d_W, d_band, d_power;     // W [n_mels][513]; first / last + 1 non-zero bin per filter; [total_frames][513]
    DevBuf d_Wn, d_bandn;            // n_fft < 1024: the filterbank on its own grid (af_mel_apply on n_fft / 2 + 1 wide rows)
    // the batch tables are rewritten by set_batch*: the last launch that reads them must be done first
    hipEvent_t last_run = nullptr;
    bool run_pending = false;
    // host copies of the tables (chunking of the host-buffer path) and its three streams
    std::vector<int64_t> h_start, h_len, h_out_off, h_unit_off;
    bool contiguous = false;         // clips back to back in increasing order: the batch can be cut by clip ranges
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev_in, ev_run;
    ~af_features()
    {
        if (last_run) (void)hipEventDestroy(last_run);
        for (hipEvent_t e : ev_in) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_run) (void)hipEventDestroy(e);
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_run) (void)hipStreamDestroy(s_run);
        if (s_out) (void)hipStreamDestroy(s_out);
    }
    // batch description
    DevBuf d_tables;                 // start[n] | len[n] | out_off[n+1] | unit_off[n+1]
    size_t n_clips = 0;
    int64_t total_pcm_frames = 0;
    int64_t total_frames = 0;
    int64_t total_units = 0;
    bool offsets_aligned = false;    // every clip base on a 16-byte boundary (the general kernel's vector loads)
    bool offsets_aligned4 = false;   // ... on a 4-byte boundary (enough for the hop-256 kernel)
    // staging for the host-buffer entry point
    DevBuf d_pcm, d_mel;
    DevBuf d_stamps;                 // diagnostics: per-phase cycle sums (af_features_debug_stamps)
    int opt_kernel = -1, opt_wg_per_cu = -1;   // af_features_set_option; -1: the process-wide af_debug_set switch
    // n_fft outside {1024, 512, 256}: the DFT as a matrix product (features_dft.hip); none of the FFT kernels' tables exist
    bool dft = false;
    DevBuf d_dft;                    // tfrag | wfrag | wmask
    size_t dft_off_w = 0, dft_off_mask = 0, dft_off_queue = 0;
    DevBuf d_queue;                  // hop-256 kernel: the counter of its unit pool (a 256-byte block of its own)
    int opt_pool = -1;               // af_features_set_option("pool"): 0 every workgroup its own contiguous range only
    int opt_dft_queue = -1;          // af_features_set_option("dft_queue"): 0 the static deal of units (diagnostics)
    int dft_kg = 0, dft_nb = 0, dft_nmt = 0, dft_row_stride = 0, dft_tile_stride = 0, dft_waves = 1;
};

// folded-row padding of the matrix-product DFT kernel in dwords: 8 is the conflict-free stride of its 16-byte LDS reads, 4
// leaves one two-way conflict per read (measured: no difference in time) and brings 400 / 160 from 22 to 21 LDS granules of
// 1 280 bytes per workgroup, i.e. from five to six workgroups per CU (1.74 against 1.89 ms at the time)
#ifndef AF_DFT_PAD
#define AF_DFT_PAD 4
#endif

namespace {

constexpr double kPi = 3.14159265358979323846;

// docs/OPS.md "Mel filterbank": HTK 2595 log10(1 + f/700); Slaney: linear (200/3 Hz per mel)
// below 1 kHz, logarithmic (27 mels per factor 6.4) above
double hz_to_mel(double hz, bool slaney)
{
    if (!slaney) return 2595.0 * std::log10(1.0 + hz / 700.0);
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return hz >= min_log_hz ? min_log_mel + std::log(hz / min_log_hz) / logstep : hz / f_sp;
}
double mel_to_hz(double m, bool slaney)
{
    if (!slaney) return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// docs/OPS.md "Mel filterbank": triangular; optional Slaney area normalisation 2/(e[m+2]-e[m])
void build_mel(const af_features_config& c, double f_max, std::vector<float>& W, std::vector<float>& Wn)
{
    const int nm = (int)c.n_mels;
    const bool sl = c.mel_scale == AF_MEL_SLANEY;
    std::vector<double> edges(nm + 2);
    const double m_lo = hz_to_mel(c.f_min, sl), m_hi = hz_to_mel(f_max, sl);
    for (int i = 0; i < nm + 2; ++i)
        edges[i] = mel_to_hz(m_lo + (m_hi - m_lo) * (double)i / (double)(nm + 1), sl);
    // n_fft < 1024 (512, 256): the kernels transform the n_fft-sample frame zero-padded to 1024 points, whose bins
    // 0, s, 2 s, ... (s = 1024 / n_fft) ARE the n_fft-point DFT; the filterbank lives on those bins, zero elsewhere
    const int n_fft = (int)c.n_fft, bins_n = n_fft / 2 + 1, stride = kNfft / n_fft;
    Wn.assign((size_t)nm * bins_n, 0.0f);
    W.assign((size_t)nm * kBins, 0.0f);
    for (int m = 0; m < nm; ++m) {
        const double lo = edges[m], ce = edges[m + 1], hi = edges[m + 2];
        for (int k = 0; k < bins_n; ++k) {
            const double f = (double)k * (double)c.sample_rate / (double)n_fft;
            const double up = (f - lo) / (ce - lo), dn = (hi - f) / (hi - ce);
            double v = up < dn ? up : dn;
            if (c.mel_norm == AF_MEL_NORM_SLANEY) v *= 2.0 / (hi - lo);
            Wn[(size_t)m * bins_n + k] = v > 0.0 ? (float)v : 0.0f;
            W[(size_t)m * kBins + (size_t)k * stride] = Wn[(size_t)m * bins_n + k];
        }
    }
}

// Turn W into MFMA A fragments.  Filters are handled in units of 4 (one float4
// of output per lane); a group is 1..4 consecutive units = one 16-row MFMA tile
// whose K range is the band of bins where any of its filters is non-zero.
// Dynamic programming picks the groups and deals them, in order, to the 4
// waves so that the largest per-wave step count is minimal (ties: fewest
// steps in total).
int flush_cost() { return 8; }

// bin_major (hop-256 kernel): a slot's 16 bins start on a multiple of 4 and lane (row, kq) of step i holds the
// weight of bin bin0 + 4 kq + i, so that a lane's four B operands of a slot are ONE 16-byte LDS read; otherwise
// (general kernel) step i covers bins bin0 + 4 i + kq.
af_status build_plan(const std::vector<float>& W, int n_mels, int kWaves, int max_slots, int uf, Plan& plan,
                     std::vector<float>& wfrag, std::vector<int>& meta, bool bin_major = false)
{
    std::memset(&plan, 0, sizeof(plan));
    plan.n_mels = n_mels;
    plan.n_waves = kWaves;
    const int kMaxStepsW = max_slots * kSlotSteps;
    // uf = filters per unit (4: every group ends in whole float4 stores; 2: finer cuts, the groups
    // that are not multiples of 4 filters are stored as single floats)
    const int U = (n_mels + uf - 1) / uf;
    const int gmax = 16 / uf;                            // units per group (one 16-row MFMA tile)
    // band of each filter
    std::vector<int> lo(n_mels, kBins), hi(n_mels, -1);
    for (int m = 0; m < n_mels; ++m)
        for (int k = 0; k < kBins; ++k)
            if (W[(size_t)m * kBins + k] != 0.0f) {
                if (k < lo[m]) lo[m] = k;
                hi[m] = k;
            }
    auto band = [&](int u0, int u1, int& b0, int& steps) {  // units [u0, u1)
        int l = kBins, h = -1;
        for (int m = uf * u0; m < uf * u1 && m < n_mels; ++m)
            if (hi[m] >= 0) { l = std::min(l, lo[m]); h = std::max(h, hi[m]); }
        if (h < l) { l = 0; h = 0; }                        // all-zero group: one harmless step
        if (bin_major) l &= ~3;
        steps = (h - l + 1 + 3) / 4;
        b0 = l;
        // the group's steps, padding included, walk consecutive bins: never read past bin 512
        // (bin_major: past bin 515 -- the kernel keeps three zero bins behind bin 512)
        const int padded = (steps + kSlotSteps - 1) / kSlotSteps * kSlotSteps;
        const int row_end = bin_major ? kBins + 3 : kBins;
        if (b0 + 4 * padded > row_end) b0 = row_end - 4 * padded;
        if (b0 < 0) b0 = 0;
    };
    const int INF = 1 << 28;
    // What a group costs its wave besides its K-steps: combining the two accumulators and the
    // (up to 16-byte) stores of 16 frames, measured at about kFlushCost MFMA steps.
    const int kFlushCost = flush_cost();
    // g[i][j]: cheapest cover of units [i, j) with groups of <= 4 units (steps + flushes)
    // gs[i][j]: the padded K-steps of that cover (a wave has room for kMaxStepsW of them)
    std::vector<std::vector<int>> g(U + 1, std::vector<int>(U + 1, INF)), gcut = g, gs = g;
    for (int i = 0; i <= U; ++i) g[i][i] = gs[i][i] = 0;
    for (int len = 1; len <= U; ++len)
        for (int i = 0; i + len <= U; ++i) {
            const int j = i + len;
            for (int k = std::max(i, j - gmax); k < j; ++k) {  // last group = units [k, j)
                int b0, st;
                band(k, j, b0, st);
                st = (st + kSlotSteps - 1) / kSlotSteps * kSlotSteps;   // groups occupy whole slots
                // a group that does not start and end on a multiple of 4 filters leaves as four predicated
                // single-float stores per lane instead of one 16-byte store: twice the flush
                const bool quads = (uf * k) % 4 == 0 && (std::min(uf * j, n_mels) - uf * k) % 4 == 0;
                const int fl = quads ? kFlushCost : 2 * kFlushCost;
                if (g[i][k] + st + fl < g[i][j]) {
                    g[i][j] = g[i][k] + st + fl;
                    gs[i][j] = gs[i][k] + st;
                    gcut[i][j] = k;
                }
            }
        }
    // f[w][j]: (max wave load, total) for the first w waves covering units [0, j)
    struct Cost { int mx, tot; };
    auto better = [](const Cost& a, const Cost& b) { return a.mx < b.mx || (a.mx == b.mx && a.tot < b.tot); };
    std::vector<std::vector<Cost>> f(kWaves + 1, std::vector<Cost>(U + 1, Cost{INF, INF}));
    std::vector<std::vector<int>> fcut(kWaves + 1, std::vector<int>(U + 1, 0));
    f[0][0] = Cost{0, 0};
    for (int w = 1; w <= kWaves; ++w)
        for (int j = 0; j <= U; ++j)
            for (int i = 0; i <= j; ++i) {
                if (f[w - 1][i].mx >= INF || gs[i][j] > kMaxStepsW) continue;
                const Cost c{std::max(f[w - 1][i].mx, g[i][j]), f[w - 1][i].tot + g[i][j]};
                if (better(c, f[w][j])) { f[w][j] = c; fcut[w][j] = i; }
            }
    if (f[kWaves][U].mx >= INF)
        return fail(AF_ERR_UNSUPPORTED, "mel plan does not fit %d waves x %d K-steps", kWaves, kMaxStepsW);
    int wave_lo[kPlanWaves + 1];
    wave_lo[kWaves] = U;
    for (int w = kWaves; w >= 1; --w) wave_lo[w - 1] = fcut[w][wave_lo[w]];

    wfrag.assign((size_t)kWaves * kMaxSteps * 64, 0.0f);      // [wave][kMaxSlotsW][64][4]
    meta.assign((size_t)kWaves * kMaxSlotsW, 0);
    for (int w = 0; w < kWaves; ++w) {
        // groups of this wave, recovered from gcut
        std::vector<std::pair<int, int>> groups;
        for (int j = wave_lo[w + 1]; j > wave_lo[w];) {
            const int k = gcut[wave_lo[w]][j];
            groups.push_back({k, j});
            j = k;
        }
        int need = 0;
        for (const auto& gr : groups) {
            int b0, st;
            band(gr.first, gr.second, b0, st);
            need += (st + kSlotSteps - 1) / kSlotSteps * kSlotSteps;
        }
        if (need > kMaxStepsW)
            return fail(AF_ERR_UNSUPPORTED, "mel plan needs %d K-steps on one wave (limit %d)", need,
                        kMaxStepsW);
        int s = 0;
        for (auto it = groups.rbegin(); it != groups.rend(); ++it) {
            int b0, st;
            band(it->first, it->second, b0, st);
            const int m0 = uf * it->first;
            const int cnt = std::min(uf * it->second, n_mels) - m0;
            const int padded = (st + kSlotSteps - 1) / kSlotSteps * kSlotSteps;
            for (int i = 0; i < padded; ++i, ++s) {
                const int bin = b0 + 4 * i;
                if (s % kSlotSteps == 0) meta[(size_t)w * kMaxSlotsW + s / kSlotSteps] = bin;
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15;
                    const int k = bin_major ? (bin - 4 * (i % kSlotSteps)) + 4 * (lane >> 4) + i % kSlotSteps
                                            : bin + (lane >> 4);
                    wfrag[(((size_t)w * kMaxSlotsW + s / kSlotSteps) * 64 + lane) * kSlotSteps +
                          s % kSlotSteps] =
                        (r < cnt && k < kBins) ? W[(size_t)(m0 + r) * kBins + k] : 0.0f;   // zero off the band
                }
            }
            plan.n_steps += st;
            plan.slot_end[w][s / kSlotSteps - 1] = 0x10000 | (cnt << 8) | m0;
            plan.n_groups += 1;
        }
        plan.wave_nslots[w] = s / kSlotSteps;
    }
    return AF_OK;
}

// Plan of the mel phase on 16 blocks of 4 x 4 x 1 (features.hpp, v2::Plan4).  Returns !ok where the filterbank does not
// fit (n_mels not a multiple of 4, more than 16 pieces per group at 32 steps): the 16 x 16 x 4 plan serves those.
v2::Plan4 build_plan4(const std::vector<float>& W, int n_mels, std::vector<float>& wfrag4, std::vector<int>& lane4)
{
    v2::Plan4 pl;
    wfrag4.assign((size_t)2 * v2::kM4Quads * 64 * 4, 0.0f);
    lane4.assign(2 * 64, 0);
    if (n_mels % 4 != 0 || n_mels <= 0) return pl;
    const int nq = n_mels / 4;
    struct Band { int q, k0, len; };
    std::vector<Band> bands;
    for (int q = 0; q < nq; ++q) {
        int lo = kBins, hi = -1;
        for (int m = 4 * q; m < 4 * q + 4; ++m)
            for (int k = 0; k < kBins; ++k)
                if (W[(size_t)m * kBins + k] != 0.0f) { lo = std::min(lo, k); hi = std::max(hi, k); }
        if (hi < lo) { lo = 0; hi = 0; }                     // an all-zero quad: one harmless step
        const int k0 = lo & ~3;
        bands.push_back({q, k0, hi - k0 + 1});
    }
    // the quads dealt to the two groups, longest first, to the lighter group
    std::vector<Band> order = bands;
    std::sort(order.begin(), order.end(), [](const Band& a, const Band& b) { return a.len > b.len || (a.len == b.len && a.q < b.q); });
    std::vector<Band> grp[2];
    int sum[2] = {0, 0};
    for (const Band& b : order) {
        const int g = sum[0] <= sum[1] ? 0 : 1;
        grp[g].push_back(b);
        sum[g] += b.len;
    }
    for (int g = 0; g < 2; ++g) {
        // the fewest steps S (a multiple of 4) at which every band fits <= 4 pieces and the pieces fit 4 rows of 4 blocks
        int S = 0;
        std::vector<int> pcs;
        for (int cand = 4; cand <= 4 * v2::kM4Quads && !S; cand += 4) {
            pcs.clear();
            bool ok = true;
            for (const Band& b : grp[g]) {
                const int pc = (b.len + cand - 1) / cand;
                if (pc > 4) { ok = false; break; }
                pcs.push_back(pc);
            }
            if (!ok) continue;
            // first fit, largest first (exact for sizes 1..4 in rows of 4)
            std::vector<int> idx(pcs.size());
            std::iota(idx.begin(), idx.end(), 0);
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return pcs[a] > pcs[b] || (pcs[a] == pcs[b] && a < b); });
            int cap[4] = {4, 4, 4, 4};
            std::vector<int> row(pcs.size(), -1);
            for (int i : idx) {
                for (int r = 0; r < 4 && row[i] < 0; ++r)
                    if (cap[r] >= pcs[i]) { row[i] = r; cap[r] -= pcs[i]; }
                if (row[i] < 0) { ok = false; break; }
            }
            if (!ok) continue;
            S = cand;
            // The arrangement: which row a band sits in, the order of the bands inside a row, and where each piece's S bins start
            // (anywhere from hi - S to lo, a multiple of 4) are free.  They decide the LDS bank conflicts of the B-operand reads:
            // lane (block b, frame j) reads the 16-byte slot (8 r' + 2 j + ks_b / 4) mod 16, and the 16-lane groups of a
            // ds_read_b128 are the blocks {4 kq, 4 kq + 3, 4 (kq + 1) + 1, 4 (kq + 1) + 2} (features.hpp): a group is
            // conflict-free when its four blocks' ks / 4 are x, x + 1, x + 8, x + 9 (mod 16).  With the pieces where first-fit
            // put them the reads cost 3.2e7 conflict cycles per launch of config 2 (11 % of the LDS time; all reads on a
            // conflict-free pattern measured 1.042 against 1.076 ms); a hill climb over row swaps and band swaps with the
            // best shifts per group brings the conflict score of the 80-mel bank at 48 kHz from 22 / 17 to 2 / 2.
            struct Ent { int band, pcs; };
            std::vector<Ent> rows[4];
            for (int i : idx) rows[row[i]].push_back({i, pcs[i]});
            struct Blk { bool used; int lo, hi; };
            const auto blocks_of = [&](const std::vector<Ent> (&rw)[4], Blk (&B)[16]) {
                for (int k = 0; k < 16; ++k) B[k] = {false, 0, 0};
                for (int r = 0; r < 4; ++r) {
                    int pos = 0;
                    for (const Ent& e : rw[r]) {
                        const Band& b = grp[g][e.band];
                        for (int pi = 0; pi < e.pcs; ++pi) {
                            const int lo = b.k0 + pi * S;
                            B[4 * r + pos + pi] = {true, lo, std::min(lo + S, b.k0 + b.len)};
                        }
                        pos += e.pcs;
                    }
                }
            };
            const int ks_max = kBins + 3 - S;                                        // never read past the row's three zero bins
            const auto cands = [&](const Blk& b, int (&c)[16]) -> int {
                int n = 0;
                if (!b.used) { c[0] = 0; c[1] = 4; c[2] = 32; c[3] = 36; return 4; }
                for (int ks = std::min(b.lo, ks_max); ks >= 0 && ks + S >= b.hi && n < 16; ks -= 4) c[n++] = ks;
                if (n == 0) c[n++] = std::max(0, std::min(b.lo, ks_max));
                return n;
            };
            const auto score4 = [](const int (&ks)[4]) {
                int cnt[16] = {0};
                for (int k = 0; k < 4; ++k)
                    for (int jj = 0; jj < 4; ++jj) cnt[((ks[k] >> 2) + 2 * jj) & 15] += 1;
                int sc = 0;
                for (int k = 0; k < 16; ++k) sc += cnt[k] > 1 ? cnt[k] - 1 : 0;
                return sc;
            };
            // best shifts of every read group for an arrangement; returns the total score
            const auto evaluate = [&](const std::vector<Ent> (&rw)[4], int (&ks_out)[16]) {
                Blk B[16];
                blocks_of(rw, B);
                int total = 0;
                for (int kq = 0; kq < 4; ++kq) {
                    const int member[4] = {4 * kq, 4 * kq + 3, 4 * ((kq + 1) & 3) + 1, 4 * ((kq + 1) & 3) + 2};
                    int c[4][16], n[4];
                    for (int m = 0; m < 4; ++m) n[m] = cands(B[member[m]], c[m]);
                    int best = 1 << 20, pick[4] = {c[0][0], c[1][0], c[2][0], c[3][0]};
                    for (int a = 0; a < n[0] && best; ++a)
                        for (int b2 = 0; b2 < n[1] && best; ++b2)
                            for (int c2 = 0; c2 < n[2] && best; ++c2)
                                for (int d = 0; d < n[3] && best; ++d) {
                                    const int ks[4] = {c[0][a], c[1][b2], c[2][c2], c[3][d]};
                                    const int sc = score4(ks);
                                    if (sc < best) { best = sc; for (int m = 0; m < 4; ++m) pick[m] = ks[m]; }
                                }
                    total += best;
                    for (int m = 0; m < 4; ++m) ks_out[member[m]] = pick[m];
                }
                return total;
            };
            int ks_of[16];
            int best_score = evaluate(rows, ks_of);
            for (int pass = 0; pass < 8 && best_score > 0; ++pass) {
                bool improved = false;
                const auto attempt = [&]() {
                    int ks_try[16];
                    const int sc = evaluate(rows, ks_try);
                    if (sc < best_score) { best_score = sc; std::memcpy(ks_of, ks_try, sizeof(ks_of)); improved = true; return true; }
                    return false;
                };
                for (int a = 0; a < 4 && best_score > 0; ++a)
                    for (int b2 = a + 1; b2 < 4 && best_score > 0; ++b2) {
                        std::swap(rows[a], rows[b2]);
                        if (!attempt()) std::swap(rows[a], rows[b2]);
                    }
                for (int r = 0; r < 4 && best_score > 0; ++r)
                    for (size_t a = 0; a < rows[r].size() && best_score > 0; ++a)
                        for (size_t b2 = a + 1; b2 < rows[r].size() && best_score > 0; ++b2) {
                            std::swap(rows[r][a], rows[r][b2]);
                            if (!attempt()) std::swap(rows[r][a], rows[r][b2]);
                        }
                if (!improved) break;
            }
            pl.read_conflict_score[g] = best_score;
            if (std::getenv("AF_M4_NO_ARRANGE")) {                                     // (diagnostics: round 5's first layout, pieces at their first bin)
                for (int r = 0; r < 4; ++r) rows[r].clear();
                for (int i : idx) rows[row[i]].push_back({i, pcs[i]});
                Blk B[16];
                blocks_of(rows, B);
                for (int k = 0; k < 16; ++k) ks_of[k] = B[k].used ? std::max(0, std::min(B[k].lo, ks_max)) : 0;
            }
            // write the tables
            for (int r = 0; r < 4; ++r) {
                int pos = 0;
                for (const Ent& e : rows[r]) {
                    const Band& b = grp[g][e.band];
                    for (int pi = 0; pi < e.pcs; ++pi) {
                        const int blk = 4 * r + pos + pi;
                        const int lo = b.k0 + pi * S, hi = std::min(lo + S, b.k0 + b.len);   // the piece's bins [lo, hi)
                        const int ks = ks_of[blk];
                        const int owner = pi == 0, add1 = (pi % 2 == 0 && pi + 1 < e.pcs), add2 = (pi == 0 && e.pcs > 2);
                        for (int i4 = 0; i4 < 4; ++i4) {
                            const int lane = 4 * blk + i4;
                            lane4[g * 64 + lane] = ks | (b.q << 10) | (owner << 17) | (add1 << 18) | (add2 << 19) | (i4 << 20);
                            for (int t = 0; t < S; ++t) {
                                const int bin = ks + t;
                                const float w = (bin >= lo && bin < hi && bin < kBins) ? W[(size_t)(4 * b.q + i4) * kBins + bin] : 0.0f;
                                wfrag4[(((size_t)g * v2::kM4Quads + t / 4) * 64 + lane) * 4 + t % 4] = w;
                            }
                        }
                    }
                    pos += e.pcs;
                    pl.n_pieces[g] += e.pcs;
                }
            }
            // (blocks no piece uses read somewhere harmless that fits the group's pattern; their weights stay zero)
            {
                Blk B[16];
                blocks_of(rows, B);
                for (int k = 0; k < 16; ++k)
                    if (!B[k].used)
                        for (int i4 = 0; i4 < 4; ++i4) lane4[g * 64 + 4 * k + i4] = ks_of[k] | (i4 << 20);
            }
        }
        if (!S) return pl;
        pl.quads[g] = S / 4;
    }
    pl.ok = true;
    return pl;
}

// Tables of the matrix-product DFT (features.hpp, namespace dft) for a frame length the FFT kernels do not factor.
af_status upload_bank_of(const std::vector<float>& w, int n_mels, int row_len, DevBuf& d_w, DevBuf& d_b);
af_status build_dft(af_features* p)
{
    const af_features_config& c = p->cfg;
    const int n_fft = (int)c.n_fft, n_bins = n_fft / 2 + 1, nm = (int)c.n_mels;
    // the symmetric window halves the products: n = 1 .. nh = n_fft / 2 paired with n_fft - n (features_dft.hip)
    const int nh = n_fft / 2;
    // (at least four groups: the kernel's fragment ring takes whole chunks of four and lets one bin tile end per chunk)
    const int kg = std::max((nh + 15) / 16, 4), nb = (n_bins + 15) / 16, nmt = (nm + 15) / 16;
    p->W.clear();                                       // (the 513-bin grid means nothing here)
    p->dft = true;
    p->dft_kg = kg; p->dft_nb = nb; p->dft_nmt = nmt;
    p->dft_row_stride = 2 * 16 * kg + (AF_DFT_PAD);    // E | O | pad
    p->dft_tile_stride = 16 * p->dft_row_stride;
    // two waves per workgroup share a unit's rows (27 KB for 400 / 160): four workgroups per CU = two waves per SIMD, each
    // hiding the other's waits.  (One wave per unit, rows private: 2.13 ms on 256 x 30 s of 400 / 160 with one wave per SIMD --
    // the LDS holds no more -- against 1.39 ms in this shape at six workgroups per CU.)
    const size_t per_wg = (size_t)p->dft_tile_stride * sizeof(float);
    if (per_wg > 160 * 1024) return fail(AF_ERR_UNSUPPORTED, "features: n_fft %u needs %zu bytes of LDS per workgroup", c.n_fft, per_wg);
    p->dft_waves = 2;
    // window x (cos | -sin), the angle reduced in integers first
    std::vector<double> win(n_fft), cs(n_fft), sn(n_fft);
    for (int i = 0; i < n_fft; ++i) {
        win[i] = 0.5 - 0.5 * std::cos(2.0 * kPi * (double)i / (double)n_fft);
        cs[i] = std::cos(2.0 * kPi * (double)i / (double)n_fft);
        sn[i] = std::sin(2.0 * kPi * (double)i / (double)n_fft);
    }
    // (zero groups behind the last one up to a multiple of 4: the kernel walks the stream in chunks of four groups)
    std::vector<float> tfrag((size_t)((nb * kg + 3) / 4 * 4) * 2 * 64 * 4, 0.0f);
    for (int bt = 0; bt < nb; ++bt)
        for (int g = 0; g < kg; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int k = 16 * g + 4 * (lane >> 4) + j, bin = 16 * bt + (lane & 15);
                    if (k >= nh || bin >= n_bins) continue;
                    const int nn = k + 1;                                  // the folded pair (nn, n_fft - nn)
                    const int r = (int)(((int64_t)nn * bin) % n_fft);
                    const size_t o = ((((size_t)bt * kg + g) * 2) * 64 + lane) * 4 + j;
                    tfrag[o] = (float)(win[nn] * cs[r]);
                    tfrag[o + 64 * 4] = (float)(-win[nn] * sn[r]);
                }
    std::vector<float> wfrag((size_t)nb * nmt * 64 * 4, 0.0f);
    std::vector<uint32_t> wmask(nb, 0u);
    for (int bt = 0; bt < nb; ++bt)
        for (int mt = 0; mt < nmt; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    const int m = 16 * mt + (lane & 15), bin = 16 * bt + 4 * (lane >> 4) + i;
                    if (m >= nm || bin >= n_bins) continue;
                    const float v = p->Wn[(size_t)m * n_bins + bin];
                    wfrag[(((size_t)bt * nmt + mt) * 64 + lane) * 4 + i] = v;
                    if (v != 0.0f) wmask[bt] |= 1u << mt;
                }
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    p->dft_off_w = up16(tfrag.size() * sizeof(float));
    p->dft_off_mask = up16(p->dft_off_w + wfrag.size() * sizeof(float));
    // (the counter in a 256-byte block of its own: on the cache line of the mask words, which every wave reads once per bin
    // tile, the atomics made the launch five times slower)
    p->dft_off_queue = (p->dft_off_mask + wmask.size() * sizeof(uint32_t) + 511) & ~(size_t)255;
    AF_TRY(p->d_dft.reserve(p->dft_off_queue + 256));
    char* b = p->d_dft.as<char>();
    if (hipMemcpy(b, tfrag.data(), tfrag.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + p->dft_off_w, wfrag.data(), wfrag.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + p->dft_off_mask, wmask.data(), wmask.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
        return fail(AF_ERR_HIP, "uploading the DFT tables failed");
    // the filterbank on its own grid for af_mel_apply
    return upload_bank_of(p->Wn, nm, n_bins, p->d_Wn, p->d_bandn);
}

af_status upload_bank_of(const std::vector<float>& w, int n_mels, int row_len, DevBuf& d_w, DevBuf& d_b)
{
    std::vector<int> band(2 * (size_t)n_mels, 0);
    for (int m = 0; m < n_mels; ++m) {
        int lo = row_len, hi = 0;
        for (int k = 0; k < row_len; ++k)
            if (w[(size_t)m * row_len + k] != 0.0f) { lo = std::min(lo, k); hi = k + 1; }
        band[2 * m] = lo < hi ? lo : 0;
        band[2 * m + 1] = lo < hi ? hi : 0;
    }
    AF_TRY(d_w.reserve(w.size() * sizeof(float)));
    AF_TRY(d_b.reserve(band.size() * sizeof(int)));
    if (hipMemcpy(d_w.p, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_b.p, band.data(), band.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail(AF_ERR_HIP, "uploading the filterbank failed");
    return AF_OK;
}

size_t bytes_per_frame(const af_features_config& c)
{
    return (size_t)c.channels * (c.sample_format == AF_PCM_I16 ? 2 : 4);
}

// More than two channels (capture.rs:30-42 downmixes any count): af_to_mono_device runs first and the
// fused kernel reads its mono output, so the kernel sees 4-byte frames.
int kernel_channels(const af_features_config& c) { return c.channels > 2 ? 1 : (int)c.channels; }
size_t kernel_bytes_per_frame(const af_features_config& c)
{
    return c.channels > 2 ? 4 : bytes_per_frame(c);
}

// spans: clip c = PCM frames [start[c], start[c] + len[c]) of the buffer (may overlap)
af_status upload_tables(af_features* p, const std::vector<int64_t>& start_in,
                        const std::vector<int64_t>& len_in)
{
    if (p->run_pending) {                      // a launch on a caller's stream may still read the old tables
        AF_HIP(hipEventSynchronize(p->last_run));
        p->run_pending = false;
    }
    // AF_CH_SEPARATE: table entry 2c + ch is channel ch of clip c (same PCM span twice)
    std::vector<int64_t> start, len;
    if (p->cfg.channel_mode == AF_CH_SEPARATE) {
        for (size_t c = 0; c < start_in.size(); ++c)
            for (int ch = 0; ch < 2; ++ch) { start.push_back(start_in[c]); len.push_back(len_in[c]); }
    } else {
        start = start_in;
        len = len_in;
    }
    p->n_clips = start.size();
    const size_t n = p->n_clips;
    // start[n] | len[n] | out_off[n+1] | unit_off[n+1]
    std::vector<int64_t> tab(2 * n + 2 * (n + 1));
    int64_t* t_start = tab.data();
    int64_t* t_len = tab.data() + n;
    int64_t* out_off = tab.data() + 2 * n;
    int64_t* unit_off = out_off + (n + 1);
    out_off[0] = 0;
    unit_off[0] = 0;
    const size_t bpf = kernel_bytes_per_frame(p->cfg);
    bool aligned = (p->cfg.hop * bpf) % 16 == 0;
    // the hop-256 kernel's 16-byte global loads only need DWORD-aligned addresses (global memory is in unaligned
    // access mode): every clip base a multiple of 4 bytes is enough -- i.e. everything but i16 mono clips that
    // start on an odd frame
    bool aligned4 = true;
    int64_t end = 0;
    for (size_t c = 0; c < n; ++c) {
        if (start[c] < 0 || len[c] < 0) return fail(AF_ERR_INVALID_ARG, "negative clip start or length");
        t_start[c] = start[c];
        t_len[c] = len[c];
        const int64_t nf = (int64_t)af_features_n_frames(p, (size_t)len[c]);
        out_off[c + 1] = out_off[c] + nf;
        unit_off[c + 1] = unit_off[c] + (nf + kFramesPerIter - 1) / kFramesPerIter;
        if ((start[c] * (int64_t)bpf) % 16 != 0) aligned = false;
        if ((start[c] * (int64_t)bpf) % 4 != 0) aligned4 = false;
        end = std::max(end, start[c] + len[c]);
    }
    p->h_start.assign(t_start, t_start + n);
    p->h_len.assign(t_len, t_len + n);
    p->h_out_off.assign(out_off, out_off + n + 1);
    p->h_unit_off.assign(unit_off, unit_off + n + 1);
    p->contiguous = true;
    for (size_t c = 0; c + 1 < n; ++c)
        if (start[c + 1] < start[c] + len[c] && !(p->cfg.channel_mode == AF_CH_SEPARATE && start[c + 1] == start[c]))
            p->contiguous = false;
    p->total_pcm_frames = end;
    p->total_frames = out_off[n];
    p->total_units = unit_off[n];
    p->offsets_aligned = aligned;
    p->offsets_aligned4 = aligned4;
    if (p->cfg.channels > 2) AF_TRY(p->d_mono.reserve((size_t)end * sizeof(float) + 64));
    AF_TRY(p->d_tables.reserve((tab.size() ? tab.size() : 1) * sizeof(int64_t)));
    AF_HIP(hipMemcpy(p->d_tables.p, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    // hop-256 kernel: one 32-byte record per 16-frame unit
    p->units_ok = false;
    if (!p->dft && ((p->dense_mel && p->cfg.output != AF_OUT_POWER) || (p->cfg.output == AF_OUT_POWER && p->cfg.n_fft != (uint32_t)kNfft)))
        AF_TRY(p->d_power.reserve(((size_t)p->total_frames * kBins + 16) * sizeof(float)));
    // (dense_mel: the hop-256 kernel only writes power rows, which needs no filterbank plan)
    const bool v2_ok = !p->dft && (p->plan8.n_waves == v2::kW || (p->dense_mel && p->cfg.hop == (uint32_t)v2::kHop));
    const bool v3_possible = !p->dft && p->cfg.hop == (uint32_t)v2::kHop && !p->dense_mel && p->cfg.output == AF_OUT_MEL;   // built on request
    // (the matrix-product DFT kernel walks the same records with the handle's own hop and takes any alignment)
    const int64_t unit_hop = p->dft ? (int64_t)p->cfg.hop : (int64_t)v2::kHop;
    if (p->dft && p->total_units >= (1ll << 31)) return fail(AF_ERR_UNSUPPORTED, "features: more than 2^31 units in one batch");
    if ((v2_ok || v3_possible || p->dft) && p->total_units > 0 && (p->dft || p->total_units < (1ll << 26))) {
        std::vector<v2::UnitDesc> units((size_t)p->total_units + 4);   // + zero records: the kernels read a unit ahead
        const bool sep = p->cfg.channel_mode == AF_CH_SEPARATE;
        size_t k = 0;
        for (size_t c = 0; c < n; ++c) {
            const int64_t nf = out_off[c + 1] - out_off[c];
            const int64_t nu = unit_off[c + 1] - unit_off[c];
            for (int64_t i = 0; i < nu; ++i, ++k) {
                v2::UnitDesc& d = units[k];
                const int64_t first = i * kFramesPerIter * unit_hop;
                d.pcm0 = t_start[c] + first;
                d.out0 = out_off[c] + i * kFramesPerIter;
                const int64_t left = t_len[c] - first;
                d.n_avail = (int32_t)std::min<int64_t>(left, (int64_t)1 << 30);
                d.n_rows = (int32_t)std::min<int64_t>(nf - i * kFramesPerIter, kFramesPerIter);
                d.flags = (i + 1 < nu ? 1 : 0) | ((int32_t)std::min<int64_t>(nu - 1 - i, 255) << 8);
                d.sel = sep ? (int32_t)(c & 1) : -1;
            }
        }
        AF_TRY(p->d_units.reserve(units.size() * sizeof(v2::UnitDesc)));
        if (!p->dft) AF_TRY(p->d_queue.reserve((64 + 4096) * sizeof(unsigned)));               // the counter of the hop-256 kernel's work pool, a block of its own
        AF_HIP(hipMemcpy(p->d_units.p, units.data(), units.size() * sizeof(v2::UnitDesc), hipMemcpyHostToDevice));
        p->units_ok = true;
    }
    return AF_OK;
}

}  // namespace

extern "C" {

void af_features_default_config(af_features_config* cfg)
{
    if (!cfg) return;
    cfg->sample_rate = 48000;
    cfg->channels = 2;
    cfg->n_fft = kNfft;
    cfg->hop = 256;
    cfg->n_mels = 80;
    cfg->f_min = 0.0f;
    cfg->f_max = 0.0f;
    cfg->sample_format = AF_PCM_F32;
    cfg->device = -1;
    cfg->output = AF_OUT_MEL;
    cfg->log_mel = 0;
    cfg->mel_scale = AF_MEL_HTK;
    cfg->mel_norm = AF_MEL_NORM_NONE;
    cfg->channel_mode = AF_CH_DOWNMIX;
}

af_status af_features_create(const af_features_config* cfg, af_features** out)
{
    AF_REQUIRE(cfg && out, "af_features_create: null argument");
    *out = nullptr;
    AF_REQUIRE(cfg->n_fft >= 32u && cfg->n_fft <= 1024u, "n_fft must be in [32, 1024] (got %u)", cfg->n_fft);
    const bool dft = !(cfg->n_fft == 1024u || cfg->n_fft == 512u || cfg->n_fft == 256u);
    AF_REQUIRE(cfg->channels >= 1 && cfg->channels <= 16, "channels must be in [1, 16] (got %u)", cfg->channels);
    if (cfg->channels > 2 && cfg->sample_format != AF_PCM_F32)
        return fail(AF_ERR_UNSUPPORTED, "more than two channels are supported for f32 PCM only");
    AF_REQUIRE(cfg->hop >= 64 && cfg->hop <= 1024 && cfg->hop % 4 == 0,
               "hop must be a multiple of 4 in [64, 1024] (got %u)", cfg->hop);
    AF_REQUIRE(cfg->n_mels >= 1 && cfg->n_mels <= 128u, "n_mels must be in [1, %d]",
               128);
    AF_REQUIRE(cfg->sample_rate > 0, "sample_rate must be positive");
    AF_REQUIRE(cfg->sample_format == AF_PCM_F32 || cfg->sample_format == AF_PCM_I16,
               "unknown sample_format %u", cfg->sample_format);
    AF_REQUIRE(cfg->output <= AF_OUT_POWER && cfg->log_mel <= 1 && cfg->mel_scale <= AF_MEL_SLANEY &&
                   cfg->mel_norm <= AF_MEL_NORM_SLANEY && cfg->channel_mode <= AF_CH_SEPARATE,
               "unknown output / mel / channel option");
    AF_REQUIRE(cfg->channel_mode == AF_CH_DOWNMIX || cfg->channels == 2,
               "AF_CH_SEPARATE needs interleaved stereo");
    const double f_max = cfg->f_max > 0.0f ? (double)cfg->f_max : 0.5 * (double)cfg->sample_rate;
    AF_REQUIRE((double)cfg->f_min >= 0.0 && (double)cfg->f_min < f_max, "need 0 <= f_min < f_max");

    AF_TRY(ensure_device(cfg->device));
    af_features* p = new (std::nothrow) af_features();
    if (!p) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    p->cfg = *cfg;
    if (hipGetDevice(&p->device) != hipSuccess) p->device = 0;
    p->n_cus = device_cu_count(p->device);

    build_mel(*cfg, f_max, p->W, p->Wn);
    if (dft) {
        const af_status ds = build_dft(p);
        if (ds != AF_OK) { delete p; return ds; }
        *out = p;
        return AF_OK;
    }
    std::vector<float> wfrag;
    std::vector<int> meta;
    af_status st = build_plan(p->W, (int)cfg->n_mels, kWaves, kMaxSlotsW, 4, p->plan, wfrag, meta);
    if (st == AF_ERR_UNSUPPORTED) {
        // a group of filters must fit the K-steps of ONE wave; few wide filters do not (20 mels over 0 .. 24 kHz):
        // power rows + the dense contraction instead
        p->dense_mel = true;
        std::memset(&p->plan, 0, sizeof(p->plan));
        wfrag.assign(4, 0.0f);
        meta.assign(4, 0);
        last_error_slot().clear();
        st = AF_OK;
    }
    if (st != AF_OK) { delete p; return st; }
    // hop-256 kernel: the same groups dealt to 8 waves, at most 8 slots each; a filterbank that
    // does not fit leaves plan8.n_waves == 0 and the general kernel serves hop 256 as well
    std::vector<float> wfrag8;
    std::vector<int> meta8;
    if (cfg->hop == (uint32_t)v2::kHop && !p->dense_mel) {
        // 6 slots per wave where the filterbank allows it (the instance with the smaller register budget)
        if (build_plan(p->W, (int)cfg->n_mels, v2::kW, 6, 2, p->plan8, wfrag8, meta8, v2::kBinMajor) != AF_OK &&
            build_plan(p->W, (int)cfg->n_mels, v2::kW, 8, 2, p->plan8, wfrag8, meta8, v2::kBinMajor) != AF_OK) {
            std::memset(&p->plan8, 0, sizeof(p->plan8));
            wfrag8.clear();
            meta8.clear();
            last_error_slot().clear();
        }
    }

    // periodic Hann over the frame's n_fft samples; the rest of the 1024-point transform is zero padding
    const int frame_len = (int)cfg->n_fft;
    const auto hann = [frame_len](int i) {
        return i < frame_len ? 0.5 - 0.5 * std::cos(2.0 * kPi * (double)i / (double)frame_len) : 0.0;
    };
    // the same kernel with its mel phase on 4 x 4 x 1 blocks (a third of the matrix work; features.hpp Plan4)
    if (p->plan8.n_waves == v2::kW && cfg->output == AF_OUT_MEL) {
        std::vector<float> wfrag4;
        std::vector<int> lane4;
        p->plan4 = build_plan4(p->W, (int)cfg->n_mels, wfrag4, lane4);
        if (p->plan4.ok) {
            const size_t o_l = wfrag4.size() * sizeof(float);
            st = p->d_mel4.reserve(o_l + lane4.size() * sizeof(int));
            if (st == AF_OK && (hipMemcpy(p->d_mel4.p, wfrag4.data(), o_l, hipMemcpyHostToDevice) != hipSuccess ||
                                hipMemcpy(p->d_mel4.as<char>() + o_l, lane4.data(), lane4.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess))
                st = fail(AF_ERR_HIP, "uploading the 4 x 4 x 1 mel plan failed");
            if (st != AF_OK) { delete p; return st; }
        }
    }
    p->win1024.resize(kNfft);
    for (int i = 0; i < kNfft; ++i) p->win1024[i] = hann(i);
    // constant block: wintab [32][kWinRow] | twtab [32][kTwRow] | wfrag | meta
    // (the two tables are copied verbatim into LDS by every workgroup; rows are
    // padded so that 16 lanes reading 16 rows with ds_read_b128 hit 16 bank groups)
    std::vector<float> wintab(32 * kWinRow, 0.0f);
    for (int n2 = 0; n2 < 32; ++n2)
        for (int n1 = 0; n1 < 32; ++n1) {
            const int i = 32 * n1 + n2;
            const double w = hann(i);
            wintab[n2 * kWinRow + n1] = 0.5f * (float)w;   // the 1/2 of the two-real-signals split, exact
        }
    std::vector<float> twtab(32 * kTwRow, 0.0f);
    for (int n2 = 0; n2 < 32; ++n2)
        for (int k1 = 0; k1 < 32; ++k1) {
            const double a = 2.0 * kPi * (double)(k1 * n2) / (double)kNfft;
            twtab[n2 * kTwRow + 2 * k1] = (float)std::cos(a);
            twtab[n2 * kTwRow + 2 * k1 + 1] = (float)std::sin(a);
        }
    // hop-256 kernel tables: window [64][kWinRow2] (row = lane, its n2; n1 = 0..15: 0.5 hann[64 n1 + n2]),
    // twiddles [64][kTw1Row] (row = lane: (cos, sin)(2 pi n2 k1 / 1024), k1 = 1..15), then the per-lane
    // stage-2 twiddles [64][kLaneTw] that the kernel keeps in registers
    std::vector<float> tabs2(v2::kTabFloats + v2::kLaneTwFloats, 0.0f);
    {
        float* win2 = tabs2.data();
        float* tw1 = win2 + 64 * v2::kWinRow2;
        float* ltw = tw1 + 64 * v2::kTw1Row;
        // rows are indexed by LANE: lane l = 16 c + m runs stage 1 for n2 = c + 4 m (features_v2.hip, T1)
        for (int l = 0; l < 64; ++l) {
            const int n2 = (l >> 4) + 4 * (l & 15);
            for (int n1 = 0; n1 < 16; ++n1) {
                const int i = 64 * n1 + n2;
                const double w = hann(i);
                win2[l * v2::kWinRow2 + n1] = 0.5f * (float)w;
                if (n1 == 0) continue;                                           // n1 plays k1; rows start at k1 = 1
                const double a = 2.0 * kPi * (double)(n1 * n2) / (double)kNfft;
                tw1[l * v2::kTw1Row + 2 * (n1 - 1)] = (float)std::cos(a);
                tw1[l * v2::kTw1Row + 2 * (n1 - 1) + 1] = (float)std::sin(a);
            }
        }
        // stage-2 twiddles per lane (kept in registers): lane (k1, j = l / 16): cos, sin of W32^j, then for
        // b' = 0..3 tan, cos of w = e^{-2 pi i (j + 4 b') / 64} (angle <= 84.4 degrees)
        for (int l = 0; l < 64; ++l) {
            const int j = l >> 4;
            ltw[l * v2::kLaneTw] = (float)std::cos(2.0 * kPi * (double)j / 32.0);
            ltw[l * v2::kLaneTw + 1] = (float)std::sin(2.0 * kPi * (double)j / 32.0);
            for (int bp = 0; bp < 4; ++bp) {
                const double a1 = 2.0 * kPi * (double)(j + 4 * bp) / 64.0;
                ltw[l * v2::kLaneTw + 2 + 2 * bp] = (float)std::tan(a1);
                ltw[l * v2::kLaneTw + 3 + 2 * bp] = (float)std::cos(a1);
            }
        }
    }
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t off_tw = wintab.size() * sizeof(float);
    const size_t off_wf = off_tw + twtab.size() * sizeof(float);
    const size_t off_meta = up16(off_wf + wfrag.size() * sizeof(float));
    const size_t off_tabs2 = up16(off_meta + meta.size() * sizeof(int));
    const size_t off_wf8 = up16(off_tabs2 + tabs2.size() * sizeof(float));
    const size_t off_meta8 = up16(off_wf8 + wfrag8.size() * sizeof(float));
    const size_t total = off_meta8 + meta8.size() * sizeof(int);
    std::vector<char> blob(total, 0);
    std::memcpy(blob.data() + off_tabs2, tabs2.data(), tabs2.size() * sizeof(float));
    if (!wfrag8.empty()) std::memcpy(blob.data() + off_wf8, wfrag8.data(), wfrag8.size() * sizeof(float));
    if (!meta8.empty()) std::memcpy(blob.data() + off_meta8, meta8.data(), meta8.size() * sizeof(int));
    std::memcpy(blob.data(), wintab.data(), off_tw);
    std::memcpy(blob.data() + off_tw, twtab.data(), twtab.size() * sizeof(float));
    std::memcpy(blob.data() + off_wf, wfrag.data(), wfrag.size() * sizeof(float));
    std::memcpy(blob.data() + off_meta, meta.data(), meta.size() * sizeof(int));
    st = p->d_const.reserve(total);
    if (st == AF_OK && hipMemcpy(p->d_const.p, blob.data(), total, hipMemcpyHostToDevice) != hipSuccess)
        st = fail(AF_ERR_HIP, "uploading feature tables failed");
    if (st != AF_OK) { delete p; return st; }
    char* base = p->d_const.as<char>();
    p->d_wintab = reinterpret_cast<const float*>(base);
    p->d_twtab = reinterpret_cast<const float*>(base + off_tw);
    p->d_wfrag = reinterpret_cast<const float*>(base + off_wf);
    p->d_meta = reinterpret_cast<const int*>(base + off_meta);
    p->d_tabs2 = reinterpret_cast<const float*>(base + off_tabs2);
    p->d_wfrag8 = reinterpret_cast<const float*>(base + off_wf8);
    p->d_meta8 = reinterpret_cast<const int*>(base + off_meta8);
    const auto upload_bank = [&](const std::vector<float>& w, int row_len, DevBuf& d_w, DevBuf& d_b) -> af_status {
        return upload_bank_of(w, (int)cfg->n_mels, row_len, d_w, d_b);
    };
    // (the matrix-core DFT kernel's tables are built when a handle asks for it: af_features_set_option(h, "kernel", 3))
    if (p->dense_mel) st = upload_bank(p->W, kBins, p->d_W, p->d_band);
    if (st == AF_OK && cfg->n_fft != (uint32_t)kNfft) st = upload_bank(p->Wn, (int)cfg->n_fft / 2 + 1, p->d_Wn, p->d_bandn);
    if (st != AF_OK) { delete p; return st; }
    *out = p;
    return AF_OK;
}

void af_features_destroy(af_features* p) { delete p; }

size_t af_features_n_frames(const af_features* p, size_t n_pcm_frames)
{
    if (!p || n_pcm_frames < p->cfg.n_fft) return 0;
    return 1 + (n_pcm_frames - p->cfg.n_fft) / p->cfg.hop;
}

size_t af_features_row_len(const af_features* p)
{
    if (!p) return 0;
    return p->cfg.output == AF_OUT_POWER ? (size_t)(p->cfg.n_fft / 2 + 1) : (size_t)p->cfg.n_mels;
}

af_status af_features_set_batch(af_features* p, const int64_t* clip_offsets, size_t n_clips,
                                size_t* total_frames_out)
{
    AF_REQUIRE(p && (clip_offsets || n_clips == 0), "af_features_set_batch: null argument");
    AF_REQUIRE(n_clips < (size_t)1 << 30, "too many clips");
    std::vector<int64_t> start(n_clips), len(n_clips);
    for (size_t c = 0; c < n_clips; ++c) {
        start[c] = clip_offsets[c];
        len[c] = clip_offsets[c + 1] - clip_offsets[c];
        if (len[c] < 0) return fail(AF_ERR_INVALID_ARG, "clip offsets must be non-decreasing");
    }
    p->n_clips = n_clips;
    AF_TRY(upload_tables(p, start, len));
    if (total_frames_out) *total_frames_out = (size_t)p->total_frames;
    return AF_OK;
}

af_status af_features_set_batch_uniform(af_features* p, size_t n_clips, size_t frames_per_clip,
                                        size_t* total_frames_out)
{
    AF_REQUIRE(p, "af_features_set_batch_uniform: null handle");
    AF_REQUIRE(n_clips < (size_t)1 << 30, "too many clips");
    std::vector<int64_t> start(n_clips), len(n_clips, (int64_t)frames_per_clip);
    for (size_t c = 0; c < n_clips; ++c) start[c] = (int64_t)(c * frames_per_clip);
    p->n_clips = n_clips;
    AF_TRY(upload_tables(p, start, len));
    if (total_frames_out) *total_frames_out = (size_t)p->total_frames;
    return AF_OK;
}

af_status af_features_set_batch_spans(af_features* p, const int64_t* starts, const int64_t* lengths,
                                      size_t n_spans, size_t* total_frames_out)
{
    AF_REQUIRE(p && ((starts && lengths) || n_spans == 0), "af_features_set_batch_spans: null argument");
    AF_REQUIRE(n_spans < (size_t)1 << 30, "too many spans");
    std::vector<int64_t> start(starts, starts + n_spans), len(lengths, lengths + n_spans);
    p->n_clips = n_spans;
    AF_TRY(upload_tables(p, start, len));
    if (total_frames_out) *total_frames_out = (size_t)p->total_frames;
    return AF_OK;
}

static af_status run_device_impl(af_features* p, const void* pcm_dev, float* mel_dev, void* stream,
                                 int64_t u_begin = 0, int64_t u_count = -1, int64_t f_begin = 0, int64_t f_end = -1,
                                 int wg_override = -1);

// remember the launch: set_batch* must not rewrite the tables under it (not inside a stream capture: the replayed
// graph owns its own ordering)
static af_status note_launch(af_features* p, hipStream_t stream)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
    if (cs == hipStreamCaptureStatusNone) {
        if (!p->last_run) AF_HIP(hipEventCreateWithFlags(&p->last_run, hipEventDisableTiming));
        AF_HIP(hipEventRecord(p->last_run, stream));
        p->run_pending = true;
    }
    return AF_OK;
}

af_status af_features_run_device(af_features* p, const void* pcm_dev, float* mel_dev, void* stream)
{
    AF_REQUIRE(p, "af_features_run_device: null handle");
    AF_TRY(run_device_impl(p, pcm_dev, mel_dev, stream));
    if (p->total_units == 0) return AF_OK;
    return note_launch(p, as_stream(stream));
}

// mel[f][m] = sum_k W[m][k] P[f][k] over the filter's own bins, 8 power rows per workgroup through LDS: the contraction
// for filterbanks the banded MFMA plan does not cover (dense_mel); f32 FMA chain in bin order
__global__ __launch_bounds__(256) void dense_mel_kernel(const float* __restrict__ power, float* __restrict__ mel,
                                                        const float* __restrict__ W, const int* __restrict__ band,
                                                        int n_mels, int64_t n_frames, int log_mel, int row_len)
{
    constexpr int F = 8;
    __shared__ float rows[F][kBins + 3];
    for (int64_t f0 = (int64_t)blockIdx.x * F; f0 < n_frames; f0 += (int64_t)gridDim.x * F) {
        __syncthreads();
        for (int i = threadIdx.x; i < F * row_len; i += blockDim.x) {
            const int f = i / row_len, k = i - f * row_len;
            rows[f][k] = f0 + f < n_frames ? power[(f0 + f) * row_len + k] : 0.0f;
        }
        __syncthreads();
        for (int o = threadIdx.x; o < F * n_mels; o += blockDim.x) {
            const int f = o / n_mels, m = o - f * n_mels;
            if (f0 + f >= n_frames) continue;
            const float* w = W + (size_t)m * row_len;
            float acc = 0.0f;
            for (int k = band[2 * m]; k < band[2 * m + 1]; ++k) acc = __builtin_fmaf(w[k], rows[f][k], acc);
            if (log_mel) acc = __log10f(fmaxf(acc, 1e-10f));
            mel[(f0 + f) * n_mels + m] = acc;
        }
    }
}

// n_fft < 1024, power output: bins 0, s, 2 s ... of the 1024-point rows are the n_fft-point spectrum
__global__ void decimate_power_kernel(const float* __restrict__ power, float* __restrict__ out, int stride, int bins_n,
                                      int64_t n_frames)
{
    const int64_t total = n_frames * bins_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / bins_n;
        const int k = (int)(i - f * bins_n);
        out[i] = power[f * kBins + (int64_t)k * stride];
    }
}

// units [u_begin, u_begin + u_count) of the batch (u_count < 0: all of it), which read PCM frames
// [f_begin, f_end) (f_end < 0: the whole batch)
static af_status run_device_impl(af_features* p, const void* pcm_dev, float* mel_dev, void* stream,
                                 int64_t u_begin, int64_t u_count, int64_t f_begin, int64_t f_end, int wg_override)
{
    if (u_count < 0) u_count = p->total_units - u_begin;
    if (p->total_units == 0 || u_count <= 0) return AF_OK;
    AF_REQUIRE(pcm_dev && mel_dev, "af_features_run_device: null buffer");
    if (p->cfg.channels > 2) {                   // to_mono first (capture.rs:30-42), then the mono kernel
        if (f_end < 0) f_end = p->total_pcm_frames;
        const size_t ch = p->cfg.channels;
        AF_TRY(af_to_mono_device(static_cast<const float*>(pcm_dev) + (size_t)f_begin * ch, (size_t)(f_end - f_begin) * ch,
                                 (uint16_t)ch, p->d_mono.as<float>() + f_begin, stream));
        pcm_dev = p->d_mono.p;
    }
    if (p->dft) {
        AF_REQUIRE(p->units_ok, "features: the batch has no unit table");
        dft::Params q{};
        q.pcm = pcm_dev;
        q.out = mel_dev;
        q.tfrag = p->d_dft.as<float>();
        q.wfrag = reinterpret_cast<const float*>(p->d_dft.as<char>() + p->dft_off_w);
        q.wmask = reinterpret_cast<const uint32_t*>(p->d_dft.as<char>() + p->dft_off_mask);
        q.units = p->d_units.as<v2::UnitDesc>() + u_begin;
        q.total_units = u_count;
        q.queue = p->opt_dft_queue == 0 ? nullptr : reinterpret_cast<unsigned*>(p->d_dft.as<char>() + p->dft_off_queue);
        if (q.queue) AF_HIP(hipMemsetAsync(q.queue, 0, sizeof(unsigned), as_stream(stream)));
        q.hop = (int)p->cfg.hop;
        q.kg = p->dft_kg;
        q.nb = p->dft_nb;
        q.n_bins = (int)p->cfg.n_fft / 2 + 1;
        q.n_mels = (int)p->cfg.n_mels;
        q.nmt = p->dft_nmt;
        q.n_fft = (int)p->cfg.n_fft;
        q.nh = (int)p->cfg.n_fft / 2;
        q.row_stride = p->dft_row_stride;
        q.tile_stride = p->dft_tile_stride;
        q.out_power = p->cfg.output == AF_OUT_POWER;
        q.log_mel = (int)p->cfg.log_mel;
        q.stamps = p->d_stamps.as<unsigned long long>();
        const size_t smem = (size_t)p->dft_tile_stride * sizeof(float);
        // (at most two waves per SIMD: the kernel holds more than 128 registers)
        // (LDS is handed out in granules of 1 280 bytes: 128 per CU)
        const size_t alloc = (smem + 1279) / 1280 * 1280;
        int per_cu = (int)((160 * 1024) / alloc);
        const int max_waves = p->dft_nmt <= 5 ? 12 : 8;              // (the 5-tile instance holds 160 registers: three waves per SIMD)
        if (per_cu > max_waves / p->dft_waves) per_cu = max_waves / p->dft_waves;
        // (whole waves per SIMD: five two-wave workgroups put ten waves on four SIMDs and every wave carries the same number
        // of units -- 1.97 ms against 1.89 with four on 400 / 160)
        if (p->dft_waves == 2 && per_cu > 2 && (per_cu & 1)) per_cu -= 1;
        if (p->opt_wg_per_cu > 0 && (size_t)p->opt_wg_per_cu * smem <= 160 * 1024) per_cu = p->opt_wg_per_cu;   // (diagnostics)
        if (per_cu < 1) per_cu = 1;
        int64_t grid = (int64_t)p->n_cus * per_cu;
        if (grid > u_count) grid = u_count;
        return dft::launch(q, kernel_channels(p->cfg), p->cfg.sample_format == AF_PCM_I16,
                           p->cfg.channel_mode == AF_CH_SEPARATE, p->dft_waves, (int)grid, as_stream(stream));
    }
    // dense_mel: the kernels write power rows, the contraction with W follows as its own launch
    const bool dense = p->dense_mel && p->cfg.output != AF_OUT_POWER;
    // n_fft < 1024 with power output: 1024-point rows first, then every (1024 / n_fft)-th bin
    const bool thin = p->cfg.output == AF_OUT_POWER && p->cfg.n_fft != (uint32_t)kNfft;
    if (dense || thin) AF_REQUIRE(u_begin == 0 && u_count == p->total_units, "two-launch feature path: whole batches only");
    float* const mel_out = mel_dev;
    if (dense || thin) mel_dev = p->d_power.as<float>();
    const auto finish = [&]() -> af_status {
        if (!(dense || thin) || p->total_frames == 0) return AF_OK;
        if (thin) {
            const int bins_n = (int)p->cfg.n_fft / 2 + 1;
            const int64_t total = p->total_frames * bins_n;
            const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)p->n_cus * 32);
            hipLaunchKernelGGL(decimate_power_kernel, dim3(grid), dim3(256), 0, as_stream(stream), p->d_power.as<float>(),
                               mel_out, kNfft / (int)p->cfg.n_fft, bins_n, p->total_frames);
            AF_HIP(hipGetLastError());
            return AF_OK;
        }
        const int64_t blocks = (p->total_frames + 7) / 8;
        const unsigned grid = (unsigned)std::min<int64_t>(blocks, (int64_t)p->n_cus * 16);
        hipLaunchKernelGGL(dense_mel_kernel, dim3(grid), dim3(256), 0, as_stream(stream), p->d_power.as<float>(), mel_out,
                           p->d_W.as<float>(), p->d_band.as<int>(), (int)p->cfg.n_mels, p->total_frames, (int)p->cfg.log_mel, kBins);
        AF_HIP(hipGetLastError());
        return AF_OK;
    };
    Params prm{};
    prm.pcm = pcm_dev;
    prm.mel = mel_dev;
    prm.wintab = p->d_wintab;
    prm.twtab = p->d_twtab;
    prm.wfrag = p->d_wfrag;
    prm.plan = p->plan;
    prm.meta = p->d_meta;
    const int64_t* tab = p->d_tables.as<int64_t>();
    prm.pcm_off = tab;
    prm.pcm_len = tab + p->n_clips;
    prm.out_off = tab + 2 * p->n_clips;
    prm.unit_off = tab + 2 * p->n_clips + (p->n_clips + 1);
    prm.total_units = u_count;
    prm.unit_begin = u_begin;
    prm.n_clips = (int)p->n_clips;
    prm.hop = (int)p->cfg.hop;
    prm.stamps = p->d_stamps.as<unsigned long long>();
    prm.out_power = p->cfg.output == AF_OUT_POWER || dense;
    prm.log_mel = (int)p->cfg.log_mel;
    prm.separate = p->cfg.channel_mode == AF_CH_SEPARATE;
    prm.aligned16 = (p->offsets_aligned && (reinterpret_cast<uintptr_t>(pcm_dev) % 16 == 0)) ? 1 : 0;
    const int wg_knob = wg_override > 0 ? wg_override
                        : p->opt_wg_per_cu >= 0 ? p->opt_wg_per_cu : debug_knob(kKnobFeaturesWgPerCu);
    // which fused kernel: 0 = the default (hop 256 on dword-aligned clips: features_v2; else the general kernel), 1 = the
    // general kernel, 2 = features_v2 where it applies (else the general kernel), 3 = features_v3, the opt-in matrix-core
    // DFT kernel.  Asked for by THIS handle (af_features_set_option) a 3 that cannot be served is an error that names the
    // condition; the process-wide switch (af_debug_set) only prefers it: handles it does not fit, or whose tables were
    // never built, run their default kernel.
    const bool own = p->opt_kernel >= 0;
    int ksel = own ? p->opt_kernel : debug_knob(kKnobFeaturesKernel);
    // (round 5: the hop-256 kernel's plain global loads take 2-byte-aligned addresses as well -- the chip runs global memory in
    // unaligned access mode, the DFT kernel's i16 row loads rely on it -- so mono i16 clips at odd frame offsets stay on it;
    // the matrix-core DFT kernel reads through range-checked buffer loads and keeps its dword rule)
    const bool aligned4_now = p->offsets_aligned4 && reinterpret_cast<uintptr_t>(pcm_dev) % 4 == 0;
    const bool hop256_ok = p->units_ok && reinterpret_cast<uintptr_t>(pcm_dev) % (kernel_bytes_per_frame(p->cfg) == 2 ? 2 : 4) == 0;
    const bool v3_applies = hop256_ok && aligned4_now && p->v3_ok && !prm.out_power && !prm.log_mel;
    if (ksel == 3 && !v3_applies) {
        if (!own) {
            ksel = 0;
        } else {
            const char* why = p->cfg.hop != (uint32_t)v2::kHop                   ? "the hop is not 256"
                              : prm.out_power                                     ? "the output is the power spectrogram (or the filterbank takes the dense contraction)"
                              : prm.log_mel                                       ? "log-mel output is not among its instances"
                              : !p->v3_ok                                         ? "the filterbank is outside its banded mel plan"
                              : !(p->offsets_aligned4 && reinterpret_cast<uintptr_t>(pcm_dev) % 4 == 0)
                                  ? "a clip does not start on a 4-byte boundary (mono i16 clips at odd offsets)"
                                  : "the batch has no unit table (empty, or beyond 2^26 units)";
            return fail(AF_ERR_UNSUPPORTED, "features: the matrix-core DFT kernel (kernel = 3) cannot serve this handle: %s", why);
        }
    }
    if (v3_applies && ksel == 3) {                       // (opt-in until it is the faster one)
        v3::Params q{};
        q.pcm = pcm_dev;
        q.out = mel_dev;
        q.tabs = p->d_tabs3.as<uint4>();
        q.wfrag = p->d_wfrag8b;
        q.meta = p->d_meta8b;
        q.units = p->d_units.as<v2::UnitDesc>() + u_begin;
        q.plan = p->plan8b;
        q.total_units = u_count;
        q.log_mel = prm.log_mel;
        q.stamps = prm.stamps;
        q.stamp_wave0 = (prm.stamps && std::getenv("AF_V3_STAMP_W0") && std::atoi(std::getenv("AF_V3_STAMP_W0")) >= 4) ? 4 : 0;   // diagnostics: waves 0-3 or 4-7
        int64_t grid3 = (int64_t)p->n_cus;               // one 8-wave workgroup per CU (256 registers per lane, 131 KB of LDS)
        if (grid3 > u_count) grid3 = u_count;
        AF_TRY(v3::launch(q, kernel_channels(p->cfg), p->cfg.sample_format == AF_PCM_I16, prm.separate != 0,
                          (int)grid3, as_stream(stream)));
        return finish();
    }
    if (hop256_ok && ksel != 1 && (p->plan8.n_waves == v2::kW || p->dense_mel)) {
        // hop 256, dword-aligned clips: the 8-wave kernel, two workgroups per CU
        v2::Params q{};
        q.pcm = pcm_dev;
        q.out = mel_dev;
        q.tabs = p->d_tabs2;
        q.wfrag = p->d_wfrag8;
        q.meta = p->d_meta8;
        q.units = p->d_units.as<v2::UnitDesc>() + u_begin;
        q.plan = p->plan8;
        q.total_units = u_count;
        q.out_power = prm.out_power;
        q.log_mel = prm.log_mel;
        q.stamps = prm.stamps;
        if (p->plan4.ok && p->opt_mel4 != 0 && !prm.out_power) {   // (the per-phase stamps build has no 4 x 4 x 1 instance: launch_fmt)
            q.wfrag4 = p->d_mel4.as<float>();
            q.lane4 = reinterpret_cast<const int*>(p->d_mel4.as<char>() + (size_t)2 * v2::kM4Quads * 64 * 4 * sizeof(float));
            q.m4_quads[0] = p->plan4.quads[0];
            q.m4_quads[1] = p->plan4.quads[1];
        }
        int64_t grid2 = (int64_t)p->n_cus * 2;
        if (wg_knob == 1) grid2 = (int64_t)p->n_cus;
        if (grid2 > u_count) grid2 = u_count;
        // the pool of the work distribution: 1 / 2^kPoolShift of every range, whole chunks (ranges of at least 64 units)
        if (p->opt_pool != 0 && u_count / grid2 >= 64) {
            const int pool = (int)((u_count / grid2) >> v2::kPoolShift) / v2::kPoolChunk * v2::kPoolChunk;
            if (pool > 0 && p->d_queue.p && grid2 <= 4096) {    // (allocated with the batch tables: nothing is allocated on the launch path)
                // [0]: the global ticket counter; [64 ...]: one deque word per range (pool mode 2)
                AF_HIP(hipMemsetAsync(p->d_queue.p, 0, (64 + (size_t)grid2) * sizeof(unsigned), as_stream(stream)));
                q.queue = p->d_queue.as<unsigned>();
                q.pool_units = pool;
            }
        }
        AF_TRY(v2::launch(q, kernel_channels(p->cfg), p->cfg.sample_format == AF_PCM_I16, prm.separate != 0,
                          (int)grid2, as_stream(stream)));
        return finish();
    }
    const size_t smem = features_smem_bytes(prm.hop);
    AF_REQUIRE(smem <= 160 * 1024, "hop %d needs %zu bytes of LDS", prm.hop, smem);
    // persistent grid: as many workgroups as stay resident (LDS-limited, at most 2 per CU)
    int per_cu = (int)((160 * 1024) / smem);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    if (wg_knob >= 1 && wg_knob <= 8) per_cu = wg_knob;
    int64_t grid = (int64_t)p->n_cus * per_cu;
    if (grid > u_count) grid = u_count;
    AF_TRY(launch_features(prm, kernel_channels(p->cfg), p->cfg.sample_format == AF_PCM_I16,
                           (int)grid, smem, as_stream(stream)));
    return finish();
}

af_status af_features_run(af_features* p, const void* pcm, size_t pcm_len, float* mel, size_t mel_cap)
{
    AF_REQUIRE(p, "af_features_run: null handle");
    const size_t need_in = (size_t)p->total_pcm_frames * p->cfg.channels;
    if (pcm_len < need_in)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "pcm buffer holds %zu samples, the batch describes %zu", pcm_len,
                    need_in);
    const size_t n_out = (size_t)p->total_frames * af_features_row_len(p);
    if (mel_cap < n_out)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "mel buffer holds %zu floats, need %zu", mel_cap, n_out);
    if (n_out == 0) return AF_OK;
    AF_REQUIRE(pcm && mel, "af_features_run: null buffer");
    const size_t bpf = bytes_per_frame(p->cfg);
    const size_t in_bytes = (size_t)p->total_pcm_frames * bpf;
    AF_TRY(p->d_pcm.reserve(in_bytes));
    AF_TRY(p->d_mel.reserve(n_out * sizeof(float)));
    // Chunks of whole clips, about 32 MB of PCM each: H2D of chunk k + 1, the kernel on chunk k and
    // D2H of chunk k - 1 run on three streams (truly concurrent when the caller's buffers are
    // page-locked; with pageable memory the runtime stages every copy and the chunks serialise).
    struct Chunk { int64_t f0, f1, r0, r1, u0, u1; };
    std::vector<Chunk> chunks;
    const size_t n = p->n_clips;
    // (pageable memory goes through the runtime's own staging, which is fastest as ONE copy)
    hipPointerAttribute_t pa;
    const bool pinned = hipPointerGetAttributes(&pa, pcm) == hipSuccess && pa.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    const bool two_launches = !p->dft && (p->dense_mel || (p->cfg.output == AF_OUT_POWER && p->cfg.n_fft != (uint32_t)kNfft));
    if (pinned && p->contiguous && n >= 2 && in_bytes >= ((size_t)64 << 20) && !two_launches) {
        const int64_t target = (int64_t)(((size_t)32 << 20) / bpf);
        size_t c0 = 0;
        while (c0 < n) {
            size_t c1 = c0 + 1;
            while (c1 < n && p->h_start[c1 - 1] + p->h_len[c1 - 1] - p->h_start[c0] < target) ++c1;
            // a separate-channel clip is two table entries over the same PCM: keep them together
            if (p->cfg.channel_mode == AF_CH_SEPARATE && (c1 & 1)) ++c1;
            if (c1 > n) c1 = n;
            chunks.push_back({p->h_start[c0], p->h_start[c1 - 1] + p->h_len[c1 - 1], p->h_out_off[c0], p->h_out_off[c1],
                              p->h_unit_off[c0], p->h_unit_off[c1]});
            c0 = c1;
        }
    }
    if (chunks.size() < 2) {
        AF_HIP(hipMemcpy(p->d_pcm.p, pcm, in_bytes, hipMemcpyHostToDevice));
        AF_TRY(af_features_run_device(p, p->d_pcm.p, p->d_mel.as<float>(), nullptr));
        AF_HIP(hipMemcpy(mel, p->d_mel.p, n_out * sizeof(float), hipMemcpyDeviceToHost));
        return AF_OK;
    }
    if (!p->s_in) {
        AF_HIP(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
        AF_HIP(hipStreamCreateWithFlags(&p->s_run, hipStreamNonBlocking));
        AF_HIP(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    }
    while (p->ev_in.size() < chunks.size()) {
        // (the pair is pushed only when both exist: the two vectors stay the same length)
        hipEvent_t a = nullptr, b = nullptr;
        AF_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        if (hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
            (void)hipEventDestroy(a);
            return fail(AF_ERR_HIP, "hipEventCreateWithFlags failed");
        }
        p->ev_in.push_back(a);
        p->ev_run.push_back(b);
    }
    const size_t row = af_features_row_len(p);
    const char* src = static_cast<const char*>(pcm);
    // The first error stops the loop, but nothing returns before the three streams have drained: the copies still in
    // flight read and write the CALLER's buffers, which are likely to be freed after an error.
    af_status st = AF_OK;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && st == AF_OK)
            st = fail(e == hipErrorOutOfMemory ? AF_ERR_OUT_OF_MEMORY : AF_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    for (size_t k = 0; k < chunks.size() && st == AF_OK; ++k) {
        const Chunk& c = chunks[k];
        if (!hip_ok(hipMemcpyAsync(p->d_pcm.as<char>() + c.f0 * bpf, src + c.f0 * bpf, (size_t)(c.f1 - c.f0) * bpf,
                                   hipMemcpyHostToDevice, p->s_in), "hipMemcpyAsync") ||
            !hip_ok(hipEventRecord(p->ev_in[k], p->s_in), "hipEventRecord") ||
            !hip_ok(hipStreamWaitEvent(p->s_run, p->ev_in[k], 0), "hipStreamWaitEvent"))
            break;
        const af_status ks = run_device_impl(p, p->d_pcm.p, p->d_mel.as<float>(), p->s_run, c.u0, c.u1 - c.u0, c.f0, c.f1);
        if (ks != AF_OK) { st = ks; break; }
        if (!hip_ok(hipEventRecord(p->ev_run[k], p->s_run), "hipEventRecord") ||
            !hip_ok(hipStreamWaitEvent(p->s_out, p->ev_run[k], 0), "hipStreamWaitEvent"))
            break;
        if (c.r1 > c.r0)
            hip_ok(hipMemcpyAsync(mel + (size_t)c.r0 * row, p->d_mel.as<float>() + (size_t)c.r0 * row,
                                  (size_t)(c.r1 - c.r0) * row * sizeof(float), hipMemcpyDeviceToHost, p->s_out),
                   "hipMemcpyAsync");
    }
    (void)hipStreamSynchronize(p->s_in);
    (void)hipStreamSynchronize(p->s_run);
    hip_ok(hipStreamSynchronize(p->s_out), "hipStreamSynchronize");
    return st;
}

af_status af_pipeline_run(const af_batch_desc* batch, const void* pcm, size_t pcm_len, float* out,
                          size_t out_cap, size_t* total_rows_out)
{
    AF_REQUIRE(batch && (batch->clip_offsets || batch->n_clips == 0), "af_pipeline_run: null batch");
    af_features* p = nullptr;
    AF_TRY(af_features_create(&batch->cfg, &p));
    size_t total = 0;
    af_status st = af_features_set_batch(p, batch->clip_offsets, batch->n_clips, &total);
    if (total_rows_out) *total_rows_out = total;
    if (st == AF_OK && !(out == nullptr && out_cap == 0)) st = af_features_run(p, pcm, pcm_len, out, out_cap);
    af_features_destroy(p);
    return st;
}

af_status af_features_mel_weights(const af_features* p, float* W, size_t cap)
{
    AF_REQUIRE(p && W, "af_features_mel_weights: null argument");
    if (cap < p->Wn.size())
        return fail(AF_ERR_BUFFER_TOO_SMALL, "need %zu floats", p->Wn.size());
    std::memcpy(W, p->Wn.data(), p->Wn.size() * sizeof(float));
    return AF_OK;
}

const char* af_features_kernel_name(void) { return "af_features_v2_kernel"; }   // the BASELINE configuration (hop 256, mel output); power output: af_features_v2_kernel; other hops: af_features_kernel

// diagnostics, host only: the mel plans -- info[81] = {groups, K-steps, slots of wave 0..3} of the
// general kernel, then {waves (0: not in use), groups, K-steps, slots of wave 0..7} of the hop-256 kernel,
// then its slot_end words [8 waves][8 slots] (0: the group continues; else 0x10000 | filters << 8 | first filter)
// The opt-in hop-256 kernel with the DFT on the matrix cores (features_v3.hip): mel output through an 8-wave banded plan
// whose slots walk bins 4 apart per step (its power rows are bin-major) + the f16-pair fragment tables of the two DFT
// stages (143 KB).  Built when a handle asks for that kernel, never for the handles that do not (every product handle,
// the ones inside chains and streams included, used to pay for it at create).
static af_status build_v3(af_features* p)
{
    if (p->v3_tried) return AF_OK;
    p->v3_tried = true;
    const af_features_config* cfg = &p->cfg;
    if (!(!p->dft && cfg->hop == (uint32_t)v2::kHop && !p->dense_mel && cfg->output == AF_OUT_MEL)) return AF_OK;
    std::vector<float> wfrag8b;
    std::vector<int> meta8b;
    // groups of whole filter quads first (one 16-byte store per group and lane: the kernel's QUADS instances), then the
    // finer cut
    if (build_plan(p->W, (int)cfg->n_mels, v3::kW, 6, 4, p->plan8b, wfrag8b, meta8b, false) != AF_OK &&
        build_plan(p->W, (int)cfg->n_mels, v3::kW, 7, 4, p->plan8b, wfrag8b, meta8b, false) != AF_OK &&
        build_plan(p->W, (int)cfg->n_mels, v3::kW, 8, 4, p->plan8b, wfrag8b, meta8b, false) != AF_OK &&
        build_plan(p->W, (int)cfg->n_mels, v3::kW, 6, 2, p->plan8b, wfrag8b, meta8b, false) != AF_OK &&
        build_plan(p->W, (int)cfg->n_mels, v3::kW, 8, 2, p->plan8b, wfrag8b, meta8b, false) != AF_OK) {
        std::memset(&p->plan8b, 0, sizeof(p->plan8b));
        last_error_slot().clear();
        return AF_OK;
    }
    std::vector<double> win(kNfft);
    for (int i = 0; i < kNfft; ++i) win[i] = p->win1024[i];
    std::vector<uint4> tabs3(v3::kTabVec);
    v3::build_tables(win.data(), tabs3.data());
    const size_t o_wf = tabs3.size() * sizeof(uint4), o_meta = o_wf + wfrag8b.size() * sizeof(float);
    AF_TRY(p->d_tabs3.reserve(o_meta + meta8b.size() * sizeof(int)));
    char* b3 = p->d_tabs3.as<char>();
    if (hipMemcpy(b3, tabs3.data(), o_wf, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b3 + o_wf, wfrag8b.data(), wfrag8b.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b3 + o_meta, meta8b.data(), meta8b.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail(AF_ERR_HIP, "uploading the DFT fragment tables failed");
    p->d_wfrag8b = reinterpret_cast<const float*>(b3 + o_wf);
    p->d_meta8b = reinterpret_cast<const int*>(b3 + o_meta);
    p->v3_ok = true;
    return AF_OK;
}

af_status af_features_set_option(af_features* p, const char* key, int value)
{
    AF_REQUIRE(p && key, "af_features_set_option: null argument");
    if (std::strcmp(key, "kernel") == 0) {
        if (value == 3) AF_TRY(build_v3(p));
        p->opt_kernel = value < 0 ? -1 : value;
    }
    else if (std::strcmp(key, "wg_per_cu") == 0) p->opt_wg_per_cu = value < 0 ? -1 : value;
    else if (std::strcmp(key, "dft_waves") == 0) {                                            // (diagnostics) waves per workgroup of the matrix-product DFT kernel
        AF_REQUIRE(p->dft && (value == 1 || value == 2), "dft_waves: 1 or 2 on a handle whose n_fft is not 1024 / 512 / 256");
        p->dft_waves = value;
    }
    else if (std::strcmp(key, "pool") == 0) p->opt_pool = value < 0 ? -1 : (value != 0);
    else if (std::strcmp(key, "dft_queue") == 0) p->opt_dft_queue = value < 0 ? -1 : (value != 0);
    else if (std::strcmp(key, "mel4") == 0) p->opt_mel4 = value < 0 ? -1 : (value != 0);    // 0: the 16 x 16 x 4 mel plan (diagnostics)
    else return fail(AF_ERR_INVALID_ARG, "af_features_set_option: unknown option '%s'", key);
    return AF_OK;
}

int af_features_kernel_describe(const af_features* p, int32_t* info)
{
    if (!p || !info) return 0;
    for (int i = 0; i < 12; ++i) info[i] = 0;
    info[0] = p->dft ? 4 : (p->cfg.hop == (uint32_t)v2::kHop && (p->plan8.n_waves == v2::kW || p->dense_mel)) ? 2 : 1;
    info[1] = p->plan4.ok ? 1 : 0;
    info[2] = p->plan4.quads[0]; info[3] = p->plan4.quads[1];
    info[4] = p->plan4.n_pieces[0]; info[5] = p->plan4.n_pieces[1];
    if (p->dft) {
        info[6] = p->dft_kg; info[7] = p->dft_nb; info[8] = p->dft_nmt; info[9] = p->dft_waves;
        info[10] = (int32_t)((size_t)p->dft_tile_stride * sizeof(float));
    }
    info[11] = p->plan4.read_conflict_score[0] + p->plan4.read_conflict_score[1];
    return 1;
}

int af_features_plan_describe(const af_features* p, int32_t* info)
{
    if (!p || !info) return 0;
    info[0] = p->plan.n_groups;
    info[1] = p->plan.n_steps;
    for (int w = 0; w < kWaves; ++w) info[2 + w] = p->plan.wave_nslots[w];
    info[6] = p->plan8.n_waves;
    info[7] = p->plan8.n_groups;
    info[8] = p->plan8.n_steps;
    for (int w = 0; w < v2::kW; ++w) info[9 + w] = p->plan8.wave_nslots[w];
    for (int w = 0; w < v2::kW; ++w)
        for (int q = 0; q < 8; ++q) info[17 + 8 * w + q] = p->plan8.slot_end[w][q];
    return 1;
}

// Diagnostics: enable == 1 allocates/zeroes the in-kernel phase stamps (subsequent runs
// record them, slightly slower); out[32] (optional) receives [wave][phase] cycle sums.
af_status af_features_debug_stamps(af_features* p, int enable, unsigned long long* out)
{
    AF_REQUIRE(p, "af_features_debug_stamps: null handle");
    if (out && p->d_stamps.p) {
        AF_HIP(hipDeviceSynchronize());
        AF_HIP(hipMemcpy(out, p->d_stamps.p, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (enable) {
        AF_TRY(p->d_stamps.reserve(32 * sizeof(unsigned long long)));
        AF_HIP(hipMemset(p->d_stamps.p, 0, 32 * sizeof(unsigned long long)));
    } else {
        p->d_stamps.release();
    }
    return AF_OK;
}

// ---- Stft: the fused kernel with AF_OUT_POWER -----------------------------------------------
struct af_stft {
    af_features* f = nullptr;
};

void af_stft_default_config(af_stft_config* cfg)
{
    if (!cfg) return;
    cfg->channels = 1;
    cfg->n_fft = kNfft;
    cfg->hop = 256;
    cfg->sample_format = AF_PCM_F32;
    cfg->device = -1;
}

af_status af_stft_create(const af_stft_config* cfg, af_stft** out)
{
    AF_REQUIRE(cfg && out, "af_stft_create: null argument");
    *out = nullptr;
    af_features_config fc;
    af_features_default_config(&fc);
    fc.sample_rate = 16000;                 // only shapes the (unused) filterbank
    fc.channels = cfg->channels;
    fc.n_fft = cfg->n_fft;
    fc.hop = cfg->hop;
    fc.sample_format = cfg->sample_format;
    fc.device = cfg->device;
    fc.output = AF_OUT_POWER;
    af_stft* s = new (std::nothrow) af_stft();
    if (!s) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    const af_status st = af_features_create(&fc, &s->f);
    if (st != AF_OK) { delete s; return st; }
    *out = s;
    return AF_OK;
}

void af_stft_destroy(af_stft* s)
{
    if (!s) return;
    af_features_destroy(s->f);
    delete s;
}

size_t af_stft_n_frames(const af_stft* s, size_t n_pcm_frames)
{
    return s ? af_features_n_frames(s->f, n_pcm_frames) : 0;
}
size_t af_stft_n_bins(const af_stft* s) { return s ? af_features_row_len(s->f) : 0; }

af_status af_stft_set_batch_uniform(af_stft* s, size_t n_clips, size_t frames_per_clip,
                                    size_t* total_frames_out)
{
    AF_REQUIRE(s, "af_stft_set_batch_uniform: null handle");
    return af_features_set_batch_uniform(s->f, n_clips, frames_per_clip, total_frames_out);
}
af_status af_stft_set_batch(af_stft* s, const int64_t* clip_offsets, size_t n_clips,
                            size_t* total_frames_out)
{
    AF_REQUIRE(s, "af_stft_set_batch: null handle");
    return af_features_set_batch(s->f, clip_offsets, n_clips, total_frames_out);
}
af_status af_stft_power_device(af_stft* s, const void* pcm_dev, float* power_dev, void* stream)
{
    AF_REQUIRE(s, "af_stft_power_device: null handle");
    return af_features_run_device(s->f, pcm_dev, power_dev, stream);
}
af_status af_stft_power(af_stft* s, const void* pcm, size_t pcm_len, float* power, size_t power_cap)
{
    AF_REQUIRE(s, "af_stft_power: null handle");
    return af_features_run(s->f, pcm, pcm_len, power, power_cap);
}

// ---- MelFilterbank: weights + the stand-alone contraction ------------------------------------
struct af_mel {
    af_features* f = nullptr;            // owns W, the MFMA plan and its device tables
    DevBuf d_in, d_out;
};

void af_mel_default_config(af_mel_config* cfg)
{
    if (!cfg) return;
    cfg->sample_rate = 16000;
    cfg->n_fft = kNfft;
    cfg->n_mels = 80;
    cfg->f_min = 0.0f;
    cfg->f_max = 0.0f;
    cfg->mel_scale = AF_MEL_HTK;
    cfg->mel_norm = AF_MEL_NORM_NONE;
    cfg->log_mel = 0;
    cfg->device = -1;
}

af_status af_mel_create(const af_mel_config* cfg, af_mel** out)
{
    AF_REQUIRE(cfg && out, "af_mel_create: null argument");
    *out = nullptr;
    af_features_config fc;
    af_features_default_config(&fc);
    fc.sample_rate = cfg->sample_rate;
    fc.channels = 1;
    fc.n_fft = cfg->n_fft;
    fc.n_mels = cfg->n_mels;
    fc.f_min = cfg->f_min;
    fc.f_max = cfg->f_max;
    fc.mel_scale = cfg->mel_scale;
    fc.mel_norm = cfg->mel_norm;
    fc.log_mel = cfg->log_mel;
    fc.device = cfg->device;
    af_mel* m = new (std::nothrow) af_mel();
    if (!m) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    const af_status st = af_features_create(&fc, &m->f);
    if (st != AF_OK) { delete m; return st; }
    *out = m;
    return AF_OK;
}

void af_mel_destroy(af_mel* m)
{
    if (!m) return;
    af_features_destroy(m->f);
    delete m;
}

af_status af_mel_weights(const af_mel* m, float* W, size_t cap)
{
    AF_REQUIRE(m, "af_mel_weights: null handle");
    return af_features_mel_weights(m->f, W, cap);
}

af_status af_mel_apply_device(af_mel* m, const float* power_dev, size_t n_frames, float* mel_dev,
                              void* stream)
{
    AF_REQUIRE(m, "af_mel_apply_device: null handle");
    if (n_frames == 0) return AF_OK;
    AF_REQUIRE(power_dev && mel_dev, "af_mel_apply_device: null buffer");
    if (m->f->dense_mel || m->f->cfg.n_fft != (uint32_t)kNfft) {
        // a filterbank the banded plan does not cover, or power rows of n_fft / 2 + 1 < 513 bins: the dense contraction
        // (on the rows' own grid: d_Wn)
        const bool narrow = m->f->cfg.n_fft != (uint32_t)kNfft;
        const int64_t blocks = ((int64_t)n_frames + 7) / 8;
        const unsigned grid = (unsigned)std::min<int64_t>(blocks, (int64_t)m->f->n_cus * 16);
        hipLaunchKernelGGL(dense_mel_kernel, dim3(grid), dim3(256), 0, as_stream(stream), power_dev, mel_dev,
                           narrow ? m->f->d_Wn.as<float>() : m->f->d_W.as<float>(),
                           narrow ? m->f->d_bandn.as<int>() : m->f->d_band.as<int>(), (int)m->f->cfg.n_mels,
                           (int64_t)n_frames, (int)m->f->cfg.log_mel, narrow ? (int)m->f->cfg.n_fft / 2 + 1 : kBins);
        AF_HIP(hipGetLastError());
        return AF_OK;
    }
    MelApplyParams prm{};
    prm.power = power_dev;
    prm.mel = mel_dev;
    prm.wfrag = m->f->d_wfrag;
    prm.meta = m->f->d_meta;
    prm.plan = m->f->plan;
    prm.n_frames = (int64_t)n_frames;
    prm.log_mel = (int)m->f->cfg.log_mel;
    const int64_t units = ((int64_t)n_frames + kFramesPerIter - 1) / kFramesPerIter;
    int64_t grid = (int64_t)m->f->n_cus * 2;
    if (grid > units) grid = units;
    return launch_mel_apply(prm, (int)grid, as_stream(stream));
}

af_status af_mel_apply(af_mel* m, const float* power, size_t n_frames, float* mel, size_t mel_cap)
{
    AF_REQUIRE(m, "af_mel_apply: null handle");
    const size_t n_out = n_frames * m->f->cfg.n_mels;
    if (mel_cap < n_out) return fail(AF_ERR_BUFFER_TOO_SMALL, "mel buffer holds %zu floats, need %zu", mel_cap, n_out);
    if (n_frames == 0) return AF_OK;
    AF_REQUIRE(power && mel, "af_mel_apply: null buffer");
    AF_TRY(m->d_in.reserve(n_frames * kBins * sizeof(float)));
    AF_TRY(m->d_out.reserve(n_out * sizeof(float)));
    AF_HIP(hipMemcpy(m->d_in.p, power, n_frames * kBins * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(af_mel_apply_device(m, m->d_in.as<float>(), n_frames, m->d_out.as<float>(), nullptr));
    AF_HIP(hipMemcpy(mel, m->d_out.p, n_out * sizeof(float), hipMemcpyDeviceToHost));
    return AF_OK;
}

}  // extern "C"


bool af::features_one_launch(const af_features* p) { return p && !p->dense_mel && p->cfg.channels <= 2; }

// The chain's pipelined form (chain.hip): the units of clips [clip_begin, clip_end) only, on the given stream, with one
// or two workgroups per CU.  Rows land at their places in mel_dev; the caller orders the launches.
af_status af::features_run_clips(af_features* p, const void* pcm_dev, float* mel_dev, hipStream_t stream, size_t clip_begin,
                                 size_t clip_end, int wg_per_cu)
{
    AF_REQUIRE(p && clip_begin <= clip_end && clip_end <= p->n_clips, "features_run_clips: bad clip range");
    AF_REQUIRE(!p->dense_mel && p->cfg.channels <= 2, "features_run_clips: one-launch configurations only");
    const int64_t u0 = p->h_unit_off[clip_begin], u1 = p->h_unit_off[clip_end];
    if (u1 <= u0) return AF_OK;
    AF_TRY(run_device_impl(p, pcm_dev, mel_dev, stream, u0, u1 - u0, 0, -1, wg_per_cu));
    return note_launch(p, stream);
}
