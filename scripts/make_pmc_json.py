"""Condense gpurun_out/rNNp (scripts/profile_rNN.sh; NN = AF_ROUND_NO, default 4) into the committed evidence under profiles/:
   r03_*.txt copies, profiles/pmc_traffic.json (calibrated HBM bytes per launch = bench.py's roofline.traffic) and
   profiles/pmc_resources.json (busy fractions of the other resources = bench.py's roofline.other).
   python3 scripts/make_pmc_json.py [gpurun_out/r04p]"""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = int(os.environ.get("AF_ROUND_NO", "5"))
RT = f"r{ROUND:02d}"
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", RT + "p")
dst = os.path.join(ROOT, "profiles")


def table(name):
    """{kernel: {counter: mean}} of a pmc_*.txt summary"""
    out = {}
    path = os.path.join(src, name)
    if not os.path.exists(path):
        return out
    for ln in open(path):
        if ln.startswith("#") or ln.startswith("kernel,"):
            continue
        k, c, n, mean, lo, hi = ln.strip().rsplit(",", 5)
        out.setdefault(k, {})[c] = float(mean)
    return out


def first(tab, key):
    for k, v in tab.items():
        if key in k:
            return k, v
    raise SystemExit(f"no kernel matching {key} in the table")


copies = {"kernel_stats.txt": RT + "_kernel_stats.txt", "pmc_fetch.txt": RT + "_pmc_fetch.txt", "pmc_write.txt": RT + "_pmc_write.txt",
          "pmc_sq.txt": RT + "_pmc_sq.txt", "pmc_sq2.txt": RT + "_pmc_sq2.txt", "pmc_cal_fetch.txt": RT + "_pmc_cal_fetch.txt",
          "pmc_cal_write.txt": RT + "_pmc_cal_write.txt", "bench.json": RT + "_bench.json",
          "bench_driver_args.json": RT + "_bench_driver_args.json", "chain_kernel_stats.txt": RT + "_chain_kernel_stats.txt",
          "pmc_chain_fetch.txt": RT + "_chain_pmc_fetch.txt", "pmc_chain_write.txt": RT + "_chain_pmc_write.txt",
          "pmc_chain_sq.txt": RT + "_chain_pmc_sq.txt", "pmc_chain_sq2.txt": RT + "_chain_pmc_sq2.txt",
          "pmc_res_fetch.txt": RT + "_res_pmc_fetch.txt", "pmc_res_write.txt": RT + "_res_pmc_write.txt",
          "pmc_res_sq.txt": RT + "_res_pmc_sq.txt", "pmc_res_sq2.txt": RT + "_res_pmc_sq2.txt",
          "resample_ratios.json": RT + "_resample_ratios.json", "ops.json": RT + "_ops.json",
          "ops_kernel_stats.txt": RT + "_ops_kernel_stats.txt",
          "wire_kernel_stats.txt": RT + "_wire_kernel_stats.txt", "pmc_wire_fetch.txt": RT + "_wire_pmc_fetch.txt",
          "pmc_wire_write.txt": RT + "_wire_pmc_write.txt", "pmc_wire_sq.txt": RT + "_wire_pmc_sq.txt", "pmc_wire_sq2.txt": RT + "_wire_pmc_sq2.txt",
          "wire_bench.json": RT + "_wire_bench.json", "wire_bench_messages320.json": RT + "_wire_bench_messages320.json",
          "wire_bench_i16.json": RT + "_wire_bench_i16.json", "chain44_kernel_stats.txt": RT + "_chain44_kernel_stats.txt",
          "nfft_kernel_stats.txt": RT + "_nfft_kernel_stats.txt", "pmc_nfft_fetch.txt": RT + "_nfft_pmc_fetch.txt", "pmc_nfft_write.txt": RT + "_nfft_pmc_write.txt", "pmc_nfft_sq.txt": RT + "_nfft_pmc_sq.txt", "pmc_nfft_sq2.txt": RT + "_nfft_pmc_sq2.txt",
          "nfft400.json": RT + "_nfft400.json", "nfft480.json": RT + "_nfft480.json",
          "chain44.txt": RT + "_chain44.txt",
          "v3_kernel_stats.txt": RT + "_v3_kernel_stats.txt", "pmc_v3_fetch.txt": RT + "_v3_pmc_fetch.txt",
          "pmc_v3_write.txt": RT + "_v3_pmc_write.txt", "pmc_v3_sq.txt": RT + "_v3_pmc_sq.txt", "pmc_v3_sq2.txt": RT + "_v3_pmc_sq2.txt"}
for a, b in copies.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))

# ---- calibration: af::to_mono2_kernel on 1 GiB in / 0.5 GiB out (scripts/pmc_calibrate.py)
_, cf = first(table("pmc_cal_fetch.txt"), "to_mono")
_, cw = first(table("pmc_cal_write.txt"), "to_mono")
cal_f = cf["FETCH_SIZE"] * 1024 / (1 << 30)          # reported / true
cal_w = cw["WRITE_SIZE"] * 1024 / (1 << 29)
kname, f = first(table("pmc_fetch.txt"), "af_features")
_, w = first(table("pmc_write.txt"), "af_features")
rd = f["FETCH_SIZE"] * 1024 / cal_f
wr = w["WRITE_SIZE"] * 1024 / cal_w
ALG = 256 * 1440000 * 8 + 1439232 * 320
old = json.load(open(os.path.join(dst, "pmc_traffic.json")))
traffic = {
    "round": ROUND, "kernel": kname, "workload": "BASELINE config 2 (256 x 30 s stereo 48 kHz f32 -> 80 mel), one launch",
    "FETCH_SIZE_kb_raw": f["FETCH_SIZE"], "WRITE_SIZE_kb_raw": w["WRITE_SIZE"],
    "calibration": {"kernel": "af::to_mono2_kernel on 1 GiB in / 0.5 GiB out (scripts/pmc_calibrate.py)",
                    "FETCH_SIZE_reported_over_true": round(cal_f, 6), "WRITE_SIZE_reported_over_true": round(cal_w, 6),
                    "note": "gfx950 FETCH_SIZE tallies 16-B-per-lane coalesced reads at exactly 1/2 (MI355X_MICROARCH.md, HBM); corrected below"},
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
    "algorithmic_bytes_per_launch": ALG, "traffic_over_algorithmic": round((rd + wr) / ALG, 4),
    "source": ["profiles/" + RT + "_pmc_fetch.txt", "profiles/" + RT + "_pmc_write.txt", "profiles/" + RT + "_pmc_cal_fetch.txt", "profiles/" + RT + "_pmc_cal_write.txt"],
    "banded_kernel_config3": old.get("banded_kernel_config3"),
}
# history: every earlier roundN record is carried forward unchanged, the previous top level goes under ITS OWN round
for k, v in old.items():
    if k.startswith("round") and k != "round" and v is not None:
        traffic[k] = v
if old.get("round") is not None and int(old["round"]) != ROUND:
    traffic[f"round{int(old['round'])}"] = {k: old[k] for k in ("kernel", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch",
                                                               "hbm_bytes_per_launch", "traffic_over_algorithmic") if k in old}
# chain
cfch, cwch = table("pmc_chain_fetch.txt"), table("pmc_chain_write.txt")
if cfch and cwch:
    kern, tot = {}, 0.0
    for k in cfch:
        r_ = cfch[k]["FETCH_SIZE"] * 1024 / cal_f
        w_ = cwch.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / cal_w
        kern[k] = {"FETCH_SIZE_kb_raw": cfch[k]["FETCH_SIZE"], "WRITE_SIZE_kb_raw": cwch.get(k, {}).get("WRITE_SIZE"),
                   "hbm_read_bytes_per_launch": int(r_), "hbm_write_bytes_per_launch": int(w_)}
        tot += r_ + w_
    alg4 = 256 * 1440000 * 8 + 479232 * 320
    traffic["chain_config4"] = {"workload": "256 x 30 s stereo 48 kHz f32 -> downmix + 48->16 kHz + FIR 129 (one launch) -> STFT -> 80 mel (one launch); scripts/profile_r%02d.sh" % ROUND,
                                "kernels": kern, "hbm_bytes_per_pass": int(tot), "algorithmic_bytes_per_pass": alg4,
                                "traffic_over_algorithmic": round(tot / alg4, 4),
                                "source": ["profiles/" + RT + "_chain_pmc_fetch.txt", "profiles/" + RT + "_chain_pmc_write.txt"]}
rf, rw = table("pmc_res_fetch.txt"), table("pmc_res_write.txt")
if rf and rw:
    k, v = first(rf, "banded_stream")
    r_ = v["FETCH_SIZE"] * 1024 / cal_f
    # the streamed kernel stages x with 8-byte-per-lane loads and fragments with 16-byte ones: FETCH_SIZE's 1/2 is
    # calibrated for 16-byte-per-lane streams only, so the read figure is indicative
    w_ = first(rw, "banded_stream")[1]["WRITE_SIZE"] * 1024 / cal_w
    alg = 1024 * (1440000 + 1323000) * 4
    traffic["banded_stream_48k_44k1"] = {"workload": "1024 x 30 s mono 48 kHz -> 44.1 kHz, one launch", "kernel": k,
                                         "hbm_read_bytes_per_launch_indicative": int(r_), "hbm_write_bytes_per_launch": int(w_),
                                         "algorithmic_bytes_per_launch": alg,
                                         "source": ["profiles/" + RT + "_res_pmc_fetch.txt", "profiles/" + RT + "_res_pmc_write.txt"]}
# the wire path in one launch
wf, ww = table("pmc_wire_fetch.txt"), table("pmc_wire_write.txt")
if wf and ww:
    k, v = first(wf, "wire_kernel")
    r_ = v["FETCH_SIZE"] * 1024 / cal_f
    w_ = first(ww, "wire_kernel")[1]["WRITE_SIZE"] * 1024 / cal_w
    algw = 1024 * 1440000 * 8 + 1024 * 4 * ((2 * 480000 + 2) // 3)
    traffic["wire_1024x30s"] = {"workload": "1024 x 30 s stereo 48 kHz f32 -> to_mono -> 16 kHz -> PCM16 -> base64 (one string per clip), ONE launch", "kernel": k,
                                "hbm_read_bytes_per_launch": int(r_), "hbm_write_bytes_per_launch": int(w_), "hbm_bytes_per_launch": int(r_ + w_),
                                "algorithmic_bytes_per_launch": algw, "traffic_over_algorithmic": round((r_ + w_) / algw, 4),
                                "source": ["profiles/" + RT + "_wire_pmc_fetch.txt", "profiles/" + RT + "_wire_pmc_write.txt"]}
# the speech framing on the matrix-product DFT kernel (compute-bound: the figure shows that the 2.5 x overlap of the frames and the
# fragment stream stay in the caches)
nf, nw_ = table("pmc_nfft_fetch.txt"), table("pmc_nfft_write.txt")
if nf and nw_:
    k, v = first(nf, "af_features_dft")
    r_ = v["FETCH_SIZE"] * 1024 / cal_f
    w_ = first(nw_, "af_features_dft")[1]["WRITE_SIZE"] * 1024 / cal_w
    algn = 256 * 480000 * 4 + 256 * 2998 * 80 * 4
    traffic["speech_nfft400_256x30s"] = {"workload": "256 x 30 s mono 16 kHz f32 -> STFT 400 / 160 -> 80 mel, one launch", "kernel": k,
                                         "hbm_read_bytes_per_launch": int(r_), "hbm_write_bytes_per_launch": int(w_), "hbm_bytes_per_launch": int(r_ + w_),
                                         "algorithmic_bytes_per_launch": algn, "traffic_over_algorithmic": round((r_ + w_) / algn, 4),
                                         "source": ["profiles/" + RT + "_nfft_pmc_fetch.txt", "profiles/" + RT + "_nfft_pmc_write.txt"]}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

# ---- the other resources of the headline kernel (HBM fraction AND VALU / MFMA / LDS fractions)
_, s1 = first(table("pmc_sq.txt"), "af_features")
_, s2 = first(table("pmc_sq2.txt"), "af_features")
cycles = s2["GRBM_GUI_ACTIVE"] / 8.0                   # the counter sums the 8 XCDs
res = {
    "round": ROUND, "kernel": kname, "workload": traffic["workload"],
    "launch_cycles": cycles,
    "lds_frac": round(s2["SQ_LDS_IDX_ACTIVE"] / 256 / cycles, 4),                  # LDS-array cycles per CU
    "valu_frac": round(s1["SQ_INSTS_VALU"] * 2 / 1024 / cycles, 4),                 # 2 cycles per wave64 instruction per SIMD
    "mfma_frac": round(s2["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cycles, 4),          # matrix pipe busy per SIMD
    "hbm_frac_of_8TBps": None,
    "counters": {"SQ_LDS_IDX_ACTIVE": s2["SQ_LDS_IDX_ACTIVE"], "SQ_LDS_BANK_CONFLICT": s2["SQ_LDS_BANK_CONFLICT"],
                 "SQ_INSTS_VALU": s1["SQ_INSTS_VALU"], "SQ_INSTS_LDS": s1["SQ_INSTS_LDS"],
                 "SQ_VALU_MFMA_BUSY_CYCLES": s2["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": s2["GRBM_GUI_ACTIVE"],
                 "SQ_WAVE_CYCLES": s1["SQ_WAVE_CYCLES"], "SQ_WAIT_ANY": s1["SQ_WAIT_ANY"],
                 "SQ_WAIT_INST_ANY": s1["SQ_WAIT_INST_ANY"], "SQ_ACTIVE_INST_ANY": s1["SQ_ACTIVE_INST_ANY"]},
    "formulas": "lds = SQ_LDS_IDX_ACTIVE / 256 CUs; valu = SQ_INSTS_VALU x 2 cycles / 1024 SIMDs; mfma = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs; all over GRBM_GUI_ACTIVE / 8 XCDs of the profiled launch",
    "source": ["profiles/" + RT + "_pmc_sq.txt", "profiles/" + RT + "_pmc_sq2.txt"],
    "note": "recorded by separate rocprofv3 --pmc passes (scripts/profile_r%02d.sh), not measured by the bench run that quotes it" % ROUND,
    "in_kernel_clock_GHz": 1.97, "in_kernel_clock_note": "s_memtime / s_memrealtime inside the product kernel (AF_V2_CLK builds, profiles/r05_v2_ablations.txt section 5): 1.97-1.98 GHz with the mel phase on 4 x 4 x 1 blocks (2.10-2.16 GHz with the 16 x 16 x 4 tiles of rounds 2-4) against 2.38 GHz for builds with a large piece left out -- the launch is power-limited",
}
# the matrix-core DFT kernel (opt-in) on the same workload, same formulas
old_res = json.load(open(os.path.join(dst, "pmc_resources.json"))) if os.path.exists(os.path.join(dst, "pmc_resources.json")) else {}
if not (table("pmc_v3_sq.txt") and table("pmc_v3_sq2.txt")) and old_res.get("features_v3"):
    res["features_v3"] = dict(old_res["features_v3"], recorded_in_round=old_res["features_v3"].get("recorded_in_round", old_res.get("round")))
if table("pmc_v3_sq.txt") and table("pmc_v3_sq2.txt"):
    k3, v1 = first(table("pmc_v3_sq.txt"), "af_features_v3")
    _, v2 = first(table("pmc_v3_sq2.txt"), "af_features_v3")
    c3 = v2["GRBM_GUI_ACTIVE"] / 8.0
    res["features_v3"] = {
        "kernel": k3, "launch_cycles": c3,
        "lds_frac": round(v2["SQ_LDS_IDX_ACTIVE"] / 256 / c3, 4), "valu_frac": round(v1["SQ_INSTS_VALU"] * 2 / 1024 / c3, 4),
        "mfma_frac": round(v2["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / c3, 4),
        "waves_waiting": round(v1["SQ_WAIT_ANY"] / v1["SQ_WAVE_CYCLES"], 4),
        "waves_stalled_on_issue": round(v1["SQ_WAIT_INST_ANY"] / v1["SQ_WAVE_CYCLES"], 4),
        "waves_issuing": round(v1["SQ_ACTIVE_INST_ANY"] / v1["SQ_WAVE_CYCLES"], 4),
        "counters": {"SQ_LDS_IDX_ACTIVE": v2["SQ_LDS_IDX_ACTIVE"], "SQ_LDS_BANK_CONFLICT": v2["SQ_LDS_BANK_CONFLICT"],
                     "SQ_INSTS_VALU": v1["SQ_INSTS_VALU"], "SQ_INSTS_LDS": v1["SQ_INSTS_LDS"],
                     "SQ_VALU_MFMA_BUSY_CYCLES": v2["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": v2["GRBM_GUI_ACTIVE"]},
        "source": ["profiles/" + RT + "_v3_pmc_sq.txt", "profiles/" + RT + "_v3_pmc_sq2.txt"]}
    if table("pmc_v3_fetch.txt") and table("pmc_v3_write.txt"):
        _, f3 = first(table("pmc_v3_fetch.txt"), "af_features_v3")
        _, w3 = first(table("pmc_v3_write.txt"), "af_features_v3")
        res["features_v3"]["traffic_over_algorithmic"] = round((f3["FETCH_SIZE"] * 1024 / cal_f + w3["WRITE_SIZE"] * 1024 / cal_w) / ALG, 4)
json.dump(res, open(os.path.join(dst, "pmc_resources.json"), "w"), indent=1)
print(json.dumps({"traffic_over_algorithmic": traffic["traffic_over_algorithmic"], "lds": res["lds_frac"], "valu": res["valu_frac"],
                  "mfma": res["mfma_frac"]}))
