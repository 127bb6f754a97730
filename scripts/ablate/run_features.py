"""Ablation runner: python3 scripts/ablate/run_features.py [--reps R] [--warm W] [--clips N] LIB [LIB ...]
Runs BASELINE config 2's fused feature launch through each given build of the library (build/ablate/libaf_*.so,
scripts/ablate/build.sh), in the order given, in ONE process: W untimed + R timed launches each, HIP events.
Under `rocprofv3 --pmc ...` use --warm 0 --reps 3: the dispatches of the feature kernel then appear in this order,
R per library (scripts/ablate/pmc_table.py splits them).  Exits non-zero if the runtime reported a fault."""
import argparse, ctypes as C, importlib.util, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20); ap.add_argument("--warm", type=int, default=30)
ap.add_argument("--clips", type=int, default=256); ap.add_argument("libs", nargs="+")
ap.add_argument("--clk", action="store_true", help="AF_V2_CLK builds: also print the in-kernel clock (s_memtime / s_memrealtime)")
ap.add_argument("--kernel", type=int, default=-1, help="af_features_set_option kernel (1 general, 2 features_v2, 3 features_v3)")
ap.add_argument("--mel4", type=int, default=-1, help="af_features_set_option mel4 (0: the 16 x 16 x 4 mel plan)")
a = ap.parse_args()
n = 1440000
dev = torch.device("cuda:0")
pcm = (torch.rand(a.clips, n, 2, device=dev) * 2 - 1) * 0.3
s = torch.cuda.current_stream().cuda_stream
for path in a.libs:
    # RTLD_LOCAL: every build resolves its own symbols (RTLD_GLOBAL would bind later builds to the first one's)
    lib = C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name in ("af_features_default_config", "af_features_create", "af_features_set_batch_uniform",
                 "af_features_run_device", "af_features_destroy", "af_last_error", "af_features_set_option",
                 "af_features_debug_stamps"):
        fn = getattr(lib, name); fn.restype, fn.argtypes = ffi.SIGNATURES[name]
    cfg = ffi.FeaturesConfig(); lib.af_features_default_config(C.byref(cfg))
    h = C.c_void_p(); assert lib.af_features_create(C.byref(cfg), C.byref(h)) == 0
    if a.kernel >= 0: assert lib.af_features_set_option(h, b"kernel", a.kernel) == 0
    if a.mel4 >= 0: assert lib.af_features_set_option(h, b"mel4", a.mel4) == 0
    tot = C.c_size_t(); assert lib.af_features_set_batch_uniform(h, a.clips, n, C.byref(tot)) == 0
    mel = torch.empty(tot.value, 80, device=dev)
    for _ in range(a.warm): assert lib.af_features_run_device(h, pcm.data_ptr(), mel.data_ptr(), s) == 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): assert lib.af_features_run_device(h, pcm.data_ptr(), mel.data_ptr(), s) == 0
    e1.record(); torch.cuda.synchronize()
    clk = ""
    if a.clk:
        import numpy as np
        assert lib.af_features_debug_stamps(h, 1, None) == 0
        n_clk = int(os.environ.get("AF_CLK_LAUNCHES", "10"))
        for _ in range(n_clk): assert lib.af_features_run_device(h, pcm.data_ptr(), mel.data_ptr(), s) == 0
        torch.cuda.synchronize()
        st = np.zeros(32, np.uint64)
        assert lib.af_features_debug_stamps(h, 0, st.ctypes.data) == 0
        if st[1]: clk = f"   in-kernel clock {float(st[0]) / float(st[1]) * 0.1:6.3f} GHz, {float(st[0]) / float(st[2]) / 1e6:6.3f} M cycles per workgroup"
        if st[24]:
            n8 = float(st[2]) / 8.0
            avg = [float(st[8 + i]) / n8 * 0.01 for i in range(8)]; mx = [float(st[16 + i]) * 0.01 for i in range(8)]
            span = (float(st[24]) - float(~st[25] & np.uint64(0xFFFFFFFFFFFFFFFF))) * 0.01
            if n_clk == 1 and st[26]:
                M = np.uint64(0xFFFFFFFFFFFFFFFF); n = float(st[2])
                s_min, s_max, s_avg = float(~st[25] & M), float(st[27]), float(st[26]) / n
                e_min, e_max, e_avg = float(~st[29] & M), float(st[24]), float(st[28]) / n
                clk += (f"\n      one launch, {int(n)} workgroups (us from the first start): starts avg {(s_avg - s_min) * 0.01:.1f} last {(s_max - s_min) * 0.01:.1f};"
                        f" ends first {(e_min - s_min) * 0.01:.1f} avg {(e_avg - s_min) * 0.01:.1f} last {(e_max - s_min) * 0.01:.1f}")
            clk += "\n      workgroup lifetimes by blockIdx % 8 (us): avg " + " ".join(f"{x:6.1f}" for x in avg) + "\n" + " " * 45 + "max " + " ".join(f"{x:6.1f}" for x in mx) + f"\n      first start to last end over the 10 launches (us): {span:.1f}"
    print(f"{os.path.basename(path):28s} {e0.elapsed_time(e1) / a.reps:8.4f} ms{clk}", flush=True)
    lib.af_features_destroy(h)
