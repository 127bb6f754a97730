"""Socket power and clocks (rocm-smi, sampled by a child process started BEFORE this process touches the GPU) while one kernel runs
in a loop: the headline launch, its FFT-arithmetic-only ablation if given (AF_LIB builds of scripts/ablate), a device copy, idle.
   python3 scripts/gpu_power_probe.py [--seconds 4] [--lib build/ablate/libaf_....so ...]"""
import argparse, ctypes as C, importlib.util, os, subprocess, sys, time
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=4.0); ap.add_argument("--lib", nargs="*", default=[])
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log = "/tmp/af_power_probe.txt"
sampler = subprocess.Popen(["bash", "-c", f"while true; do date +%s.%N; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -iE 'power|sclk|mclk|junction|edge' ; sleep 0.2; done > {log} 2>&1"])
time.sleep(1.0)
import torch
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
dev = torch.device("cuda:0")
pcm = (torch.rand(256, 1440000, 2, device=dev) * 2 - 1) * 0.3
s = torch.cuda.current_stream().cuda_stream
marks = []
def phase(name, fn):
    torch.cuda.synchronize(); t0 = time.time()
    while time.time() - t0 < a.seconds:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    marks.append((name, t0, time.time()))
    time.sleep(1.0)
marks.append(("idle", time.time(), time.time() + 1.0)); time.sleep(1.0)
libs = [os.path.join(ROOT, "audio-flow-rs_amd", "libaudioflow_dsp.so")] + [os.path.abspath(x) for x in a.lib]
keep = []
for path in libs:
    lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name in ("af_features_default_config", "af_features_create", "af_features_set_batch_uniform", "af_features_run_device", "af_last_error"):
        fn = getattr(lib, name); fn.restype, fn.argtypes = ffi.SIGNATURES[name]
    cfg = ffi.FeaturesConfig(); lib.af_features_default_config(C.byref(cfg))
    h = C.c_void_p(); assert lib.af_features_create(C.byref(cfg), C.byref(h)) == 0
    tot = C.c_size_t(); assert lib.af_features_set_batch_uniform(h, 256, 1440000, C.byref(tot)) == 0
    mel = torch.empty(tot.value, 80, device=dev); keep.append((lib, h, mel))
    phase(os.path.basename(path), lambda: lib.af_features_run_device(h, pcm.data_ptr(), mel.data_ptr(), s))
dst = torch.empty_like(pcm)
phase("device copy 2.95 GB", lambda: dst.copy_(pcm))
time.sleep(0.5)
sampler.terminate()
rows, t = [], None
for ln in open(log):
    ln = ln.strip()
    try:
        t = float(ln); rows.append([t, ""]); continue
    except ValueError:
        pass
    if rows: rows[-1][1] += " | " + " ".join(ln.split())
import re
for name, t0, t1 in marks:
    sel = [r[1] for r in rows if t0 + 0.8 <= r[0] <= t1]
    pw = [float(m.group(1)) for x in sel for m in [re.search(r"Power \(W\): ([0-9.]+)", x)] if m]
    ck = [float(m.group(1)) for x in sel for m in [re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", x)] if m]
    tj = [float(m.group(1)) for x in sel for m in [re.search(r"junction\) \(C\): ([0-9.]+)", x)] if m]
    if pw:
        print(f"{name:34s} {len(pw):2d} samples   socket power {min(pw):6.0f} .. {max(pw):6.0f} W (mean {sum(pw) / len(pw):6.0f})   sclk {min(ck):5.0f} .. {max(ck):5.0f} MHz   junction {max(tj):4.0f} C")
    else:
        print(f"{name:34s} no samples")
