"""Print the hop-256 kernel's mel plan (groups, steps, slots per wave) for a few filterbanks.
   python3 scripts/print_mel_plan.py [LIB ...]   (default: the product library; needs a GPU: handles own device buffers)"""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
for path in (sys.argv[1:] or [ffi.LIB_PATH]):
    lib = C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name in ("af_features_default_config", "af_features_create", "af_features_destroy", "af_last_error"):
        fn = getattr(lib, name); fn.restype, fn.argtypes = ffi.SIGNATURES[name]
    print("==", path)
    for sr, nm in [(48000, 80), (16000, 80), (48000, 128), (16000, 40), (22050, 64), (48000, 30)]:
        cfg = ffi.FeaturesConfig(); lib.af_features_default_config(C.byref(cfg))
        cfg.sample_rate, cfg.channels, cfg.n_mels = sr, 1, nm
        h = C.c_void_p()
        if lib.af_features_create(C.byref(cfg), C.byref(h)) != 0:
            print(sr, nm, "unsupported:", lib.af_last_error().decode()); continue
        info = (C.c_int32 * 81)()
        lib.af_features_plan_describe(h, info)
        v = list(info)
        print(sr, nm, "general: groups", v[0], "steps", v[1], "slots", v[2:6], "| hop-256: waves", v[6], "groups", v[7],
              "steps", v[8], "slots", v[9:17])
        for w in range(8):
            g = [(e & 255, (e >> 8) & 31) for e in v[17 + 8 * w:25 + 8 * w] if e]
            print('   wave', w, 'groups (first filter, filters):', g)
        lib.af_features_destroy(h)
