"""Print the mel plan (groups, steps, slots per wave) for a few filterbanks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_flow_rs_amd as af
lib = af.library()
for sr, nm in [(48000, 80), (16000, 80), (48000, 128), (16000, 40), (22050, 64), (48000, 30)]:
    try:
        p = af.FeaturePipeline(sample_rate=sr, channels=1, n_mels=nm)
    except af.AudioError as e:
        print(sr, nm, "unsupported:", e); continue
    info = (C.c_int32 * 81)()
    lib.af_features_plan_describe(p._h, info)
    v = list(info)
    print(sr, nm, v[:17])
    for w in range(8):
        g = [(e & 255, (e >> 8) & 31) for e in v[17 + 8 * w:25 + 8 * w] if e]
        print('   wave', w, 'groups (first filter, filters):', g)
