import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import audio_flow_rs_amd as af
bad = []
for sr in (8000, 16000, 22050, 32000, 44100, 48000, 96000):
    for nm in (1, 2, 3, 4, 5, 8, 10, 13, 16, 20, 23, 26, 32, 40, 64, 80, 96, 128):
        for hop in (256, 160):
            try:
                af.FeaturePipeline(sample_rate=sr, channels=1, hop=hop, n_mels=nm)
            except Exception as e:
                bad.append((sr, nm, hop, str(e)[:90]))
print(len(bad), "failing")
for b in bad: print(b)
