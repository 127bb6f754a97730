import sys, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import audio_flow_rs_amd as af
from oracle import pyoracle as po
from util import synth_clip
x = synth_clip(0, 6000, 2, 48000)
pipe = af.FeaturePipeline(); pipe.set_batch([0, 3000 * 0 + 6000]); mel = pipe.run(x)
ref = po.features(x, 2, 48000.0)
err = np.abs(mel - ref) / np.maximum(np.abs(ref), 1e-30)
print("per-mel max rel err (x1e-6):"); print(np.round(err.max(0) * 1e6, 1))
print("per-frame:", np.round(err.max(1) * 1e6, 1))
np.set_printoptions(linewidth=200, suppress=True)
print("ratio frame0:"); print(np.round(mel[0] / ref[0], 3))
print("ratio frame7:"); print(np.round(mel[7] / ref[7], 3))
