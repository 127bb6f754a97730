import os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
dev = torch.device("cuda:0")
k = np.arange(129) - 64.0
taps = (0.9 * np.sinc(0.9 * k) * np.hamming(129)).astype(np.float32)
n = 1440000
pcm = (torch.rand(256, n, 2, device=dev) * 2 - 1) * 0.3
s = torch.cuda.current_stream().cuda_stream
chain = af.ChainPipeline(fir_taps=taps)
tot = chain.set_batch_uniform(256, n)
mel = torch.empty(tot, 80, device=dev)
chain.set_option("pipeline", 8)
for _ in range(6): chain.run_device(pcm.data_ptr(), mel.data_ptr(), s)
torch.cuda.synchronize()
