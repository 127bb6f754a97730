"""The 4 x 4 x 1 mel plan with and without the arrangement search of build_plan4 (LDS read conflicts of its B operands), config 2,
alternating in one process.   python3 scripts/ablate/m4_arrange.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, audio_flow_rs_amd as af
lib = af.library(); dev = torch.device("cuda:0")
x = (torch.rand(256, 1440000, 2, device=dev) * 2 - 1) * 0.3
s = torch.cuda.current_stream().cuda_stream
pipes = {}
for name in ("arranged", "first_fit"):
    if name == "first_fit": os.environ["AF_M4_NO_ARRANGE"] = "1"   # (read at af_features_create)
    p = af.FeaturePipeline(); tot = p.set_batch_uniform(256, 1440000); pipes[name] = p
    import ctypes as C
    info = (C.c_int32 * 12)(); lib.af_features_kernel_describe(p._h, info); print(name, "plan:", list(info)[:6], "read-conflict score", info[11])
mel = torch.empty(tot, 80, device=dev)
t0 = time.time()
while time.time() - t0 < 1.0: pipes["arranged"].run_device(x.data_ptr(), mel.data_ptr(), s); torch.cuda.synchronize()
res = {k: [] for k in pipes}
for rep in range(8):
    for k, p in pipes.items():
        for _ in range(5): p.run_device(x.data_ptr(), mel.data_ptr(), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): p.run_device(x.data_ptr(), mel.data_ptr(), s)
        e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 50)
for k, r in res.items():
    r = sorted(r); print(k, f"min {r[0]:.4f} median {r[len(r)//2]:.4f} max {r[-1]:.4f}")
