"""Registers / spills / scratch of every kernel in one csrc file:  python3 scripts/kernel_resources.py features_v2 [hipcc flags]"""
import os, re, subprocess, sys
f, extra = sys.argv[1], sys.argv[2:]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-flow-rs_amd", "csrc")
p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                    "--cuda-device-only", "-S", f + ".hip", "-o", f"/tmp/{f}.s", "-Rpass-analysis=kernel-resource-usage"] + extra,
                   cwd=csrc, capture_output=True, text=True)
cur, vals = None, {}
for ln in p.stderr.splitlines():
    if "error" in ln: print(ln)
    m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur, vals = t.split(":", 1)[1].strip(), {}
    else:
        k, v = t.rsplit(":", 1); vals[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            print(f"{name:78s} VGPR {vals.get('VGPRs'):>4s} spill {vals.get('VGPRs Spill'):>3s} scratch {vals.get('ScratchSize [bytes/lane]'):>4s} "
                  f"SGPR {vals.get('TotalSGPRs')} sspill {vals.get('SGPRs Spill')} occ {vals.get('Occupancy [waves/SIMD]')}")
