"""ISA census of one kernel: python3 scripts/isa_count.py FILE.hip KERNEL_SUBSTR [hipcc flags...]
Compiles FILE.hip to gfx950 assembly with the product's flags (+ the given ones) and prints, per basic block of the
first kernel whose mangled name contains KERNEL_SUBSTR, the instruction counts by class (VALU, of which FMA with three
distinct VGPR sources; MFMA; LDS; VMEM; SALU; waits; branches)."""
import re, subprocess, sys, os, tempfile
src, sub, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = tempfile.mktemp(suffix=".s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "--offload-arch=gfx950", "-w",
                "-I" + os.path.join(root, "audio-flow-rs_amd/csrc"), "-S", "--cuda-device-only", "-o", out, src] + flags, check=True)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sub in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
def cls(op, args):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "scratch" if op.startswith("scratch_") else "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur = [], None
for l in lines[start:end + 1]:
    t = l.strip()
    if not t or t.startswith((";", "//")): continue
    if re.match(r"^[\.\w]+:", t):
        cur = {"name": t.split(":")[0], "n": {}, "fma3": 0, "lits": 0}
        blocks.append(cur); continue
    if t.startswith("."): continue
    if cur is None: continue
    m = re.match(r"^(\S+)\s*(.*?)(;.*)?$", t)
    op, args = m.group(1), m.group(2)
    c = cls(op, args)
    cur["n"][c] = cur["n"].get(c, 0) + 1
    if c == "valu" and op.startswith(("v_fma_f32", "v_fmac_f32")):
        regs = re.findall(r"\bv(\d+)\b", args)
        srcs = set(regs[1:]) if op.startswith("v_fma_f32") else set(regs)
        if len(srcs) >= 3: cur["fma3"] += 1
keys = ["valu", "mfma", "lds", "vmem", "scratch", "smem", "salu", "wait", "nop", "barrier", "branch"]
print("%-12s " % "block" + " ".join("%7s" % k for k in keys) + "    fma3")
tot = {k: 0 for k in keys}; f3 = 0
for b in blocks:
    n = sum(b["n"].values())
    if n < 12 and "-v" not in flags: 
        for k in keys: tot[k] += b["n"].get(k, 0)
        f3 += b["fma3"]; continue
    print("%-12s " % b["name"][-12:] + " ".join("%7d" % b["n"].get(k, 0) for k in keys) + "  %5d" % b["fma3"])
    for k in keys: tot[k] += b["n"].get(k, 0)
    f3 += b["fma3"]
print("%-12s " % "TOTAL" + " ".join("%7d" % tot[k] for k in keys) + "  %5d" % f3)
for l in lines[end:end + 400]:
    if sub in l and ".name" in l: break
m = re.search(r"\.vgpr_count:\s+(\d+)", "\n".join(lines)); 
os.unlink(out)
