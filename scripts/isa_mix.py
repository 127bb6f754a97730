"""Instruction mix of one kernel in a hipcc --save-temps .s file, whole body and per basic block.
usage: isa_mix.py file.s kernel-name-substring [--blocks]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
show_blocks = "--blocks" in sys.argv
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0]:
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "vlane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"
tot = collections.Counter(); blocks = []; cur = ["entry", collections.Counter()]
ops = collections.Counter()
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith((";", "//")): continue
    t = t.split(";")[0].strip()
    if not t: continue
    if t.endswith(":"):
        blocks.append(cur); cur = [t[:-1], collections.Counter()]; continue
    if t.startswith("."): continue
    op = t.split()[0]
    k = klass(op); tot[k] += 1; cur[1][k] += 1; ops[op] += 1
blocks.append(cur)
print("total:", dict(tot))
if show_blocks:
    for name, c in blocks:
        n = sum(c.values())
        if n >= 20: print(f"{name:12s} {n:6d}", dict(c))
if "--ops" in sys.argv:
    for op, n in ops.most_common(60): print(f"{op:28s}{n}")
