"""Print the instruction-class stream of one kernel's assembly (from hipcc -save-temps): M = MFMA, v = VALU, t = transcendental, D = LDS read, W = LDS write,
G = global/LDS-DMA, s = scalar, B = barrier, w = s_waitcnt, one line per basic block.  usage: isa_stream.py file.s kernel-substring"""
import re, sys

def classify(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "t"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"): return "D"
    if op.startswith("ds_"): return "W"
    if op.startswith(("global_", "buffer_", "flat_")): return "G"
    if op == "s_barrier": return "B"
    if op == "s_waitcnt": return "w"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "j"
    if op.startswith("s_"): return "s"
    return "?"

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
out, cur, label = [], "", "entry"
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end") or "s_endpgm" in l and False: break
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        out.append((label, cur)); label, cur = m.group(1), ""; continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")): continue
    cur += classify(t.split()[0])
out.append((label, cur))
for lab, s in out:
    if len(s) >= int(sys.argv[3]) if len(sys.argv) > 3 else 1:
        print(f"{lab:12s} n={len(s):4d} M={s.count('M'):3d} v={s.count('v'):3d} t={s.count('t'):3d} D={s.count('D'):3d}\n  {s}")
