"""Static VALU instruction count of the hot loop of a kernel in a hipcc -S dump: finds the largest backward-branch
loop body in the named kernel and counts instructions by class.  usage: count_loop_instrs.py dev.s <kernel substring> [steps per loop]"""
import re, sys, collections
src, pat = sys.argv[1], sys.argv[2]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        n = i - labels[m.group(1)]
        if n > best[0]: best = (n, labels[m.group(1)], i)
n, lo, hi = best
cnt = collections.Counter()
for l in body[lo:hi]:
    t = l.strip().split()
    if not t or t[0].endswith(":") or t[0].startswith((";", ".")): continue
    op = t[0]
    cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "lds" if op.startswith("ds_") else "other"
    cnt[cls] += 1
    if cls == "valu": cnt["pk" if op.startswith("v_pk_") else "nonpk"] += 1
print(f"{pat}: loop of {n} lines; per step (/{per}):", {k: round(v / per, 1) for k, v in cnt.items()})
