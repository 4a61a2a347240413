"""Static instruction mix of the hot loop (largest backward-branch region) of one kernel in a hipcc -S dump.
usage: loopstats.py dev.s <mangled-name substring> [steps in the loop body]"""
import re, sys, collections
def loopstats(path, pat):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    loop = body[best[1]:best[2] + 1]
    return collections.Counter(l.strip().split()[0] for l in loop if l.startswith("\t") and not l.strip().startswith((";", ".")))
if __name__ == "__main__":
    c = loopstats(sys.argv[1], sys.argv[2])
    per = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"VALU {valu / per:.1f}  total {sum(c.values()) / per:.1f} per step")
    print({k: round(v / per, 1) for k, v in c.most_common(40)})
