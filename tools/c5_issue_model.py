"""In-order issue model of one step of the (16, 8) MFMA sweep (csrc/backward_mfma16.hpp), fast path (Q_uu positive
definite), from the compiler's assembly -- tools/loopsim.py's idea for a loop body with branches, SGPR operands and the
matrix pipe.  Costs are the lone-wave measurements of tools/micro/issue_rate.hip: independent vector instruction 4.1 cycles
(result usable after 8.7), transcendental 9 / 12.5, v_readlane 8.75, MFMA f32 16x16x4: the pipe is busy 32 cycles, the
result usable after 40; v_permlane*_swap 8 / 12; ds_bpermute 8 / 120.  Prints the mix, the predicted cycles per step, the
time of every MFMA / rsqrt, and the largest stalls.  (The next step's loads are waited for a step later: the model's one
"s_waitcnt" stall of ~400 cycles on them is an artefact of simulating a single trip.)

    cat > /tmp/one.hip <<EOT
    #include "solver.hpp"
    namespace ilqr { template __global__ void backward_mfma16_kernel<float, true>(KArgs<float>); }
    EOT
    hipcc <the Makefile's CXXFLAGS> --cuda-device-only -save-temps=obj -c -o /tmp/one.o /tmp/one.hip      (in csrc/)
    python tools/c5_issue_model.py /tmp/one-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re, sys
S = open(sys.argv[1]).read().splitlines()
def find(lbl): return next(i for i,l in enumerate(S) if l.startswith(lbl))
# the fast path of the loop: header .. the branch that skips the LU fallback (s_cbranch_vccz T) | T: .. the back edge (s_cbranch_scc);
# the s_cbranch_vccnz in between is the not-taken side of "not positive definite"
i6 = next(i for i, l in enumerate(S) if l.startswith(".LBB0_6:"))
br = next(i for i in range(i6, len(S)) if "s_cbranch_vccz" in S[i])
i8 = find(S[br].split()[1] + ":")
end = next(i for i in range(i8, len(S)) if S[i].strip().startswith("s_cbranch_scc"))
body = S[i6+1:br] + S[i8+1:end+1]
body = [l.split(";")[0].strip() for l in body]
body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
REG = re.compile(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]")
def regs(t):
    o=set()
    for m in REG.finditer(t):
        if m.group(1): o.add(m.group(1)+m.group(2))
        else: o.update(m.group(3)+str(k) for k in range(int(m.group(4)), int(m.group(5))+1))
    return o
def cls(m):
    if m.startswith(("buffer_load","global_load")): return "load"
    if m.startswith(("buffer_store","global_store")): return "store"
    if m.startswith(("v_rcp","v_rsq","v_sqrt")): return "trans"
    if m.startswith(("v_readlane","v_readfirstlane")): return "readlane"
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith("v_permlane"): return "perm"
    if m.startswith("ds_bpermute"): return "bperm"
    if m.startswith("ds_read"): return "ldsr"
    if m.startswith("ds_write"): return "ldsw"
    if m.startswith("v_"): return "valu"
    if m.startswith("s_nop"): return "nop"
    if m.startswith("s_waitcnt"): return "wait"
    return "salu"
COST = {"valu":(4.1,8.7),"trans":(9.0,12.5),"readlane":(8.75,8.75),"mfma":(4.1,40.0),"load":(13.5,700.0),"store":(21.0,0),"wait":(0,0),"salu":(2.0,2.0),"perm":(8.0,12.0),"bperm":(8.0,120.0),"nop":(0,0),"ldsr":(14.0,100.0),"ldsw":(14.0,0.0)}
t=0.0; ready={}; mix={}; mf_free=0.0; stalls=[]; lg=[]
marks=[]
for idx,l in enumerate(body):
    m=l.split()[0]; ops=l[len(m):]; c=cls(m); mix[c]=mix.get(c,0)+1
    f=[x.strip() for x in ops.split(",")]
    writes = c in ("valu","trans","mfma","load","readlane","perm","bperm","salu","ldsr")
    dst = regs(f[0]) if writes and f and f[0] else set()
    src=set()
    for x in (f[1:] if writes else f): src|=regs(x)
    if "fmac" in m or "fmaak" in m or c=="perm": src|=dst
    if c=="perm": dst|=regs(f[1])
    if c=="nop":
        t += int(ops.strip() or 0)+1; continue
    if c=="wait":
        if "lgkmcnt(0)" in ops or "vmcnt(0)" in ops:
            need=max([x for x in lg]+[0]); 
            if need>t: stalls.append((need-t,l,idx)); t=need
            lg=[]
        continue
    need=max([ready.get(r,0.0) for r in src]+[0.0])
    if c=="mfma": need=max(need,mf_free)
    if need>t:
        stalls.append((need-t,l,idx)); t=need
    iss,lat=COST[c]
    if c=="mfma": mf_free=t+32.0
    for r in dst: ready[r]=t+lat
    if c in ("bperm","load","ldsr"): lg.append(t+lat)
    t+=iss
    if c=="trans": marks.append((idx,t,"rsq"))
    if c=="mfma": marks.append((idx,t,"mfma"))
print("instructions",len(body),mix)
print("cycles per step %.0f, stalls %.0f"%(t,sum(s for s,_,_ in stalls)))
for k,(idx,tt,w) in enumerate(marks): print(w,idx,int(tt),end=" | ")
print()
big=sorted(stalls,reverse=True)[:25]
for s,l,idx in sorted(big,key=lambda x:x[2]): print("%5.0f @%d %s"%(s,idx,l[:90]))
