"""Generates tile16m2_step_gen.inc: the fp32 Riccati step of the (n_x, n_u) = (4, 2) DPP sweep (backward_tile16m2.hpp,
iLQR_class.py:100-114) as a hand-ORDERED instruction stream, the way tile16_step_f32 is for n_u = 1 -- except that here the
order comes out of a list scheduler instead of being written down by hand.

Why: a sweep wave is alone on its SIMD and issues in order.  An independent vector instruction costs it ~4.1 cycles, one
that reads the result of the instruction before it ~8.7, a DPP read of a fresh result ~16.5 (tools/micro/issue_rate.hip),
and hipcc neither fuses the lane moves into the accumulating v_fmac nor spreads the step's twelve independent chains:
its order of the 84 instructions measured ~650-710 cycles per step (tools/fused_stamps.py dp).  The scheduler below places
every instruction at least 3 slots behind the producers of its plain operands and 4 behind those it reads through DPP
(which also satisfies the 2 wait states the hardware demands there: nothing inside an asm statement is padded), longest
remaining chain first, and prints the stream as volatile asm statements of at most 30 operands each (the compiler's limit).

    python3 gen_tile16m2_step.py > tile16m2_step_gen.inc          (committed; re-run after editing the table)
"""
import sys

DPP = " row_mask:0xf bank_mask:0xf bound_ctrl:1"
QB = ["quad_perm:[0,0,0,0]", "quad_perm:[1,1,1,1]", "quad_perm:[2,2,2,2]", "quad_perm:[3,3,3,3]"]
QP = [None, "quad_perm:[1,2,3,0]", "quad_perm:[2,3,0,1]", "quad_perm:[3,0,1,2]"]
QSW = "quad_perm:[1,0,3,2]"
ROR = [None, "row_ror:12", "row_ror:8", "row_ror:4"]       # rows i+1, i+2, i+3

# inputs of the step: the carried state, the tile, the lane masks
INPUTS = ["V", "vx", "a", "c", "sj0", "sj1", "sj2", "sj3", "fu0j", "fu1j", "lxj", "lux0j", "lux1j", "e0", "e1", "e2",
          "lxx", "m0", "m1", "mk_i1", "mk_j", "mk_r23"]
SGPR_INPUTS = {"mk_i1", "mk_j", "mk_r23"}
TILE_ONLY = set(INPUTS) - {"V", "vx"}

I = []      # (name of the instruction, text with {operands}, destination, plain sources, DPP sources, accumulates?, kind)


def ins(text, dst, plain=(), dpp=(), acc=False, kind="valu"):
    I.append(dict(text=text, dst=dst, plain=list(plain), dpp=list(dpp), acc=acc, kind=kind))


# ---- SK[i][1..3] and l_ux[.][i] broadcast out of the quad (tile only) -------------------------------------------------------
for d in (1, 2, 3):
    ins(f"v_mov_b32_dpp {{si{d}}}, {{a}} {QB[d]}{DPP}", f"si{d}", dpp=["a"])
# ---- A: P = f_x' V_xx, then Q_xx = l_xx + P f_x -------------------------------------------------------------------------------
ins(f"v_mul_f32_dpp {{P}}, {{a}}, {{V}} {QB[0]}{DPP}", "P", plain=["V"], dpp=["a"])
for d in (1, 2, 3):
    ins(f"v_fmac_f32_dpp {{P}}, {{V}}, {{si{d}}} {ROR[d]}{DPP}", "P", plain=[f"si{d}"], dpp=["V"], acc=True)
ins("v_fma_f32 {Qxx}, {sj0}, {P}, {lxx}", "Qxx", plain=["sj0", "P", "lxx"])
for d in (1, 2, 3):
    ins(f"v_fmac_f32_dpp {{Qxx}}, {{P}}, {{sj{d}}} {QP[d]}{DPP}", "Qxx", plain=[f"sj{d}"], dpp=["P"], acc=True)
# ---- B_c: pu_c[j] = sum_i f_u[i][c] V[i][j] (down the rows), Q_ux[c][j] = l_ux[c][j] + (pu_c f_x)[j] ----------------------------
for cc in (0, 1):
    ins(f"v_mul_f32_dpp {{pu{cc}}}, {{c}}, {{V}} {QB[cc]}{DPP}", f"pu{cc}", plain=["V"], dpp=["c"])
    ins(f"v_add_f32_dpp {{tb{cc}}}, {{pu{cc}}}, {{pu{cc}}} row_ror:8{DPP}", f"tb{cc}", plain=[f"pu{cc}"], dpp=[f"pu{cc}"])
    ins(f"v_add_f32_dpp {{pv{cc}}}, {{tb{cc}}}, {{tb{cc}}} row_ror:12{DPP}", f"pv{cc}", plain=[f"tb{cc}"], dpp=[f"tb{cc}"])
    ins(f"v_fma_f32 {{Qux{cc}}}, {{sj0}}, {{pv{cc}}}, {{lux{cc}j}}", f"Qux{cc}", plain=["sj0", f"pv{cc}", f"lux{cc}j"])
    for d in (1, 2, 3):
        ins(f"v_fmac_f32_dpp {{Qux{cc}}}, {{pv{cc}}}, {{sj{d}}} {QP[d]}{DPP}", f"Qux{cc}", plain=[f"sj{d}"], dpp=[f"pv{cc}"], acc=True)
# ---- Q_ux in row form (lane (i, j) needs Q_ux[c][i] for the value update): the column form's lane i of quad i, broadcast
# inside each quad by four masked DPP moves (bank_mask = one quad of every 16-lane row; the other quads keep what they have).
# (Until round 3 the row form was a second contraction chain through V_xx's symmetry: 16 instructions for these 8.)
for cc in (0, 1):
    for k in range(4):
        ins(f"v_mov_b32_dpp {{Qi{cc}}}, {{Qux{cc}}} {QB[k]} row_mask:0xf bank_mask:{1 << k:#x}", f"Qi{cc}", dpp=[f"Qux{cc}"], acc=(k > 0))
# ---- C: Q_x[j] = l_x[j] + (f_x' V_x)[j] -------------------------------------------------------------------------------------------
ins("v_fma_f32 {qx}, {sj0}, {vx}, {lxj}", "qx", plain=["sj0", "vx", "lxj"])
for d in (1, 2, 3):
    ins(f"v_fmac_f32_dpp {{qx}}, {{vx}}, {{sj{d}}} {QP[d]}{DPP}", "qx", plain=[f"sj{d}"], dpp=["vx"], acc=True)
# ---- U_c: Q_u[c] = l_u[c] + f_u[.][c]' V_x (l_u rides lane j = 0 into the quad sum) ----------------------------------------------
for cc in (0, 1):
    ins(f"v_mul_f32 {{qu{cc}}}, {{fu{cc}j}}, {{vx}}", f"qu{cc}", plain=[f"fu{cc}j", "vx"])
    ins(f"v_fmac_f32 {{qu{cc}}}, {{m0}}, {{e{cc}}}", f"qu{cc}", plain=["m0", f"e{cc}"], acc=True)
    ins(f"v_add_f32_dpp {{tu{cc}}}, {{qu{cc}}}, {{qu{cc}}} {QSW}{DPP}", f"tu{cc}", plain=[f"qu{cc}"], dpp=[f"qu{cc}"])
    ins(f"v_add_f32_dpp {{Qu{cc}}}, {{tu{cc}}}, {{tu{cc}}} {QP[2]}{DPP}", f"Qu{cc}", plain=[f"tu{cc}"], dpp=[f"tu{cc}"])
# ---- Q_uu = l_uu + pu f_u: q00 (l_uu00 sits in e2 of lane j = 0 alone), q01 (e0 of lane 1), q11 (e1 of lane 1) -------------------
ins("v_fma_f32 {q00}, {pv0}, {fu0j}, {e2}", "q00", plain=["pv0", "fu0j", "e2"])
ins(f"v_add_f32_dpp {{t00}}, {{q00}}, {{q00}} {QSW}{DPP}", "t00", plain=["q00"], dpp=["q00"])
ins(f"v_add_f32_dpp {{Q00}}, {{t00}}, {{t00}} {QP[2]}{DPP}", "Q00", plain=["t00"], dpp=["t00"])
for nm, pv, fu, e in (("01", "pv0", "fu1j", "e0"), ("11", "pv1", "fu1j", "e1")):
    ins(f"v_mul_f32 {{q{nm}}}, {{{pv}}}, {{{fu}}}", f"q{nm}", plain=[pv, fu])
    ins(f"v_fmac_f32 {{q{nm}}}, {{m1}}, {{{e}}}", f"q{nm}", plain=["m1", e], acc=True)
    ins(f"v_add_f32_dpp {{t{nm}}}, {{q{nm}}}, {{q{nm}}} {QSW}{DPP}", f"t{nm}", plain=[f"q{nm}"], dpp=[f"q{nm}"])
    ins(f"v_add_f32_dpp {{Q{nm}}}, {{t{nm}}}, {{t{nm}}} {QP[2]}{DPP}", f"Q{nm}", plain=[f"t{nm}"], dpp=[f"t{nm}"])
# ---- the 2 x 2 solve in closed form (iLQR_class.py:109-110): det, 1/det with one Newton step (fast_rcp) -----------------------------
ins("v_mul_f32 {det}, {Q00}, {Q11}", "det", plain=["Q00", "Q11"])
ins("v_fma_f32 {det}, -{Q01}, {Q01}, {det}", "det", plain=["Q01"], acc=True)
ins("v_rcp_f32 {inv}, {det}", "inv", plain=["det"], kind="trans")
ins("v_fma_f32 {er}, -{det}, {inv}, 1.0", "er", plain=["det", "inv"])
ins("v_fmac_f32 {inv}, {er}, {inv}", "inv", plain=["er"], acc=True)
ins("v_mul_f32_e64 {nia}, -{Q11}, {inv}", "nia", plain=["Q11", "inv"])     # -(d / det)
ins("v_mul_f32 {ib}, {Q01}, {inv}", "ib", plain=["Q01", "inv"])            # +(b / det) = -(inverse's off-diagonal)
ins("v_mul_f32_e64 {nid}, -{Q00}, {inv}", "nid", plain=["Q00", "inv"])     # -(a / det)
# K = -Quu^-1 Qux, k = -Quu^-1 Qu
ins("v_mul_f32 {K0}, {nia}, {Qux0}", "K0", plain=["nia", "Qux0"])
ins("v_fmac_f32 {K0}, {ib}, {Qux1}", "K0", plain=["ib", "Qux1"], acc=True)
ins("v_mul_f32 {K1}, {ib}, {Qux0}", "K1", plain=["ib", "Qux0"])
ins("v_fmac_f32 {K1}, {nid}, {Qux1}", "K1", plain=["nid", "Qux1"], acc=True)
ins("v_mul_f32 {k0}, {nia}, {Qu0}", "k0", plain=["nia", "Qu0"])
ins("v_fmac_f32 {k0}, {ib}, {Qu1}", "k0", plain=["ib", "Qu1"], acc=True)
ins("v_mul_f32 {k1}, {ib}, {Qu0}", "k1", plain=["ib", "Qu0"])
ins("v_fmac_f32 {k1}, {nid}, {Qu1}", "k1", plain=["nid", "Qu1"], acc=True)
# the scalar this lane stores into the gain record: K[0][j] (row 0), K[1][j] (row 1), k[0] / k[1] (rows 2, 3: lanes j = 0 / j > 0)
# -- three selects on loop-invariant lane masks (hipcc turned the C conditional into exec-mask branches: ~12 instructions)
ins("v_cndmask_b32_e64 {oK}, {K0}, {K1}, {mk_i1}", "oK", plain=["K0", "K1"])
ins("v_cndmask_b32_e64 {ok}, {k0}, {k1}, {mk_j}", "ok", plain=["k0", "k1"])
ins("v_cndmask_b32_e64 {outv}, {oK}, {ok}, {mk_r23}", "outv", plain=["oK", "ok"])
# short form (:113-114): V_xx = Q_xx + Q_ux' K, V_x = Q_x + K' Q_u
ins("v_fma_f32 {Vn}, {Qi0}, {K0}, {Qxx}", "Vn", plain=["Qi0", "K0", "Qxx"])
ins("v_fmac_f32 {Vn}, {Qi1}, {K1}", "Vn", plain=["Qi1", "K1"], acc=True)
ins("v_fma_f32 {vxn}, {K0}, {Qu0}, {qx}", "vxn", plain=["K0", "Qu0", "qx"])
ins("v_fmac_f32 {vxn}, {K1}, {Qu1}", "vxn", plain=["K1", "Qu1"], acc=True)

N = len(I)
# ---- dependencies: the last writer of every source (and, for an accumulating write, of the destination) -----------------------------
last = {}
for k, x in enumerate(I):
    x["deps"] = []           # (producer index, is_dpp_read)
    for s in x["plain"] + ([x["dst"]] if x["acc"] else []):
        if s in last:
            x["deps"].append((last[s], False))
    for s in x["dpp"]:
        if s in last:
            x["deps"].append((last[s], True))
    # an accumulating DPP instruction reads its own destination plainly: handled above; WAR on temporaries cannot occur
    # (every temporary has one writer chain)
    last[x["dst"]] = k
users = [[] for _ in range(N)]
for k, x in enumerate(I):
    for p, _ in x["deps"]:
        users[p].append(k)
LAT = {"valu": 3, "trans": 4}          # slots a plain consumer stays behind its producer


def gap(p, is_dpp):
    return (4 if is_dpp else LAT[I[p]["kind"]])


prio = [0] * N
for k in range(N - 1, -1, -1):
    prio[k] = 1 + max([prio[u] + (gap(k, any(d == k and f for d, f in I[u]["deps"])) - 1) for u in users[k]] + [0])

pos = {}
order = []
slot = 0
# the carried V, V_x were written by the previous step's last instructions: DPP reads of them wait 4 slots
carried_ready = {"V": 3, "vx": 3}
while len(order) < N:
    best, best_key = None, None
    for k in range(N):
        if k in pos or any(p not in pos for p, _ in I[k]["deps"]):
            continue
        short = 0           # how many slots too early this instruction would be
        hard_ok = True
        for p, f in I[k]["deps"]:
            short = max(short, pos[p] + gap(p, f) - slot)
            if f and slot - pos[p] < 3:
                hard_ok = False      # the hardware's 2 wait states before a DPP read of a VALU result
        for s in I[k]["dpp"]:
            if s in carried_ready and slot < carried_ready[s]:
                hard_ok = False
        if not hard_ok:
            continue
        key = (max(short, 0), -prio[k])
        if best_key is None or key < best_key:
            best, best_key = k, key
    if best is None:
        order.append(None)           # nothing may issue: a wait state
    else:
        pos[best] = slot
        order.append(best)
    slot += 1

stalls = sum(1 for k in order if k is None)
soft = 0
for k in order:
    if k is None:
        continue
    for p, f in I[k]["deps"]:
        soft += max(0, pos[p] + gap(p, f) - pos[k])

# ---- emission: volatile asm statements of at most MAXOPS distinct operands ---------------------------------------------------------
MAXOPS = 28
import re
NAME = re.compile(r"\{(\w+)\}")
blocks, cur, cur_names = [], [], set()
for k in order:
    names = set(NAME.findall(I[k]["text"])) if k is not None else set()
    if cur and len(cur_names | names) > MAXOPS:
        blocks.append(cur)
        cur, cur_names = [], set()
    cur.append(k)
    cur_names |= names
if cur:
    blocks.append(cur)

out = []
w = out.append
w("// GENERATED by gen_tile16m2_step.py -- do not edit; edit the table there and re-run.")
w(f"// {N} instructions in {len(order)} issue slots ({stalls} forced wait states, {soft} slot(s) of soft-gap shortfall), {len(blocks)} asm statements.")
w("// sel: lane masks (i == 1), (j != 0), (i >= 2); outv: the scalar this lane stores (K[0][j], K[1][j], k[0] or k[1])")
w("ILQR_DEV void tile16m2_step_f32(const TileQ2& tq, const LaneConst<float>& lc, const GainSel& sel, float& V, float& vx, float& outv, bool& pd) {")
temps = []
for x in I:
    if x["dst"] not in temps and x["dst"] not in ("outv",):
        temps.append(x["dst"])
w("    float " + ", ".join(temps) + ";")
cexpr = {"V": "V", "vx": "vx", "a": "tq.a", "c": "tq.c", "sj0": "tq.sj[0]", "sj1": "tq.sj[1]", "sj2": "tq.sj[2]",
         "sj3": "tq.sj[3]", "fu0j": "tq.g[0]", "fu1j": "tq.g[1]", "lxj": "tq.g[2]", "lux0j": "tq.g[3]", "lux1j": "tq.g[4]",
         "e0": "tq.g[5]", "e1": "tq.g[6]", "e2": "tq.g[7]", "lxx": "tq.lxx", "m0": "lc.m0", "m1": "lc.m1",
         "mk_i1": "sel.i1", "mk_j": "sel.jn0", "mk_r23": "sel.r23"}
first = True
for b in blocks:
    lines, rbw, writes, reads = [], [], [], []
    for k in b:
        if k is None:
            lines.append("s_nop 0")
            continue
        x = I[k]
        lines.append(NAME.sub(lambda m: "%[" + m.group(1) + "]", x["text"]))
        names = NAME.findall(x["text"])
        d = x["dst"]
        srcs = names[1:] + ([d] if x["acc"] else [])
        for nme in srcs:
            if nme not in writes and nme not in rbw:
                rbw.append(nme)          # read before any write in this statement: must come in with a value
        if d not in writes:
            writes.append(d)
    if first:
        lines.insert(0, "s_nop 1")       # a DPP read of an operand the compiler may just have copied: its 2 wait states
        first = False
    outs_new = [n for n in writes if n not in rbw]
    outs_rw = [n for n in writes if n in rbw]
    ins_ = [n for n in rbw if n not in writes]
    body = "\\n\\t\"\n        \"".join(lines)
    o = ", ".join([f'[{n}] "=&v"({n})' for n in outs_new] + [f'[{n}] "+v"({n})' for n in outs_rw])
    i_ = ", ".join(f'[{n}] "{"s" if n in SGPR_INPUTS else "v"}"({cexpr.get(n, n)})' for n in ins_)
    w("    asm volatile(")
    w(f'        "{body}"')
    w(f"        : {o}")
    w(f"        : {i_});")
w("    pd = (Q00 > 0.0f) && (det > 0.0f);")
w("    V = Vn;")
w("    vx = vxn;")
w("}")
sys.stdout.write("\n".join(out) + "\n")
sys.stderr.write(f"{N} instructions, {len(order)} slots, {stalls} wait states, soft shortfall {soft}, {len(blocks)} asm blocks\n")
