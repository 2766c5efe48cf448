#!/usr/bin/env python3
"""Live VGPRs per instruction of one kernel's gfx950 assembly (from scripts/kasm.sh): backward dataflow over the basic blocks.
Prints the pressure profile: for every basic block its line range, the maximum number of live VGPRs and where the maximum is.
`python scripts/vlive.py /tmp/k4/k.s [--at LINE]` (--at: the registers live at that line, with the line that defines each)."""
import re, sys
STORE = ("ds_write", "global_store", "scratch_store", "buffer_store", "global_atomic", "ds_add", "s_", "v_cmp", "v_cmpx", "ds_bpermute_b32_nodst")
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None: out.append(int(m.group(3)))
        else: out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
def parse(path):
    ins = []  # (lineno, text, defs, uses, label, branch_target, is_uncond)
    for n, raw in enumerate(open(path), 1):
        t = raw.split(";")[0].strip()
        if not t: continue
        if t.endswith(":"):
            ins.append((n, t, [], [], t[:-1], None, False)); continue
        if t.startswith("."): continue
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in rest.split(",")] if rest else []
        tgt = None
        if op.startswith("s_cbranch") or op == "s_branch": tgt = ops[0]
        defs, uses = [], []
        nodst = op.startswith(STORE) and not (op.startswith("global_atomic") and "sc0" in t and False)
        if op.startswith(("v_readlane", "v_readfirstlane")): nodst = True
        for i, o in enumerate(ops):
            rs = regs(o)
            if i == 0 and not nodst: defs += rs
            else: uses += rs
        if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot")) or "dpp" in t or "sdwa" in t or op.startswith(("v_writelane", "v_cndmask")) and False:
            uses += defs
        if op.startswith("v_writelane"): uses += defs
        ins.append((n, t, defs, uses, None, tgt, op == "s_branch" or op == "s_endpgm"))
    return ins
def main():
    path = sys.argv[1]
    at = int(sys.argv[sys.argv.index("--at") + 1]) if "--at" in sys.argv else None
    ins = parse(path)
    # basic blocks
    starts = {0}
    lab = {}
    for i, x in enumerate(ins):
        if x[4]: starts.add(i); lab[x[4]] = i
        if x[5] is not None or x[6]: starts.add(i + 1)
    starts = sorted(s for s in starts if s < len(ins))
    blocks = [(s, e) for s, e in zip(starts, starts[1:] + [len(ins)])]
    bidx = {s: k for k, (s, e) in enumerate(blocks)}
    succ = []
    for s, e in blocks:
        last = ins[e - 1]
        sc = []
        if last[5] is not None and last[5] in lab: sc.append(bidx[lab[last[5]]])
        if not last[6] and e < len(ins): sc.append(bidx[e])
        succ.append(sc)
    live_in = [set() for _ in blocks]; live_out = [set() for _ in blocks]
    changed = True
    while changed:
        changed = False
        for k in reversed(range(len(blocks))):
            s, e = blocks[k]
            out = set().union(*[live_in[j] for j in succ[k]]) if succ[k] else set()
            cur = set(out)
            for i in reversed(range(s, e)):
                cur -= set(ins[i][2]); cur |= set(ins[i][3])
            if out != live_out[k] or cur != live_in[k]:
                live_out[k], live_in[k] = out, cur; changed = True
    for k, (s, e) in enumerate(blocks):
        cur = set(live_out[k]); mx, where = len(cur), e - 1
        for i in reversed(range(s, e)):
            if at is not None and ins[i][0] == at:
                print("live at line %d: %d" % (at, len(cur)))
                for rr in sorted(cur):
                    d = next((ins[q] for q in range(i - 1, -1, -1) if rr in ins[q][2]), None)
                    print("  v%-3d last def above: %5s  %s" % (rr, d[0] if d else "-", d[1][:70] if d else ""))
            cur -= set(ins[i][2]); cur |= set(ins[i][3])
            if len(cur) > mx: mx, where = len(cur), i
        if at is None and e - s > 3:
            print("lines %5d-%5d  max live %3d at line %5d  (in %3d, out %3d)  %s" % (ins[s][0], ins[e - 1][0], mx, ins[where][0], len(live_in[k]), len(live_out[k]), ins[s][1][:40] if ins[s][4] else ""))
main()
