#!/usr/bin/env python3
"""rebank.py -- re-allocate the VGPRs of a kernel's hot loop so that its fused multiply-adds read their two factors
from different register banks.

Measured on gfx950 (profiles/r02_issue_costs.txt, 7a): the VGPR file reads as two banks, even and odd register
numbers.  A `v_fma_f32` / `v_fmac_f32` whose src0 and src1 are two VGPRs of ONE parity issues in 4 cycles instead of
2 (the addend's bank does not matter; two-source instructions and v_bitop3_b32 are not affected; a VOP3 v_fma_f32 that
names the same register twice reads it once and is not affected, the VOP2 v_fmac_f32 is).  hipcc's register allocator
does not know the banks: in the default scene's trace loop 11 of 24 such instructions are conflicted, ~22 of the
loop's ~256 cycles per trip.

The pass works on the compiler's assembly (hipcc --cuda-device-only -S) and touches ONE loop: the single-block loop
with two v_sin_f32 (phase 1 of trace_static_kernel, where a frame spends 95 % of its trips), entered only through its
preheader and left only by falling out of its backward branch.  Inside that block it
  1. splits every register into its live ranges ("webs": a full write starts one; a write under a narrowed exec mask
     or an accumulating v_fmac continues one; what is read before it is written is live around the back edge and
     shares a web with the block's last write of that register);
  2. keeps every web that is ever part of a register tuple v[a:b] (64-bit products, LDS vector reads) where the
     compiler put it, and searches registers for the others -- any register below the kernel's accumulation offset
     that is free over the web's whole range -- minimising the number of same-bank multiply-adds;
  3. rewrites the block, turns `v_fmac_f32 d, x, x` into `v_fma_f32 d, x, x, d`, and places the moves that carry values
     into their new registers in front of the loop and back behind it (a parallel copy, v_swap_b32 for cycles).
Data flow, order, hazards and wait states are unchanged: they depend on which instruction feeds which, not on the
numbers.  Anything the pass does not understand (an unknown VGPR-touching mnemonic, DPP/SDWA, memory instructions in
the loop, a second exit, exec left narrowed at the back edge) makes it stop and leave the file as it is.

usage: rebank.py in.s out.s [--report]         (exit status 0 also when nothing was done: out.s is then a copy)
"""
import random
import re
import sys

FMA = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_mad_f32", "v_mac_f32")
VSINGLE = re.compile(r"(?<![\w\[:])v(\d+)\b")
VTUPLE = re.compile(r"\bv\[(\d+):(\d+)\]")


class Skip(Exception):
    pass


def kernel_span(lines):
    b = e = None
    for k, ln in enumerate(lines):
        if b is None and re.match(r"^_Z\w+:", ln):
            b = k + 1
        elif b is not None and re.match(r"^\.Lfunc_end\d+:", ln):
            e = k
            break
    if b is None or e is None:
        raise Skip("no kernel body found")
    return b, e


def hot_loop(lines, b, e):
    label_at = {}
    for k in range(b, e):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[k])
        if m:
            label_at[m.group(1)] = k
    best = None
    for k in range(b, e):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[k])
        if not (m and m.group(1) in label_at and label_at[m.group(1)] < k):
            continue
        lo, hi = label_at[m.group(1)], k
        body = lines[lo + 1:hi]
        if sum("v_sin_f32" in x for x in body) != 2:
            continue
        if any(re.match(r"\s+s_(c?branch|setpc|swappc|call|endpgm)", x) or re.match(r"^\.LBB", x) for x in body):
            continue
        if best is None or hi - lo > best[1] - best[0]:
            best = (lo, hi)
    if best is None:
        raise Skip("the trace loop was not found")
    lo, hi = best
    label = re.match(r"^(\.LBB\d+_\d+):", lines[lo]).group(1)
    refs = [k for k in range(b, e) if k != lo and re.search(re.escape(label) + r"(?!\d)", lines[k])]
    if refs != [hi]:
        raise Skip(f"{label} is entered from elsewhere than its preheader and its back edge")
    return lo, hi


def split_ops(rest):
    return [x.strip() for x in rest.split(",")] if rest.strip() else []


class Instr:
    def __init__(self, line_no, text):
        self.line_no = line_no
        code, sep, self.comment = text.partition(";")
        self.sep = sep
        self.indent = re.match(r"\s*", code).group(0)
        body = code.strip()
        self.mnem, _, rest = body.partition(" ")
        # trailing modifiers (bitop3:0x96, offset:16, op_sel ...) stay with the last operand's text; they hold no VGPRs
        self.ops = split_ops(rest)
        self.op_webs = []            # per operand: {register: web} for the single registers in it


def analyse(instrs):
    """Webs of the block.  Returns (webs, reg_final, live_in) where webs[w] = dict(reg, pinned, mask)."""
    n = len(instrs)
    parent = []
    info = []                        # per sub-web: reg, first_def (or -1 for live-in), last_touch, pinned

    def new_web(reg, first_def):
        parent.append(len(parent))
        info.append({"reg": reg, "first_def": first_def, "last": first_def, "pinned": False, "to_end": False})
        return len(parent) - 1

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    cur, live_in = {}, {}
    partial = False
    for p, ins in enumerate(instrs):
        m = ins.mnem
        ops = ins.ops
        if not m or m.startswith(("s_waitcnt", "s_nop")):
            ins.op_webs = [{} for _ in ops]
            continue
        touches_v = any(VSINGLE.search(o) or VTUPLE.search(o) for o in ops)
        if m.startswith("s_"):
            if touches_v:
                raise Skip(f"scalar instruction with a VGPR operand: {m}")
            if "saveexec" in m:
                partial = True
            elif ops and ops[0] == "exec":
                partial = not (m == "s_mov_b64")
            ins.op_webs = [{} for _ in ops]
            continue
        if any(k in " ".join(ops) for k in ("dpp", "sdwa", "row_", "quad_perm", "dst_sel", "src0_sel")):
            raise Skip(f"DPP/SDWA in the loop: {m}")
        if m.startswith("v_cmpx") or m.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_swap",
                                                   "v_mfma", "v_accvgpr", "v_div_scale", "v_div_fmas")):
            raise Skip(f"not handled: {m}")
        if m.startswith("v_cmp"):
            def_ops, tied = set(), False
        elif m.startswith(("v_fmac", "v_mac")):
            def_ops, tied = {0}, True
        elif m.startswith("v_"):
            def_ops, tied = {0}, False
        elif m.startswith(("ds_read", "ds_load")):
            def_ops, tied = {0}, False
        elif m.startswith(("ds_write", "ds_store")):
            def_ops, tied = set(), False
        else:
            raise Skip(f"not handled: {m}")
        ins.op_webs = [{} for _ in ops]
        # uses first (an instruction reads before it writes), then definitions
        for phase in ("use", "def"):
            for k, o in enumerate(ops):
                is_def = k in def_ops
                regs = [(int(x.group(1)), False) for x in VSINGLE.finditer(o)]
                for x in VTUPLE.finditer(o):
                    regs += [(r, True) for r in range(int(x.group(1)), int(x.group(2)) + 1)]
                for r, in_tuple in regs:
                    reads = (not is_def) or tied or partial
                    if phase == "use" and reads:
                        if r not in cur:
                            live_in[r] = cur[r] = new_web(r, -1)
                        w = cur[r]
                        info[w]["last"] = max(info[w]["last"], p)
                        if in_tuple:
                            info[w]["pinned"] = True
                        if not is_def:
                            ins.op_webs[k][r] = w
                    if phase == "def" and is_def:
                        if tied or partial:
                            w = cur[r]                 # continues (the use phase made sure it exists)
                            info[w]["last"] = max(info[w]["last"], p + 1)
                        else:
                            w = cur[r] = new_web(r, p)
                            info[w]["last"] = p + 1
                        if in_tuple:
                            info[w]["pinned"] = True
                        ins.op_webs[k][r] = w
    if partial:
        raise Skip("exec is left narrowed at the back edge")
    # the back edge: what is read before written shares a web with the block's last write of that register; every
    # register's last web is kept to the end of the block (it may be read behind the loop)
    for r, w in cur.items():
        info[w]["to_end"] = True
        if r in live_in and find(live_in[r]) != find(w):
            parent[find(w)] = find(live_in[r])
    webs = {}
    for w, d in enumerate(info):
        root = find(w)
        g = webs.setdefault(root, {"reg": d["reg"], "pinned": False, "mask": 0})
        assert g["reg"] == d["reg"]
        g["pinned"] |= d["pinned"]
        lo = 0 if d["first_def"] < 0 else d["first_def"] + 1          # points: k = between instruction k-1 and k
        hi = n if d["to_end"] else max(d["last"], lo)
        g["mask"] |= ((1 << (hi + 1)) - 1) & ~((1 << lo) - 1)
    for ins in instrs:
        ins.op_webs = [{r: find(w) for r, w in d.items()} for d in ins.op_webs]
    return webs, {r: find(w) for r, w in cur.items()}, {r: find(w) for r, w in live_in.items()}


def fma_factors(ins):
    """-> the two (register, web) factors of a multiply-add whose src0 and src1 are plain VGPRs, else None."""
    if not ins.mnem.startswith(FMA):
        return None
    out = []
    for k in (1, 2):
        if k >= len(ins.ops):
            return None
        m = re.fullmatch(r"-?\|?v(\d+)\|?", ins.ops[k])
        if not m:
            return None
        r = int(m.group(1))
        out.append(ins.op_webs[k][r])
    return tuple(out)


def parallel_copy(copies):
    """copies: {dst: src}, all read before any is written -> a list of instructions."""
    pending = {d: s for d, s in copies.items() if d != s}
    out = []
    while pending:
        srcs = set(pending.values())
        ready = [d for d in pending if d not in srcs]
        if ready:
            d = ready[0]
            out.append(f"\tv_mov_b32_e32 v{d}, v{pending.pop(d)}")
            continue
        d = next(iter(pending))                    # only cycles are left
        s = pending.pop(d)
        out.append(f"\tv_swap_b32 v{d}, v{s}")      # d has its value; what d held is in s now
        for d2 in list(pending):
            if pending[d2] == d:
                pending[d2] = s
            if pending[d2] == d2:
                del pending[d2]
    return out


def rebank(lines, report=False):
    b, e = kernel_span(lines)
    lo, hi = hot_loop(lines, b, e)
    instrs = [Instr(k, lines[k]) for k in range(lo + 1, hi)
              if lines[k].startswith(("\t", " ")) and lines[k].split(";")[0].strip()]
    webs, final, live_in = analyse(instrs)
    n = len(instrs)

    text = "\n".join(lines)
    next_free = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", text).group(1))
    m = re.search(r"\.amdhsa_accum_offset\s+(\d+)", text)
    limit = int(m.group(1)) if m else next_free    # free registers below the accumulation offset cost no occupancy
    limit = min(limit, (next_free + 7) // 8 * 8)   # ... as long as the allocation granule (8) does not grow
    referenced = {g["reg"] for g in webs.values()}
    full = (1 << (n + 1)) - 1
    occ = [0] * limit
    for r in range(min(next_free, limit)):
        if r not in referenced:
            occ[r] = full                          # not named in the loop: may hold a value that lives across it
    assign = {}
    for w, g in webs.items():
        assign[w] = g["reg"]
        if g["reg"] >= limit:
            raise Skip("a register above the accumulation offset")
        assert occ[g["reg"]] & g["mask"] == 0, "the compiler's own allocation overlaps: the web analysis is wrong"
        occ[g["reg"]] |= g["mask"]
    edges = [f for f in (fma_factors(i) for i in instrs) if f]
    squares = sum(1 for x, y in edges if x == y)
    edges = [(x, y) for x, y in edges if x != y]
    free = [w for w, g in webs.items() if not g["pinned"]]
    by_web = {}
    for x, y in edges:
        by_web.setdefault(x, []).append(y)
        by_web.setdefault(y, []).append(x)

    def cost(a):
        return sum(1 for x, y in edges if (a[x] ^ a[y]) & 1 == 0)

    def local(a, o, w):
        return sum(1 for y in by_web.get(w, ()) if (a[w] ^ a[y]) & 1 == 0)

    before = cost(assign)
    rng = random.Random(1)
    best, best_c, best_moved = dict(assign), before, 0
    for restart in range(60):
        a, o = dict(assign), list(occ)

        def move(w, c):
            o[a[w]] &= ~webs[w]["mask"]
            a[w] = c
            o[c] |= webs[w]["mask"]

        if restart:
            for _ in range(rng.randint(2, 12)):        # shake: random legal moves
                w = rng.choice(free)
                cands = [c for c in range(limit) if c != a[w] and o[c] & webs[w]["mask"] == 0]
                if cands:
                    move(w, rng.choice(cands))
        improved = True
        while improved:
            improved = False
            for w in free:
                here = local(a, o, w)
                if here == 0:
                    continue
                home = a[w]
                for c in range(limit):
                    if (c ^ home) & 1 == 0 or o[c] & webs[w]["mask"]:
                        continue
                    move(w, c)
                    if local(a, o, w) < here:
                        improved = True
                        break
                    move(w, home)
        c = cost(a)
        for w in free:                                 # a move that bought nothing goes back
            home = webs[w]["reg"]
            if a[w] != home and o[home] & webs[w]["mask"] == 0:
                was = a[w]
                move(w, home)
                if cost(a) > c:
                    move(w, was)
        moved = sum(1 for w in free if a[w] != webs[w]["reg"])
        if c < best_c or (c == best_c and moved < best_moved):
            best, best_c, best_moved = dict(a), c, moved

    # ---- rewrite the block
    n_sq = 0
    for ins in instrs:
        ops = list(ins.ops)
        for k, o in enumerate(ops):
            mp = ins.op_webs[k] if k < len(ins.op_webs) else {}
            ops[k] = VSINGLE.sub(lambda mo: f"v{best[mp[int(mo.group(1))]]}" if int(mo.group(1)) in mp else mo.group(0), o)
        mnem = ins.mnem
        if mnem == "v_fmac_f32_e32" and len(ops) == 3 and ops[1] == ops[2] and re.fullmatch(r"v\d+", ops[1]):
            mnem, ops = "v_fma_f32", [ops[0], ops[1], ops[2], ops[0]]      # the VOP3 form reads the repeated register once
            n_sq += 1
        if mnem != ins.mnem or ops != ins.ops:         # (untouched lines keep the compiler's text)
            lines[ins.line_no] = f"{ins.indent}{mnem} {', '.join(ops)}".rstrip() + (" " + ins.sep + ins.comment if ins.sep else "")
    entry = {best[w]: r for r, w in live_in.items()}
    exit_ = {r: best[w] for r, w in final.items()}
    pre = parallel_copy(entry)
    post = parallel_copy(exit_)
    note = "\t; rebank.py: the loop below runs on re-allocated VGPRs (multiply-add factors in different banks)"
    lines[hi + 1:hi + 1] = post
    lines[lo:lo] = [note] + pre
    top = max(list(best.values()) + [0]) + 1
    out = "\n".join(lines)
    if top > next_free:
        out = re.sub(r"(\.amdhsa_next_free_vgpr\s+)\d+", lambda mo: mo.group(1) + str(top), out, count=1)
        out = re.sub(r"(\.vgpr_count:\s+)\d+", lambda mo: mo.group(1) + str(top), out, count=1)
    msg = (f"rebank: loop at lines {lo}-{hi}: {n} instructions, {len(webs)} webs ({len(webs) - len(free)} kept in place: tuples), "
           f"{len(edges) + squares} multiply-adds with two VGPR factors: same-bank {before} + {squares} squares -> {best_c} + "
           f"{squares - n_sq} squares; {best_moved} webs moved, {len(pre)} + {len(post)} moves around the loop; registers used: {max(top, next_free)}")
    if report:
        for w in free:
            if best[w] != webs[w]["reg"]:
                msg += f"\n  web of v{webs[w]['reg']} -> v{best[w]}"
        msg += "\n  entry: " + "; ".join(x.strip() for x in pre) + "\n  exit:  " + "; ".join(x.strip() for x in post)
    return out.split("\n"), msg


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) != 2:
        raise SystemExit(__doc__)
    lines = open(args[0]).read().split("\n")
    try:
        out, msg = rebank(list(lines), "--report" in sys.argv)
    except Skip as why:
        out, msg = lines, f"rebank: nothing done ({why})"
    open(args[1], "w").write("\n".join(out))
    print(msg, file=sys.stderr)


if __name__ == "__main__":
    main()
