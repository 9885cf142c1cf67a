#!/usr/bin/env python
"""Generator of morpheus_amd/csrc/field_bwd_b3_gen.h: the tile loops of the fused field backward (bf16 x 3 form), software-pipelined
at C++ STATEMENT granularity.

Why a generator.  The kernels run ONE wave per SIMD (160-192 accumulator registers), so nothing hides a vector instruction but the
wave's own MFMAs: 5-6 single-issue instructions per v_mfma_f32_32x32x16_bf16 are free (tools/micro/mfma_valu_gap.hip), the rest is
paid in full.  hipcc's scheduler does not produce that interleave for these kernels: sched_group_barrier pipelines are dropped in the
register-tight regions (the schedule reverts to source order: round 5's FUSED_FILL experiment, round 6's first two forms of this
kernel -- 80 and 357 spilled registers).  So the SOURCE ORDER is made the schedule: every MFMA is followed by the few statements
that ride in its shadow and a full scheduling barrier (__builtin_amdgcn_sched_barrier(0)), and which statements go where is decided
here, by a list scheduler over the tile's statement DAG:

  * the MFMAs keep their order (backward-data of a layer, its weight gradient, the next layer ...);
  * every other statement ("filler": one split2 = 11 VALU, one mask = 3, one LDS / global access = 1 ...) has a DEADLINE, the MFMA that
    first needs its result minus a latency margin (LDS reads are issued >= 4 MFMAs ahead, a layer's input rows a whole layer ahead),
    and is ready once its producers were emitted (+ their latency);
  * after each MFMA the scheduler spends a credit of K instruction slots on the ready fillers with the earliest deadlines; a filler
    whose deadline arrives is emitted regardless (a longer gap, never a wrong order);
  * the loop is rotated by one stage: the per-point inputs of the NEXT tile are fetched at the top of the body and its first slices
    and weight fragments are made under the current tile's last MFMAs.

    tools/experiments/field_bwd_pipeline/build_variant.sh [--fill 6] [-DFB_ASM_MEM=1]    (an EXPERIMENT: not part of the product build)
"""
import argparse
import sys

A_PL = ["l", "m", "h", "m", "h", "h"]      # plane of the A operand in the six slice products, small terms first (mlp.hip: dw_mma_b3)
B_PL = ["h", "m", "l", "h", "m", "h"]
PLN = {"h": 0, "m": 1, "l": 2}
MFMA = "__builtin_amdgcn_mfma_f32_32x32x16_bf16"


class Op:
    __slots__ = ("text", "cost", "reads", "writes", "lat", "asap", "earliest", "mfma", "idx", "after", "mem", "dests", "loads", "post")

    def __init__(self, text, cost=1, reads=(), writes=(), lat=0, mfma=False, earliest=0, asap=False, after=(), deadline=None, mem=None, dests=(), post=None):
        self.text, self.cost, self.reads, self.writes, self.lat = text, cost, tuple(reads), tuple(writes), lat
        self.mfma, self.earliest, self.after = mfma, earliest, tuple(after)
        self.asap = asap or deadline is not None             # (an explicit deadline = "as early as possible")
        # mem: None, or the counter an inline-asm memory operation ticks ("lds" -> lgkmcnt, "vm" -> vmcnt); dests: the C++ lvalues an asm
        # LOAD writes -- hipcc keeps no book on them: Program.insert_waits() puts a counted s_waitcnt in front of their first consumer
        # post: a statement to run right behind the wait that pins the destinations (an asm load NEVER writes part of a larger object:
        # hipcc would copy the half-written tuple at once -- garbage -- and free the destination register for reuse before the data
        # lands; it writes a variable of its own, and `post` moves that into place once it is there)
        self.mem, self.dests, self.loads, self.post = mem, tuple(dests), (), post


class Program:
    """statements of one tile in PROGRAM ORDER (a legal sequential order); schedule() re-orders the fillers around the MFMAs"""

    def __init__(self):
        self.ops = []
        self.initial = set()      # tokens available at loop entry (made by the previous iteration / the pre-loop prologue)

    def add(self, *a, **k):
        op = Op(*a, **k)
        op.idx = len(self.ops)
        self.ops.append(op)
        return op

    def n_mfma(self):
        return sum(1 for o in self.ops if o.mfma)

    def schedule(self, fill):
        """-> the statements in emission order (None = scheduling barrier).  Gap g = the slot behind MFMA g (gap -1: in front of the first).
        Fillers that FREE registers or start long-latency loads (asap=True: stores to LDS, global loads, the adds that consume a loaded
        row) go to the earliest gap their producers allow; fillers that ALLOCATE registers (slices, fragment reads, masks) go to the
        LATEST gap their consumers allow that still has room (capacity `fill` instruction slots per gap), so that a value is made
        just before it is used: the schedule's register pressure is what a hand-written pipeline would have."""
        ops = self.ops
        writer = {}
        for o in ops:
            for t in o.writes:
                writer.setdefault(t, []).append(o)
        touch = {}                         # token -> ops reading or writing it, program order (for write-after-read / -write edges)
        for o in ops:
            for t in set(o.reads) | set(o.writes):
                touch.setdefault(t, []).append(o)
        prod, cons = {o.idx: [] for o in ops}, {o.idx: [] for o in ops}
        carried = {}
        for o in ops:
            ps = {}
            for t in o.reads:
                ws = [w for w in writer.get(t, []) if w.idx < o.idx]
                if ws:
                    ps[ws[-1].idx] = (ws[-1], True)          # true dependence: the producer's latency applies
                elif t not in self.initial:
                    raise RuntimeError(f"token {t!r} read by {o.text[:60]!r} has no producer")
                elif writer.get(t):
                    carried.setdefault(o.idx, []).append(writer[t][-1])      # made by the PREVIOUS iteration's last writer
            for t in o.writes:
                for q in touch[t]:
                    if q.idx < o.idx and q.idx not in ps:
                        ps[q.idx] = (q, False)               # ordering only
            for q in o.after:
                ps.setdefault(q.idx, (q, False))
            prod[o.idx] = list(ps.values())
            o.loads = tuple(q for q, true_dep in prod[o.idx] if true_dep and q.mem and q.dests) + \
                tuple(q for q in carried.get(o.idx, []) if q.mem and q.dests)
            for q, true_dep in prod[o.idx]:
                cons[q.idx].append((o, true_dep))
        mf_index, g = {}, 0
        for o in ops:
            if o.mfma:
                mf_index[o.idx] = g
                g += 1
        n_mf = g
        # earliest gap (forward, no capacity): a filler behind MFMA m sits in gap >= m; latencies in bundles
        est = {}
        for o in ops:
            if o.mfma:
                est[o.idx] = mf_index[o.idx]
                continue
            e = max(o.earliest - 1, -1)
            for q, true_dep in prod[o.idx]:
                lat = q.lat if true_dep else 0
                e = max(e, est[q.idx] + (lat if not q.mfma else max(lat, 0)))
            est[o.idx] = e
        # latest gap (backward): in front of MFMA m means gap <= m - 1
        lst = {}
        for o in reversed(ops):
            if o.mfma:
                lst[o.idx] = mf_index[o.idx] - 1       # (as a producer bound for ITS producers: they must be in gaps <= m - 1)
                continue
            l = n_mf - 1
            for c, true_dep in cons[o.idx]:
                lat = o.lat if true_dep else 0
                l = min(l, (lst[c.idx] if c.mfma else lst[c.idx]) - lat)
            lst[o.idx] = l
        gap, load = {}, {}
        fillers = [o for o in ops if not o.mfma]
        late = 0
        # ALAP fillers, consumers first (reverse program order is a reverse topological order)
        for o in reversed(fillers):
            if o.asap:
                continue
            hi = lst[o.idx]
            for c, true_dep in cons[o.idx]:
                if not c.mfma and c.idx in gap:
                    hi = min(hi, gap[c.idx] - (o.lat if true_dep else 0))
            lo = est[o.idx]
            if hi < lo:
                late += 1
                hi = lo
            gpos = hi
            while gpos > lo and load.get(gpos, 0) + o.cost > fill + 0.01 and load.get(gpos, 0) > 0:
                gpos -= 1
            if load.get(gpos, 0) + o.cost > fill + 0.01 and load.get(gpos, 0) > 0:
                gpos = min(range(lo, hi + 1), key=lambda x: (load.get(x, 0), -x))      # no room anywhere: the emptiest gap of the window
            gap[o.idx] = gpos
            load[gpos] = load.get(gpos, 0) + o.cost
        # ASAP fillers, forward; then a forward fix-up so that nothing sits in front of a producer
        for o in fillers:
            e = max(o.earliest - 1, -1)
            for q, true_dep in prod[o.idx]:
                lat = q.lat if true_dep else 0
                e = max(e, (mf_index[q.idx] if q.mfma else gap[q.idx]) + lat)
            if o.asap:
                while o.cost > 0 and e < lst[o.idx] and load.get(e, 0) + o.cost > fill + 2:      # (room permitting, never past its own deadline)
                    e += 1
                gap[o.idx] = e
                load[e] = load.get(e, 0) + o.cost
            elif gap[o.idx] < e:
                load[gap[o.idx]] -= o.cost
                gap[o.idx] = e
                load[e] = load.get(e, 0) + o.cost
        # latencies are wishes (the hardware interlocks), ORDER is not: where the two passes disagree (a layer too short for its
        # latencies), clamp backward then forward with order-only constraints -- program order is a legal order, so this settles
        for o in reversed(fillers):
            lim = n_mf - 1
            for c, _ in cons[o.idx]:
                lim = min(lim, mf_index[c.idx] - 1 if c.mfma else gap[c.idx])
            gap[o.idx] = min(gap[o.idx], lim)
        for o in fillers:
            e = max(o.earliest - 1, -1)
            for q, _ in prod[o.idx]:
                e = max(e, mf_index[q.idx] if q.mfma else gap[q.idx])
            gap[o.idx] = max(gap[o.idx], e)
        bad = 0
        for o in fillers:
            for c, _ in cons[o.idx]:
                if (c.mfma and gap[o.idx] > mf_index[c.idx] - 1) or (not c.mfma and gap[o.idx] > gap[c.idx]):
                    bad += 1
            for q, _ in prod[o.idx]:
                if (q.mfma and gap[o.idx] < mf_index[q.idx]) or (not q.mfma and gap[o.idx] < gap[q.idx]):
                    bad += 1
        assert bad == 0, f"{bad} statements out of order"
        load = {}
        for o in fillers:
            load[gap[o.idx]] = load.get(gap[o.idx], 0) + o.cost
        by_gap = {}
        for o in fillers:
            by_gap.setdefault(gap[o.idx], []).append(o)
        out = []
        for o in sorted(by_gap.get(-1, []), key=lambda q: q.idx):
            out.append(o)
        if by_gap.get(-1):
            out.append(None)
        for o in ops:
            if not o.mfma:
                continue
            gi = mf_index[o.idx]
            out.append(o)
            for q in sorted(by_gap.get(gi, []), key=lambda q: q.idx):
                out.append(q)
            out.append(None)
        loads = [load.get(x, 0) for x in range(-1, n_mf)]
        self.stats = dict(mfma=n_mf, filler_cost=sum(o.cost for o in fillers), fill=fill, gap_max=max(loads), gaps_over=sum(1 for x in loads if x > fill + 2),
                          late=late, misplaced=bad, model_cycles=int(sum(max(33.3 + 0.3 * x, 5 * x + 7.8) for x in loads)))
        return self.insert_waits(out)

    def insert_waits(self, seq):
        """Counted waits for the inline-asm loads.  The asm statements keep their relative order (volatile), LDS operations of a wave and
        its vector-memory operations each complete in issue order, so "all but the N youngest operations of the counter" covers a
        load exactly when N = the operations of its counter issued behind it.  The body is walked TWICE (a load of the previous
        iteration -- the rotated stage, the row slots -- is consumed in this one): the second walk's waits are the steady state, and
        right for the first iteration too (fewer operations outstanding than assumed: the wait is satisfied at once)."""
        CAP = {"lds": 15, "vm": 63}
        real = [o for o in seq if o is not None]
        issued = {"lds": 0, "vm": 0}
        at = {}                                   # load op -> (counter, its ordinal) of its most recent issue
        covered = {"lds": 0, "vm": 0}             # operations of the counter with ordinal <= covered are known complete
        out, n_waits = [], 0
        pinned = {}                               # load op -> ordinal of the issue whose destinations were last pinned behind a wait
        for walk in range(2):
            for o in seq:
                if o is None:
                    if walk == 1:
                        out.append(None)
                    continue
                need, pin_only = {}, []
                for q in o.loads:
                    if q in at:
                        c, k = at[q]
                        if k > covered[c]:
                            need.setdefault(c, []).append(q)
                        elif pinned.get(q) != k:
                            # complete by an earlier wait that did not NAME it: without a statement ordered behind that wait the
                            # compiler may schedule this consumer above it -- an empty asm on the destinations is that statement
                            pin_only.append(q)
                for c, qs in need.items():
                    n = min(issued[c] - at[q][1] for q in qs)
                    n = min(n, CAP[c])
                    covered[c] = max(covered[c], issued[c] - n)
                    for q in qs:
                        pinned[q] = at[q][1]
                    if walk == 1:
                        ds = []
                        for q in qs:
                            for d in q.dests:
                                if d not in ds:
                                    ds.append(d)
                        cnt = f"lgkmcnt({n})" if c == "lds" else f"vmcnt({n})"
                        for k0 in range(0, len(ds), 12):      # (an asm statement takes at most 30 operands; 12 keeps the lines readable)
                            ops = ", ".join(f'"+v"({d})' for d in ds[k0:k0 + 12])
                            out.append(Op(f'FB_WAIT("{cnt}", {ops});', cost=0))
                            n_waits += 1
                        for q in qs:
                            if q.post:
                                out.append(Op(q.post, cost=0))
                for q in pin_only:
                    pinned[q] = at[q][1]
                    if walk == 1:
                        ops = ", ".join(f'"+v"({d})' for d in q.dests)
                        out.append(Op(f'FB_PIN({ops});', cost=0))
                        if q.post:
                            out.append(Op(q.post, cost=0))
                if o.mem:
                    issued[o.mem] += 1
                    if o.dests:
                        at[o] = (o.mem, issued[o.mem])
                if walk == 1:
                    out.append(o)
        self.stats["waits"] = n_waits
        return out


BARRIERS = False      # a scheduling barrier behind every MFMA bundle: measured 2.3 x SLOWER (see main())


def emit_text(seq, indent="        "):
    lines = []
    for o in seq:
        if o is None:
            if BARRIERS:
                lines.append(indent + "__builtin_amdgcn_sched_barrier(0);")
        else:
            for ln in o.text.split("\n"):
                lines.append(indent + ln)
    return "\n".join(lines)


# ------------------------------------------------------------------------------------------------ statement builders
def bb(L, s, pl):
    return f"bb_{L}[{s}][{PLN[pl]}]"


def split_pair(P, L, vin, e, **kw):
    """split2 of column pair e in two statements of 5 and 6 instructions (a gap holds one of them): hi slice + residuals, then mid / lo"""
    s, k = e >> 2, e & 3
    P.add(f"split2a({vin}[{2 * e}], {vin}[{2 * e + 1}], {bb(L, s, 'h')}.u[{k}], rr_{L}[{2 * e}], rr_{L}[{2 * e + 1}]);",
          cost=5, reads=[f"{vin}.{2 * e}", f"{vin}.{2 * e + 1}"], writes=[f"{L}.r.{e}"], **kw)
    return P.add(f"split2b(rr_{L}[{2 * e}], rr_{L}[{2 * e + 1}], {bb(L, s, 'm')}.u[{k}], {bb(L, s, 'l')}.u[{k}]);",
                 cost=6, reads=[f"{L}.r.{e}"], writes=[f"{L}.p.{e}"], **kw)


def zero_pair(P, L, e, **kw):
    s, k = e >> 2, e & 3
    return P.add(f"{bb(L, s, 'h')}.u[{k}] = {bb(L, s, 'm')}.u[{k}] = {bb(L, s, 'l')}.u[{k}] = 0u;", cost=0, writes=[f"{L}.p.{e}"], **kw)


def wload(P, L, wt, S, MT, s, **kw):
    """A fragments of k16 step s: wt = (per-lane LDS address variable of the layer's block, ...) -- inline asm: the compiler would sink a
    plain LDS read to its use and wait there (measured: a third of the kernel at s_waitcnt)"""
    PL = MT * S * 64
    for t in range(MT):
        for pl in "hml":
            P.add(f"FB_DS_READ128(W_{L}[{s}][{PLN[pl]}][{t}].f, {wt}, {(PLN[pl] * PL + (t * S + s) * 64) * 16});", cost=1, writes=[f"{L}.W.{s}.{pl}.{t}"],
                  lat=4, mem="lds", dests=[f"W_{L}[{s}][{PLN[pl]}][{t}].f"], **kw)


def bd_mfmas(P, L, MT, s, first, acc):
    for k in range(6):
        for t in range(MT):
            c = "zero16" if (first and k == 0) else f"{acc}[{t}]"
            P.add(f"{acc}[{t}] = {MFMA}(W_{L}[{s}][{PLN[A_PL[k]]}][{t}].h, {bb(L, s, B_PL[k])}.h, {c}, 0, 0, 0);", mfma=True,
                  reads=[f"{L}.W.{s}.{A_PL[k]}.{t}"] + [f"{L}.p.{e}" for e in range(4 * s, 4 * s + 4)] + ([] if (first and k == 0) else [f"{L}.acc.{t}"]),
                  writes=[f"{L}.acc.{t}"], lat=2)


def put(P, L, vin, T, **kw):
    """out tile T of layer L's sliced column -> plane image + fp32 scratch (the single 32-row buffer of the wave)"""
    ops = []
    for q in range(4):
        e = 8 * T + 2 * q
        for pl in "hml":
            ops.append(P.add(f"FB_DS_WRITE64(L.img_w, (u32x2_t{{{bb(L, e >> 2, pl)}.u[{e & 3}], {bb(L, e >> 2, pl)}.u[{(e & 3) + 1}]}}), {PLN[pl]} * FB_PLANE_BYTES + {q} * 2 * FB_CHUNKS * 8);",
                             cost=1, reads=[f"{L}.p.{e}", f"{L}.p.{e + 1}", "buf.free"], writes=[f"{L}.img.{T}"], asap=True, mem="lds", **kw))
    for r in range(16):
        ops.append(P.add(f"FB_DS_WRITE32(L.scr_w, {vin}[{16 * T + r}], {(r & 3) + 8 * (r >> 2)} * SCR_STRIDE * 4);", cost=1,
                         reads=[f"{vin}.{16 * T + r}", "buf.free"], writes=[f"{L}.img.{T}"], asap=True, mem="lds", **kw))
    return ops


def get(P, L, T, **kw):
    ops = []
    for s in range(2):
        for u in range(2):
            for pl in "hml":
                ops.append(P.add(f"FB_DS_READTR(tA_{L}[{T}][{PLN[pl]}][{s}][{u}], L.img_r, {PLN[pl]} * FB_PLANE_BYTES + {s * 64 + u * 32});",
                                 cost=1, reads=[f"{L}.img.{T}"], writes=[f"{L}.A.{T}.{pl}.{s}.{u}"], lat=4, mem="lds",
                                 dests=[f"tA_{L}[{T}][{PLN[pl]}][{s}][{u}]"],
                                 post=f"A_{L}[{T}][{PLN[pl]}][{s}].u[{2 * u}] = tA_{L}[{T}][{PLN[pl]}][{s}][{u}][0]; A_{L}[{T}][{PLN[pl]}][{s}].u[{2 * u + 1}] = tA_{L}[{T}][{PLN[pl]}][{s}][{u}][1];", **kw))
    return ops


def bsum(P, L, T, var, **kw):
    lds = []
    for j in range(4):
        lds.append(P.add(f"FB_DS_READ128(bs_{L}[{j}], L.scr_r, {16 * j});", cost=1, reads=[f"{L}.img.{T}"],
                         writes=[f"{L}.bs.{T}.{j}"], lat=4, asap=True, mem="lds", dests=[f"bs_{L}[{j}]"], **kw))
    for j in range(4):
        P.add(f"{var} += (bs_{L}[{j}][0] + bs_{L}[{j}][1]) + (bs_{L}[{j}][2] + bs_{L}[{j}][3]);", cost=4, reads=[f"{L}.bs.{T}.{j}"],
              writes=[f"{var}.acc"], asap=True)
    return lds


def slice_rows(P, L, n, slot, **kw):
    """parked input rows in register slot `slot` -> the B slices of in tile n (8 split2)"""
    for s in range(2):
        for e2 in range(4):
            j, q = 2 * s + (e2 >> 1), 2 * (e2 & 1)
            P.add(f"split2a(raw[{slot}].v[{j}][{q}], raw[{slot}].v[{j}][{q + 1}], Bs_{L}[{n}][0][{s}].u[{e2}], rb_{L}[{n}][{8 * s + 2 * e2}], rb_{L}[{n}][{8 * s + 2 * e2 + 1}]);",
                  cost=5, reads=[f"raw.{slot}"], writes=[f"{L}.Br.{n}.{s}.{e2}", f"raw.{slot}.used"], **kw)
            P.add(f"split2b(rb_{L}[{n}][{8 * s + 2 * e2}], rb_{L}[{n}][{8 * s + 2 * e2 + 1}], Bs_{L}[{n}][1][{s}].u[{e2}], Bs_{L}[{n}][2][{s}].u[{e2}]);",
                  cost=6, reads=[f"{L}.Br.{n}.{s}.{e2}"], writes=[f"{L}.B.{n}.{s}"], **kw)


def reload(P, slot, tile, row, **kw):
    """the slot's next content, as soon as the slot is free (these loads want the longest possible head start)"""
    for j in range(4):
        P.add(f"FB_GLOAD128(raw[{slot}].v[{j}], v_rowoff, {tile} + {row} * TILE, {16 * j});", cost=1, writes=[f"raw.{slot}"],
              asap=True, mem="vm", dests=[f"raw[{slot}].v[{j}]"], **kw)


def release(P, L, tiles):
    """the wave's transposition buffer is free once the layer's last transposing reads and row-sum reads were ISSUED (LDS is in order)"""
    P.add("", cost=0, reads=[f"{L}.A.{T}.{pl}.{s}.{u}" for T in tiles for pl in "hml" for s in range(2) for u in range(2)] + [f"{L}.bs.{T}.{j}" for T in tiles for j in range(4)],
          writes=["buf.free"], asap=True)


def dw_mfmas(P, L, outs, n_list, acc_of):
    """weight-gradient MFMAs: two out tiles against one in tile at a time (n-major, the accumulators of the two out tiles in rotation),
    or ONE out tile against two in tiles (rotation over the in tiles)"""
    if len(outs) == 2:
        for n in n_list:
            for s in range(2):
                for k in range(6):
                    for mt in outs:
                        a = acc_of(mt, n)
                        P.add(f"{a} = {MFMA}(A_{L}[{mt}][{PLN[A_PL[k]]}][{s}].h, Bs_{L}[{n}][{PLN[B_PL[k]]}][{s}].h, {a}, 0, 0, 0);", mfma=True,
                              reads=[f"{L}.A.{mt}.{A_PL[k]}.{s}.0", f"{L}.A.{mt}.{A_PL[k]}.{s}.1", f"{L}.B.{n}.{s}"], writes=[f"dw.{a}"])
    else:
        mt = outs[0]
        for s in range(2):
            for k in range(6):
                for n in n_list:
                    a = acc_of(mt, n)
                    P.add(f"{a} = {MFMA}(A_{L}[{mt}][{PLN[A_PL[k]]}][{s}].h, Bs_{L}[{n}][{PLN[B_PL[k]]}][{s}].h, {a}, 0, 0, 0);", mfma=True,
                          reads=[f"{L}.A.{mt}.{A_PL[k]}.{s}.0", f"{L}.A.{mt}.{A_PL[k]}.{s}.1", f"{L}.B.{n}.{s}"], writes=[f"dw.{a}"])


def mask_ops(P, mw, acc, L, vout, **kw):
    for j in range(32):
        P.add(f"{vout}[{j}] = mask_bit({mw}, {j}, {acc}[{j >> 4}][{j & 15}]);", cost=3, reads=[f"{L}.acc.{j >> 4}", "mw"], writes=[f"{vout}.{j}"], **kw)


def hidden_layer(P, L, vin, wt, MT, acc, NI, slots, reloads, acc_of, bias, nxt):
    """a full layer: 4 k16 steps of backward-data into `acc`, then dW against NI in tiles.  slots[n] = register slot holding in tile n,
    reloads[n] = (tile expr, row) the slot takes afterwards.  `nxt(P)` adds the statements that only depend on the backward-data result
    (masking into the next layer's column, outputs) -- in program order they sit between this layer's backward-data and its dW, so that
    they may ride under the dW MFMAs."""
    for e in range(4, 16):
        split_pair(P, L, vin, e)
    for s in range(1, 4):
        wload(P, L, wt, 4, MT, s)
    for s in range(4):
        bd_mfmas(P, L, MT, s, s == 0, acc)
    p0 = put(P, L, vin, 0)
    g0 = get(P, L, 0)
    b0 = bsum(P, L, 0, bias[0])
    put(P, L, vin, 1, after=g0 + b0)
    get(P, L, 1)
    bsum(P, L, 1, bias[1])
    nxt(P)
    for n in range(NI):
        slice_rows(P, L, n, slots[n])
        reload(P, slots[n], *reloads[n])
    dw_mfmas(P, L, [0, 1], list(range(NI)), acc_of)
    release(P, L, [0, 1])


# ------------------------------------------------------------------------------------------------ colour launch
COLOR_HEAD = r'''
// ---- color_net: Q2 (3 rows) <- g_albedo, Q1, Q0; hands d(geo) to the sdf launch (which also takes the geo rows of dW_s2) -------------
__global__ __launch_bounds__(FUSED_THREADS, 1) void field_fused_color_b3_kernel(
    const float *__restrict__ albedo, const float *__restrict__ g_albedo, const float *__restrict__ wpackT,
    const float *__restrict__ acts, float *__restrict__ dgeo_scr, float *__restrict__ g_feat_c, float *__restrict__ ws,
    FusedPart part, uint32_t *__restrict__ gmax, int64_t M, int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave: provably uniform -> scalar tile pointers)
    const int pt = lane & 31, h = lane >> 5, i = lane & 31;
    constexpr int W_F4 = FUSED_TC2(1) + FUSED_TC1(1) + FUSED_TC0(1);
    stage_fused<W_F4>(wpackT, 0);     // TC2 | TC1 | TC0 slices
    const FbLds L = fb_lds(fb_lds_u32(lds_fused + W_F4) + wave * FB_WAVE_BYTES, lane);
    __syncthreads();
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 w2[2], w1[2][2], w0[2][2];                      // dW of c2 [1 out tile][2 in], c1, c0 [2][2]: 160 registers
    f32x16 wg[2];                                          // + the geo rows of the SDF net's last layer, dW_s2[geo][:] = d(geo) H2^T: d(geo) is made HERE
    float bg = 0.f;
    acc_zero<2>(wg);
    acc_zero<2>(w2);
    acc_zero<2>(w1[0]);
    acc_zero<2>(w1[1]);
    acc_zero<2>(w0[0]);
    acc_zero<2>(w0[1]);
    float b2 = 0.f, b1[2] = {0.f, 0.f}, b0[2] = {0.f, 0.f};
    uint32_t max_c = 0;
    const int chunk = blockIdx.x * (FUSED_THREADS / 64) + wave, n_chunks = gridDim.x * (FUSED_THREADS / 64);
    const float use_ga = g_albedo ? 1.f : 0.f;              // an absent gradient reads a valid array and counts for nothing (no branch)
    const float *ga_src = g_albedo ? g_albedo : albedo;
    // parked input rows one layer ahead in register slots that live across the tile loop: slots 0, 1 = C2 -> C0 (= [hash_c | geo]) ->
    // the next tile's C2; slots 2, 3 = C1 -> S2 (the geo rows' weight gradient at the end of the tile) -> the next tile's C1
    RowFrag raw[4];
    float n_alb[3], n_ga[3];
    uint32_t n_mw3 = 0, n_mw2 = 0;
    uint32_t pf_p;
    const uint32_t *pf_mk;
    const uint32_t v_rowoff = (uint32_t)(i * TILE + 16 * h) * 4u, v_lane4 = (uint32_t)lane * 4u, v_lane64 = (uint32_t)lane * 64u;      // loop-invariant lane offsets (bytes)
    const uint32_t wl2 = fb_lds_u32(lds_fused) + lane * 16, wl1 = wl2 + FUSED_TC2(1) * 16, wl0 = wl1 + FUSED_TC1(1) * 16;   // per-lane LDS byte addresses of the layers' fragments
    Frag bb_c2[4][3], bb_c1[4][3], bb_c0[4][3], bb_g[4][3]; // column slices [k16 step][plane] (the B operand of the backward-data product)
    Frag W_c2[1][3][2], W_c1[4][3][2], W_c0[4][3][2];      // transposed weight fragments [k16 step][plane][out tile]
    Frag A_c2[1][3][2], A_c1[2][3][2], A_c0[2][3][2], A_g[1][3][2];      // row-form dPre fragments [out tile][plane][k16 step] (transposing reads)
    u32x2_t tA_c2[1][3][2][2], tA_c1[2][3][2][2], tA_c0[2][3][2][2], tA_g[1][3][2][2];      // ... as the reads deliver them: 64-bit halves
    Frag Bs_c2[2][3][2], Bs_c1[2][3][2], Bs_c0[2][3][2], Bs_g[2][3][2];   // sliced parked input rows [in tile][plane][k16 step]
    f32x4 bs_c2[4], bs_c1[4], bs_c0[4], bs_g[4];
    float rr_c2[32], rr_c1[32], rr_c0[32], rr_g[32], rb_c2[2][16], rb_c1[2][16], rb_c0[2][16], rb_g[2][16];      // residuals between the two halves of a split
    f32x16 acc2[2], acc1[2], acc0[2];
    float v_c2[32], v_c1[32], v_c0[32], v_g[32], eh[16];
    if (chunk < n_tiles) {                  // (a wave without tiles still joins the workgroup's reduction)
    {
        const float *at0 = acts + (int64_t)chunk * (int64_t)(FIELD_ACT_ROWS * TILE);
        fb_row_load_asm(raw[0], at0 + 352 * TILE, v_rowoff);
        fb_row_load_asm(raw[1], at0 + 384 * TILE, v_rowoff);
        fb_row_load_asm(raw[2], at0 + 288 * TILE, v_rowoff);
        fb_row_load_asm(raw[3], at0 + 320 * TILE, v_rowoff);
    }
'''

COLOR_TAIL = r'''
    }
    if (gmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) max_c = max(max_c, (uint32_t)__shfl_xor((int)max_c, o));
        if (lane == 0 && max_c) atomicMax(gmax + 1, max_c);
    }
    bg += __shfl_xor(bg, 32);
    b2 += __shfl_xor(b2, 32);
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        b1[mt] += __shfl_xor(b1[mt], 32);
        b0[mt] += __shfl_xor(b0[mt], 32);
    }
    // the workgroup's partial = the sum of its four waves' (acc_to_lds): 12 accumulator tiles + 6 bias rows = ~50 KB of LDS
    float *red = reinterpret_cast<float *>(lds_fused);
    for (int src = 1; src < FUSED_THREADS / 64; src++) {
        __syncthreads();
        if (wave == src) {
            float *q = acc_to_lds<2>(w2, red, lane);
            q = acc_to_lds<2>(w1[0], q, lane);
            q = acc_to_lds<2>(w1[1], q, lane);
            q = acc_to_lds<2>(w0[0], q, lane);
            q = acc_to_lds<2>(w0[1], q, lane);
            q = acc_to_lds<2>(wg, q, lane);
            q[5 * 64 + lane] = bg;
            q[0 * 64 + lane] = b2;
            q[1 * 64 + lane] = b1[0];
            q[2 * 64 + lane] = b1[1];
            q[3 * 64 + lane] = b0[0];
            q[4 * 64 + lane] = b0[1];
        }
        __syncthreads();
        if (wave == 0) {
            const float *q = acc_add_lds<2>(w2, red, lane);
            q = acc_add_lds<2>(w1[0], q, lane);
            q = acc_add_lds<2>(w1[1], q, lane);
            q = acc_add_lds<2>(w0[0], q, lane);
            q = acc_add_lds<2>(w0[1], q, lane);
            q = acc_add_lds<2>(wg, q, lane);
            bg += q[5 * 64 + lane];
            b2 += q[0 * 64 + lane];
            b1[0] += q[1 * 64 + lane];
            b1[1] += q[2 * 64 + lane];
            b0[0] += q[3 * 64 + lane];
            b0[1] += q[4 * 64 + lane];
        }
    }
    if (wave != 0) return;
    // partial sums of this workgroup: layers in the launch's order c0, c1, c2 = part.dw[0..2]
    const int64_t pchunk = blockIdx.x;
    dw_store<2>(ws + part.dw[2] + pchunk * 32 * 64, w2, 0, 64, i, h);
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        dw_store<2>(ws + part.dw[1] + pchunk * 64 * 64, w1[mt], mt, 64, i, h);
        dw_store<2>(ws + part.dw[0] + pchunk * 64 * 64, w0[mt], mt, 64, i, h);
    }
    dw_store<2>(ws + part.dw_s2 + pchunk * 64 * 64, wg, 0, 64, i, h);      // out tile 0 (the geo rows) of the sdf net's last layer
    if (h == 0) {
        ws[part.db[2] + pchunk * 32 + i] = b2;
        ws[part.db_s2 + pchunk * 64 + i] = bg;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            ws[part.db[1] + pchunk * 64 + 32 * mt + i] = b1[mt];
            ws[part.db[0] + pchunk * 64 + 32 * mt + i] = b0[mt];
        }
    }
}
'''


def color_prefetch(P, tile, **kw):
    """the per-point inputs of tile `tile` -> n_* (inline-asm global loads; the first stage waits for them with a counted vmcnt)"""
    P.add(f"pf_p = min((uint32_t)(({tile}) * TILE) + (uint32_t)pt, (uint32_t)(M - 1)) * 12u;\n"
          f"pf_mk = reinterpret_cast<const uint32_t *>(acts + ({tile}) * (int64_t)(FIELD_ACT_ROWS * TILE) + FIELD_HID_ROWS * TILE) + 2 * 64;",
          cost=4, writes=["pf.addr"], **kw)
    for c in range(3):
        P.add(f"FB_GLOAD32(n_alb[{c}], pf_p, albedo, {4 * c});", cost=1, reads=["pf.addr"], writes=[f"n.alb.{c}"], mem="vm", dests=[f"n_alb[{c}]"], **kw)
        P.add(f"FB_GLOAD32(n_ga[{c}], pf_p, ga_src, {4 * c});", cost=1, reads=["pf.addr"], writes=[f"n.ga.{c}"], mem="vm", dests=[f"n_ga[{c}]"], **kw)
    P.add("FB_GLOAD32(n_mw2, v_lane4, pf_mk, 0);", cost=1, reads=["pf.addr"], writes=["n.mw2"], mem="vm", dests=["n_mw2"], **kw)
    P.add("FB_GLOAD32(n_mw3, v_lane4, pf_mk, 256);", cost=1, reads=["pf.addr"], writes=["n.mw3"], mem="vm", dests=["n_mw3"], **kw)


def color_first_stage(P, tile, **kw):
    """first pipeline stage of tile `tile`: dQ2 from the prefetched inputs, its slices, the first weight fragments"""
    P.add(f"{{ const bool on_ = (({tile}) * TILE + pt < M) && h == 0;\n"
          "  _Pragma(\"unroll\") for (int c = 0; c < 3; c++) v_c2[c] = on_ ? n_ga[c] * use_ga * n_alb[c] * (1.0f - n_alb[c]) : 0.f;      /* dQ2 = g_albedo a (1 - a) */\n"
          "  _Pragma(\"unroll\") for (int r = 3; r < 32; r++) v_c2[r] = 0.f; }", cost=12, reads=[f"n.alb.{c}" for c in range(3)] + [f"n.ga.{c}" for c in range(3)],
          writes=[f"v_c2.{j}" for j in range(32)], **kw)
    split_pair(P, "c2", "v_c2", 0, **kw)
    split_pair(P, "c2", "v_c2", 1, **kw)
    for e in range(2, 8):
        zero_pair(P, "c2", e, **kw)
    wload(P, "c2", "wl2", 2, 2, 0, **kw)


def gen_color(fill):
    P = Program()
    # tokens alive at loop entry: made by the previous iteration's rotated stage / the pre-loop prologue
    P.initial |= {f"c2.p.{e}" for e in range(8)} | {f"c2.W.0.{pl}.{t}" for pl in "hml" for t in range(2)} | {f"v_c2.{j}" for j in range(32)}
    P.initial |= {f"raw.{k}" for k in range(4)} | {"buf.free", "n.mw2", "n.mw3"}
    P.add("mw3 = n_mw3; mw2 = n_mw2;", cost=2, reads=["n.mw2", "n.mw3"], writes=["mw"], asap=True)      # (the prefetch below overwrites n_mw*)
    color_prefetch(P, "tile_n", deadline=8)
    # ---- layer c2: dQ2 = three live rows (k16 step 0; step 1 is zeros): dC2 = W2^T dQ2 -> masked dQ1; dW2 += dQ2 C2^T
    bd_mfmas(P, "c2", 2, 0, True, "acc2")
    put(P, "c2", "v_c2", 0)
    get(P, "c2", 0)
    bsum(P, "c2", 0, "b2")
    mask_ops(P, "mw3", "acc2", "c2", "v_c1")
    split_pair(P, "c1", "v_c1", 0)
    split_pair(P, "c1", "v_c1", 1)
    split_pair(P, "c1", "v_c1", 2)
    split_pair(P, "c1", "v_c1", 3)
    wload(P, "c1", "wl1", 4, 2, 0)
    slice_rows(P, "c2", 0, 0)
    reload(P, 0, "atile", 224)           # slot 0 <- C0 input tile 0 (hash_c)
    slice_rows(P, "c2", 1, 1)
    reload(P, 1, "atile", 256)           # slot 1 <- C0 input tile 1 (geo)
    dw_mfmas(P, "c2", [0], [0, 1], lambda mt, n: f"w2[{n}]")
    release(P, "c2", [0])

    # ---- layer c1
    def after_c1(P):
        mask_ops(P, "mw2", "acc1", "c1", "v_c0")
        for e in range(4):
            split_pair(P, "c0", "v_c0", e)
        wload(P, "c0", "wl0", 4, 2, 0)
    hidden_layer(P, "c1", "v_c1", "wl1", 2, "acc1", 2, [2, 3], [("atile", 160), ("atile", 192)], lambda mt, n: f"w1[{mt}][{n}]",
                 ["b1[0]", "b1[1]"], after_c1)

    # ---- layer c0
    def after_c0(P):
        for r in range(16):
            P.add(f"eh[{r}] = acc0[0][{r}];", cost=1, reads=["c0.acc.0"], writes=[f"eh.{r}"])
        for q in range(4):
            P.add(f"FB_GSTORE128(v_lane64, (f32x4{{acc0[1][{4 * q}], acc0[1][{4 * q + 1}], acc0[1][{4 * q + 2}], acc0[1][{4 * q + 3}]}}), dgeo_scr + tile_id * (64 * 16), {16 * q});",
                  cost=5, reads=["c0.acc.1"], writes=[f"dgeo.{q}"], mem="vm")
        P.add("_Pragma(\"unroll\") for (int r = 0; r < 16; r++) max_c = max(max_c, (live && g_feat_c) ? __float_as_uint(fabsf(eh[r])) : 0u);", cost=24,
              reads=[f"eh.{r}" for r in range(16)], writes=["max_c"])
    hidden_layer(P, "c0", "v_c0", "wl0", 2, "acc0", 2, [0, 1], [("atile_n", 352), ("atile_n", 384)], lambda mt, n: f"w0[{mt}][{n}]",
                 ["b0[0]", "b0[1]"], after_c0)
    # ---- dW_s2, geo rows: A = d(geo) rows (this launch's last backward-data tile), B = the sdf net's parked S2 rows
    for r in range(32):
        P.add(f"v_g[{r}] = " + (f"acc0[1][{r}];" if r < 16 else "0.f;"), cost=1 if r < 16 else 0, reads=["c0.acc.1"] if r < 16 else [], writes=[f"v_g.{r}"])
    for e in range(8):
        split_pair(P, "g", "v_g", e)
    put(P, "g", "v_g", 0)
    get(P, "g", 0)
    bsum(P, "g", 0, "bg")
    slice_rows(P, "g", 0, 2)
    reload(P, 2, "atile_n", 288)         # slot 2 <- the next tile's C1 tile 0
    slice_rows(P, "g", 1, 3)
    reload(P, 3, "atile_n", 320)
    dw_mfmas(P, "g", [0], [0, 1], lambda mt, n: f"wg[{n}]")
    release(P, "g", [0])
    # ---- the next tile's first stage (rotated): after this tile's readers of what it overwrites
    n_c2 = 12 + 24
    color_first_stage(P, "tile_n", earliest=n_c2)
    seq = P.schedule(fill)
    # the pre-loop prologue: the first tile's inputs and first stage, in program order
    Q = Program()
    color_prefetch(Q, "(int64_t)chunk")
    Q.add('FB_WAIT("vmcnt(0)", "+v"(n_alb[0]), "+v"(n_alb[1]), "+v"(n_alb[2]), "+v"(n_ga[0]), "+v"(n_ga[1]), "+v"(n_ga[2]), "+v"(n_mw2), "+v"(n_mw3));')
    color_first_stage(Q, "(int64_t)chunk")
    # the counted waits of the loop body assume the steady state's issue sequence: everything the prologue started is drained here
    Q.add('FB_WAIT("vmcnt(0) lgkmcnt(0)", ' + ", ".join(f'"+v"(raw[{k}].v[{j}])' for k in range(2) for j in range(4)) + ");")
    Q.add('FB_PIN(' + ", ".join(f'"+v"(raw[{k}].v[{j}])' for k in range(2, 4) for j in range(4)) + ");")
    Q.add('FB_PIN(' + ", ".join(f'"+v"(W_c2[0][{pl}][{t}].f)' for pl in range(3) for t in range(2)) + ");")
    pro = "\n".join("    " + ln for o in Q.ops for ln in o.text.split("\n"))
    body = emit_text(seq)
    loop = f'''{pro}
    for (int64_t tile_id = chunk; tile_id < n_tiles; tile_id += n_chunks) {{
        const int64_t p = tile_id * TILE + pt;
        const bool live = p < M;
        const float *atile = acts + tile_id * (int64_t)(FIELD_ACT_ROWS * TILE);
        const int64_t tile_n = tile_id + n_chunks < n_tiles ? tile_id + n_chunks : tile_id;
        const float *atile_n = acts + tile_n * (int64_t)(FIELD_ACT_ROWS * TILE);
        uint32_t mw3, mw2;
{body}
        if (live && g_feat_c) {{
            f32x4 *o = reinterpret_cast<f32x4 *>(g_feat_c + p * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] = f32x4{{eh[4 * q], eh[4 * q + 1], eh[4 * q + 2], eh[4 * q + 3]}};
        }}
        __builtin_amdgcn_sched_barrier(0);
    }}'''
    return COLOR_HEAD + loop + COLOR_TAIL, P.stats



# ------------------------------------------------------------------------------------------------ sdf launch
def sdf_head(name, wc, dx):
    return r"""
// ---- sdf_net (+ Laplace density): P2 <- [d(geo) | g_sdf, g_sigma], P1, P0, d(inputs)""" + (r"""; the geo rows of dW2 (the colour launch made
//      d(geo); THIS launch has the S2 rows in registers for the sdf row anyway, and those 24 MFMAs fill the one hole of the pipeline:
//      between the last layer's backward-data and the masked dP1)""" if wc else r""" -- the sdf-only pass (finite-difference taps)""") + r"""
__global__ __launch_bounds__(FUSED_THREADS, 1) void """ + name + r"""(
    const float *__restrict__ xc, const float *__restrict__ sdf, const float *__restrict__ g_sdf,
    const float *__restrict__ g_sigma, const float *__restrict__ wpackT, const float *__restrict__ beta_p, int n_bands,
    const float *__restrict__ acts, const float *__restrict__ dgeo_scr, float *__restrict__ g_xc,
    float *__restrict__ g_feat_s, float *__restrict__ g_topo, float *__restrict__ ws,
    FusedPart part, uint32_t *__restrict__ gmax, int64_t M, int64_t n_tiles) {
    constexpr bool WITH_COLOR = """ + ("true" if wc else "false") + r""", HAS_DX = """ + ("true" if dx else "false") + r""";
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave: provably uniform -> scalar tile pointers)
    const int pt = lane & 31, h = lane >> 5, i = lane & 31;
    constexpr int W_F4 = FUSED_TS2(1) + FUSED_TS1(1) + FUSED_TS0(1);
    stage_fused<W_F4>(wpackT + 4 * (FUSED_TC2(1) + FUSED_TC1(1) + FUSED_TC0(1)), 0);     // TS2 | TS1 | TS0 slices
    const FbLds L = fb_lds(fb_lds_u32(lds_fused + W_F4) + wave * FB_WAVE_BYTES, lane);
    __syncthreads();
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 w1[2][2], w0[2][3];                             // 64 + 96 accumulator registers (the geo rows of dW2: the colour launch)
    float wsdf[2] = {0.f, 0.f}, bsdf = 0.f;
    acc_zero<2>(w1[0]);
    acc_zero<2>(w1[1]);
    acc_zero<3>(w0[0]);
    acc_zero<3>(w0[1]);
    float b1[2] = {0.f, 0.f}, b0[2] = {0.f, 0.f};
    uint32_t max_s = 0;
    float gb_acc = 0.f;                                    // d(loss)/d(beta) of this lane's points (lanes h == 0 carry it)
    const float beta = *beta_p;
    const int chunk = blockIdx.x * (FUSED_THREADS / 64) + wave, n_chunks = gridDim.x * (FUSED_THREADS / 64);
    const uint32_t wl2 = fb_lds_u32(lds_fused) + lane * 16, wl1 = wl2 + FUSED_TS2(1) * 16, wl0 = wl1 + FUSED_TS1(1) * 16;   // per-lane LDS byte addresses of the layers' fragments
    // absent gradients read a valid array and count for nothing (no branch in the tile loop)
    const float use_gs = g_sdf ? 1.f : 0.f, use_gsg = g_sigma ? 1.f : 0.f;
    const float *gs_src = g_sdf ? g_sdf : sdf, *gsg_src = g_sigma ? g_sigma : sdf;
    // parked input rows, at least one layer ahead, in three register slots that live across the tile loop:
    //   slot 0: S2 tile 0 -> S1 tile 1 -> S0 tile 2 -> the next tile's S2 tile 0
    //   slot 1: S2 tile 1 -> S0 tile 0 -> the next tile's S2 tile 1
    //   slot 2: S1 tile 0 -> S0 tile 1 -> the next tile's S1 tile 0
    RowFrag raw[3];
    float n_s = 0.f, n_gs = 0.f, n_gsg = 0.f;              // the next tile's per-point inputs
    uint32_t n_mw1 = 0, n_mw0 = 0;
    f32x4 n_dg[4];
    uint32_t pf_p;
    const uint32_t *pf_mk;
    const float *pf_dg;
    const uint32_t v_rowoff = (uint32_t)(i * TILE + 16 * h) * 4u, v_lane4 = (uint32_t)lane * 4u, v_lane64 = (uint32_t)lane * 64u;      // loop-invariant lane offsets (bytes)
    const uint32_t v_encoff = (uint32_t)((1 - h) * TILE + pt) * 4u;      // the OTHER half's row of a k-step of the parked encoding
    (void)pf_dg; (void)v_encoff; (void)v_lane64;
    Frag bb_s2[4][3], bb_s1[4][3], bb_s0[4][3];            // column slices [k16 step][plane] (the B operand of the backward-data product)
    Frag W_s2[3][3][2], W_s1[4][3][2], W_s0[4][3][3];      // transposed weight fragments [k16 step][plane][out tile]
    Frag A_s1[2][3][2], A_s0[2][3][2];      // row-form dPre fragments [out tile][plane][k16 step] (transposing reads)
    u32x2_t tA_s1[2][3][2][2], tA_s0[2][3][2][2];      // ... as the asm reads deliver them: 64-bit halves
    Frag Bs_s1[2][3][2], Bs_s0[3][3][2];   // sliced parked input rows [in tile][plane][k16 step]
    f32x4 bs_s1[4], bs_s0[4], gsv[2][4];
    float gs_cur = 0.f;
    float rr_s2[32], rr_s1[32], rr_s0[32], rb_s1[2][16], rb_s0[3][16];      // residuals between the two halves of a split
    f32x16 acc2[2], acc1[2], e[3];
    float v_s2[32], v_s1[32], v_s0[32], ef[16], enc[18], gx[3] = {0.f, 0.f, 0.f}, gtopo = 0.f;
    (void)xc; (void)n_bands; (void)enc; (void)gx;
    if (chunk < n_tiles) {                  // (a wave without tiles still joins the workgroup's reduction)
    {
        const float *at0 = acts + (int64_t)chunk * (int64_t)(FIELD_ACT_ROWS * TILE);
        fb_row_load_asm(raw[0], at0 + 160 * TILE, v_rowoff);
        fb_row_load_asm(raw[1], at0 + 192 * TILE, v_rowoff);
        fb_row_load_asm(raw[2], at0 + 96 * TILE, v_rowoff);
    }
"""


SDF_TAIL = r"""
    }
    if (gmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) max_s = max(max_s, (uint32_t)__shfl_xor((int)max_s, o));
        if (lane == 0 && max_s) atomicMax(gmax + 0, max_s);
    }
    wsdf[0] += __shfl_xor(wsdf[0], 32);
    wsdf[1] += __shfl_xor(wsdf[1], 32);
    bsdf += __shfl_xor(bsdf, 32);
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        b1[mt] += __shfl_xor(b1[mt], 32);
        b0[mt] += __shfl_xor(b0[mt], 32);
    }
    // the workgroup's partial = the sum of its four waves' (acc_to_lds): up to 12 accumulator tiles + 9 rows = 51 KB over the weights
    float *red = reinterpret_cast<float *>(lds_fused);
    for (int src = 1; src < FUSED_THREADS / 64; src++) {
        __syncthreads();
        if (wave == src) {
            float *q = red;
            q = acc_to_lds<2>(w1[0], q, lane);
            q = acc_to_lds<2>(w1[1], q, lane);
            q = acc_to_lds<3>(w0[0], q, lane);
            q = acc_to_lds<3>(w0[1], q, lane);
            q[1 * 64 + lane] = bsdf;
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                q[(2 + mt) * 64 + lane] = b1[mt];
                q[(4 + mt) * 64 + lane] = b0[mt];
                q[(7 + mt) * 64 + lane] = wsdf[mt];
            }
            q[6 * 64 + lane] = gb_acc;
        }
        __syncthreads();
        if (wave == 0) {
            const float *q = red;
            q = acc_add_lds<2>(w1[0], q, lane);
            q = acc_add_lds<2>(w1[1], q, lane);
            q = acc_add_lds<3>(w0[0], q, lane);
            q = acc_add_lds<3>(w0[1], q, lane);
            bsdf += q[1 * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                b1[mt] += q[(2 + mt) * 64 + lane];
                b0[mt] += q[(4 + mt) * 64 + lane];
                wsdf[mt] += q[(7 + mt) * 64 + lane];
            }
            gb_acc += q[6 * 64 + lane];
        }
    }
    if (wave != 0) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gb_acc += __shfl_xor(gb_acc, o);
    if (lane == 0) ws[part.gb + blockIdx.x] = gb_acc;
    // partial sums of this workgroup: layers s0, s1, s2 = part.dw[0..2]; s2: out tile 0 = the geo rows (zeros on the sdf-only pass),
    // out tile 1 = the sdf row and 31 zero rows
    const int64_t pchunk = blockIdx.x;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        dw_store<3>(ws + part.dw[0] + pchunk * 64 * 96, w0[mt], mt, 96, i, h);
        dw_store<2>(ws + part.dw[1] + pchunk * 64 * 64, w1[mt], mt, 64, i, h);
    }
    {
        f32x16 z[2];
        acc_zero<2>(z);
        if (!WITH_COLOR) dw_store<2>(ws + part.dw[2] + pchunk * 64 * 64, z, 0, 64, i, h);      // (with colour: the colour launch's)
        if (h == 0) {                 // tile 1: row 0 = the sdf row (accumulator row r = 0 of the lanes h == 0), the other 31 are zero
            z[0][0] = wsdf[0];
            z[1][0] = wsdf[1];
        }
        dw_store<2>(ws + part.dw[2] + pchunk * 64 * 64, z, 1, 64, i, h);
    }
    if (h == 0) {
        if (!WITH_COLOR) ws[part.db[2] + pchunk * 64 + i] = 0.f;
        ws[part.db[2] + pchunk * 64 + 32 + i] = i == 0 ? bsdf : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            ws[part.db[1] + pchunk * 64 + 32 * mt + i] = b1[mt];
            ws[part.db[0] + pchunk * 64 + 32 * mt + i] = b0[mt];
        }
    }
}
"""


def sdf_prefetch(P, tile, wc, **kw):
    P.add(f"pf_p = min((uint32_t)(({tile}) * TILE) + (uint32_t)pt, (uint32_t)(M - 1)) * 4u;\n"
          f"pf_mk = reinterpret_cast<const uint32_t *>(acts + ({tile}) * (int64_t)(FIELD_ACT_ROWS * TILE) + FIELD_HID_ROWS * TILE);" +
          (f"\npf_dg = dgeo_scr + ({tile}) * (64 * 16);" if wc else ""), cost=4, writes=["pf.addr"], **kw)
    P.add("FB_GLOAD32(n_s, pf_p, sdf, 0);", cost=1, reads=["pf.addr"], writes=["n.s"], mem="vm", dests=["n_s"], **kw)
    P.add("FB_GLOAD32(n_gs, pf_p, gs_src, 0);", cost=1, reads=["pf.addr"], writes=["n.gs"], mem="vm", dests=["n_gs"], **kw)
    P.add("FB_GLOAD32(n_gsg, pf_p, gsg_src, 0);", cost=1, reads=["pf.addr"], writes=["n.gsg"], mem="vm", dests=["n_gsg"], **kw)
    P.add("FB_GLOAD32(n_mw0, v_lane4, pf_mk, 0);", cost=1, reads=["pf.addr"], writes=["n.mw0"], mem="vm", dests=["n_mw0"], **kw)
    P.add("FB_GLOAD32(n_mw1, v_lane4, pf_mk, 256);", cost=1, reads=["pf.addr"], writes=["n.mw1"], mem="vm", dests=["n_mw1"], **kw)
    if wc:
        for q in range(4):
            P.add(f"FB_GLOAD128(n_dg[{q}], v_lane64, pf_dg, {16 * q});", cost=1, reads=["pf.addr"], writes=[f"n.dg.{q}"], mem="vm", dests=[f"n_dg[{q}]"], **kw)


def sdf_first_stage(P, tile, wc, **kw):
    """first pipeline stage of tile `tile`: dP2 = [d geo | d sdf] from the prefetched inputs, its first slices and weight fragments"""
    P.add(f"{{ const bool on_ = (({tile}) * TILE + pt < M) && h == 0;\n"
          "  const float s_ = n_s, gsg_ = n_gsg * use_gsg;\n"
          "  const float sg_ = (s_ > 0.f) ? 1.f : ((s_ < 0.f) ? -1.f : 0.f);\n"
          "  const float ex_ = expf(-(fabsf(s_) / beta));\n"
          "  const float gs_v = n_gs * use_gs + gsg_ * (-(0.5f / (beta * beta)) * sg_ * sg_ * ex_);\n"
          "  const float gb_v = gsg_ * (-(1.0f / (beta * beta)) * laplace_unit(s_, beta) + (1.0f / beta) * (0.5f * sg_ * ex_ * (fabsf(s_) / (beta * beta))));\n"
          "  gs_cur = on_ ? gs_v : 0.f;\n"
          "  gb_acc += on_ ? gb_v : 0.f;\n" +
          ("  _Pragma(\"unroll\") for (int r = 0; r < 16; r++) v_s2[r] = n_dg[r >> 2][r & 3];\n" if wc else
           "  _Pragma(\"unroll\") for (int r = 0; r < 16; r++) v_s2[r] = 0.f;\n") +
          "  _Pragma(\"unroll\") for (int r = 17; r < 32; r++) v_s2[r] = 0.f;\n"
          "  v_s2[16] = gs_cur;         /* tile 1, row 0 (only h == 0 lanes carry a non-zero value) */ }", cost=70,
          reads=["n.s", "n.gs", "n.gsg"] + ([f"n.dg.{q}" for q in range(4)] if wc else []), writes=[f"v_s2.{j}" for j in range(32)] + ["gs.cur"], **kw)
    P.add("FB_DS_WRITE32(L.gs_w, gs_cur, 0);", cost=1, reads=["gs.cur"], writes=["gsrow"], mem="lds", **kw)
    if wc:
        for e in range(4):
            split_pair(P, "s2", "v_s2", e, **kw)
        wload(P, "s2", "wl2", 4, 2, 0, **kw)
    else:
        split_pair(P, "s2", "v_s2", 8, **kw)
        for e in (9, 10, 11):
            zero_pair(P, "s2", e, **kw)
        wload(P, "s2", "wl2", 4, 2, 2, **kw)


def wsdf_ops(P, n, slot, with_bias):
    """the sdf row of the last layer on the VALU: dW2[sdf][in tile n] += g . S2 rows (raw, unsliced), g = LDS broadcast of the tile's d(loss)/d(sdf)"""
    for j in range(4):
        P.add(f"FB_DS_READ128(gsv[{n}][{j}], L.gs_r, {16 * j});", cost=1, reads=["gsrow"],
              writes=[f"gsv.{n}.{j}"], lat=4, asap=True, mem="lds", dests=[f"gsv[{n}][{j}]"])
    for j in range(4):
        body = "".join(f" wsdf[{n}] = fmaf(gsv[{n}][{j}][{q}], raw[{slot}].v[{j}][{q}], wsdf[{n}]);" + (f" bsdf += gsv[{n}][{j}][{q}];" if with_bias else "")
                       for q in range(4))
        P.add("{" + body + " }", cost=8 if with_bias else 4, reads=[f"gsv.{n}.{j}", f"raw.{slot}"], writes=[f"wsdf.{n}", f"raw.{slot}.used"], asap=True)


def gen_sdf(name, wc, dx, fill):
    P = Program()
    first_pairs = range(4) if wc else (8, 9, 10, 11)
    first_step = 0 if wc else 2
    P.initial |= {f"s2.p.{e}" for e in first_pairs} | {f"s2.W.{first_step}.{pl}.{t}" for pl in "hml" for t in range(2)}
    P.initial |= {f"v_s2.{j}" for j in range(32)} | {"gsrow"} | {f"raw.{k}" for k in range(3)} | {"buf.free", "n.mw0", "n.mw1"}
    P.add("mw1 = n_mw1; mw0 = n_mw0;", cost=2, reads=["n.mw0", "n.mw1"], writes=["mw"], asap=True)      # (the prefetch below overwrites n_mw*)
    sdf_prefetch(P, "tile_n", wc, deadline=8)
    if dx:
        # the parked encoding of xc (S0 rows 0..35, the OTHER half's row of each k-step): d/dx of the frequency encoding at the end of the tile
        for k in range(18):
            P.add(f"FB_GLOAD32(enc[{k}], v_encoff, atile + 16 * TILE, {(2 * k - 16) * 32 * 4});", cost=1, writes=[f"enc.{k}"], deadline=4, mem="vm", dests=[f"enc[{k}]"])
    # ================= layer s2
    if wc:
        for e in range(4, 9):
            split_pair(P, "s2", "v_s2", e)
        for e in (9, 10, 11):
            zero_pair(P, "s2", e)
        wload(P, "s2", "wl2", 4, 2, 1)
        wload(P, "s2", "wl2", 4, 2, 2)
        for s in range(3):
            bd_mfmas(P, "s2", 2, s, s == 0, "acc2")
    else:
        bd_mfmas(P, "s2", 2, 2, True, "acc2")
    wsdf_ops(P, 0, 0, True)
    wsdf_ops(P, 1, 1, False)
    mask_ops(P, "mw1", "acc2", "s2", "v_s1")
    for e in range(4):
        split_pair(P, "s1", "v_s1", e)
    wload(P, "s1", "wl1", 4, 2, 0)
    reload(P, 0, "atile", 128)                  # slot 0 <- S1 tile 1
    reload(P, 1, "atile", 0)                    # slot 1 <- S0 tile 0

    # ================= layer s1
    def after_s1(P):
        mask_ops(P, "mw0", "acc1", "s1", "v_s0")
        for e in range(4):
            split_pair(P, "s0", "v_s0", e)
        wload(P, "s0", "wl0", 4, 3, 0)
    hidden_layer(P, "s1", "v_s1", "wl1", 2, "acc1", 2, [2, 0], [("atile", 32), ("atile", 64)], lambda mt, n: f"w1[{mt}][{n}]",
                 ["b1[0]", "b1[1]"], after_s1)

    # ================= layer s0: d(inputs) = W0^T dP0 (tile 0 = enc kk 0..15, tile 1 = enc kk 16..19 + topo at r = 4, tile 2 = hash)
    def after_s0(P):
        if dx:
            P.add("gx[0] = gx[1] = gx[2] = 0.f;", cost=3, writes=["gx"])
            for k in range(18):
                de = f"e[0][{k}]" if k < 16 else f"e[1][{k - 16}]"
                f = float(1 << (k // 3))
                P.add(f"gx[{k % 3}] += {de} * (h ? -{f}f * enc[{k}] : {f}f * enc[{k}]);", cost=4, reads=[f"enc.{k}", "s0.acc.0", "s0.acc.1"], writes=["gx"])
            P.add("{ const float ex0_ = e[1][2], ex1_ = e[1][3];\n  gx[0] += h == 0 ? ex0_ : 0.f; gx[2] += h == 0 ? ex1_ : 0.f; gx[1] += h == 0 ? 0.f : ex0_; }", cost=8,
                  reads=["s0.acc.1"], writes=["gx"])
            P.add("_Pragma(\"unroll\") for (int d = 0; d < 3; d++) gx[d] += __shfl_xor(gx[d], 32);", cost=6, reads=["gx"], writes=["gx"])
        P.add("gtopo = e[1][4];", cost=1, reads=["s0.acc.1"], writes=["gtopo"])
        for r in range(16):
            P.add(f"ef[{r}] = e[2][{r}];", cost=1, reads=["s0.acc.2"], writes=[f"ef.{r}"])
        P.add("_Pragma(\"unroll\") for (int r = 0; r < 16; r++) max_s = max(max_s, live ? __float_as_uint(fabsf(ef[r])) : 0u);", cost=24,
              reads=[f"ef.{r}" for r in range(16)], writes=["max_s"])
    hidden_layer(P, "s0", "v_s0", "wl0", 3, "e", 3, [1, 2, 0], [("atile_n", 192), ("atile_n", 96), ("atile_n", 160)],
                 lambda mt, n: f"w0[{mt}][{n}]", ["b0[0]", "b0[1]"], after_s0)
    n_s2 = 36 if wc else 12
    sdf_first_stage(P, "tile_n", wc, earliest=n_s2)
    seq = P.schedule(fill)
    Q = Program()
    sdf_prefetch(Q, "(int64_t)chunk", wc)
    Q.add('FB_WAIT("vmcnt(0)", "+v"(n_s), "+v"(n_gs), "+v"(n_gsg), "+v"(n_mw0), "+v"(n_mw1)' +
          (', "+v"(n_dg[0]), "+v"(n_dg[1]), "+v"(n_dg[2]), "+v"(n_dg[3])' if wc else "") + ");")
    sdf_first_stage(Q, "(int64_t)chunk", wc)
    Q.add('FB_WAIT("vmcnt(0) lgkmcnt(0)", ' + ", ".join(f'"+v"(raw[{k}].v[{j}])' for k in range(3) for j in range(4)) + ");")
    Q.add('FB_PIN(' + ", ".join(f'"+v"(W_s2[{0 if wc else 2}][{pl}][{t}].f)' for pl in range(3) for t in range(2)) + ");")
    pro = "\n".join("    " + ln for o in Q.ops for ln in o.text.split("\n"))
    body = emit_text(seq)
    loop = f"""{pro}
    for (int64_t tile_id = chunk; tile_id < n_tiles; tile_id += n_chunks) {{
        const int64_t p = tile_id * TILE + pt;
        const bool live = p < M;
        const float *atile = acts + tile_id * (int64_t)(FIELD_ACT_ROWS * TILE);
        const int64_t tile_n = tile_id + n_chunks < n_tiles ? tile_id + n_chunks : tile_id;
        const float *atile_n = acts + tile_n * (int64_t)(FIELD_ACT_ROWS * TILE);
        uint32_t mw1, mw0;
{body}
        if (live) {{                           // the tile's outputs: the only predicated block of the loop
            if (HAS_DX && h == 0) {{
                g_xc[p * 3 + 0] = gx[0];
                g_xc[p * 3 + 1] = gx[1];
                g_xc[p * 3 + 2] = gx[2];
            }}
            if (g_topo) g_topo[p * 2 + h] = gtopo;
            f32x4 *o = reinterpret_cast<f32x4 *>(g_feat_s + p * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] = f32x4{{ef[4 * q], ef[4 * q + 1], ef[4 * q + 2], ef[4 * q + 3]}};
        }}
        __builtin_amdgcn_sched_barrier(0);
    }}"""
    return sdf_head(name, wc, dx) + loop + SDF_TAIL, P.stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fill", type=float, default=5.0)
    ap.add_argument("--barriers", action="store_true", help="pin the order with a scheduling barrier behind every bundle (A/B only)")
    a = ap.parse_args()
    global BARRIERS
    BARRIERS = a.barriers
    color, st_c = gen_color(a.fill)
    print("// GENERATED by tools/gen_field_bwd.py --fill %g -- do not edit; see that file for what the order means." % a.fill)
    print("#pragma once")
    print("// colour launch: %s" % st_c)
    print(color)
    print("colour", st_c, file=sys.stderr)
    for wc in (True, False):
        for dx in (True, False):
            name = "field_fused_sdf_b3_kernel_c%dd%d" % (wc, dx)
            txt, st = gen_sdf(name, wc, dx, a.fill)
            print("// %s: %s" % (name, st))
            print(txt)
            print(name, st, file=sys.stderr)


if __name__ == "__main__":
    main()
