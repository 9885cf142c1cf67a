"""Audit of inline-asm loads: between an asm load and the next asm statement (a wait / pin), no compiler instruction may touch its destination."""
import re, sys
src = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(name + ':'))
end = next(i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end'))
body = src[start:end]
def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
inasm = False
pending = {}      # reg -> (line no, text) of asm loads not yet followed by an asm wait/pin block
bad = 0
for ln, l in enumerate(body):
    t = l.strip()
    if t.startswith(';;#ASMSTART'): inasm = True; continue
    if t.startswith(';;#ASMEND'): inasm = False; continue
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    ops = re.findall(r'v\[\d+:\d+\]|v\d+', t)
    if inasm:
        if t.startswith(('ds_read', 'global_load')):
            d = regs(t.split()[1].rstrip(','))
            for r in d: pending[r] = (ln, t)
        elif t.startswith('s_waitcnt') or t == '':
            pass
        continue
    used = set()
    for o in ops: used |= regs(o)
    hit = used & set(pending)
    if hit:
        # a compiler instruction touches a register with a load in flight: legal only if a wait came in between -- approximated: any asm
        # s_waitcnt since the load (the generator's waits are asm); flag if none
        for r in sorted(hit):
            lno, txt = pending[r]
            waited = any(body[k].strip().startswith('s_waitcnt') for k in range(lno, ln))
            if not waited:
                bad += 1
                if bad <= 12: print("TOUCHED BEFORE ANY WAIT: v%d loaded by [%s] used by [%s]" % (r, txt[:60], t[:70]))
            pending.pop(r, None)
print("violations:", bad)
