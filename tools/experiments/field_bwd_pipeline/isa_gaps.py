"""Static issue model of a kernel's main loop: cycles = sum over MFMA gaps of max(33.3 + 0.3 n, 5 n + 7.8), n = non-MFMA vector/LDS/VMEM
instructions between consecutive MFMAs (tools/micro/mfma_valu_gap: 5-6 fillers free per gap, +5 cycles each beyond)."""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(name + ":"))
end = next(i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end'))
body = src[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
loops.sort(reverse=True)
which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
_, lo, hi = loops[which]
gaps, n, seen = [], 0, False
kinds = collections.Counter()
pre = 0
for l in body[lo:hi]:
    l = l.strip()
    if not l or l.startswith('.') or l.startswith(';') or l.endswith(':'): continue
    ins = l.split()[0]
    if ins.startswith('v_mfma'):
        if seen: gaps.append(n)
        else: pre = n
        seen, n = True, 0
    elif ins.startswith(('v_', 'ds_', 'global_', 'buffer_', 'scratch_', 'flat_')):
        n += 1
        kinds[ins.split('_')[0] if not ins.startswith('v_accvgpr') else 'accvgpr'] += 1
post = n
cyc = sum(max(33.3 + 0.3 * g, 5 * g + 7.8) for g in gaps) + 5 * (pre + post) + 33
hist = collections.Counter(min(g, 60) // 4 * 4 for g in gaps)
print(f"mfma {len(gaps)+1}  other {sum(gaps)+pre+post}  kinds {dict(kinds)}")
print(f"model cycles {cyc:.0f}   (mfma-only {33.3*(len(gaps)+1):.0f}, serial {33.3*(len(gaps)+1)+5*(sum(gaps)+pre+post):.0f})")
print("gap histogram (bucket of 4):", sorted(hist.items()))
