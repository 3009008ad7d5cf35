"""Static view of one kernel's ISA (hipcc --save-temps .s): basic blocks with instruction mixes and the loops (backward branches).
usage: isa_blocks.py file.s [kernel-name-substring]"""
import re, sys, collections
src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else None
# find kernel body
start = 0
if want:
    for i, l in enumerate(src):
        if re.match(r'^[A-Za-z_][\w$]*:', l) and want in l.split(':')[0]:
            start = i; break
blocks = []  # (label, line_idx, instrs)
cur = ['entry', start, []]
for i in range(start + 1, len(src)):
    l = src[i]
    if l.startswith('.Lfunc_end'): break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), i, []]; continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur[2].append(t.split(';')[0].strip())
blocks.append(cur)
pos = {b[0]: k for k, b in enumerate(blocks)}
def cls(op):
    if op.startswith('v_pk_fma') or op.startswith('v_pk_mul'): return 'pk'
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane'
    if op.startswith('v_'): return 'v'
    if op.startswith('ds_'): return 'ds'
    if op.startswith('buffer_') or op.startswith('global_') or op.startswith('flat_') or op.startswith('scratch_'): return 'mem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 'sld'
    return 's'
for k, (lab, li, ins) in enumerate(blocks):
    c = collections.Counter(cls(x.split()[0]) for x in ins)
    tgts = [x.split()[-1] for x in ins if x.startswith('s_cbranch') or x.startswith('s_branch')]
    back = [t for t in tgts if t in pos and pos[t] <= k]
    print(f"{k:4d} {lab:12s} L{li+1:<7d} n={len(ins):4d} " + ' '.join(f"{a}={c[a]}" for a in ('pk','v','lane','s','sld','ds','mem','wait','bar','nop','br') if c[a]) + (f"  -> {','.join(tgts)}" if tgts else '') + (f"   LOOP<-{','.join(back)}" if back else ''))
