"""s_nop / v_pk count per kernel of a --save-temps .s (compiler-inserted wait states between inline-asm FMAs).  usage: isa_nops.py file.s [filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^; codeLenInByte = (\d+)', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name: continue
    lines = [l.split(';')[0].strip() for l in body.splitlines()]
    n = sum(1 for l in lines if l and not l.startswith('.') and not l.startswith(';'))
    print(f"{name[:80]:80s} instr {n:6d} pk {sum(l.startswith('v_pk_') for l in lines):5d} fma64 {sum(l.startswith('v_fma_f64') for l in lines):5d} nop {sum(l.startswith('s_nop') for l in lines):5d}")
