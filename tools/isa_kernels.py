"""Per-kernel resource summary of a hipcc --save-temps .s file: code bytes, VGPRs, scratch, scalar spills (v_readlane / v_writelane),
readfirstlane ("waterfall") loops.  usage: isa_kernels.py file.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^; codeLenInByte = (\d+).*?^; NumVgprs: (\d+).*?^; ScratchSize: (\d+).*?^; Occupancy: (\d+)', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name: continue
    wf = len(re.findall(r'v_readfirstlane_b32 (s\d+), (v\d+)\n(?:\s*s_nop \d+\n)?\s*v_cmp_eq_u32\w* vcc, \1, \2', body))
    print(f"{name[:90]:90s} code {m.group(3):>6s} vgpr {m.group(4):>3s} scratch {m.group(5):>4s} occ {m.group(6)} readlane {body.count('v_readlane'):4d} writelane {body.count('v_writelane'):4d} waterfall {wf}")
