"""Whole calls of config 2 (wavedec2 db4 level 3 on 64 x 1024^2 f32) in ONE output regime, for a kernel trace whose AVERAGE is the
statistic: `dropped` = every result dropped at once (the driver's K timed steps: each call rewrites one output
block, partly absorbed by the 256 MiB Infinity Cache), `rotating` = the last three results kept alive (every byte to HBM).
A spin-up of untimed calls first (idle clocks), a marker kernel (torch.zeros of a recognisable size) between spin-up and the traced
loop so that the trace can be cut there.  usage: trace_regime.py dropped|rotating [calls]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
regime = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
held = [None] * 3
def call(i):
    r = ptwt_amd.wavedec2(xs[i % 3], 'db4', mode='reflect', level=3)
    if regime == 'rotating':
        held[i % 3] = r
for i in range(300):
    call(i)
torch.cuda.synchronize()
marker = torch.zeros(12345, device='cuda')  # (the cut: launches before this fill kernel are spin-up)
torch.cuda.synchronize()
for i in range(n):
    call(i)
torch.cuda.synchronize()
