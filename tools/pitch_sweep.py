"""Config 2 whole calls against the row pitch of the planes kernel 16 writes (MIFWT_PYRAMID_ROW_ALIGN = -k: k extra floats per row),
results dropped / three rotating output sets.  One process per pitch (the plan cache keys on the setting)."""
import os, subprocess, sys
extras = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4, 5, 7, 9, 13, 17, 21, 29, 33, 45, 61]
for k in extras:
    env = dict(os.environ, MIFWT_PYRAMID_ROW_ALIGN=str(-k) if k else "1")
    out = subprocess.run([sys.executable, "-W", "ignore", "tools/pyr_ab.py", "0"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(f"extra {k:3d}: " + (out[-1].split("strides")[1] if out else "?"), flush=True)
