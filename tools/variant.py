"""Experiment build for A/B runs: ONE source recompiled with extra flags, linked with the product build's other objects into
``libmifwt_<tag>.so`` next to the product library (selected at run time with MIFWT_LIB=libmifwt_<tag>.so, tools/ only).

    python tools/variant.py <tag> <source.hip>[,<source2.hip>] "<cflags>"
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

tag, srcs, cflags = sys.argv[1], sys.argv[2].split(","), sys.argv[3]
g.build(verbose=False)  # the product objects are up to date
objdir = os.path.join(g.PKG, "build")
vdir = os.path.join(g.PKG, "build_" + tag)
os.makedirs(vdir, exist_ok=True)
flags = [f"--offload-arch={g.ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + g.CSRC, *cflags.split()]
objs = []
for src in g._sources():
    base = os.path.basename(src)
    if base in srcs:
        obj = os.path.join(vdir, base[:-4] + ".o")
        subprocess.run([g._hipcc(), *flags, "-c", src, "-o", obj], check=True)
    else:
        obj = os.path.join(objdir, base[:-4] + ".o")
    objs.append(obj)
lib = os.path.join(g.PKG, f"libmifwt_{tag}.so")
subprocess.run([g._hipcc(), f"--offload-arch={g.ARCH}", "-shared", "-fPIC", *objs, "-o", lib], check=True)
print(lib)
