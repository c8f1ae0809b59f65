import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import edm_amd.hip as H
import golden_util as GU
os.makedirs("gpurun_out/diag", exist_ok=True)
cfg = "gpurun_out/diag/nb.edm"
open(cfg, "w").write(open(os.path.join(GU.FIXTURES, "notebook_input.edm")).read() + "\nhills_filename gpurun_out/diag/H1\n")
b = H.Bias(cfg); b.setup(1, 1); b.subdivide([0], [10], [0], [10], [0], [0])
b.pre_add_hill(1); b.add_hill([0.25], 1.0); b.post_add_hill()
b.gauss.multi_write("gpurun_out/diag/mw.grid", 0); b.gauss.multi_write("gpurun_out/diag/lt.ltab", 1)
for got, want in (("mw.grid", "file_notebook_multiwrite.grid"), ("lt.ltab", "file_notebook_lammps.ltab")):
    A = open("gpurun_out/diag/" + got).read(); Wt = open(os.path.join(GU.GOLDEN, want)).read()
    print(got, "identical" if A == Wt else "DIFFERS")
    al, wl = A.split("\n"), Wt.split("\n")
    print(len(al), len(wl))
    for i, (x, y) in enumerate(zip(al, wl)):
        if x != y: print(i, repr(x), repr(y))
