"""Pairwise stream probe (gsum_debug_pipe_probe, lab build): overlap of two 100-us kernels on every pair of a context's four streams
and of streams created after them; first context of the process, then a second and a third one.  -> profiles/r05_pipe_probe.log"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsum_amd  # noqa: E402
from gsum_amd import _lib  # noqa: E402

np.set_printoptions(precision=2, suppress=True, linewidth=200)
names = ["main(lo)", "chain(hi)", "aux(hi)", "grp3(hi)", "extra1", "extra2", "extra3", "extra4"]
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.zeros(1, device="cuda").sum().item()
    print("(torch initialised its stream first)")
a = _lib.HipContext(0, lab=True)
print("first context of the process: pipes_ok =", a.get_option("pipes_ok"), "min overlap", a.get_option("pipe_overlap_permille") / 1000)
print("streams:", names)
print(a.pipe_probe(extra=4))
b = _lib.HipContext(0, lab=True)
print("second context: pipes_ok =", b.get_option("pipes_ok"), "min overlap", b.get_option("pipe_overlap_permille") / 1000, "streams replaced:", b.get_option("pipe_heals"))
print(b.pipe_probe(extra=0))
c = _lib.HipContext(0, lab=True)
print("third context: pipes_ok =", c.get_option("pipes_ok"), "min overlap", c.get_option("pipe_overlap_permille") / 1000, "streams replaced:", c.get_option("pipe_heals"))
print(c.pipe_probe(extra=0))
print("first context again:")
print(a.pipe_probe(extra=0))
more = [_lib.HipContext(0, lab=True) for _ in range(5)]
print("contexts 4-8: pipes_ok", [m.get_option("pipes_ok") for m in more], "streams replaced", [m.get_option("pipe_heals") for m in more])
for x in more + [c, b, a]:
    x.close()
