import os, sys, time, importlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # first: the library must bind to the HIP runtime torch ships
import __graft_entry__ as ge
amd = ge.load_package()
# wrap the C entry points with host wall-clock timing
L = amd.lib()
calls = []
def wrap(name):
    f = getattr(L, name)
    def g(*a):
        t0 = time.perf_counter(); r = f(*a); dt = time.perf_counter() - t0
        calls.append((name, dt)); return r
    setattr(L, name, g)
for n in ("nvbio_full_gotoh_score", "nvbio_full_gotoh_traceback", "nvbio_banded_gotoh_score", "nvbio_banded_gotoh_traceback", "nvbio_opposite_mate_windows", "nvbio_fm_match_seed_diagonals"):
    wrap(n)
sys.argv = ["bench_pe.py"]
import runpy
runpy.run_path(os.path.join(ROOT, "scripts", "bench_pe.py"), run_name="__main__")
for n, dt in calls:
    if dt > 0.02: print("SLOW", n, "%.1f ms" % (dt * 1e3))
print("calls", len(calls))
