import cProfile, pstats, io, os, sys, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.environ["R4D_PHASE_TIMING"] = "0"
import numpy as np, torch
import annotation_e2e as e2e
from rag4dyg_amd import annotation
root = tempfile.mkdtemp(prefix="r4d_ann_")
e2e.hepth_workdir(root); os.chdir(root)
np.random.seed(0)
from contextlib import redirect_stdout
with redirect_stdout(io.StringIO()):
    annotation.main(["x", "hepth", "11", "0.8"])
pr = cProfile.Profile(); pr.enable()
with redirect_stdout(io.StringIO()):
    annotation.main(["x", "hepth", "11", "0.8"])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
