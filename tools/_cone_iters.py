import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ML_ONLY_CONE"]="1"; os.environ["ML_NO_CHECK"]="1"
sys.argv=[sys.argv[0],"4096"]
exec(open(os.path.join(os.path.dirname(__file__),"multilevel_bench.py")).read().split("def gpu_ms")[0])
ri, ts, te, pk, term = C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, 1e-3, 0.004, with_terminate_planes=True)
it = term.cpu().numpy()
print("iterations per ray: mean %.0f max %.0f min %.0f; samples/ray %.1f" % (it.mean(), it.max(), it.min(), ri.shape[0]/4096))
print("per wave (8 rays) max mean:", it.reshape(-1,8).max(1).mean())
