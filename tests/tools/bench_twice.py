import os, sys, gc
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from common import BASELINE_160MS as g, zoo
from obs_rvc_amd import weights as W
job = bench.Job(dry=False)
z = zoo("full")
vecs = W.make_index()
seq = sys.argv[1].split(",")
for item in seq:
    idx = item.startswith("i")
    roof = "r" in item
    rec, eng, rings, d = bench.run_config(job, z, g, 1, idx, 60, 10, False, vecs, want_roofline=roof)
    print(item, rec["ms_per_step"], rec["latency_ms"], flush=True)
    if "k" not in item:
        del eng, d
        gc.collect(); job.torch.cuda.empty_cache()
