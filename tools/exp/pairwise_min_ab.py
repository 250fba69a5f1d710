"""odw_pairwise_sim_ws: the one-launch panel kernel against the split + LDS-DMA pair, both forced, per P (round 6)."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import sys, json, torch
sys.path.insert(0, %r)
import bench
print(json.dumps(bench.pairwise_sim_live(torch.device("cuda", 0), sizes=(2000, 4000, 5000, 5600, 6000, 7000, 8000), iters=30)))
''' % root
res = {}
for tag, v in (("panel", str(1 << 30)), ("planes+dma", "0")):
    env = dict(os.environ, ODW_PAIRWISE_PLANES_MIN=v)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    res[tag] = json.loads(out)
print("%8s %12s %12s" % ("P", "panel us", "planes+dma us"))
for k in res["panel"]:
    print("%8s %12.2f %12.2f" % (k.split("=")[1], res["panel"][k]["avg_launch_us"], res["planes+dma"][k]["avg_launch_us"]))
