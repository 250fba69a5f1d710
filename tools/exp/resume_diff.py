"""Which momentum buffers differ between an uninterrupted 4-iteration run and one resumed at iteration 2
(tests/test_pipeline_gpu.py's scenario), per parameter.  python tools/exp/resume_diff.py [out_dir]"""
import os, shutil, subprocess, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
opts = ["MODEL.WSOD_ON", "True", "MODEL.FASTER_RCNN", "False", "MODEL.BACKBONE.CONV_BODY", "VGG16-OICR",
        "MODEL.ROI_BOX_HEAD.NUM_CLASSES", "21", "MODEL.ROI_BOX_HEAD.POOLER_METHOD", "ROIPool",
        "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", "7", "MODEL.ROI_BOX_HEAD.POOLER_SCALES", "(0.125,)",
        "MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR", "VGG16.roi_head", "MODEL.ROI_WEAK_HEAD.PREDICTOR", "MISTPredictor",
        "MODEL.ROI_WEAK_HEAD.LOSS", "RoIRegLoss", "MODEL.ROI_WEAK_HEAD.REGRESS_ON", "True", "DB.METHOD", "dropblock",
        "SOLVER.CONTRA", "True", "SOLVER.BASE_LR", "1e-5", "SOLVER.CHECKPOINT_PERIOD", "2", "SEED", "7", "MODEL.WEIGHT", ""]


def run(out_dir, max_iter, env):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_net.py"), "--synthetic", "--size", "160", "--proposals", "60",
           "--log-period", "1"] + opts + ["SOLVER.MAX_ITER", str(max_iter), "OUTPUT_DIR", out_dir]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def scenario(tag, extra_env):
    env = dict(os.environ, ODW_NO_TIMER="1", **extra_env)
    tmp = tempfile.mkdtemp()
    a, b, c = os.path.join(tmp, "a"), os.path.join(tmp, "b"), os.path.join(tmp, "c")
    oa = run(a, 4, env)
    oc = run(c, 4, env)                 # a second uninterrupted run: run-to-run noise
    os.makedirs(b)
    shutil.copy(os.path.join(a, "model_0000002.pth"), os.path.join(b, "model_0000002.pth"))
    with open(os.path.join(b, "last_checkpoint"), "w") as f:
        f.write(os.path.join(b, "model_0000002.pth"))
    ob = run(b, 4, env)
    print("==== %s" % tag)
    for name, o in (("straight", oa), ("resumed", ob)):
        print(name, [l for l in o.splitlines() if "iter:" in l][-2:])
    A = torch.load(os.path.join(a, "model_final.pth"), weights_only=False)
    B = torch.load(os.path.join(b, "model_final.pth"), weights_only=False)
    C = torch.load(os.path.join(c, "model_final.pth"), weights_only=False)
    names = list(A["model"].keys())
    for i, sa in A["optimizer"]["state"].items():
        ma = sa["momentum_buffer"].double()
        mb = B["optimizer"]["state"][i]["momentum_buffer"].double()
        mc = C["optimizer"]["state"][i]["momentum_buffer"].double()
        if ma.abs().max() > 0:
            d, d2 = float((ma - mb).norm() / ma.norm()), float((ma - mc).norm() / ma.norm())
            if d > 0.01 or d2 > 0.01:
                print("  state %s %s: resumed vs straight %.4f   straight vs straight %.4f  |m| %.3e" % (i, tuple(ma.shape), d, d2, float(ma.norm())))
    shutil.rmtree(tmp)


scenario("device lists", {})
scenario("host lists", {"ODW_HOST_LISTS": "1"})
