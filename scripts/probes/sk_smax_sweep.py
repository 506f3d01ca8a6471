"""single-video VASNet score / train-step (graph) time under the slice caps of the in-launch split-K launches (diagnostic build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests/golden")
import torch
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda:0")
r = bench.single_video_leg(dev)["vasnet"]
print(os.environ.get("SUMK_SK_SMAX"), "score", r["score_eager"]["us_per_video"], "train graph", r["train_step_graph"]["us_per_video"])
