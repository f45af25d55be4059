"""The content / shape spread of the ADAPTIVE repair pass (bench.py: repair_spread) on its own, for A/B builds and switches:
MCA_HIP_LIB=abtest/lib_x.so MCA_HIP_ADAPT_TAU_SCALE=0.6 python tools/repair_spread.py"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="adaptive")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream
out = bench.repair_spread(args, dev, st, 0, 0.0)
for o in out:
    print("%-82s %6.2f M frames/s  %.3f ms per 32768 frames  repair %.3f ms  flagged %.4f  recomputed %.4f" %
          (o["input"], o["value"] / 1e6, o["ms_per_32768_frames"], o["repair_ms"], o["flagged_fraction"], o["recomputed_fraction"]))
print(json.dumps({"lib": os.environ.get("MCA_HIP_LIB", "default"), "tau_scale": os.environ.get("MCA_HIP_ADAPT_TAU_SCALE", "1"), "spread": out}))
