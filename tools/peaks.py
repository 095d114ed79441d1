"""Print the measured stream-copy / read bandwidth and dense MFMA rates of this GPU (gp_microbench_* kernels)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprediction_amd import peaks  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(peaks.measure("cuda:0", gib=float(sys.argv[1]) if len(sys.argv) > 1 else 2.0, reps=10, mfma_iters=8192)))
