"""rocprofv3 target: the train-step leg of bench.py alone (forward + backward of the sparse branch, one OPT-1.3B sequence)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args([])
print(json.dumps(bench.train_step_leg("opt-1.3b", args, torch.device("cuda:0"))))
