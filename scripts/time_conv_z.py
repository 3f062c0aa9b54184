#!/usr/bin/env python3
"""Round 5 A/B: the layer's per-region times with the 1x1 convolution in conv2's epilogue (`conv_z_epilogue`) and without it.
usage (GPU box): python scripts/time_conv_z.py [workload batch] ..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as B
import sea_attention_amd as S

def run(wname, nb, z):
    lb = B.LayerBench(wname, nb, "bf16", torch.device("cuda", 0), ctx_dtype_name="fp32", layer_attrs=dict(conv_z_epilogue=z))
    for _ in range(3): lb.forward()
    lb.capture("gather")
    lb.timed(5)
    best = min(lb.timed(20)[0] for _ in range(4)) / 20 * 1e3
    bench = S.get_bench(); bench.disabled, bench.synchronize = False, True; bench.reset_measures()
    lb.layer.attention.sparse_kernel = "gather"
    for _ in range(5): lb.forward()
    torch.cuda.synchronize()
    reg = {k: round(v * 1e3, 4) for k, v in sorted(bench.todict().items()) if k.startswith(("cnn", "predictor"))}
    bench.disabled, bench.synchronize = True, False; bench.reset_measures()
    lb.release()
    return {"ms_per_step_best_of_4x20": round(best, 4), "regions_ms": reg}

if __name__ == "__main__":
    legs = [("opt-1.3b", 8), ("llama-13b", 1), ("opt-2.7b", 1), ("opt-125m", 8)]
    for w, nb in legs:
        for z in (False, True, False, True):
            print(json.dumps({"workload": w, "batch": nb, "conv_z_epilogue": z, **run(w, nb, z)}), flush=True)
