#!/usr/bin/env python3
"""Is the layer's selection reproducible?  Entry count of the CSR over repeated forwards of one layer and over freshly built
layers (same seeds), at the one-sequence shapes (segmented Performer, streamed-weight MLP, 5-tile convolutions)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as B
for wname, nb in (("llama-13b", 1), ("opt-2.7b", 1), ("opt-1.3b", 8)):
    zs, sums = [], []
    for rep in range(3):
        lb = B.LayerBench(wname, nb, "bf16", torch.device("cuda", 0), ctx_dtype_name="fp32")
        for it in range(3):
            out = lb.forward()
            torch.cuda.synchronize()
            zs.append(int(out.partial_attention_mask.crow[:, -1].sum().item()))
            pv = out.estimated_attention_probs_m.float()
            sums.append(float(pv.double().sum().item()))
        lb.release()
    print(json.dumps({"workload": wname, "batch": nb, "nnz": zs, "probs_sum": [round(s, 6) for s in sums]}), flush=True)
