"""bench.py's fwd_bwd probe on its own (batch 16 of 128x128 LR, fp16 train-mode forward + backward): the command
the rocprofv3 kernel traces / PMC passes of the backward were taken with.  Usage (GPU box): python tools/fwd_bwd_probe.py [timed steps, default 4]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, argparse
a = argparse.Namespace(batch=16, lr=128, precision='fp16')
print(bench.fwd_bwd_probe(a, torch.device('cuda:0'), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 4))
