import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, argparse
a = argparse.Namespace(batch=16, lr=128, precision='fp16')
print(bench.fwd_bwd_probe(a, torch.device('cuda:0'), steps=4))
