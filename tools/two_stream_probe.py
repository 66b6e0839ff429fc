#!/usr/bin/env python
"""Does splitting the batch over concurrent streams hide per-launch fixed costs? (GPU probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth

dev = torch.device('cuda:0')
sd = synth.rrdbnet_state_dict(23, 0)


def run(nsplit, B=16, steps=10):
    nets = [arch.RRDBNet(3, 3, 64, 23).to(dev).eval().set_precision('fp16') for _ in range(nsplit)]
    for n in nets:
        n.load_state_dict(sd)
    xs = [synth.image_batch(i, B // nsplit, 3, 128, 128).to(dev) for i in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    with torch.no_grad():
        for _ in range(2):
            for n, x, s in zip(nets, xs, streams):
                with torch.cuda.stream(s):
                    n(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for n, x, s in zip(nets, xs, streams):
                with torch.cuda.stream(s):
                    n(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print('streams=%d  sub-batch=%d  %.3f ms/step  %.1f HR-Mpix/s' % (nsplit, B // nsplit, dt * 1e3, B * 0.262144 / dt))


for ns in (1, 2, 4):
    run(ns)
