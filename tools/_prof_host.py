import sys, os, cProfile, pstats
sys.path.insert(0, '/root/repo')
sys.argv = ['x', '3']
import runpy
ns = runpy.run_path('/root/repo/tools/train_phases.py')
step = ns['step']
import torch
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step([])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
