import torch, sys
sys.path.insert(0,".")
from gym_amd.atari import AtariPreprocessor
n=4096
pp=AtariPreprocessor(n)
fr=torch.randint(0,256,(n,2,210,160,3),dtype=torch.uint8,device="cuda")
pp.reset(fr)
for _ in range(5): pp.step(fr)
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): pp.step(fr)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/50
print("atari ms", ms, "GB/s", 208656*n/ms/1e6)
