import sys, os, torch
sys.path.insert(0, os.getcwd())
import quickstep_amd.capi as capi
from quickstep_amd import types as T
dev=torch.device('cuda',0)
g=torch.Generator(device=dev); g.manual_seed(1)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
for nb in (1_000_000, 10_000_000):
    keys=(torch.randperm(nb, device=dev, generator=g, dtype=torch.int64)*7+3).to(torch.int32)
    t=capi.JoinTable(T.INT, nb)
    c=timed(lambda: t.clear())
    cb=timed(lambda: (t.clear(), t.build(keys)))
    print(nb, 'clear', round(c,4), 'clear+build', round(cb,4), 'build', round(cb-c,4))
