import sys, torch
sys.path.insert(0, "/root/repo")
import quickstep_amd.capi as capi
from quickstep_amd import types as T
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
out = (torch.empty(100_000_000, dtype=torch.int32, device=dev), torch.empty(100_000_000, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
for nb in (1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000):
    b = torch.randperm(nb, device=dev, generator=g, dtype=torch.int32)
    p = torch.randint(0, nb, (100_000_000,), device=dev, generator=g, dtype=torch.int32)
    t = capi.JoinTable(T.INT, nb, key_range=(0, nb - 1)); t.build(b)
    h = capi.JoinTable(T.INT, nb); h.build(b)
    print(nb, "dense probe %.3f ms" % timed(lambda: t.probe(p, capacity=100_000_000, out=out)), "hashed probe %.3f ms" % timed(lambda: h.probe(p, capacity=100_000_000, out=out)), flush=True)
    t.close(); h.close()

# Range-partitioned probe: scatter the probe keys (with their row numbers) by the 4 MiB slice of the head array they hit,
# then one ordinary probe over the scattered keys — the tiles in flight stay inside one slice, which L2 holds.
nb = 8_000_000
b = torch.randperm(nb, device=dev, generator=g, dtype=torch.int32)
p = torch.randint(0, nb, (100_000_000,), device=dev, generator=g, dtype=torch.int32)
t = capi.JoinTable(T.INT, nb, key_range=(0, nb - 1)); t.build(b)
tids = torch.arange(p.numel(), device=dev, dtype=torch.int32)
def partitioned():
    (pk, pt), off = capi.partition_scatter(p >> 20, 8, [p, tids])
    op, ob, cnt = t.probe(pk, capacity=100_000_000, out=out)
    return capi.gather(pt, op)
print("8 M keys: direct probe %.3f ms, range-partitioned (shift + K9 + probe + tid gather) %.3f ms" %
      (timed(lambda: t.probe(p, capacity=100_000_000, out=out)), timed(partitioned)))
(pk, pt), off = capi.partition_scatter(p >> 20, 8, [p, tids])
print("  the probe alone over range-partitioned keys: %.3f ms" % timed(lambda: t.probe(pk, capacity=100_000_000, out=out)))
