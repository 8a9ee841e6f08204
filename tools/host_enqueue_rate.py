import time, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from evo_amd import EvoFst
S, N, P = 1_000_000, 200, 10
sp = (np.arange(N) % P).astype(np.int32)
es = []
streams = [torch.cuda.Stream() for _ in range(3)]
for b in range(6):
    e = EvoFst(0); e.set_layout(sp, P); e.set_pairs(); e.set_windows(100, 100)
    e.set_stream(streams[b % 3].cuda_stream); e.reserve(S); e.synth(S, seed=1, missing_rate=0.0, first_site_global=b * S); e.sync(); es.append(e)
for i in range(20): es[i % 6].run()
for e in es: e.sync()
n = 600
t0 = time.perf_counter()
for i in range(n): es[i % 6].run()
t1 = time.perf_counter()
for e in es: e.sync()
t2 = time.perf_counter()
print("host enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
