import ctypes as C, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shark_amd import load
L = load()
torch.cuda.init()
for mb in (10, 21, 160):
    n = 20
    t0 = time.perf_counter()
    ps = [L.shk_alloc_pinned(mb << 20) for _ in range(n)]
    t1 = time.perf_counter()
    # touch
    for p in ps:
        C.memset(p, 1, mb << 20)
    t2 = time.perf_counter()
    for p in ps:
        L.shk_free_pinned(p)
    t3 = time.perf_counter()
    print("%d MB x %d: alloc %.2f ms each (%.2f s/GB), first touch %.2f ms each, free %.2f ms each" % (mb, n, (t1 - t0) / n * 1e3, (t1 - t0) / (n * mb / 1024), (t2 - t1) / n * 1e3, (t3 - t2) / n * 1e3))
