import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from ces_amd.engine import Engine
p, J = 256, 65536
pin = torch.empty((p, J), dtype=torch.float32).pin_memory(); pin.normal_()
out = Engine._HostOutPool._fresh((p, J))
def T(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return "median %.2f ms min %.2f" % (float(np.median(ts)), min(ts))
for th in (1, 4, 16, 32, 64):
    torch.set_num_threads(th)
    print("torch copy_ f32->f64 threads", th, T(lambda: torch.from_numpy(out).copy_(pin)))
print("np.copyto single", T(lambda: np.copyto(out, pin.numpy())))
for th in (4, 8, 16, 32):
    ex = ThreadPoolExecutor(th)
    src = pin.numpy()
    rows = [(k * p // th, (k + 1) * p // th) for k in range(th)]
    def par():
        list(ex.map(lambda r: np.copyto(out[r[0]:r[1]], src[r[0]:r[1]]), rows))
    print("np.copyto %d python threads" % th, T(par))
src64 = np.random.default_rng(0).standard_normal((p, J))
pin32 = torch.empty((p, J), dtype=torch.float32).pin_memory()
for th in (8, 16):
    ex = ThreadPoolExecutor(th)
    rows = [(k * p // th, (k + 1) * p // th) for k in range(th)]
    dst = pin32.numpy()
    def par2():
        list(ex.map(lambda r: np.copyto(dst[r[0]:r[1]], src64[r[0]:r[1]], casting="same_kind"), rows))
    print("f64->pinned f32 np %d threads" % th, T(par2))
torch.set_num_threads(16)
print("f64->pinned f32 torch 16", T(lambda: pin32.copy_(torch.from_numpy(src64))))
