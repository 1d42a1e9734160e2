import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from ces_amd import engine
sys.path.insert(0, 'tests')
from test_gpu_fullsize import torch_factored_aldi
P = N = 512
for J, dtype, tol in ((8192, "float64", 1e-6), (8192, "float32", 1e-3)):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((N, P)) / np.sqrt(P)
    ustar = rng.standard_normal((P, 1))
    prob = dict(A=A, ustar=ustar, Gamma=0.01 * np.eye(N), y=(A @ ustar).ravel() + 0.1 * rng.standard_normal(N), mu=np.zeros((P, 1)), sigma=100.0 * np.eye(P))
    g = torch.Generator(device="cuda").manual_seed(7)
    U = torch.as_tensor(ustar, device="cuda") + torch.randn((P, J), generator=g, device="cuda", dtype=torch.float64)
    G = torch.as_tensor(A, device="cuda") @ U
    xi = torch.randn((P, J), generator=g, device="cuda", dtype=torch.float64)
    eng = engine.Engine(P, N, J, dtype=dtype)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    Ud, Gd, xid = eng.to_device(U), eng.to_device(G), eng.to_device(xi)
    out = eng.step(engine.step_params(update="aldi"), Ud, Gd, xi=xid)
    res = eng.result()
    ref, hk, met = torch_factored_aldi(prob, Ud.double(), Gd.double(), xid.double())
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    print(dtype, "J", J, "rel err", err, "hk", res.hk, hk, "OK" if err < tol else "FAIL")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5):
        eng.step(engine.step_params(update="aldi", step_index=i, first_step=False, t_len=1, t_last=0.1), Ud, Gd, xi=None, out=out, recenter=False); eng.result()
    torch.cuda.synchronize()
    print("  ms/step", (time.perf_counter() - t0) / 5 * 1e3)
