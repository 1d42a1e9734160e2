"""numpy stand-in for one ``ces_amd.engine.Engine`` shard (CPU tests only).

TEST INFRASTRUCTURE (see oracle/__init__.py).  It restates, in numpy and fp64,
what the split entry points of the C ABI compute (include/cesx.h: cesx_colsum,
cesx_set_shift, cesx_moments, cesx_apply, cesx_apply_drift, cesx_apply_finish,
cesx_result), with the same packed moment layout, so that the multi-rank driver
``ces_amd.dist`` can be exercised over ``gloo`` without a GPU.  The arithmetic
is the factored form of oracle/ces_numpy.py (ces/calibrate.py:418-529).
"""
import types

import numpy as np
import torch


class FakeEngine:
    def __init__(self, p, n_obs, J, J_global=None, j_offset=0, seed=1234):
        self.calls = []
        self.p, self.n_obs, self.J = p, n_obs, J
        self.seed = seed
        self.drawn_steps = []           # Philox step indices the driver asked noise for
        self.J_global = J if J_global is None else J_global
        self.j_offset = j_offset
        self.shift = None
        self.metric_sums = np.zeros(2)
        self._res = None

    # -- buffers ---------------------------------------------------------
    def to_device(self, a, rows=None):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))

    def empty(self, rows):
        return torch.empty((rows, self.J), dtype=torch.float64)

    def moments_len(self):
        p, n = self.p, self.n_obs
        return 1 + p + n + p * p + p * n + n * n + 2

    def set_problem(self, y, Gamma, mu, sigma, ustar):
        self.y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
        self.Gamma = np.asarray(Gamma, dtype=np.float64)
        self.mu = np.asarray(mu, dtype=np.float64).reshape(-1, 1)
        self.sigma = np.asarray(sigma, dtype=np.float64)
        self.ustar = np.asarray(ustar, dtype=np.float64).reshape(-1, 1)

    def _xi(self, prm, xi):
        """Injected noise, or the block the device would draw (Philox keyed by the GLOBAL particle index)."""
        if xi is not None:
            return xi.numpy()
        from . import philox
        self.drawn_steps.append(int(prm.step_index))
        return philox.noise_block(self.p, self.J, self.seed, int(prm.step_index), j_offset=self.j_offset,
                                  dtype=np.float64)

    # -- split entry points --------------------------------------------------
    def colsum(self, U, G):
        U, G = U.numpy(), G.numpy()
        return torch.as_tensor(np.concatenate([[float(self.J)], U.sum(axis=1), G.sum(axis=1)]))

    def set_shift(self, sums):
        s = sums.numpy()
        self.shift = s[1:] / s[0]

    # packed layout of include/cesx.h: [N, sum a, S_aa | sum b, S_ab, S_bb, lagged q sums]
    def moments_uu_len(self):
        return 1 + self.p + self.p * self.p

    def moments_uu(self, U, G, out=None):
        p = self.p
        a = U.numpy() - self.shift[:p, None]
        mom = torch.zeros(self.moments_len(), dtype=torch.float64) if out is None else out
        mom[: self.moments_uu_len()] = torch.as_tensor(np.concatenate([[float(self.J)], a.sum(axis=1), (a @ a.T).ravel()]))
        return mom

    def chol_async(self, prm, mom):
        pass                                     # the stand-in factors C inside apply()

    # the sharded driver's stream-overlap branch (ces_amd/dist.py): on the CPU there are no streams, the order of
    # the calls and of the collectives is what a gloo test of that branch checks
    def moments_uu_handover(self, U, G, out=None):
        self.calls.append("uu_handover")
        return self.moments_uu(U, G, out=out)

    def side_stream(self):
        return None                              # torch.cuda.stream(None) is a no-op context

    def moments_rest(self, U, G, mom):
        p = self.p
        a = U.numpy() - self.shift[:p, None]
        b = G.numpy() - self.shift[p:, None]
        mom[self.moments_uu_len():] = torch.as_tensor(np.concatenate([b.sum(axis=1), (a @ b.T).ravel(), (b @ b.T).ravel(),
                                                                       self.metric_sums]))
        return mom

    def moments(self, U, G):
        return self.moments_rest(U, G, self.moments_uu(U, G))

    def _dense(self, prm, mom):
        p, n = self.p, self.n_obs
        m = mom.numpy()
        N = m[0]
        sa = m[1:1 + p]
        o = 1 + p
        Saa = m[o:o + p * p].reshape(p, p); o += p * p
        sb = m[o:o + n]; o += n
        Sab = m[o:o + p * n].reshape(p, n); o += p * n
        Sbb = m[o:o + n * n].reshape(n, n); o += n * n
        lag = m[o:o + 2] / N
        ubar = (self.shift[:p] + sa / N)[:, None]
        gbar = (self.shift[p:] + sb / N)[:, None]
        S_uu = Saa - np.outer(sa, sa) / N
        S_ug = Sab - np.outer(sa, sb) / N
        S_ee = Sbb - np.outer(sb, sb) / N
        mm = gbar - self.y
        S_rr = S_ee + N * (mm @ mm.T)
        div = N if prm.update == 0 else N - 1
        C = S_uu / div + 1e-8 * np.eye(p)
        L = np.linalg.cholesky(C)
        K = np.linalg.solve(self.Gamma.T, (S_ug / N).T).T
        M = np.linalg.solve(self.sigma.T, C.T).T
        X = np.linalg.solve(self.Gamma, S_rr)
        X = np.linalg.solve(self.Gamma, X.T).T
        frob = np.sqrt(max(float((X * S_ee).sum()), 0.0)) / N
        d = types.SimpleNamespace(N=N, ubar=ubar, gbar=gbar, S_ee=S_ee, C=C, L=L, K=K, M=M, frob=frob,
                                  S_ug=S_ug, alpha=(p + 1.0) / N, lag=lag,
                                  self_bias=np.trace(S_uu) / N,
                                  bias=np.trace(S_uu) / N + float(((ubar - self.ustar) ** 2).sum()))
        return d

    def _hk(self, prm, d):
        ts = prm.time_step
        radspec = 0.0
        if ts == 0:
            hk = 1.0 / (d.frob + 1e-8)
        elif ts == 1:
            radspec = max(float(np.linalg.eigvals(np.linalg.solve(self.Gamma, d.S_ee) / d.N).real.max()), 0.0)
            hk = 1.0 / radspec
        elif ts == 2:
            hk = prm.delta_t
        elif ts == 4:
            hk = 1.0 / (d.frob + 1e-8) if (prm.t_len == 0 or prm.t_last < prm.spinup) else prm.delta_t
        else:
            raise AttributeError("'sampling' object has no attribute 'LM_procedure'")
        return hk, radspec

    def _data_metrics(self, G, d):
        Ginv = np.linalg.inv(self.Gamma)
        E = G.numpy() - d.gbar
        R = G.numpy() - self.y
        qe = ((Ginv @ E) * E).sum(axis=0)
        qr = ((Ginv @ R) * R).sum(axis=0)
        self.metric_sums = np.array([(qr ** 2).sum(), (qe ** 2).sum()])

    def _finish(self, prm, d, hk, radspec):
        t_new = hk if prm.first_step else hk + prm.t_last
        self._res = types.SimpleNamespace(
            hk=hk, t_new=t_new, self_bias=d.self_bias, bias=d.bias, radspec=radspec,
            bias_data=self.metric_sums[0] / d.N, self_bias_data=self.metric_sums[1] / d.N,
            lag_bias_data=d.lag[0], lag_self_bias_data=d.lag[1], status=0)

    def apply(self, prm, mom, U, G, xi=None, out=None):
        p = self.p
        d = self._dense(prm, mom)
        hk, radspec = self._hk(prm, d)
        t_new = hk if prm.first_step else hk + prm.t_last
        K = d.K
        if prm.time_step == 2 or (prm.update == 1 and prm.time_step == 4 and t_new > 1):
            K = np.linalg.solve((hk * d.S_ee / d.N + self.Gamma).T, (d.S_ug / d.N).T).T
        if prm.update == 1:
            W = np.hstack([(1 + hk * d.alpha) * np.eye(p) - hk * d.M, -hk * K, np.sqrt(2 * hk) * d.L])
            b = hk * (K @ self.y + d.M @ self.mu - d.alpha * d.ubar)
        else:
            P = np.linalg.inv(np.eye(p) + hk * d.M)
            W = np.hstack([P, -hk * (P @ K), np.sqrt(2 * hk) * d.L])
            b = P @ (hk * (K @ self.y + d.M @ self.mu))
        res = W @ np.vstack([U.numpy(), G.numpy(), self._xi(prm, xi)]) + b
        self._data_metrics(G, d)
        self._finish(prm, d, hk, radspec)
        return torch.as_tensor(res)

    def apply_drift(self, prm, mom, U, G, out):
        p = self.p
        d = self._dense(prm, mom)
        sw = prm.switch_mult * d.alpha
        Wd = np.hstack([sw * np.eye(p) - d.M, -d.K])
        bd = d.K @ self.y + d.M @ self.mu - sw * d.ubar
        drift = Wd @ np.vstack([U.numpy(), G.numpy()]) + bd
        out.copy_(torch.as_tensor(drift))
        self._data_metrics(G, d)
        self._pending = d
        return torch.tensor([float(np.abs(drift).max())], dtype=torch.float64)

    def apply_finish(self, prm, absmax, U, xi, out):
        d = self._pending
        hk = 0.1 / float(absmax[0])
        res = U.numpy() + hk * out.numpy() + np.sqrt(2 * hk) * (d.L @ self._xi(prm, xi))
        self._finish(prm, d, hk, 0.0)
        out.copy_(torch.as_tensor(res))
        return out

    def result(self):
        return self._res
