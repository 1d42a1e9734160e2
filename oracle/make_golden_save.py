"""Generate tests/golden/ref_save/ by letting the REAL reference write its on-disk format (build container only).

    python -m oracle.make_golden_save            # from the repo root

The reference's ``sampling.run(save_online=True)`` dumps one ``ensemble_NNNN.npy`` / ``Gensemble_NNNN.npy`` pair per
iteration plus ``metrics.pkl`` (ces/calibrate.py:371-385 -> enka.save(online=True), :170-197), and ``enka.save(all=True)``
writes ``ensemble.npy``, ``Gensemble.npy``, ``metrics.pkl``, ``ensemble_path.npy``, ``Gensemble_path.npy``.  Both
directories are committed as DATA files (numpy arrays and a pickled dict of float lists, a few KB): the build's
``enka.load`` must read what the reference wrote (tests/test_save_format.py), and the reference's ``load`` what the build
writes.  ``expected.npz`` holds the arrays / metric lists the reference's own ``load`` returns for them.  No reference
source text is stored.
"""
import contextlib
import io
import os
import shutil

import numpy as np

from . import _refload

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_save")


def main():
    ref = _refload.load_reference_calibrate()
    utils = _refload.load_reference_utils()
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    p, n, J, T = 2, 10, 20, 3
    np.random.seed(1)                                   # examples/notebooks/linear.ipynb:73-74
    A = np.ones((n, p))
    A[:, 1] = 2.0 * np.random.normal(0, 1, n)
    model = utils.lineal(A)
    model.l_window = 5                                  # run(save_online=True) names the directory with it (:376)
    eks = ref.sampling(p=p, n_obs=n, J=J)
    eks.T = T
    eks.ustar = np.array([[-1.0], [2.0]])
    eks.mu, eks.sigma = np.zeros((p, 1)), 100.0 * np.eye(p)
    eks.directory, eks.nexp = OUT, 1
    Gamma = 0.1 * np.eye(n)
    y = (A @ eks.ustar).ravel()
    np.random.seed(7)
    U0 = np.random.normal(0, 1, [p, J])
    with contextlib.redirect_stdout(io.StringIO()):
        eks.run(y, U0, model, Gamma, np.linalg.cholesky(Gamma), save_online=True, t_tol=1e9)
        eks.save(path=OUT + "/", file="final/", all=True)
    online = os.path.join("ensembles", "lineal-eks-005-%s-01" % str(J).zfill(4))
    assert os.path.isdir(os.path.join(OUT, online)), os.listdir(os.path.join(OUT, "ensembles"))
    # what the reference's own load returns for the two directories
    a = ref.sampling(p=p, n_obs=n, J=J)
    assert a.load(path=OUT + "/", eks_dir="final/")
    b = ref.sampling(p=p, n_obs=n, J=1)
    assert b.load(path=os.path.join(OUT, online) + "/", eks_dir="", ix_ensemble=True)
    np.savez(os.path.join(OUT, "expected.npz"), final_Uall=a.Uall, final_Gall=a.Gall,
             online_Uall=b.Uall, online_Gall=b.Gall, online_J=b.J, online_dir=online,
             **{"metric_" + k.replace("-", "_"): np.asarray(v) for k, v in a.metrics.items()},
             numpy_version=np.__version__)
    for root, _, files in os.walk(OUT):
        for f in sorted(files):
            print(os.path.relpath(os.path.join(root, f), OUT), os.path.getsize(os.path.join(root, f)))


if __name__ == "__main__":
    main()
