"""The on-disk format of ``enka.save`` / ``enka.load`` (ces/calibrate.py:170-237) against files the REFERENCE wrote.

tests/golden/ref_save/ holds two directories produced by the reference itself (oracle/make_golden_save.py: its
``run(save_online=True)`` dumps and its ``save(all=True)``), data files only, and ``expected.npz`` with what the
reference's own ``load`` returns for them.  The build's ``load`` must return the same, and what the build's ``save``
writes must have the same file names, dtypes and shapes -- so that a directory can pass between the two either way.
No GPU: save / load are host code.
"""
import os
import pickle

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "ref_save")


def _expected():
    return np.load(os.path.join(REF, "expected.npz"), allow_pickle=False)


def test_load_reads_what_the_reference_saved():
    from ces_amd.calibrate import sampling
    ex = _expected()
    a = sampling(p=2, n_obs=10, J=20)
    assert a.load(path=REF + "/", eks_dir="final/")
    assert np.array_equal(a.Uall, ex["final_Uall"]) and np.array_equal(a.Gall, ex["final_Gall"])
    for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t"):
        assert np.array_equal(np.asarray(a.metrics[k]), ex["metric_" + k.replace("-", "_")]), k
    online = str(ex["online_dir"])
    b = sampling(p=2, n_obs=10, J=1)
    assert b.load(path=os.path.join(REF, online) + "/", eks_dir="", ix_ensemble=True)
    assert np.array_equal(b.Uall, ex["online_Uall"]) and np.array_equal(b.Gall, ex["online_Gall"])
    assert b.J == int(ex["online_J"]) and np.array_equal(b.Ustar, ex["online_Uall"][-1])
    c = sampling(p=2, n_obs=10, J=1)                                    # count taken from the metrics instead of the listing
    assert c.load(path=os.path.join(REF, online) + "/", eks_dir="", ix_ensemble=True, flag_metrics=True)
    assert np.array_equal(c.Uall, ex["online_Uall"])
    assert not sampling(p=2, n_obs=10, J=1).load(path=os.path.join(REF, online) + "/", eks_dir="")   # no path arrays there


def test_save_writes_the_files_the_reference_writes(tmp_path):
    from ces_amd.calibrate import sampling
    ex = _expected()
    a = sampling(p=2, n_obs=10, J=20)
    assert a.load(path=REF + "/", eks_dir="final/")
    a.Ustar, a.Gstar = a.Uall[-1], a.Gall[-1][:10]
    a.save(path=str(tmp_path) + "/", file="again/", all=True)
    ref_files = sorted(os.listdir(os.path.join(REF, "final")))
    assert sorted(os.listdir(tmp_path / "again")) == ref_files
    for f in ref_files:
        if f.endswith(".npy"):
            mine, theirs = np.load(tmp_path / "again" / f), np.load(os.path.join(REF, "final", f))
            assert mine.dtype == theirs.dtype and mine.shape == theirs.shape and np.array_equal(mine, theirs), f
    with open(tmp_path / "again" / "metrics.pkl", "rb") as fh:
        mine = pickle.load(fh)
    with open(os.path.join(REF, "final", "metrics.pkl"), "rb") as fh:
        theirs = pickle.load(fh)
    assert mine == theirs and sorted(mine) == sorted(["self-bias", "self-bias-data", "bias-data", "bias", "t"])
    # online dumps: the names of ces/calibrate.py:192-194
    a.save(path=str(tmp_path) + "/", file="on/", online=True, counter=7)
    assert sorted(os.listdir(tmp_path / "on")) == ["Gensemble_0007.npy", "ensemble_0007.npy", "metrics.pkl"]
    assert np.array_equal(np.load(tmp_path / "on" / "ensemble_0007.npy"), ex["final_Uall"][-1])
