"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/cesx.h declares; struct layouts agree; the host
mirror keeps the reference's interface (ces/calibrate.py:14-22, :241-529).
No compute calls (no GPU here)."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ces_amd import build, engine
    build.build_lib()
    return engine.load_library()


def _declared():
    text = open(os.path.join(ROOT, "include", "cesx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cesx_[a-z_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from ces_amd import engine
    names = _declared()
    assert len(names) >= 21
    for name in names:
        assert hasattr(lib, name), "libcesx.so does not export %s" % name
    assert sorted(engine.EXPORTS) == names
    assert lib.cesx_abi_version() == engine.ABI_VERSION


def test_struct_layouts_match_header():
    from ces_amd import engine
    # sizes implied by include/cesx.h on LP64
    assert ctypes.sizeof(engine.Config) == 4 * 5 + 4 + 8 * 4      # 5 x 32-bit + pad, 4 x 64-bit
    assert ctypes.sizeof(engine.StepParams) == 4 * 6 + 8 * 5
    assert ctypes.sizeof(engine.StepResult) == 8 * 9 + 4 * 2


def test_create_rejects_bad_config_without_gpu(lib):
    from ces_amd import engine
    h = ctypes.c_void_p()
    cfg = engine.Config()
    cfg.struct_bytes = 3
    assert lib.cesx_create(ctypes.byref(cfg), ctypes.byref(h)) == engine.EINVAL
    assert not h
    cfg.struct_bytes = ctypes.sizeof(engine.Config)
    cfg.p, cfg.n_obs, cfg.J_local, cfg.J_global, cfg.dtype = 0, 3, 10, 10, 0
    assert lib.cesx_create(ctypes.byref(cfg), ctypes.byref(h)) == engine.EINVAL
    assert b"invalid" in lib.cesx_last_error(None)


def test_step_params_follow_reference_kwargs():
    from ces_amd import engine
    prm = engine.step_params(update="aldi", time_step="constant", T=30)
    assert prm.delta_t == pytest.approx(1.0 / 15)             # ces/calibrate.py:253
    assert engine.step_params(time_step="mix").spinup == 4.0   # :257
    assert engine.step_params(update="aldi_constant").switch_mult == 1.0   # :517
    with pytest.raises(UnboundLocalError):                      # :262 with an unknown rule
        engine.step_params(time_step="bogus")
    with pytest.raises(ValueError):
        engine.step_params(update="bogus")


def test_host_mirror_keeps_reference_interface():
    from ces_amd.calibrate import enka, sampling
    s = sampling(p=2, n_obs=10, J=40)
    assert (s.p, s.n_obs, s.J, s.T, s.epsilon, s.parallel, s.mute_bar) == (2, 10, 40, 30, 1e-7, False, True)
    assert repr(s) == "enka-0040-eks"                          # ces/calibrate.py:24-28
    assert issubclass(sampling, enka)
    sig = inspect.signature(sampling.run)
    assert list(sig.parameters)[:8] == ["self", "y_obs", "U0", "model", "Gamma", "Jnoise", "save_online", "trace"]
    assert sig.parameters["save_online"].default is False and sig.parameters["trace"].default is True
    for name in ("eks_update", "eks_update_aldi", "eks_update_aldi_constant"):
        assert list(inspect.signature(getattr(sampling, name)).parameters)[:6] == \
            ["self", "y_obs", "U0", "Geval", "Gamma", "iter"]
    assert list(inspect.signature(sampling.timestep_method).parameters)[:6] == \
        ["self", "D", "Geval", "y_obs", "Gamma", "Jnoise"]
    assert sampling.run_eks is sampling.run
    # the base class's placeholders (ces/calibrate.py:50-93): same signatures, no-ops that return None
    e = enka(p=2, n_obs=10, J=40)
    assert list(inspect.signature(enka.run).parameters) == ["self", "y_obs", "U0", "model", "Gamma", "Jnoise"]
    assert list(inspect.signature(enka.run_sde).parameters) == ["self", "y_obs", "U0", "model", "Gamma", "Jnoise"]
    assert list(inspect.signature(enka.eks_update).parameters) == ["self", "Geval"]
    assert e.run(None, None, None, None, None) is None and e.run_sde(None, None, None, None, None) is None
    assert e.eks_update(None) is None


def test_g_ens_matches_reference_loop():
    from ces_amd.calibrate import sampling
    from ces_amd.utils import lineal
    rng = np.random.default_rng(0)
    A = rng.standard_normal((5, 3))
    model = lineal(A, b=0.5)
    assert model.type == "map" and model.n_obs == 5 and model.model_name == "lineal"
    s = sampling(p=3, n_obs=5, J=7)
    U = rng.standard_normal((3, 7))
    assert np.allclose(s.G_ens(U, model), A @ U + 0.5)


def test_timestep_shim_small_D():
    """ces/calibrate.py:243-267 on an explicit D (compatibility shim)."""
    from ces_amd.calibrate import sampling
    s = sampling(p=2, n_obs=3, J=4)
    s.Uall = [None]
    D = np.arange(16.0).reshape(4, 4) / 10
    hk = s.timestep_method(D, None, None, None, None)
    assert hk == pytest.approx(1.0 / (np.linalg.norm(D) + 1e-8)) and s.metrics["t"] == [hk]
    s.Uall = [None, None]
    hk2 = s.timestep_method(D, None, None, None, None, time_step="constant")
    assert hk2 == pytest.approx(1.0 / 15) and s.metrics["t"][-1] == pytest.approx(hk + hk2)
    with pytest.raises(AttributeError):
        s.timestep_method(D, None, None, None, None, time_step="adaptive")


def test_product_path_fails_loudly_without_gpu():
    import torch
    from ces_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        engine.Engine(2, 3, 8)


def test_gram_work_partition_invariants(lib):
    """The work partition of both moments launches (make_gram_plan, kernels_gram.hip; host code, no device): for a
    sweep of shapes, shard sizes, dtypes and workgroup budgets every wanted block of the lower triangle is dealt to
    exactly one wave of one type, the slices of a type cover J in whole tiles, the staged rows fit in LDS and the
    launch stays within its workgroup budget; at the benchmark shape the busiest workgroup of the second launch is
    within 3 % of a perfectly level one."""
    info = (ctypes.c_int * 6)()
    shapes = [(256, 256), (2, 2), (10, 6), (33, 17), (64, 50), (96, 80), (300, 40), (40, 300), (250, 250), (512, 512),
              (700, 96), (130, 520)]
    for p, n in shapes:
        for dtype in (0, 1):
            for J in (32, 1004, 4096, 65536, 524288):
                for budget in (256, 248, 224, 64, 3):
                    for part in (0, 1):
                        bad = lib.cesx_debug_gram_plan(p, n, dtype, part, budget, J, info)
                        assert bad == 0, (p, n, dtype, J, budget, part, bad, list(info))
    # C2, second launch, 248 workgroups: 100 blocks x 2048 tiles over 4 SIMDs x 248 workgroups = 206.5 block-tiles each
    assert lib.cesx_debug_gram_plan(256, 256, 0, 1, 248, 65536, info) == 0
    assert info[2] == 100 and info[1] <= 248 and info[3] <= 1.03 * 100 * 2048 / 4 / info[1] + 1
    assert lib.cesx_debug_gram_plan(256, 256, 0, 0, 256, 65536, info) == 0
    assert info[2] == 36 and info[1] == 256
    assert lib.cesx_debug_gram_plan(0, 4, 0, 0, 256, 64, None) < 0
