"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/toricenv.h declares; pure-host entry points behave; nothing here launches a kernel."""
import ctypes as C
import os
import re

import pytest
import torch

import toric_rl_decoder_amd as T
from toric_rl_decoder_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    T.build()
    return T.load()


def header_functions():
    text = open(os.path.join(ROOT, "include", "toricenv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tq_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = header_functions()
    assert len(declared) >= 25
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert set(declared) == bound
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_version_and_block_layout(lib):
    assert lib.tq_version() == 200
    # d=7: W=1 -> 4*8 + 4 + 4 + 4 + 1 bytes per transition (planes, action, reward, priority, terminal),
    # sections 8-byte aligned
    assert lib.tq_transition_block_bytes(7, 8) == 4 * 8 * 8 + 32 + 32 + 32 + 8
    assert lib.tq_transition_block_bytes(9, 1000) == 4 * 8 * 2 * 1000 + 3 * 4000 + 1000
    from toric_rl_decoder_amd import wire
    for d in (3, 5, 7, 9, 11, 13, 15, 17, 19, 21):
        for cap in (1, 7, 64, 1000, 65536):
            assert lib.tq_transition_block_bytes(d, cap) == wire.block_bytes(d, cap)
    assert lib.tq_transition_block_bytes(4, 8) == -1
    assert lib.tq_transition_block_bytes(7, -1) == -1


def test_errors_are_codes_not_aborts(lib):
    h = C.c_void_p(None)
    rc = lib.tq_create(C.byref(h), 0, 7, 0, 1, 0)
    assert rc == _lib.TQ_E_INVALID and b"n_envs" in lib.tq_last_error()
    rc = lib.tq_create(C.byref(h), 8, 6, 0, 1, 0)
    assert rc == _lib.TQ_E_INVALID and b"unsupported lattice size" in lib.tq_last_error()
    assert lib.tq_set_params(None, 0.1, 100.0, 75) == _lib.TQ_E_INVALID
    assert lib.tq_destroy(None) == 0
    assert lib.tq_states_reserve(4, 10) == _lib.TQ_E_INVALID and lib.tq_states_reserve(7, 0) == _lib.TQ_E_INVALID
    assert lib.tq_block_priorities(7, None, 8, 4, 2, None, 0.95, None) == _lib.TQ_E_INVALID
    if not torch.cuda.is_available():
        rc = lib.tq_create(C.byref(h), 8, 7, 0, 1, 0)       # no HIP device: an error code, not a crash
        assert rc < 0 and lib.tq_last_error()


def test_workgroup_share_setting_is_range_checked(lib):
    """tq_set_xcd_bias / tq_get_xcd_bias (toricenv.h): a process-wide host setting, 0..16, no device needed."""
    before = lib.tq_get_xcd_bias()
    assert 0 <= before <= 16
    try:
        for b in (0, 16, 5):
            assert lib.tq_set_xcd_bias(b) == 0 and lib.tq_get_xcd_bias() == b
        for b in (-1, 17, 1 << 20):
            assert lib.tq_set_xcd_bias(b) < 0 and b"xcd bias" in lib.tq_last_error()
            assert lib.tq_get_xcd_bias() == 5
        assert lib.tq_env_set_xcd_bias(None, 3) < 0 and lib.tq_env_get_xcd_bias(None) < 0      # no handle without a device
    finally:
        lib.tq_set_xcd_bias(before)


def test_python_surface_fails_loudly_without_gpu():
    env = T.make("toric-code-v0", {"size": 5, "min_qubit_errors": 0, "p_error": 0.1})
    assert env.system_size == 5 and int(env.action_space.high[-1]) == 3
    with pytest.raises(ValueError):
        T.make("toric-code-v0", {"size": 4})
    with pytest.raises(ValueError):
        T.make("cartpole", {})
    if not torch.cuda.is_available():
        with pytest.raises(T.ToricEnvError):
            T.EnvSet(env, 4)
    dt = T.transition_dtype(7)
    assert dt.itemsize == 1609                               # reference record at d=7


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: with libtoricenv.so absent, load() raises and nothing else is tried."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libtoricenv.so"))
    with pytest.raises(T.ToricEnvError, match="no CPU fallback"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load() is not None


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no module of the package may import it."""
    import glob
    pkg = os.path.join(ROOT, "toric-rl-decoder_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True) + glob.glob(os.path.join(pkg, "csrc", "*")):
        text = open(path, errors="ignore").read()
        assert "import oracle" not in text and "from oracle" not in text and "toric_oracle" not in text, path
        assert "host_twin" not in text, path                 # the host twin of the ABI is test infrastructure too
