"""The PRODUCT library csrc/libclc_hip.so itself (the rest of the GPU suite runs on the -DCLC_TEST_HOOKS build of the same
translation units, tests/conftest.py): loaded next to the hooks build, it solves C1, a cooperative-size problem, a batch and a
stored-scan selection to the oracle's tolerances and BIT FOR BIT like the hooks build; its hooks are absent and say so."""
import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import _build, simdata as sd

pytestmark = pytest.mark.gpu

X0 = sd.pose7_from_T(np.eye(4))


@pytest.fixture(scope="module")
def pair():
    prod = clc.Solver(0, library=_build.PRODUCT_LIB_PATH)
    hooks = clc.Solver(0, library="hooks")
    yield prod, hooks
    prod.close()
    hooks.close()


def test_product_library_is_loaded_and_has_no_hooks(pair):
    prod, hooks = pair
    assert not prod._L.has_hooks and hooks._L.has_hooks
    maps = open("/proc/self/maps").read()
    import os
    hooks_name = os.path.basename(_build.LIB_PATH if _build.LIB_PATH != _build.PRODUCT_LIB_PATH else _build.HOOKS_LIB_PATH)  # (the hooks build)
    assert "camlasercalibratool_amd/csrc/libclc_hip.so" in maps and "camlasercalibratool_amd/csrc/" + hooks_name in maps
    assert "gfx950" in prod.device_info()[0]
    prod.upload(clc.flatten_observations(sd.GenerateSimData(1, noise_sigma=0.01), False))
    with pytest.raises(RuntimeError, match="test hook"):
        prod.time_eval(X0)
    with pytest.raises(RuntimeError, match="test hook"):
        prod.debug_coop_control(drop_next=1)


def test_product_library_solves_like_the_hooks_build_and_the_oracle(pair, oracle_mod):
    prod, hooks = pair
    cases = [clc.flatten_observations(sd.GenerateSimData(3, noise_sigma=0.01), False),           # C1: single-workgroup resident kernel
             clc.flatten_observations(sd.sim_fixed_count(5, 120, 500, noise_sigma=0.01), False)]  # 6e4 observations: cooperative kernel
    for rec in cases:
        res = []
        for sv in (prod, hooks):
            sv.upload(rec)
            res.append(sv.solve(X0))
        a, b = res
        assert np.array_equal(a.pose, b.pose) and a.summary.final_cost == b.summary.final_cost and a.summary.num_iterations == b.summary.num_iterations
        assert [(t.cost, t.step_norm) for t in a.trace] == [(t.cost, t.step_norm) for t in b.trace]
        ref = oracle_mod.solve(rec, X0, linear_solver="qr")
        assert np.abs(sd.T_from_pose7(a.pose) - sd.T_from_pose7(ref.pose)).max() <= 1e-6 and abs(a.summary.final_cost - ref.summary.final_cost) <= 1e-8
        assert a.summary.num_iterations == ref.summary.num_iterations
    pi = prod.path_info()
    assert pi.coop_resident == 1 and pi.coop_solves == 1 and pi.coop_timeouts == 0 and pi.single_resident == 0
    # a batch on the resident kernel
    rec, off, x0, gt = sd.sim_shard_records(7, 0, 64, 20, 500, 0.01)
    out = []
    for sv in (prod, hooks):
        sv.upload_batched(rec, off)
        out.append(sv.solve_batched(x0))
    assert prod.path_info().batched_resident == 1 and prod.path_info().batched_lanes == 256
    assert np.array_equal(out[0][0], out[1][0])
    assert [s.final_cost for s in out[0][1]] == [s.final_cost for s in out[1][1]]
    for k in (0, 31, 63):
        ref = oracle_mod.solve(rec[off[k]:off[k + 1]], x0[k], linear_solver="qr")
        assert np.abs(sd.T_from_pose7(out[0][0][k]) - sd.T_from_pose7(ref.pose)).max() <= 1e-6
