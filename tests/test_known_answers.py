"""Known answers for the whole step (tests/golden/portable_run_hashes.json): with the portable
libm on both sides nothing on the path depends on the platform, so whole-run state hashes can be
committed.  The oracle must reproduce them on this CPU; the HIP engine on the MI355X -- without
the oracle in the loop."""
import json
import os

import pytest

import dynearthsol_amd as des
from oracle_binding import OracleEngine, portable_libm

import importlib.util
_spec = importlib.util.spec_from_file_location(
    "make_portable_run_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_portable_run_golden.py"))
gold = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gold)

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "portable_run_hashes.json")) as f:
    WANT = json.load(f)


def _run(engine_cls):
    with portable_libm():
        for name, host, nsteps in gold.cases():
            e = engine_cls(host)
            e.init_from_host(host)
            e.step(nsteps)
            got = gold.digest(e)
            assert got == WANT[name], (name, [k for k in got if got[k] != WANT[name][k]])


def test_oracle_reproduces_the_committed_run_hashes():
    _run(OracleEngine)


@pytest.mark.gpu
def test_device_reproduces_the_committed_run_hashes():
    _run(des.DeviceEngine)
