"""SURVEY.md 8 f3 -- the marker work that stays on the host (csrc/host/markers.cpp):
phase_changes (phasechanges.cxx:109-152, every 10 steps: dynearthsol.cxx:881-894) and
markers.init_marker_option = 2 (markerset.cxx:556-663).  The host code is the same under the CPU
oracle and the HIP engine (des_run drives either through the engine table), so here the loop runs
over the oracle and the marker logic is checked against an independent numpy restatement of the
reference's rules; tests/test_gpu_driver.py repeats the run on the device and compares frames."""
import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from dynearthsol_amd import driver
from test_driver_output import oracle_api, read_frame, in_tmp  # noqa: F401

# eight materials (phase_change_option = 1 needs the full set, input.cxx:1402): upper crust on
# sediment on serpentinised mantle on mantle, under a young oceanic geotherm -- the sediments below
# 20 km are hotter than 650 C (-> schist) and the serpentinite is past its stability field (-> mantle)
PHASE_OV = ("mat.num_materials = 8\nmat.phase_change_option = 1\n"
            "ic.mattype_option = 1\nic.num_mattype_layers = 4\nic.layer_mattypes = [6, 4, 1, 0]\n"
            "ic.mattype_layer_depths = [0.2, 0.5, 0.8]\n"
            "sim.max_steps = 30\nsim.output_step_interval = 10\nsim.has_marker_output = yes\nmesh.quality_check_step_interval = 100\n")
PHASE_KW = dict(cfgs.EVP, lx=60e3, ly=10e3, lz=60e3, res=5e3, ic="oceanic_plate_age_in_yr = 1e6\n")


def phase_host(extra=""):
    return des.Host(cfg_text=cfgs.make(**PHASE_KW), overrides=PHASE_OV + extra)


def numpy_phase_rules(mt, Z, P, T):
    """phasechanges.cxx:10-90 restated with arrays (no hydrous markers: mantle never hydrates)"""
    new = mt.copy()
    new[(mt == 2) & (T > 500 + 273) & (P > -0.3e9 + 2.2e6 * T)] = 3
    new[(mt == 4) & (T > 650 + 273) & (Z < -20e3)] = 5
    new[(mt == 1) & (T > 550 + 273) & (P > 2.1e9 + (7.5e9 - 2.1e9) * (T - (730 + 273)) / (500 - 730))] = 0
    return new


def test_phase_changes_move_markers_as_the_reference_rules_say(in_tmp):
    host = phase_host("sim.modelname = pc\n")
    nmat, ne, nn = host.params.nmat, host.nelem, host.nnode
    mt0 = host.array("markerset.mattype")
    elem = host.array("markerset.elem")
    eta = host.array("markerset.eta").reshape(4, -1)
    conn = host.array("connectivity").reshape(4, ne)
    assert set(np.unique(mt0)) == {0, 1, 4, 6}
    st = driver.run(host, api=oracle_api())
    assert (st.steps, st.frames, st.exit_code) == (30, 4, 0) and st.phase_changed_markers > 0
    # frame 1 = state at step 10, right after the first phase_changes: apply the rules to the
    # frame's own coordinates / temperatures (the loop used exactly those)
    f0, f1 = read_frame("pc.save.000000"), read_frame("pc.save.000001")
    z = f1["coordinate"].view(np.float64).reshape(nn, 3)[:, 2]
    T = f1["temperature"].view(np.float64)
    Zm = (z[conn[:, elem]] * eta).sum(axis=0)
    Tm = (T[conn[:, elem]] * eta).sum(axis=0)
    P = 3300 * 10.0 * 0 + host.params.rho0[host.params.mattype_ref] * host.params.gravity * (-Zm)      # ref_pressure_option 0
    want = numpy_phase_rules(mt0, Zm, P, Tm)
    got = f1["markerset.mattype"].view(np.int32)
    assert np.array_equal(f0["markerset.mattype"].view(np.int32), mt0)
    assert np.array_equal(got, want)
    changed = (want != mt0)
    assert changed.sum() > 0 and set(np.unique(want[changed])) == {0, 5}          # serpentinite -> mantle, sediment -> schist
    # the per-element counts the engine works with follow the markers
    counts = np.zeros((ne, nmat), dtype=np.int64)
    np.add.at(counts, (elem, got), 1)
    assert np.array_equal(f1["material"].view(np.float64).astype(np.int64), counts.argmax(axis=1))


def test_phase_change_options_are_checked_like_the_reference():
    base = cfgs.make(**PHASE_KW)
    for ov, code in ((PHASE_OV.replace("num_materials = 8", "num_materials = 7"), 11),       # input.cxx:1402-1404
                     (PHASE_OV.replace("phase_change_option = 1", "phase_change_option = 2"), 11),   # phasechanges.cxx:135-137
                     (PHASE_OV + "control.has_hydration_processes = yes\n", 31)):
        with pytest.raises(des.DesError) as e:
            des.Host(cfg_text=base, overrides=ov)
        assert e.value.code == code, ov
    # option 101 is the reference's empty template: accepted, changes nothing
    h = des.Host(cfg_text=base, overrides=PHASE_OV.replace("phase_change_option = 1", "phase_change_option = 101"))
    assert h.cfg_int("mat.phase_change_option") == 101


def test_regularly_spaced_markers():
    """init_marker_option = 2: one marker per grid point of spacing init_marker_spacing * resolution
    (truncated to whole metres, as the reference's `const int d`), each inside the element it is
    filed under, with the material of its depth."""
    kw = dict(cfgs.EP, lx=20e3, ly=10e3, lz=10e3, res=2.5e3)
    ov = ("markers.init_marker_option = 2\nmarkers.init_marker_spacing = 0.2\nmat.num_materials = 2\n"
          "ic.mattype_option = 1\nic.num_mattype_layers = 2\nic.layer_mattypes = [0, 1]\nic.mattype_layer_depths = [0.5]\n")
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=ov)
    nn, ne = host.nnode, host.nelem
    d = int(0.2 * 2.5e3)
    nx, ny, nz = int(20e3 / d + 1), int(10e3 / d + 1), int(10e3 / d + 1)
    eta = host.array("markerset.eta").reshape(4, -1)
    elem, mt = host.array("markerset.elem"), host.array("markerset.mattype")
    assert eta.shape[1] == nx * ny * nz                    # a box: every grid point finds its element
    assert np.all(eta >= -5e-11) and np.allclose(eta.sum(axis=0), 1, atol=1e-12)
    coord = host.array("coord").reshape(3, nn)
    conn = host.array("connectivity").reshape(4, ne)
    pos = np.einsum("dkm,km->dm", coord[:, conn[:, elem]], eta)
    n = np.arange(nx * ny * nz)
    grid = np.stack([(n % nx) * d, ((n // nx) % ny) * d, -10e3 + (n // (nx * ny)) * d]).astype(float)
    assert np.abs(pos - grid).max() < 1e-6                 # marker n sits on grid point n
    assert np.array_equal(mt, np.where(grid[2] >= -5e3, 0, 1))
    em = host.array("elemmarkers").reshape(ne, 2)
    counts = np.zeros((ne, 2), dtype=np.int64)
    np.add.at(counts, (elem, mt), 1)
    assert np.array_equal(em, counts) and em.sum(axis=1).min() >= 1
    # too coarse a grid leaves elements empty: the reference's constructor dies (markerset.cxx:56-66)
    with pytest.raises(des.DesError) as e:
        des.Host(cfg_text=cfgs.make(**kw), overrides=ov.replace("init_marker_spacing = 0.2", "init_marker_spacing = 2"))
    assert e.value.code == 52


def test_regularly_spaced_markers_2d():
    """The same in the 2-D build (ny = 1; triangle barycentric coordinates, barycentric-fn.cxx:230-244; tolerance 1e-12)."""
    kw = dict(cfgs.EP, lx=20e3, lz=10e3, res=2.5e3)
    ov = ("markers.init_marker_option = 2\nmarkers.init_marker_spacing = 0.2\nmat.num_materials = 2\n"
          "ic.mattype_option = 1\nic.num_mattype_layers = 2\nic.layer_mattypes = [0, 1]\nic.mattype_layer_depths = [0.5]\n")
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=ov, ndims=2)
    nn, ne = host.nnode, host.nelem
    d = int(0.2 * 2.5e3)
    nx, nz = int(20e3 / d + 1), int(10e3 / d + 1)
    eta = host.array("markerset.eta").reshape(3, -1)
    elem, mt = host.array("markerset.elem"), host.array("markerset.mattype")
    assert eta.shape[1] == nx * nz                         # a box: every grid point finds its element
    assert np.all(eta >= -1e-12) and np.allclose(eta.sum(axis=0), 1, atol=1e-12)
    coord = host.array("coord").reshape(2, nn)
    conn = host.array("connectivity").reshape(3, ne)
    pos = np.einsum("dkm,km->dm", coord[:, conn[:, elem]], eta)
    n = np.arange(nx * nz)
    grid = np.stack([(n % nx) * d, -10e3 + (n // nx) * d]).astype(float)
    assert np.abs(pos - grid).max() < 1e-6                 # marker n sits on grid point n
    assert np.array_equal(mt, np.where(grid[1] >= -5e3, 0, 1))
    em = host.array("elemmarkers").reshape(ne, 2)
    counts = np.zeros((ne, 2), dtype=np.int64)
    np.add.at(counts, (elem, mt), 1)
    assert np.array_equal(em, counts) and em.sum(axis=1).min() >= 1
    with pytest.raises(des.DesError) as e:
        des.Host(cfg_text=cfgs.make(**kw), overrides=ov.replace("init_marker_spacing = 0.2", "init_marker_spacing = 2"), ndims=2)
    assert e.value.code == 52
