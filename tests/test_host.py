"""Host library: .cfg front-end (input.cxx), regular mesher and topology builders
(mesh.cxx), initial conditions (ic.cxx)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des

NOF = np.array([[1, 2, 3], [0, 3, 2], [0, 1, 3], [0, 2, 1]])


@pytest.fixture(scope="module")
def host():
    return des.Host(cfg_text=cfgs.make(**cfgs.EP))


def arr(ptr, n, dtype=np.int32):
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, dtype)


def test_cfg_defaults_and_normalisation(host):
    p = host.params
    assert p.gravity == 10 and p.damping_factor == 0.8 and p.damping_option == 1     # input.cxx defaults
    assert p.has_thermal_diffusion == 1 and p.is_using_mixed_stress == 1
    assert p.rheol_type == 5 and p.nmat == 1
    assert p.has_winkler_foundation == 1 and p.vbc_types[4] == 0                      # input.cxx:1252-1255
    assert p.max_vbc_val == 1e-9                                                      # bc.cxx:66-91
    assert p.compensation_pressure == 2700 * 10 * 8e3                                 # ic.cxx:361
    assert host.cfg_int("sim.checkpoint_frame_interval") == 10
    assert host.cfg_double("mat.max_tension") == 1e9


@pytest.mark.parametrize("text,code", [
    ("[sim]\nbogus_key = 1\n", 10),                       # unknown option -> EXIT_CONFIG
    ("[mesh]\nxlength = abc\n", 10),                      # bad value
    ("", 10),                                             # required options missing
])
def test_cfg_errors_carry_the_reference_exit_codes(text, code):
    with pytest.raises(des.DesError) as e:
        des.Host(cfg_text=text)
    assert e.value.code == code


def test_cfg_rejects_what_the_reference_rejects():
    base = cfgs.make(**cfgs.EP)
    with pytest.raises(des.DesError) as e:
        des.Host(cfg_text=base, overrides="mat.rheology_type = plastic\n")
    assert e.value.code == 11
    with pytest.raises(des.DesError) as e:
        des.Host(cfg_text=base, overrides="control.dt_fraction = 2\n")
    assert e.value.code == 11
    with pytest.raises(des.DesError) as e:
        des.Host(cfg_text=base, overrides="mat.rho0 = [1, 2, 3]\n")      # wrong list length
    assert e.value.code == 11
    with pytest.raises(des.DesError) as e:
        des.Host(cfg_text=base.replace("[mat]", "[mat]\nrho0 = [1]"))    # option given twice
    assert e.value.code == 11


def test_features_outside_the_offloaded_path_are_refused_not_ignored():
    """Options that would change the results or the files written, but are not built: code 31,
    never a silent no-op."""
    base = cfgs.make(**cfgs.EP)
    for ov in ("control.has_hydraulic_diffusion = yes\n", "control.surface_process_option = 101\n",
               "control.has_hydration_processes = yes\n", "monitor.enabled = yes\nmonitor.num_points = 1\n",
               "ic.temperature_option = 90\n", "mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n",
               "mesh.meshing_option = 95\nmesh.meshing_elem_shape = 0\n"):
        with pytest.raises(des.DesError) as e:
            des.Host(cfg_text=base, overrides=ov)
        assert e.value.code == 31, ov


def test_cfg_list_broadcast_and_bool_spellings():
    h = des.Host(cfg_text=cfgs.make(nmat=2, **cfgs.EVP), overrides="bc.has_water_loading = on\n")
    assert list(h.params.alpha[:2]) == [3e-5, 3e-5]                       # input.cxx:983-989
    assert h.params.has_water_loading == 1 and h.params.vbc_types[5] == 0


def test_regular_mesh_counts_and_geometry(host):
    nx, ny, nz = 21, 5, 5
    assert host.nnode == nx * ny * nz and host.nelem == 5 * 20 * 4 * 4
    coord = host.array("coord").reshape(3, -1)
    conn = host.array("connectivity").reshape(4, -1)
    d = coord[:, conn]                      # [3, 4, ne]
    a, b, c, dd = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    vol = np.einsum("ie,ie->e", a - dd, np.cross((b - dd).T, (c - dd).T).T) / 6
    assert np.all(np.abs(vol) > 0)
    assert abs(np.abs(vol).sum() - 40e3 * 8e3 * 8e3) < 1e-3 * 40e3 * 8e3 * 8e3
    # renumbering_mesh sorts nodes along x (mesh.cxx:2742-2766)
    key = coord[0] + 1e-3 * coord[1] + 1e-6 * coord[2]
    assert np.all(np.diff(key) >= 0)


def test_boundary_flags_nodes_and_facets(host):
    m = host.mesh
    coord = host.array("coord").reshape(3, -1)
    flag = arr(m.bcflag, m.nnode, np.uint32)
    exp = ((coord[0] == 0) * 1 + (coord[0] == 40e3) * 2 + (coord[1] == 0) * 4 + (coord[1] == 8e3) * 8
           + (coord[2] == -8e3) * 16 + (coord[2] == 0) * 32).astype(np.uint32)
    assert np.array_equal(flag, exp)
    conn = host.array("connectivity").reshape(4, -1)
    ncell_face = [4 * 4, 4 * 4, 20 * 4, 20 * 4, 20 * 4, 20 * 4]
    outward = [(0, -1), (0, 1), (1, -1), (1, 1), (2, -1), (2, 1)]
    for i in range(6):
        nf = m.nbfacets[i]
        assert nf == 2 * ncell_face[i]
        fe, ff = arr(m.bfacet_elem[i], nf), arr(m.bfacet_facet[i], nf)
        assert np.all(np.diff(fe) >= 0)                       # sorted by element (mesh.cxx:3240-3246)
        nodes = conn[NOF[ff].T, fe]                           # [3, nf]
        assert np.all(flag[nodes] & (1 << i))
        p = coord[:, nodes]
        n = np.cross((p[:, 1] - p[:, 0]).T, (p[:, 2] - p[:, 0]).T)
        ax, sgn = outward[i]
        assert np.all(n[:, ax] * sgn > 0)                     # counter-clockwise seen from outside
        bn = arr(m.bnodes[i], m.nbnodes[i])
        assert np.array_equal(bn, np.nonzero(flag & (1 << i))[0])


def test_support_csr_matches_a_numpy_rebuild(host):
    m = host.mesh
    ne, nn = m.nelem, m.nnode
    conn = host.array("connectivity").reshape(4, -1)
    idx = arr(m.support_idx, nn + 1)
    sa, sl = arr(m.support_arr, 4 * ne), arr(m.support_lidx, 4 * ne)
    e_ids = np.repeat(np.arange(ne), 4)
    l_ids = np.tile(np.arange(4), ne)
    nodes = conn.T.ravel()
    order = np.lexsort((l_ids, e_ids, nodes))                 # by node, then ascending element
    assert np.array_equal(sa, e_ids[order]) and np.array_equal(sl, l_ids[order])
    assert np.array_equal(idx, np.concatenate([[0], np.cumsum(np.bincount(nodes, minlength=nn))]))
    assert np.array_equal(conn[sl, sa], np.repeat(np.arange(nn), np.diff(idx)))


def test_surface_info(host):
    m = host.mesh
    top = arr(m.top_nodes, m.ntop)
    coord = host.array("coord").reshape(3, -1)
    assert np.all(coord[2, top] == 0) and m.ntop == 21 * 5 and m.etop == 2 * 20 * 4
    assert np.all(np.diff(coord[0, top]) >= 0)                # sorted by x (mesh.cxx:3047-3055)
    ean = arr(m.elem_and_nodes, 3 * m.etop).reshape(3, -1)
    cs = arr(m.connectivity_surface, 4 * m.etop).reshape(4, -1)
    assert np.array_equal(top[ean], cs[:3])
    te = arr(m.top_elems, m.ntop_elems)
    conn = host.array("connectivity").reshape(4, -1)
    is_top = np.isin(conn, top).any(axis=0)
    assert np.array_equal(te, np.nonzero(is_top)[0])
    bn = np.ctypeslib.as_array(m.bnormals, shape=(30,)).reshape(3, 10)
    assert np.allclose(bn[:, :6].T, [[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]])


def test_initial_conditions(host):
    coord = host.array("coord").reshape(3, -1)
    conn = host.array("connectivity").reshape(4, -1)
    zc = coord[2, conn].sum(axis=0) / 4
    s = host.array("stress").reshape(6, -1)
    e = host.array("strain").reshape(6, -1)
    p = 2700 * 10 * (-zc)
    assert np.allclose(s[:3], -p, rtol=1e-15) and np.all(s[3:] == 0)           # ic.cxx:322-362
    assert np.allclose(e[:3], -p / 50e9 / 3, rtol=1e-15)
    pls = host.array("plstrain")
    assert set(np.unique(pls)) == {0.0, 0.5} and 0 < (pls > 0).sum() < pls.size // 4
    assert np.all(host.array("temperature") == 273)                            # erf profile with Tm = Ts
    assert np.all(host.array("elemmarkers") == 4)
    assert np.all(host.array("viscosity") == 1e24)


def test_layered_markers_follow_the_reference_rng():
    # mattype_option 1: markers_per_element random markers per element, material by marker
    # depth (markerset.cxx:524-553, 703-716) with libc rand() seeded by markers.random_seed
    h = des.Host(cfg_text=cfgs.make(nmat=2, **cfgs.EVP))
    mk = h.array("elemmarkers").reshape(-1, 2)
    assert np.all(mk.sum(axis=1) == 4)
    coord = h.array("coord").reshape(3, -1)
    conn = h.array("connectivity").reshape(4, -1)
    zmax, zmin = coord[2, conn].max(axis=0), coord[2, conn].min(axis=0)
    assert np.all(mk[zmin >= -4e3, 0] == 4) and np.all(mk[zmax < -4e3, 1] == 4)
    h2 = des.Host(cfg_text=cfgs.make(nmat=2, **cfgs.EVP))
    assert np.array_equal(mk, h2.array("elemmarkers").reshape(-1, 2))


def test_mesh_file_round_trip(tmp_path, host):
    path = str(tmp_path / "m.desmesh")
    host.save_mesh(path)
    h2 = des.Host(cfg_text=cfgs.make(**cfgs.EP), mesh_file=path)
    for name in ("coord", "connectivity", "segment", "segflag", "stress", "plstrain"):
        assert np.array_equal(host.array(name), h2.array(name))
    assert h2.mesh.ntop == host.mesh.ntop


def _declared(header):
    import re
    txt = open(os.path.join(des.REPO_ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(des_(?:dev|host|part|output|run)(?:_[a-z_0-9]+)?)\s*\(", txt)))


def _exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_host_library_exports_every_declared_symbol():
    syms = _exported(des.HOST_LIB_PATH)
    decl = _declared("des_host.h") + _declared("des_run.h")
    assert "des_run" in decl and "des_output_write" in decl and "des_part_halo" in decl
    missing = [s for s in decl if s not in syms]
    assert not missing, missing


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(des.HIP_LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(des.REPO_ROOT, "dynearthsol_amd", "csrc"), "../libdes_hip.so"])
    lib = C.CDLL(des.HIP_LIB_PATH)                   # loads without a GPU
    syms = _exported(des.HIP_LIB_PATH)
    decl = _declared("des_dev.h")
    assert len(decl) >= 15
    missing = [s for s in decl if s not in syms]
    assert not missing, missing
    assert lib.des_dev_device_count() >= 0


def test_device_path_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = des.Host(cfg_text=cfgs.make(**cfgs.EP))
    with pytest.raises(des.DesError) as e:
        des.DeviceEngine(h)
    assert e.value.code == 31


def test_multi_segment_weak_zone():
    """ic.weakzone_option = 5 (Multi_planar_zone, ic.cxx:72-179, 577-622): two bounded planes."""
    ov = ("ic.weakzone_option = 5\nic.weakzone_num_segments = 2\nic.weakzone_segments_xcenter = [0.3, 0.7]\n"
          "ic.weakzone_segments_zcenter = [0.5, 0.5]\nic.weakzone_segments_inclination = [90, 60]\n"
          "ic.weakzone_segments_halfwidth = [1.2, 1.2]\nic.weakzone_segments_x_min = [0, 0.5]\n"
          "ic.weakzone_segments_x_max = [0.5, 1]\nic.weakzone_segments_depth_max = [1, 0.6]\n")
    h = des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides=ov)
    pls = h.array("plstrain")
    conn = h.array("connectivity").reshape(4, -1)
    c = h.array("coord").reshape(3, -1)[:, conn].mean(axis=1)
    lx, lz, res = 40e3, 8e3, 2e3
    inc = np.deg2rad(60.0)
    seg1 = (c[0] < 0.5 * lx) & (np.abs(c[0] - 0.3 * lx) < 1.2 * res)                       # vertical plane
    d2 = -np.sin(inc) * (c[0] - 0.7 * lx) - np.cos(inc) * (c[2] + 0.5 * lz)
    seg2 = (c[0] > 0.5 * lx) & (c[2] > -0.6 * lz) & (np.abs(d2) < 1.2 * res)
    expect = np.where(seg1 | seg2, 0.5, 0.0)
    # centroids exactly on a bound are decided by rounding: compare away from the bounds
    sure = (np.abs(np.abs(c[0] - 0.3 * lx) - 1.2 * res) > 1) & (np.abs(np.abs(d2) - 1.2 * res) > 1)
    assert np.array_equal(pls[sure], expect[sure]) and pls.max() == 0.5 and (pls > 0).sum() > 20


@pytest.mark.skipif(not os.path.isdir("/root/reference/benchmarks-cores"), reason="reference tree only in the build container")
def test_every_reference_cfg_is_accepted_or_refused_with_a_reference_exit_code(monkeypatch):
    """All .cfg files the reference ships: the front-end either builds the model or stops with
    one of the reference's exit codes -- 10 for the three stale files whose options input.cxx no
    longer declares (the reference stops on them too), 30 for 2-D-only settings, 31 for features
    outside the offloaded path (TetGen meshing without a mesh file, PT loop, terrigenous surface
    processes, Exodus ...); never a crash, never a silent acceptance of an unbuilt feature."""
    import glob
    ref = "/root/reference"
    files = sorted(glob.glob(ref + "/examples/*.cfg") + glob.glob(ref + "/examples/*/*.cfg") +
                   glob.glob(ref + "/benchmarks-cores/*.cfg") + glob.glob(ref + "/*.cfg"))
    assert len(files) >= 30
    outcome = {}
    for f in files:
        monkeypatch.chdir(os.path.dirname(f))
        try:
            des.Host(cfg_path=f)
            outcome[os.path.basename(f)] = 0
        except des.DesError as e:
            outcome[os.path.basename(f)] = e.code
    assert set(outcome.values()) <= {0, 10, 30, 31}, outcome
    assert outcome["test-3d-equ-long.cfg"] == 0 and outcome["test-3d-equ-tiny.cfg"] == 0
    assert sorted(k for k, v in outcome.items() if v == 10) == ["core-complex-mmg.cfg", "test-3d-equ-big.cfg", "test_gospl_coupling.cfg"]
    assert outcome["kenner_and_segall.cfg"] == 31 and outcome["test-rect-tiny.cfg"] == 30
