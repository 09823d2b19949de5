"""The 2-D (triangle) build on the CPU side: the host library with ndims = 2 (mesh, topology, ICs),
the oracle compiled -DDES_NDIMS=2 (oracle/libdes_oracle2d.so), and the reference's own 2-D case
benchmarks-cores/test-tiny.cfg (BASELINE configs[0]) on the mesh the reference's vendored Triangle
makes for it (tests/golden/test-tiny.desmesh, tests/golden/make_test_tiny_mesh.py).

Nothing pins the 2-D restatement to the reference's binary (its translation units do not build here:
DESIGN.md "Oracle"): what is checked are known answers of the 2-D formulas -- Hooke's law, the Mohr
circle, returns onto the yield surface, triangle areas, boundary normals -- and the run's own
regression hashes.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from dynearthsol_amd import driver
from oracle_binding import OracleEngine, load_oracle, dptr
from test_driver_output import oracle_api, read_frame, in_tmp  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
TINY_MESH = os.path.join(HERE, "golden", "test-tiny.desmesh")
K, G = 50e9, 30e9


def host2d(kw=None, **extra):
    return des.Host(cfg_text=cfgs.make(**(kw or cfgs.EP)), ndims=2, **extra)


# ---- host: mesh, topology ---------------------------------------------------------------------
def test_regular_2d_mesh_has_the_references_counts_and_orientation():
    h = host2d()                                             # 40 km x 8 km at 2 km
    nx, nz = 21, 5
    assert (h.nnode, h.nelem) == (nx * nz, 2 * (nx - 1) * (nz - 1))      # dynearthsol.cxx:127-132
    coord = h.array("coord").reshape(2, -1)
    conn = h.array("connectivity").reshape(3, -1)
    a, b, c = (coord[:, conn[i]] for i in range(3))
    signed = 0.5 * ((b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]))
    assert (signed > 0).all() and signed.sum() == pytest.approx(40e3 * 8e3, rel=1e-13)    # counter-clockwise triangles tile the box
    assert coord[1].max() == 0 and coord[1].min() == -8e3 and coord[0].min() == 0 and coord[0].max() == 40e3
    # renumbering_mesh (mesh.cxx:2696-2821): nodes sorted along x (then z), elements by centroid
    assert (np.diff(coord[0]) >= 0).all()
    seg = h.array("segment").reshape(2, -1)
    flags = h.array("segflag")
    assert len(flags) == 2 * (nx + nz - 2) and sorted(set(flags)) == [1, 2, 16, 32]
    assert seg.min() >= 0 and seg.max() < h.nnode


def test_2d_topology_lists():
    h = host2d()
    m = h.mesh
    nx, nz = 21, 5
    nbf = [m.nbfacets[i] for i in range(10)]
    assert nbf == [nz - 1, nz - 1, 0, 0, nx - 1, nx - 1, 0, 0, 0, 0]
    assert [m.nbnodes[i] for i in range(10)] == [nz, nz, 0, 0, nx, nx, 0, 0, 0, 0]
    assert (m.ntop, m.etop) == (nx, nx - 1)
    coord = h.array("coord").reshape(2, -1)
    conn = h.array("connectivity").reshape(3, -1)
    bc = np.ctypeslib.as_array(m.bcflag, shape=(h.nnode,))
    assert ((bc & 1) != 0).sum() == nz and (coord[0][(bc & 1) != 0] == 0).all()
    assert ((bc & 32) != 0).sum() == nx and (coord[1][(bc & 32) != 0] == 0).all()
    # every boundary facet (element, edge) lies on its boundary: NODE_OF_FACET of the 2-D build (constants.hpp:71-75)
    nof = [(1, 2), (2, 0), (0, 1)]
    for ib, bit in ((0, 1), (1, 2), (4, 16), (5, 32)):
        el = np.ctypeslib.as_array(m.bfacet_elem[ib], shape=(nbf[ib],))
        fa = np.ctypeslib.as_array(m.bfacet_facet[ib], shape=(nbf[ib],))
        assert (np.diff(el) >= 0).all()                                   # sorted by element (mesh.cxx:3238-3244)
        for e, f in zip(el, fa):
            n0, n1 = conn[nof[f][0], e], conn[nof[f][1], e]
            assert bc[n0] & bit and bc[n1] & bit
    # outward unit normals (bc.cxx:42-50, 94-150): x0 -> (-1, 0), x1 -> (1, 0), z0 -> (0, -1), z1 -> (0, 1)
    bn = np.ctypeslib.as_array(m.bnormals, shape=(2, 10))
    for ib, n in ((0, (-1, 0)), (1, (1, 0)), (4, (0, -1)), (5, (0, 1))):
        assert tuple(bn[:, ib]) == n
    # support CSR: every (element, local node) incidence once, elements ascending per node
    idx = np.ctypeslib.as_array(m.support_idx, shape=(h.nnode + 1,))
    arr = np.ctypeslib.as_array(m.support_arr, shape=(3 * h.nelem,))
    lid = np.ctypeslib.as_array(m.support_lidx, shape=(3 * h.nelem,))
    assert idx[-1] == 3 * h.nelem
    for n in range(h.nnode):
        es, ls = arr[idx[n]:idx[n + 1]], lid[idx[n]:idx[n + 1]]
        assert (np.diff(es) > 0).all() and (conn[ls, es] == n).all()
    # the top nodes are sorted by x: simple_diffusion walks consecutive pairs (bc.cxx:1021-1033)
    top = np.ctypeslib.as_array(m.top_nodes, shape=(m.ntop,))
    assert (np.diff(coord[0][top]) > 0).all()


def test_2d_initial_conditions_have_2d_shapes():
    h = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, nmat=2)), ndims=2)
    nn, ne = h.nnode, h.nelem
    assert h.array("vel").size == 2 * nn and h.array("stress").size == 3 * ne and h.array("strain").size == 3 * ne
    assert h.array("stressyy").size == ne and h.array("markerset.eta").size == 3 * 4 * ne     # NODES_PER_ELEM shape functions
    eta = h.array("markerset.eta").reshape(3, -1)
    assert np.allclose(eta.sum(0), 1) and (eta >= 0).all()
    st = h.array("stress").reshape(3, -1)
    assert (st[0] == st[1]).all() and (st[0] < 0).all() and (st[2] == 0).all()      # lithostatic, ic.cxx:322-362
    em = h.array("elemmarkers").reshape(ne, 2)
    assert (em.sum(1) == 4).all() and em[:, 0].sum() > 0 and em[:, 1].sum() > 0     # two layers


def test_what_the_2d_host_does_not_build_is_refused():
    with pytest.raises(des.DesError) as ei:
        des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides="mesh.meshing_elem_shape = 2\n")        # 2-D only (input.cxx:1072-1076)
    assert ei.value.code == 30
    with pytest.raises(des.DesError) as ei:
        des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides="mesh.meshing_elem_shape = 0\n", ndims=2)   # needs Triangle
    assert ei.value.code == 31
    with pytest.raises(des.DesError) as ei:
        des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides="bc.vbc_z0 = 5\nbc.has_winkler_foundation = no\n", ndims=2)
    assert ei.value.code == 11                                            # input.cxx:1272-1277: 0..4 in 2-D
    des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides="bc.vbc_z0 = 4\nbc.has_winkler_foundation = no\n", ndims=2)
    with pytest.raises(des.DesError) as ei:
        des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides="bc.vbc_z0 = 4\nbc.has_winkler_foundation = no\n")       # 0..3 in 3-D
    assert ei.value.code == 11
    with pytest.raises(des.DesError) as ei:                               # a 3-D mesh file is not a 2-D mesh
        des.Host(cfg_text=cfgs.TEST_TINY, mesh_file=os.path.join(HERE, "golden", "test-3d.desmesh"), ndims=2)
    assert ei.value.code == 30


def test_2d_mesh_is_cut_into_node_slabs_like_the_3d_one():
    """host/partition.cpp on triangles: owned node ranges tile the mesh, every element is owned once, the local top
    nodes stay a run of the sorted surface (simple_diffusion walks consecutive pairs, bc.cxx:1021-1033)."""
    import dynearthsol_amd.decomp as decomp
    h = host2d()
    parts = [decomp.Partition(h, 3, r) for r in range(3)]
    assert sum(p.owned[1] - p.owned[0] for p in parts) == h.nnode
    assert sum(int(p.elem_owned.sum()) for p in parts) == h.nelem
    gx = h.array("coord").reshape(2, -1)[0]
    for p in parts:
        assert (p.ndims, p.node_width, p.elem_width) == (2, 6, 8)
        assert p.mesh.etop == p.mesh.ntop - 1
        top = np.ctypeslib.as_array(p.mesh.top_nodes, shape=(p.mesh.ntop,))
        x = gx[p.l2g_node[top]]
        assert (np.diff(x) > 0).all()
        assert p.local("coord").size == 2 * p.nnode and p.local("stressyy").size == p.nelem
        assert [q for q in p.nbr_rank] == [r for r in (p.rank - 1, p.rank + 1) if 0 <= r < 3]


# ---- oracle: the 2-D constitutive formulas ----------------------------------------------------
def mc_params(coh=4.4e7, phi=30.0, psi=0.0, tension_max=1e9):
    sphi, spsi = np.sin(np.radians(phi)), np.sin(np.radians(psi))
    anphi, anpsi = (1 + sphi) / (1 - sphi), (1 + spsi) / (1 - spsi)
    return 2 * coh * np.sqrt(anphi), anphi, anpsi, min(tension_max, coh / np.tan(np.radians(phi)))


def ep2(s, de, plane_strain_syy=None, hardn=0.0, **kw):
    lib = load_oracle(ndims=2)
    amc, anphi, anpsi, ten_max = mc_params(**kw)
    s, de = np.array(s, dtype=np.float64), np.array(de, dtype=np.float64)
    fm = C.c_int(0)
    d3 = C.POINTER(C.c_double)
    if plane_strain_syy is None:
        lib.des_oracle_elasto_plastic.restype = C.c_double
        lib.des_oracle_elasto_plastic.argtypes = [C.c_double] * 7 + [d3, d3, C.POINTER(C.c_int)]
        depls = lib.des_oracle_elasto_plastic(K, G, amc, anphi, anpsi, hardn, ten_max, dptr(de), dptr(s), C.byref(fm))
        return s, depls, fm.value, (amc, anphi, anpsi, ten_max)
    syy = C.c_double(plane_strain_syy)
    lib.des_oracle_elasto_plastic2d.restype = C.c_double
    lib.des_oracle_elasto_plastic2d.argtypes = [C.c_double] * 7 + [d3, d3, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    depls = lib.des_oracle_elasto_plastic2d(K, G, amc, anphi, anpsi, hardn, ten_max, dptr(de), dptr(s), C.byref(syy), C.byref(fm))
    return s, syy.value, depls, fm.value, (amc, anphi, anpsi, ten_max)


def principal2(s):
    return np.linalg.eigvalsh(np.array([[s[0], s[2]], [s[2], s[1]]]))


def test_2d_below_yield_is_hookes_law():
    s0, de = np.array([-2e8, -2.1e8, 1e6]), np.array([1e-6, -2e-6, 3e-7])
    s, depls, fm, _ = ep2(s0, de)
    lam = K - 2.0 / 3 * G
    ref = s0.copy()
    ref[:2] += 2 * G * de[:2] + lam * (de[0] + de[1])
    ref[2] += 2 * G * de[2]
    assert depls == 0 and fm == 0 and np.array_equal(s, ref)


def test_2d_shear_return_lands_on_the_mohr_coulomb_surface_and_keeps_the_principal_axes():
    rng = np.random.default_rng(5)
    for _ in range(200):
        s0 = np.array([-rng.uniform(5e8, 1.5e9), -rng.uniform(5e7, 3e8), rng.uniform(-2e8, 2e8)])
        s, depls, fm, (amc, anphi, anpsi, ten_max) = ep2(s0, np.zeros(3), psi=float(rng.choice([0.0, 10.0])))
        p0 = principal2(s0)
        if p0[0] - p0[1] * anphi + amc > 0:
            assert fm == 0 and np.array_equal(s, s0)
            continue
        assert fm == 10 and depls > 0
        p = principal2(s)
        assert abs(p[0] - p[1] * anphi + amc) <= 1e-9 * abs(p).max()         # fs = 0 (rheology.cxx:371-372)
        # the Mohr circle keeps its orientation: same principal directions before and after
        th0 = 0.5 * np.arctan2(2 * s0[2], s0[0] - s0[1])
        th = 0.5 * np.arctan2(2 * s[2], s[0] - s[1])
        assert th == pytest.approx(th0, abs=1e-9)


def test_2d_tensile_return_caps_the_larger_principal_stress():
    s0 = np.array([7.5e7, 9e7, 1e5])              # both principal stresses near the tension cut-off: h >= 0 (rheology.cxx:381-386)
    s, depls, fm, (amc, anphi, anpsi, ten_max) = ep2(s0, np.zeros(3))
    assert fm == 1 and depls > 0
    assert principal2(s)[1] == pytest.approx(ten_max, rel=1e-12)


def test_plane_strain_law_agrees_with_hooke_below_yield_and_tracks_the_out_of_plane_stress():
    s0, de = np.array([-2e8, -2.1e8, 1e6]), np.array([1e-6, -2e-6, 3e-7])
    s, syy, depls, fm, _ = ep2(s0, de, plane_strain_syy=-2.05e8)
    lam = K - 2.0 / 3 * G
    ref = s0.copy()
    ref[:2] += 2 * G * de[:2] + lam * (de[0] + de[1])
    ref[2] += 2 * G * de[2]
    assert depls == 0 and fm == 0
    assert np.allclose(s, ref, rtol=1e-14, atol=0)
    assert syy == pytest.approx(-2.05e8 + lam * (de[0] + de[1]), rel=1e-15)      # rheology.cxx:520


def test_plane_strain_shear_return_uses_the_extreme_pair_of_three_principal_stresses():
    rng = np.random.default_rng(7)
    seen, checked = set(), 0
    for _ in range(300):
        s0 = np.array([-rng.uniform(2e8, 1.5e9), -rng.uniform(2e8, 1.5e9), rng.uniform(-1e8, 1e8)])
        syy0 = -rng.uniform(1e8, 1.6e9)
        s, syy, depls, fm, (amc, anphi, anpsi, ten_max) = ep2(s0, np.zeros(3), plane_strain_syy=syy0)
        v0 = np.append(principal2(s0), syy0)                   # {s_I, s_II, s_yy} before
        role = np.argsort(v0)                                  # who is the minor / intermediate / major stress
        if v0[role[0]] - v0[role[2]] * anphi + amc >= 0:
            assert fm == 0 and depls == 0
            continue
        assert fm == 10 and depls > 0
        v = np.append(principal2(s), syy)
        # the law corrects the (minor, major) pair it chose BEFORE the return onto the surface and does not
        # re-sort afterwards (geoFLAC's form): some pair of the three values satisfies fs = 0, the third is untouched
        # (zero dilation: a2 - a2 anpsi = 0, rheology.cxx:640-645)
        fs = min(abs(v[i] - v[j] * anphi + amc) for i in range(3) for j in range(3) if i != j)
        assert fs <= 1e-9 * abs(v).max()
        assert min(abs(v - v0[role[1]])) <= 1e-13 * abs(v0[role[1]])
        seen.add(int(role[1]))
        checked += 1
    assert seen == {0, 1, 2} and checked > 50                  # all three orderings (rheology.cxx:553-582)


# ---- oracle: a whole step ----------------------------------------------------------------------
def test_2d_step_geometry_known_answers():
    h = host2d(dict(cfgs.EVP, res=1e3))
    o = OracleEngine(h)
    o.init_from_host(h)
    vol = o.download("VOLUME")
    assert vol.sum() == pytest.approx(40e3 * 8e3, rel=1e-13) and np.allclose(vol, 0.5e6)
    assert o.download("VOLUME_N").sum() == pytest.approx(3 * vol.sum(), rel=1e-13)       # every triangle counted at its three nodes
    m = o.download("MASS")
    assert (m > 0).all()
    sc = o.step(20)
    assert sc.steps == 20 and sc.status == 0 and o.check_nan() == 0
    # divergence-free corner: strain rate of a uniform stretch v = (a x, 0) is (a, 0, 0) in every element
    nn = h.nnode
    coord = o.download("COORD").reshape(2, -1)
    a = 1e-14
    v = np.zeros((2, nn))
    v[0] = a * coord[0]
    o.upload("VEL", v.ravel())
    lib = load_oracle(ndims=2)
    # one more step would move the mesh; the strain-rate kernel alone is not exported, so take it from the
    # stress update's input: strain_rate is written before anything else reads the new velocities
    o.step(1)
    sr = o.download("STRAIN_RATE").reshape(3, -1)
    # after the anti-locking correction (rheology.cxx:786-793) the trace is redistributed, the shear part stays 0
    assert np.abs(sr[2]).max() < 1e-6 * a
    assert (sr[0] + sr[1]) == pytest.approx(np.full(h.nelem, a), rel=0.2)


def test_reference_2d_case_test_tiny_runs_and_keeps_its_hashes():
    """benchmarks-cores/test-tiny.cfg on its Triangle mesh: four steps of the 2-D oracle.  The hashes
    are this oracle's own (tests/golden/test_tiny_hashes.json, written by this test when absent): a
    regression net under the device parity test, not a pin to the reference binary."""
    h = des.Host(cfg_text=cfgs.TEST_TINY, mesh_file=TINY_MESH, ndims=2)
    assert (h.nnode, h.nelem, h.mesh.ntop, h.mesh.etop) == (97, 164, 9, 8)
    assert h.params.nmat == 8 and h.params.rheol_type == 7 and h.params.quality_check_step_interval == 2
    o = OracleEngine(h)
    dt = o.init_from_host(h)
    # quasi-static: dt_elastic = 0.5 * minl / (max_vbc_val * inertial_scaling) governs (geometry.cxx:1618-1632);
    # the smallest element height is a fraction of the 10-km resolution
    minl = dt * 1e-10 * h.cfg_double("control.inertial_scaling") / (0.5 * h.cfg_double("control.dt_fraction"))
    assert 1e3 < minl < 1e4
    sc = o.step(4)
    assert sc.steps == 4 and sc.status == 0 and o.check_nan() == 0 and sc.n_return_mapping == 164
    got = {"dt": repr(dt)}
    for f in ("COORD", "VEL", "STRESS", "STRAIN", "PLSTRAIN", "TEMPERATURE", "VISCOSITY"):
        got[f] = hashlib.sha256(o.download(f).tobytes()).hexdigest()
    path = os.path.join(HERE, "golden", "test_tiny_hashes.json")
    if not os.path.exists(path):
        json.dump(got, open(path, "w"), indent=1, sort_keys=True)
    assert got == json.load(open(path))


def test_driver_loop_writes_2d_frames(in_tmp):
    """des_run over the 2-D oracle on test-tiny.cfg: a frame every step, the reference's 2-D file layout
    (header ndims=2, coordinate [nnode][2], connectivity [nelem][3], stress [nelem][3]; binaryio.cxx:39-40,
    output.cxx:300-384)."""
    host = des.Host(cfg_text=cfgs.TEST_TINY, mesh_file=TINY_MESH, ndims=2, overrides="sim.modelname = tiny\n")
    st = driver.run(host, api=oracle_api(2))
    assert (st.steps, st.frames, st.exit_code) == (4, 5, 0)
    fr = read_frame("tiny.save.000004", ndims=2)
    nn, ne = 97, 164
    assert fr["coordinate"].size == 2 * nn * 8 and fr["connectivity"].size == 3 * ne * 4
    assert fr["stress"].size == 3 * ne * 8 and fr["velocity"].size == 2 * nn * 8 and fr["strain-rate"].size == 3 * ne * 8
    assert fr["markerset.eta"].size == 3 * 4 * ne * 8 and fr["markerset.coord"].size == 2 * 4 * ne * 8
    assert fr["steps"].view(np.int32)[0] == 4
    q = fr["mesh quality"].view(np.float64)
    assert q.size == ne and (q > 0.3).all() and (q <= 1 + 1e-12).all()           # elem_quality, geometry.cxx:1901-1906
    # the frame holds what a straight oracle run holds after four steps
    o = OracleEngine(host)
    o.init_from_host(host)
    o.step(4)
    assert np.array_equal(fr["stress"].view(np.float64).reshape(ne, 3).T.ravel(), o.download("STRESS"))
    assert np.array_equal(fr["coordinate"].view(np.float64).reshape(nn, 2).T.ravel(), o.download("COORD"))


def test_2d_restart_continues_bit_for_bit(in_tmp):
    """restart() (dynearthsol.cxx:231-435) of a plane-strain 2-D model from our own frame + checkpoint
    (stressyy travels in the checkpoint, output.cxx:394-395 / dynearthsol.cxx:380-381): 20 steps, restart,
    20 more == 40 straight, to the bit."""
    base = ("sim.max_steps = 40\nsim.output_step_interval = 20\nsim.checkpoint_frame_interval = 1\n"
            "mesh.quality_check_step_interval = 10\nsim.is_outputting_averaged_fields = yes\nmat.is_plane_strain = yes\n")
    kw = dict(cfgs.EVP, nmat=2, res=1e3)
    driver.run(des.Host(cfg_text=cfgs.make(**kw), overrides=base + "sim.modelname = a\n", ndims=2), api=oracle_api(2))
    hb = des.Host(cfg_text=cfgs.make(**kw), ndims=2, overrides=base + "sim.modelname = b\nsim.is_restarting = yes\n"
                  "sim.restarting_from_modelname = a\nsim.restarting_from_frame = 1\n")
    assert np.abs(hb.array("stressyy")).max() > 0
    st = driver.run(hb, api=oracle_api(2))
    assert (st.steps, st.frames) == (40, 2)
    a, b = read_frame("a.save.000002", ndims=2), read_frame("b.save.000002", ndims=2)
    assert sorted(a) == sorted(b)
    for name in a:
        if name != "walltime_sec":
            assert np.array_equal(a[name], b[name]), name
    ca, cb = read_frame("a.chkpt.000002", ndims=2), read_frame("b.chkpt.000002", ndims=2)
    assert "stressyy" in ca
    for name in ca:
        assert np.array_equal(ca[name], cb[name]), name
    # a 3-D host refuses the 2-D files by their header (binaryio.cxx:230-238)
    with pytest.raises(des.DesError) as ei:
        des.Host(cfg_text=cfgs.make(**kw), overrides=base + "sim.is_restarting = yes\nsim.restarting_from_modelname = a\n"
                 "sim.restarting_from_frame = 1\n")
    assert ei.value.code == 22


REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Dynearthsol.py")),
                    reason="the reference's reader is only present in the build container")
def test_reference_reader_reads_our_2d_frames(in_tmp):
    """The reference's own Dynearthsol.py on the frames of a 2-D run: header, shapes and values."""
    import sys
    host = des.Host(cfg_text=cfgs.TEST_TINY, mesh_file=TINY_MESH, ndims=2, overrides="sim.modelname = tiny\n")
    driver.run(host, api=oracle_api(2))
    sys.path.insert(0, REF)
    try:
        import Dynearthsol as refpy
    finally:
        sys.path.remove(REF)
    d = refpy.Dynearthsol("tiny")
    assert d.ndims == 2 and d.revision == 4 and d.frames == [0, 1, 2, 3, 4] and d.steps == [0, 1, 2, 3, 4]
    o = OracleEngine(host)
    o.init_from_host(host)
    o.step(4)
    nn, ne = host.nnode, host.nelem
    assert d.read_field(4, "coordinate").shape == (nn, 2) and d.read_field(4, "connectivity").shape == (ne, 3)
    assert d.read_field(4, "stress").shape == (ne, 3) and d.read_field(4, "velocity").shape == (nn, 2)
    assert np.array_equal(d.read_field(4, "coordinate").T.ravel(), o.download("COORD"))
    assert np.array_equal(d.read_field(4, "connectivity").T.ravel(), host.array("connectivity"))
    assert np.array_equal(d.read_field(4, "stress").T.ravel(), o.download("STRESS"))
    assert np.array_equal(d.read_field(4, "temperature"), o.download("TEMPERATURE"))
    mk = d.read_markers(4, "markerset")
    assert mk["size"] == 4 * ne and mk["markerset.eta"].shape == (4 * ne, 3) and mk["markerset.coord"].shape == (4 * ne, 2)
    coord, conn = d.read_field(4, "coordinate"), d.read_field(4, "connectivity")
    expect = np.einsum("mkd,mk->md", coord[conn[mk["markerset.elem"]]], mk["markerset.eta"])
    assert np.allclose(mk["markerset.coord"], expect, rtol=1e-14, atol=1e-6)


def test_equilateral_2d_mesher_of_test_rect_tiny():
    """mesh.meshing_elem_shape = 2 (new_mesh_regular_equilateral, mesh.cxx:578-684): benchmarks-cores/test-rect-tiny.cfg
    is meshed by the host itself -- the reference's counts, counter-clockwise triangles that tile the box, flat
    boundaries, equilateral interior rows; then four steps of the oracle."""
    h = des.Host(cfg_text=cfgs.TEST_TINY, overrides=cfgs.TEST_RECT_TINY_OVERRIDES, ndims=2)
    Lx, Lz, res = 150e3, 50e3, 1e4
    nx = int((Lx / 2 - 0.5 * res) / res) * 2 + 2                       # mesh.cxx:660-665
    nz = int(Lz * 2 / np.sqrt(3.0) / res) + 1
    assert (nx, nz) == (16, 6)
    assert h.nnode == nx * ((nz + 1) // 2) + (nx + 1) * (nz // 2) == 99
    assert h.nelem == (2 * nx - 1) * (nz - 1) == 155
    m = h.mesh
    assert [m.nbfacets[i] for i in range(10)] == [nz - 1, nz - 1, 0, 0, nx, nx - 1, 0, 0, 0, 0]     # nz even: the last row has nx + 1 nodes
    assert (m.ntop, m.etop) == (nx, nx - 1)
    coord = h.array("coord").reshape(2, -1)
    conn = h.array("connectivity").reshape(3, -1)
    a, b, c = (coord[:, conn[i]] for i in range(3))
    area = 0.5 * ((b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]))
    assert (area > 0).all() and area.sum() == pytest.approx(Lx * Lz, rel=1e-13)
    bc = np.ctypeslib.as_array(m.bcflag, shape=(h.nnode,))
    assert (coord[0][(bc & 1) != 0] == 0).all() and (coord[0][(bc & 2) != 0] == Lx).all()
    assert (coord[1][(bc & 32) != 0] == 0).all() and (coord[1][(bc & 16) != 0] == -Lz).all()
    # interior triangles of the upper strips are equilateral with side = resolution
    side = np.stack([np.hypot(*(b - a)), np.hypot(*(c - b)), np.hypot(*(a - c))])
    equil = np.isclose(side, res, rtol=1e-12).all(axis=0)
    assert equil.sum() > 0.5 * h.nelem
    o = OracleEngine(h)
    o.init_from_host(h)
    sc = o.step(4)
    assert sc.steps == 4 and sc.status == 0 and o.check_nan() == 0
