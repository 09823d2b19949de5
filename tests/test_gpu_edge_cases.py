"""Edge cases of the path: the smallest mesh the mesher makes (one hexahedral cell = 5 tets, 8
nodes: every node on the boundary, every kernel with one partly filled wavefront), the largest
material count the parameter block carries (DES_MAX_MAT = 16, with a 16-layer model so that
elements with mixed markers exist), zero steps, and a mesh whose element count is one more than
a multiple of the workgroup size."""
import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from oracle_binding import OracleEngine, portable_libm
from test_gpu_parity import STATE

pytestmark = pytest.mark.gpu


def same(dev, ora, note=""):
    for f in STATE:
        a, b = dev.download(f), ora.download(f)
        assert np.array_equal(a, b), "%s: %d of %d entries differ %s" % (f, int((a != b).sum()), a.size, note)


def pair(host):
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    return dev, ora


@pytest.mark.parametrize("rheol", ["elasto-plastic", "elasto-visco-plastic"])
def test_one_cell_mesh(rheol):
    kw = dict(cfgs.EVP, rheol=rheol, lx=2e3, ly=2e3, lz=2e3, res=2e3)
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**kw))
        assert (host.nnode, host.nelem) == (8, 5)
        dev, ora = pair(host)
        sd, so = dev.step(0), ora.step(0)
        assert (sd.dt, sd.steps, sd.time) == (so.dt, 0, 0.0)
        for n in (1, 9, 30):
            sd, so = dev.step(n), ora.step(n)
            assert (sd.dt, sd.time, sd.steps) == (so.dt, so.time, so.steps)
            same(dev, ora, "after %d steps" % so.steps)
        assert dev.check_nan() == 0


def test_sixteen_materials():
    n = 16
    lst = lambda f: "[" + ", ".join(f(i) for i in range(n)) + "]"
    mat = "\n".join([
        "num_materials = %d" % n,
        "rho0 = " + lst(lambda i: "%g" % (2600 + 50 * i)),
        "alpha = [3e-5]",
        "bulk_modulus = " + lst(lambda i: "%g" % (50e9 + 4e9 * i)),
        "shear_modulus = " + lst(lambda i: "%g" % (30e9 + 2e9 * i)),
        "visc_exponent = " + lst(lambda i: "%g" % (3.0 + 0.05 * i)),
        "visc_coefficient = " + lst(lambda i: "%g" % (1.25e-1 * (1 + i))),
        "visc_activation_energy = " + lst(lambda i: "%g" % (2.76e5 + 1e4 * i)),
        "pls0 = [0]", "pls1 = " + lst(lambda i: "%g" % (0.1 + 0.02 * i)),
        "cohesion0 = [4.4e7]", "cohesion1 = " + lst(lambda i: "%g" % (4e6 + 1e5 * i)),
        "friction_angle0 = [30]", "friction_angle1 = " + lst(lambda i: "%g" % (30 - i)),
        "min_viscosity = 1e19", ""])
    ic = ("oceanic_plate_age_in_yr = 2e5\nmattype_option = 1\nnum_mattype_layers = %d\nlayer_mattypes = %s\n"
          "mattype_layer_depths = %s\n" % (n, lst(str), "[" + ", ".join("%g" % ((i + 1) / n) for i in range(n - 1)) + "]"))
    cfg = cfgs.BASE.format(rheol="elasto-visco-plastic", lx=40e3, ly=8e3, lz=8e3, res=2e3, spo=1, qcsi=100, vx0=-1e-9, vx1=1e-9,
                           tmantle=1573, control="", bc="", ic=ic, mat=mat, water="no")
    with portable_libm():
        host = des.Host(cfg_text=cfg)
        assert host.params.nmat == 16
        mk = host.array("elemmarkers").reshape(-1, 16)
        assert ((mk > 0).sum(axis=1) > 1).any() and (mk.sum(axis=0) > 0).all()     # mixed elements, every material present
        dev, ora = pair(host)
        dev.step(60); ora.step(60)
        same(dev, ora)


def test_element_count_one_past_a_workgroup_multiple():
    # 13 x 5 x 4 cells: 1300 tets = 5 * 256 + 20
    kw = dict(cfgs.YIELD, lx=26e3, ly=10e3, lz=8e3, res=2e3)
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**kw))
        assert host.nelem % 256 not in (0,) and host.nelem == 1300
        dev, ora = pair(host)
        dev.step(80); ora.step(80)
        assert ora.step(0).n_return_mapping > 0
        same(dev, ora)
