"""Small .cfg texts used by the tests (regular meshes: built by the host library itself,
no TetGen needed)."""

BASE = """
[sim]
modelname = t
max_steps = 100
output_step_interval = 100
is_outputting_averaged_fields = no
[mesh]
meshing_option = 1
meshing_elem_shape = 1
xlength = {lx}
ylength = {ly}
zlength = {lz}
resolution = {res}
quality_check_step_interval = {qcsi}
[control]
surface_process_option = {spo}
surface_diffusivity = 1e-6
inertial_scaling = 1e4
{control}
[bc]
vbc_x0 = 1
vbc_x1 = 1
vbc_val_x0 = {vx0}
vbc_val_x1 = {vx1}
vbc_y0 = 1
vbc_y1 = 1
has_water_loading = {water}
surface_temperature = 273
mantle_temperature = {tmantle}
{bc}
[ic]
weakzone_option = 1
weakzone_azimuth = 15
weakzone_inclination = -60
weakzone_halfwidth = 1.2
weakzone_depth_min = 0.5
weakzone_depth_max = 1.0
weakzone_xcenter = 0.5
weakzone_ycenter = 0.5
weakzone_zcenter = 0
weakzone_plstrain = 0.5
{ic}
[mat]
rheology_type = {rheol}
{mat}
"""

MAT_1 = """
rho0 = [2700]
alpha = [{alpha}]
bulk_modulus = [50e9]
shear_modulus = [30e9]
pls0 = [0]
pls1 = [0.5]
cohesion0 = [4.4e7]
cohesion1 = [4e6]
friction_angle0 = [30]
friction_angle1 = [30]
min_viscosity = {vmin}
"""

MAT_2 = """
num_materials = 2
rho0 = [2700, 3300]
alpha = [3e-5]
bulk_modulus = [50e9, 120e9]
shear_modulus = [30e9, 70e9]
visc_exponent = [3.05, 3.5]
visc_coefficient = [1.25e-1, 1.1e5]
visc_activation_energy = [2.76e5, 5.3e5]
pls0 = [0]
pls1 = [0.5, 0.1]
cohesion0 = [4.4e7]
cohesion1 = [4e6]
friction_angle0 = [30]
friction_angle1 = [30, 15]
min_viscosity = {vmin}
"""


def make(rheol="elasto-plastic", lx=40e3, ly=8e3, lz=8e3, res=2e3, spo=1, qcsi=100, vx=1e-9,
         tmantle=273, alpha=0, vmin="1e24", nmat=1, control="", bc="", ic="", water="no", mat_extra=""):
    mat = (MAT_1 if nmat == 1 else MAT_2).format(alpha=alpha, vmin=vmin) + mat_extra
    if nmat == 2:
        ic += "\nmattype_option = 1\nnum_mattype_layers = 2\nlayer_mattypes = [0,1]\nmattype_layer_depths = [0.5]\n"
    return BASE.format(rheol=rheol, lx=lx, ly=ly, lz=lz, res=res, spo=spo, qcsi=qcsi, vx0=-vx, vx1=vx,
                       tmantle=tmantle, control=control, bc=bc, ic=ic, mat=mat, water=water)


# the reference's benchmarks-cores/test-3d.cfg physics on a regular mesh
EP = dict(rheol="elasto-plastic")
# visco-elasto-plastic with a real geotherm (thermal expansion on, creep active)
EVP = dict(rheol="elasto-visco-plastic", tmantle=1573, alpha=3e-5, vmin="1e19",
           ic="oceanic_plate_age_in_yr = 2e5\n")
# fast loading so that elements yield (shear and tensile return mapping) within ~100 steps
YIELD = dict(rheol="elasto-plastic", vx=1e-6, control="characteristic_speed = 1e-9\n")


# Parameter values of the reference's benchmarks-cores/test-3d.cfg (BASELINE configs[1]); the
# mesh itself (meshing_option = 2 needs TetGen) comes from tests/golden/test-3d.desmesh.
TEST3D = """
[sim]
modelname = benchmark
max_time_in_yr = 200
output_time_interval_in_yr = 50
output_step_interval = 100
has_output_during_remeshing = no
is_outputting_averaged_fields = no
checkpoint_frame_interval = 5
[mesh]
meshing_option = 2
xlength = 100e3
ylength = 10e3
zlength = 10e3
resolution = 1e3
largest_size = 10
smallest_size = 0.001
refined_zonex = [0.3, 0.7]
refined_zoney = [0.0, 1.0]
refined_zonez = [0.0, 1.0]
quality_check_step_interval = 100
min_quality = 0.2
max_boundary_distortion = 0.2
remeshing_option = 11
[control]
surface_process_option = 1
surface_diffusivity = 1e-6
dt_fraction = 1.0
inertial_scaling = 1e4
[bc]
vbc_x0 = 1
vbc_x1 = 1
vbc_val_x0 = -1e-9
vbc_val_x1 = 1e-9
vbc_y0 = 1
vbc_y1 = 1
vbc_val_y0 = 0
vbc_val_y1 = 0
has_water_loading = no
surface_temperature = 273
mantle_temperature = 273
[ic]
weakzone_option = 1
weakzone_azimuth = 15
weakzone_inclination = -60
weakzone_halfwidth = 1.2
weakzone_depth_min = 0.5
weakzone_depth_max = 1.0
weakzone_xcenter = 0.5
weakzone_ycenter = 0.5
weakzone_zcenter = 0
weakzone_plstrain = 0.5
[mat]
rheology_type = elasto-plastic
rho0 = [2700]
alpha = [0]
bulk_modulus = [50e9]
shear_modulus = [30e9]
pls0 = [0]
pls1 = [0.5]
cohesion0 = [4.4e7]
cohesion1 = [4e6]
friction_angle0 = [30]
friction_angle1 = [30]
min_viscosity = 1e24
"""


def apply_overrides(text, overrides):
    """Rewrite "section.key = value" lines into a .cfg text (for the executable, which takes
    the reference's single config-file argument)."""
    import re
    for line in overrides.strip().splitlines():
        key, val = [x.strip() for x in line.split("=", 1)]
        sec, k = key.split(".")
        if re.search(r"^%s = " % re.escape(k), text, flags=re.M):
            text = re.sub(r"(?m)^%s = .*$" % re.escape(k), "%s = %s" % (k, val), text)
        else:
            text = text.replace("[%s]\n" % sec, "[%s]\n%s = %s\n" % (sec, k, val), 1)
    return text


# Parameter values of the reference's examples/oblique-rift-3d.cfg (BASELINE configs[4]: two
# materials, oblique extension through vbc type 6, PREM reference pressure, dt_fraction 0.5, no
# surface process); the mesh (meshing_option = 2: TetGen) comes from
# tests/golden/oblique-rift-3d.desmesh (676 nodes / 2,991 tets).
OBLIQUE = """
[sim]
modelname = result
max_time_in_yr = 0.1e6
output_time_interval_in_yr = 50000
is_outputting_averaged_fields = no
[mesh]
meshing_option = 2
xlength = 200e3
ylength = 100e3
zlength = 50e3
resolution = 5e3
smallest_size = 0.01
refined_zonex = [0.3, 0.7]
refined_zoney = [0.3, 0.7]
refined_zonez = [0.7, 1.0]
quality_check_step_interval = 500
remeshing_option = 11
[control]
ref_pressure_option = 1
dt_fraction = 0.5
[bc]
vbc_x0 = 6
vbc_x1 = 6
vbc_val_x0 = -3.17e-10
vbc_val_x1 = 3.17e-10
vbc_val_x0_l = 1.59e-10
vbc_val_x1_l = -1.59e-10
vbc_y0 = 1
vbc_y1 = 1
vbc_val_y0 = 0
vbc_val_y1 = 0
vbc_n0 = 1
vbc_val_n0 = 0
has_water_loading = no
surface_temperature = 273
mantle_temperature = 1573
[ic]
weakzone_option = 1
weakzone_azimuth = 0
weakzone_inclination = 60
weakzone_halfwidth = 1.5
weakzone_depth_min = 0.0
weakzone_depth_max = 1.0
weakzone_xcenter = 0.5
weakzone_ycenter = 0.5
weakzone_zcenter = 0.5
weakzone_plstrain = 0.5
oceanic_plate_age_in_yr = 60e6
[mat]
rheology_type = elasto-visco-plastic
num_materials = 2
rho0 = [2800, 3300]
alpha = [3e-5]
bulk_modulus = [50e9]
shear_modulus = [30e9]
visc_exponent = [3.05]
visc_coefficient = [1.25e-1]
visc_activation_energy = [3.76e5]
heat_capacity = [1000]
therm_cond = [3.3]
pls0 = [0]
pls1 = [0.1]
cohesion0 = [4e7]
cohesion1 = [4e6]
friction_angle0 = [30]
friction_angle1 = [30]
dilation_angle0 = [0]
dilation_angle1 = [0]
max_viscosity = 1e24
min_viscosity = 1e19
"""


# Parameter values of the reference's benchmarks-cores/test-3d-equ-tiny.cfg (12,500 tets) -- and,
# through make_equ(long=True), of test-3d-equ-long.cfg (984,375 tets / 209,664 nodes): the
# reference's own regular-mesh 3-D benchmarks (meshing_option = 1, meshing_elem_shape = 1), which
# the host library meshes itself: seven materials in two layers, continental geotherm with
# radiogenic heating (temperature_option = 3), water loading, surface diffusion, evp.
EQU = """
[sim]
modelname = benchmark
{stop}
has_output_during_remeshing = no
is_outputting_averaged_fields = no
checkpoint_frame_interval = {ckpt}
[mesh]
meshing_option = 1
meshing_elem_shape = 1
xlength = 250e3
ylength = {ly}
zlength = 125e3
resolution = {res}
quality_check_step_interval = {qcsi}
min_quality = 0.2
max_boundary_distortion = {mbd}
remeshing_option = 13
[markers]
markers_per_element = 8
[control]
surface_process_option = 1
surface_diffusivity = {sdiff}
dt_fraction = 1.0
inertial_scaling = 1e4
surf_base_level = 15.e3
gravity = 9.81
[bc]
vbc_x0 = 1
vbc_x1 = 1
vbc_val_x0 = -1.57e-10
vbc_val_x1 = 1.57e-10
vbc_y0 = 1
vbc_y1 = 1
vbc_val_y0 = 0
vbc_val_y1 = 0
has_water_loading = yes
surface_temperature = 273
mantle_temperature = 1723
[ic]
mattype_option = 1
num_mattype_layers = 2
layer_mattypes = [1,2]
mattype_layer_depths = [0.24]
temperature_option = 3
num_radiogenic_heat_layer = 3
radiogenic_heat_boundry = [-1, 6e3, 8e3, -1]
radiogenic_heat_mat_in_layer = [1, 1, 2]
radiogenic_heat_dome_amplitude = 10.e3
radiogenic_heat_dome_width = 30.e3
surface_heat_flux = 70e-3
weakzone_option = 0
[mat]
rheology_type = elasto-visco-plastic
num_materials = 7
mattype_ref = 2
mattype_oceanic_crust = 0
mattype_crust = 1
mattype_mantle = 2
mattype_asthenosphere = 3
mattype_sed = 4
mattype_depleted_mantle = 5
mattype_mor_extrusion = 6
rho0 = [ 2900, 2750, 3300, 3300, 2400, 3300, 2900 ]
alpha = [ 3e-5 ]
bulk_modulus = [  4.0e10, 4.3e10, 6.7e10, 6.7e10, 5.5e10, 6.7e10, 4.3e10 ]
shear_modulus = [ 4.3e10, 3.5e10, 6.7e10, 6.7e10, 3.6e10, 6.7e10, 4.3e10 ]
visc_exponent =          [ 3.5,    3.05,     3.5,    3.5,   4.0,    3.5,   3.05 ]
visc_coefficient =       [ 1.1e5, 1.25e-1, 1.1e+5, 1.1e+5,  5.e2,  1.1e+5, 1.25e-1 ]
visc_activation_energy = [ 2.23e5, 2.76e5, 5.30e5, 5.30e5, 4.3e5, 5.30e+5,  2.76e5 ]
visc_activation_volume = [ 27.E-6, 27.E-6, 27.E-6, 27.E-6, 27.E-6, 27.E-6, 27.E-6 ]
radiogenic_heat_prod =   [ 0, 4e-10, 2e-11, 0.00E+00, 0, 2e-11, 0 ]
heat_capacity = [ 1000 ]
therm_cond = [ 3.3 ]
pls0 = [ 0 ]
pls1 = {pls1}
cohesion0 = [ 4e7, 4e7, 4e7, 4e7, 4e7, 4e7, 4.4e7 ]
cohesion1 = [ 4e6 ]
friction_angle0 = {fa0}
friction_angle1 = {fa1}
dilation_angle0 = [ 0 ]
dilation_angle1 = [ 0 ]
max_viscosity = 1e24
min_viscosity = 1e19
"""


def make_equ(long=False):
    if long:
        return EQU.format(stop="max_time_in_yr = 5.e6\noutput_time_interval_in_yr = 50.e3", ckpt=50, ly="50e3", res="2e3",
                          qcsi=100, mbd=2.5, sdiff="7e-6", pls1="[ 0.1, 0.1, 0.5, 0.5, 0.1, 0.5, 0.1 ]",
                          fa0="[ 30, 30, 30, 30, 20, 30, 30 ]", fa1="[ 10, 15, 15, 15, 15, 15, 5 ]")
    return EQU.format(stop="max_steps = 400\noutput_step_interval = 100", ckpt=1, ly="10e3", res="5e3", qcsi=5, mbd="1e-3",
                      sdiff="2e-5", pls1="[ 0.1, 0.1, 5, 5, 0.1, 5, 0.1 ]", fa0="[ 30, 30, 30, 30, 30, 30, 30 ]",
                      fa1="[ 10, 20, 30, 30, 15, 30, 5 ]")


# Parameter values of the reference's examples/conjugate-faults-3d.cfg: the oblique-rift model with
# plain extension, a uniform TetGen mesh (meshing_option = 1: tests/golden/conjugate-faults-3d.desmesh,
# 4,313 nodes / 20,334 tets) and two conjugate weak zones (weakzone_option = 5).
CONJUGATE = (OBLIQUE
             .replace("max_time_in_yr = 0.1e6", "max_time_in_yr = 5e6")
             .replace("""meshing_option = 2
xlength = 200e3
ylength = 100e3
zlength = 50e3
resolution = 5e3
smallest_size = 0.01
refined_zonex = [0.3, 0.7]
refined_zoney = [0.3, 0.7]
refined_zonez = [0.7, 1.0]
quality_check_step_interval = 500
remeshing_option = 11""", """meshing_option = 1
xlength = 200e3
ylength = 100e3
zlength = 50e3
resolution = 5e3
quality_check_step_interval = 500
remeshing_option = 1
min_quality = 0
smallest_size = 0.0
max_boundary_distortion = 1e30""")
             .replace("""vbc_x0 = 6
vbc_x1 = 6
vbc_val_x0 = -3.17e-10
vbc_val_x1 = 3.17e-10
vbc_val_x0_l = 1.59e-10
vbc_val_x1_l = -1.59e-10""", """vbc_x0 = 1
vbc_x1 = 1
vbc_val_x0 = -3.17e-10
vbc_val_x1 = 3.17e-10""")
             .replace("""weakzone_option = 1
weakzone_azimuth = 0
weakzone_inclination = 60
weakzone_halfwidth = 1.5
weakzone_depth_min = 0.0
weakzone_depth_max = 1.0
weakzone_xcenter = 0.5
weakzone_ycenter = 0.5
weakzone_zcenter = 0.5
weakzone_plstrain = 0.5""", """weakzone_option = 5
weakzone_plstrain = 0.5
weakzone_num_segments = 2
weakzone_segments_xcenter = [0.35, 0.65]
weakzone_segments_ycenter = [0.5,  0.5]
weakzone_segments_zcenter = [0.5,  0.5]
weakzone_segments_azimuth      = [0,   180]
weakzone_segments_inclination  = [60,  60]
weakzone_segments_halfwidth    = [1.5, 1.5]
weakzone_segments_x_min = [0.0, 0.5]
weakzone_segments_x_max = [0.5, 1.0]
weakzone_segments_y_min     = [0.0, 0.0]
weakzone_segments_y_max     = [1.0, 1.0]
weakzone_segments_depth_min = [0.0, 0.0]
weakzone_segments_depth_max = [1.0, 1.0]"""))
assert "weakzone_option = 5" in CONJUGATE and "meshing_option = 1" in CONJUGATE and "vbc_x0 = 1" in CONJUGATE


# Parameter values of the reference's benchmarks-cores/test-tiny.cfg (BASELINE configs[0]: the 2-D
# build's plumbing case -- eight materials, elasto-visco-plastic, a Gaussian weak zone, continental
# geotherm, surface diffusion, four steps with a quality check every second one); the mesh
# (meshing_option = 91: benchmarks-cores/cube.poly through Triangle) comes from
# tests/golden/test-tiny.desmesh (97 nodes / 164 triangles).
TEST_TINY = """
[sim]
modelname = benchmark
max_steps = 4
output_step_interval = 1
has_marker_output = yes
has_output_during_remeshing = no
is_outputting_averaged_fields = no
[mesh]
meshing_option = 91
poly_filename = cube.poly
meshing_sediment = no
xlength = 150e3
ylength = 100e3
zlength = 100e3
resolution = 1e4
largest_size = 1.e5
smallest_size = 5.e-2
quality_check_step_interval = 2
min_angle = 30.
min_quality = 0.2
max_boundary_distortion = 1e0
remeshing_option = 11
is_discarding_internal_segments = yes
[markers]
init_marker_option = 1
[control]
ref_pressure_option = 1
surface_process_option = 1
surface_diffusivity = 1e-7
[bc]
vbc_x0 = 1
vbc_val_x0 = -1e-10
vbc_x1 = 1
vbc_val_x1 = 0.e-11
has_water_loading = no
surface_temperature = 273
mantle_temperature = 1573
[ic]
weakzone_option = 3
weakzone_standard_deviation = 3e3
weakzone_xcenter = 0.5
weakzone_ycenter = 0
weakzone_zcenter = 0.3
weakzone_plstrain = 1.0
temperature_option = 1
continental_plate_age_in_yr = 200e6
radiogenic_crustal_thickness = 33e3
radiogenic_folding_depth = 10.e3
radiogenic_heating_of_crust = 3.e-10
lithospheric_thickness = 120.e3
[mat]
rheology_type = elasto-visco-plastic
num_materials = 8
mattype_crust = 3
mattype_mantle = 1
mattype_sed = 4
rho0 = [ 3300, 3280, 2850, 2700, 2400, 3280, 2850, 2700 ]
alpha = [ 3e-5 ]
bulk_modulus = [ 122e9, 122e9, 63e9, 55e9, 55e9, 122e9, 63e9, 55e9 ]
shear_modulus = [ 74e9,  74e9, 40e9, 36e9, 36e9,  74e9, 40e9, 36e9 ]
visc_exponent =          [   3.5,   3.5,    3.05,     4.0,    4.0,   3.5,    3.05,     4.0 ]
visc_coefficient =       [  7.e4,  7.e4, 1.25e-1, 1.25e-1,   5.e2,  7.e4, 1.25e-1, 1.25e-1 ]
visc_activation_energy = [ 4.8e5, 5.3e5,   3.0e5,  2.23e5, 2.23e5, 5.3e5,   3.0e5,  2.23e5 ]
heat_capacity = [ 1000 ]
therm_cond = [ 3.3 ]
pls0 = [ 0 ]
pls1 = [ 0.5 ]
cohesion0 = [ 4e7 ]
cohesion1 = [ 4e6 ]
friction_angle0 = [ 30, 30, 30, 30, 5, 40, 40, 40 ]
friction_angle1 = [ 15, 15, 15,  5, 1, 40, 40, 40 ]
dilation_angle0 = [ 0 ]
dilation_angle1 = [ 0 ]
max_viscosity = 1e24
min_viscosity = 1e19
"""


# benchmarks-cores/test-rect-tiny.cfg = test-tiny.cfg on the 2-D build's own mesh of near-equilateral
# triangles (meshing_option = 1 with meshing_elem_shape = 2: no Triangle needed), a 50-km-deep box
TEST_RECT_TINY_OVERRIDES = ("mesh.meshing_option = 1\nmesh.meshing_elem_shape = 2\nmesh.meshing_verbosity = -1\n"
                            "mesh.zlength = 50e3\nmesh.min_angle = 10.\nmesh.max_boundary_distortion = 1e1\n")

# benchmarks-cores/test-topo.cfg = test-tiny.cfg with topo.poly (10 km of relief on the top boundary) at 5 km,
# strong surface diffusion, 2000 steps; mesh: tests/golden/test-topo.desmesh (361 nodes / 653 triangles)
TEST_TOPO_OVERRIDES = ("mesh.poly_filename = topo.poly\nmesh.resolution = 5e3\ncontrol.surface_diffusivity = 1e-2\n"
                       "sim.max_steps = 2000\nsim.output_step_interval = 500\nsim.checkpoint_frame_interval = 4\n")
