"""ctypes mirrors of include/des_params.h (POD structs crossing the C-ABI)."""
import ctypes as C

DES_MAX_MAT = 16
DES_NBDRY = 10
DES_NBDRY_SIDE = 6
DES_MAX_PERIOD = 8

_dbl_mat = C.c_double * DES_MAX_MAT


class DesParams(C.Structure):
    _fields_ = [
        ("ndims", C.c_int), ("nmat", C.c_int), ("rheol_type", C.c_int), ("mattype_ref", C.c_int),
        ("gravity", C.c_double), ("inertial_scaling", C.c_double), ("damping_factor", C.c_double),
        ("dt_fraction", C.c_double), ("fixed_dt", C.c_double), ("characteristic_speed", C.c_double),
        ("surface_diffusivity", C.c_double), ("surf_base_level", C.c_double),
        ("damping_option", C.c_int), ("ref_pressure_option", C.c_int),
        ("surface_process_option", C.c_int), ("is_quasi_static", C.c_int),
        ("has_thermal_diffusion", C.c_int), ("is_using_mixed_stress", C.c_int),
        ("has_moving_mesh", C.c_int), ("quality_check_step_interval", C.c_int),
        ("surface_temperature", C.c_double), ("winkler_delta_rho", C.c_double),
        ("elastic_foundation_constant", C.c_double), ("sea_water_density", C.c_double),
        ("vbc_val_z1_loading_period", C.c_double),
        ("has_winkler_foundation", C.c_int), ("has_elastic_foundation", C.c_int),
        ("has_water_loading", C.c_int), ("is_outputting_averaged_fields", C.c_int),
        ("vbc_types", C.c_int * DES_NBDRY), ("vbc_values", C.c_double * DES_NBDRY),
        ("vbc_val_l", C.c_double * 4),
        ("stress_bc_types", C.c_int * DES_NBDRY_SIDE), ("stress_bc_values", C.c_double * DES_NBDRY_SIDE),
        ("xlength", C.c_double), ("ylength", C.c_double), ("zlength", C.c_double),
        ("visc_min", C.c_double), ("visc_max", C.c_double), ("tension_max", C.c_double),
        ("therm_diff_max", C.c_double),
        ("rho0", _dbl_mat), ("alpha", _dbl_mat), ("bulk_modulus", _dbl_mat), ("shear_modulus", _dbl_mat),
        ("visc_exponent", _dbl_mat), ("visc_coefficient", _dbl_mat),
        ("visc_activation_energy", _dbl_mat), ("visc_activation_volume", _dbl_mat),
        ("heat_capacity", _dbl_mat), ("therm_cond", _dbl_mat),
        ("pls0", _dbl_mat), ("pls1", _dbl_mat), ("cohesion0", _dbl_mat), ("cohesion1", _dbl_mat),
        ("friction_angle0", _dbl_mat), ("friction_angle1", _dbl_mat),
        ("dilation_angle0", _dbl_mat), ("dilation_angle1", _dbl_mat), ("porosity", _dbl_mat),
        ("max_vbc_val", C.c_double), ("compensation_pressure", C.c_double),
        ("has_PT", C.c_int), ("PT_max_iter", C.c_int), ("PT_relative_tolerance", C.c_double),
        # read by the 2-D build only
        ("is_plane_strain", C.c_int), ("mattype_oceanic_crust", C.c_int),
        ("num_vbc_period_x0", C.c_int), ("num_vbc_period_x1", C.c_int),
        ("vbc_period_x0_time_in_yr", C.c_double * DES_MAX_PERIOD), ("vbc_period_x0_ratio", C.c_double * DES_MAX_PERIOD),
        ("vbc_period_x1_time_in_yr", C.c_double * DES_MAX_PERIOD), ("vbc_period_x1_ratio", C.c_double * DES_MAX_PERIOD),
        ("vbc_vertical_div_x0", C.c_double * 4), ("vbc_vertical_div_x1", C.c_double * 4),
        ("vbc_vertical_ratio_x0", C.c_double * 4), ("vbc_vertical_ratio_x1", C.c_double * 4),
        ("bottom_shear_zone_thickness", C.c_double),
        ("surf_diff_ratio_terrig", C.c_double), ("surf_diff_ratio_marine", C.c_double),
    ]


_pint = C.POINTER(C.c_int)
_pdbl = C.POINTER(C.c_double)


class DesMesh(C.Structure):
    _fields_ = [
        ("nnode", C.c_int), ("nelem", C.c_int),
        ("connectivity", _pint),
        ("support_idx", _pint), ("support_arr", _pint), ("support_lidx", _pint),
        ("bcflag", C.POINTER(C.c_uint)),
        ("nbfacets", C.c_int * DES_NBDRY),
        ("bfacet_elem", _pint * DES_NBDRY), ("bfacet_facet", _pint * DES_NBDRY),
        ("nbnodes", C.c_int * DES_NBDRY), ("bnodes", _pint * DES_NBDRY),
        ("bnormals", _pdbl), ("edge_vec", _pdbl), ("nedge", C.c_int),
        ("edge_slot", C.c_int * (DES_NBDRY * DES_NBDRY)),
        ("ntop", C.c_int), ("etop", C.c_int), ("ntop_elems", C.c_int),
        ("top_nodes", _pint), ("elem_and_nodes", _pint), ("connectivity_surface", _pint),
        ("support_surf_idx", _pint), ("support_surf_arr", _pint), ("top_elems", _pint),
        ("coord", _pdbl), ("owned_begin", C.c_int), ("owned_end", C.c_int),
    ]


class DesHalo(C.Structure):
    _fields_ = [
        ("owned_begin", C.c_int), ("owned_end", C.c_int), ("nlayers", C.c_int), ("nnbr", C.c_int),
        ("nbr_rank", _pint), ("send_ptr", _pint), ("send_idx", _pint),
        ("recv_ptr", _pint), ("recv_idx", _pint),
        ("esend_ptr", _pint), ("esend_idx", _pint), ("erecv_ptr", _pint), ("erecv_idx", _pint),
        ("owned_global_begin", C.c_int),
    ]


class DesScalars(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("time", C.c_double), ("l2_residual", C.c_double),
        ("max_surf_vel", C.c_double), ("max_global_vel_mag", C.c_double),
        ("global_dt_min", C.c_double), ("steps", C.c_longlong), ("status", C.c_int), ("n_return_mapping", C.c_int),
        ("avg_time0", C.c_double), ("n_pt_iterations", C.c_longlong),
    ]


# enum des_field, include/des_params.h
FIELDS = ["COORD", "VEL", "FORCE", "FORCE_RESIDUAL", "COORD0", "TEMPERATURE", "VOLUME_N", "MASS",
          "TMASS", "DHACC", "STRESS", "STRAIN", "STRAIN_RATE", "PLSTRAIN", "DELTA_PLSTRAIN",
          "VISCOSITY", "VOLUME", "VOLUME_OLD", "DPRESSURE", "EDVOLDT", "RADIOGENIC", "ELEMMARKERS",
          "EDVACC_SURF", "DH", "NTMP", "STRESS_AVG", "DPLSTRAIN_AVG", "STRAIN0", "COORD_AVG0", "STRESSYY"]
F = {name: i for i, name in enumerate(FIELDS)}
INT_FIELDS = {"ELEMMARKERS"}
