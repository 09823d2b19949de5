/* des_run.h -- output writers and the driver loop of the host side (libdes_host.so).
 *
 *   des_output_*   Output::write / write_exact / write_checkpoint / write_info
 *                  (output.cxx:41-274, 372-409) in the reference's own binary format
 *                  (binaryio.cxx:18-204: 4096-byte ASCII header "# DynEarthSol ndims=3 revision=4",
 *                  "name\toffset" lines, raw little-endian blobs in AoS order), so Dynearthsol.py,
 *                  2vtk.py and compare.py read frames of a device run unchanged.
 *   des_run        main()'s loop (dynearthsol.cxx:738-982): steps the engine, schedules output
 *                  frames / checkpoints / the mesh-quality check exactly as the reference does.
 *
 * The loop is written against a table of engine entry points (des_engine_api) whose members
 * have the signatures of include/des_dev.h; the product fills it with des_dev_* (see
 * dynearthsol_amd/csrc/driver/main.cpp).  libdes_host.so itself never touches a GPU.
 */
#ifndef DES_RUN_H
#define DES_RUN_H

#include "des_host.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host copies of the arrays one frame needs, SoA as des_dev_download returns them. */
typedef struct des_frame {
    long long steps;
    double time, dt, max_global_vel_mag;
    const double *coord, *vel, *temperature, *radiogenic, *plstrain, *delta_plstrain, *strain_rate,
                 *strain, *stress, *viscosity, *volume, *force, *coord0;
    const int *elemmarkers;
    /* checkpoint only */
    const double *volume_old, *edvacc_surf, *dhacc;
    double reference_frame_time, last_remesh_time;
    int info_display_next_step, pad_;
    /* Output::average_fields state; all NULL unless sim.is_outputting_averaged_fields */
    const double *coord_avg0, *strain0, *stress_avg, *dplstrain_avg;
    double avg_time0;
    const double *stressyy;         /* checkpoint only, plane-strain 2-D models (output.cxx:394-395); else NULL */
} des_frame;

typedef struct des_output des_output;

/* Output::Output (output.cxx:23-35); the clock for the wall-time column starts here */
des_output *des_output_create(const des_host *host, int start_frame);
void des_output_destroy(des_output *o);
int des_output_frame(const des_output *o);            /* number of the next frame */
/* Output::write (exact = 0: averaged variants when the option is on) / write_exact (exact = 1).
 * Writes <modelname>.save.NNNNNN, appends the .info row, prints the "Output #" line. */
int des_output_write(des_output *o, const des_frame *f, int exact);
/* Output::write_checkpoint: <modelname>.chkpt.NNNNNN, marker set included (the host keeps it);
 * an extra "elemmarkers" array carries the per-element counts the device works with */
int des_output_write_checkpoint(des_output *o, const des_frame *f);

/* Entry points of an engine, same signatures and meaning as include/des_dev.h. */
typedef struct des_engine_api {
    void *(*create)(int device, const des_params *params, const des_mesh *mesh, int *err);
    void (*destroy)(void *h);
    int (*upload)(void *h, int field, const void *host, long long count);
    int (*download)(void *h, int field, void *host, long long count);
    long long (*field_count)(const void *h, int field);
    int (*set_clock)(void *h, double dt, double time, long long steps);
    int (*init_geometry)(void *h);
    int (*compute_dt)(void *h, double *dt);
    int (*step)(void *h, int nsteps, des_scalars *out);
    int (*check_nan)(void *h, long long *n_nan);
    int (*mesh_quality)(void *h, double smallest_vol, double bottom, double bottom_dist, des_quality *out);
    const char *(*last_error)(void);
    /* isostasy_adjustment mode on / off (des_dev_set_isostasy); may be NULL, in which case a
     * config with ic.isostasy_adjustment_time_in_yr > 0 is refused with code 31 */
    int (*set_isostasy)(void *h, int on);
    /* One-process-per-GPU runs: every rank runs the same loop over an engine table whose entries
     * are collective (upload scatters, download gathers the global arrays, step / compute_dt /
     * mesh_quality / check_nan reduce across ranks: dynearthsol_amd/distributed.py); ranks with
     * no_files != 0 take part in every gather but write no frame, checkpoint or progress line. */
    int no_files;
    /* initial_body_force_adjustment (des_dev_body_force_adjustment; dynearthsol.cxx:753-761); may be NULL, in which
     * case a config with ic.has_body_force_adjustment is refused with code 31 */
    int (*body_force_adjustment)(void *h, des_scalars *out);
} des_engine_api;

/* What the loop did, for callers that do not parse stdout. */
typedef struct des_run_stats {
    long long steps;
    double time, dt;
    int frames, checkpoints;
    int exit_code;              /* 0, or the reference ExitCode the run stopped with          */
    int remesh_needed;          /* bad_mesh_quality's code (1..3) if the run stopped for it   */
    double compute_seconds;     /* wall time inside engine steps                              */
    long long phase_changed_markers;   /* markers whose material phase_changes() moved (phasechanges.cxx:109-152) */
    int last_frame;             /* number of the last frame written (with remesh_needed: the state to remesh, with its checkpoint) */
    int pad_;
} des_run_stats;

/* init() tail + main loop.  Returns stats->exit_code.  `quiet` suppresses the progress lines.
 *
 * Remeshing is host work with a mesher the product does not have: where the
 * reference would call remesh() (remeshing.cxx:2869-3189) the loop writes the state as an exact
 * frame + checkpoint (stats->last_frame), sets stats->remesh_needed and returns 31.  The callers
 * (driver/main.cpp --remesher / DES_REMESH_CMD, dynearthsol_amd/driver.py run(remesher=...)) turn
 * that into a round trip: run `<command> <modelname> <frame>`, which must leave the remeshed model
 * as frame + 1 (save, chkpt and the .info row, the reference's formats -- the reference binary
 * itself, restarted from the pair, is such a tool), then restart from it (sim.is_restarting) with
 * a new engine on the new mesh, and so on until the run ends. */
int des_run(des_host *host, const des_engine_api *api, int device, int quiet, des_run_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
