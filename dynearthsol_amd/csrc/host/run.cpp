// run.cpp -- main()'s tail and loop (dynearthsol.cxx:643-982) over an engine given as a table of
// entry points (include/des_run.h).  Everything the reference does between two output / mesh-
// quality events is one des_dev_step(n) call: the engine keeps dt, time and steps on the device,
// so the host only joins the stream at the events the reference itself schedules:
//   * an output frame (step- or time-triggered, dynearthsol.cxx:906-931),
//   * every mesh.quality_check_step_interval steps: bad_mesh_quality + the progress line (:933-971),
//   * the end of the run (:977-980).
#include "des_run.h"
#include "des_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>
#include <string>
#include <vector>

namespace des {
const std::string &output_last_error();
void output_set_quiet(des_output *o, bool q);
}
extern "C" int des_output_skip(des_output *o);      // advance the frame counter without writing

namespace {

const double YEAR2SEC = 365.2422 * 86400;             // constants.hpp:77
const double sizefactor = 0.118;                      // remeshing.cxx:41 (3D)

double seconds_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Run {
    des_host *host;
    const des_engine_api *api;
    void *eng;
    des_output *out;
    des_scalars sc;
    bool averaged;
    bool plane_strain = false;    // mat.is_plane_strain of a 2-D model: stressyy goes into checkpoints
    bool body_force_adjustment = false;   // ic.has_body_force_adjustment: the engine's PT loop runs once before the first step
    int info_display_next_step;
    double reference_frame_time, last_remesh_time;
    double restored_vmax;         // max_global_vel_mag of the checkpoint, until compute_dt renews it

    void check(int rc, const char *what)
    {
        if (rc != DES_OK)
            throw des::Error(rc, std::string(what) + ": " + (api->last_error ? api->last_error() : ""));
    }

    std::vector<double> get(int field)
    {
        long long n = api->field_count(eng, field);
        if (n < 0) throw des::Error(60, "engine does not hold field " + std::to_string(field));
        std::vector<double> v((size_t)n);
        if (n) check(api->download(eng, field, v.data(), n), "download");
        return v;
    }

    // gathers what Output::_write / write_checkpoint read from Variables
    void write(bool exact, bool checkpoint)
    {
        std::vector<double> coord = get(DES_F_COORD), vel = get(DES_F_VEL), T = get(DES_F_TEMPERATURE),
            radio = get(DES_F_RADIOGENIC), pls = get(DES_F_PLSTRAIN), dpls = get(DES_F_DELTA_PLSTRAIN),
            edot = get(DES_F_STRAIN_RATE), strain = get(DES_F_STRAIN), stress = get(DES_F_STRESS),
            visc = get(DES_F_VISCOSITY), vol = get(DES_F_VOLUME), force = get(DES_F_FORCE), coord0 = get(DES_F_COORD0);
        std::vector<int> markers((size_t)api->field_count(eng, DES_F_ELEMMARKERS));
        check(api->download(eng, DES_F_ELEMMARKERS, markers.data(), (long long)markers.size()), "download");
        des_frame f = des_frame();
        if (sc.max_global_vel_mag == 0 && restored_vmax != 0) sc.max_global_vel_mag = restored_vmax;
        f.steps = sc.steps; f.time = sc.time; f.dt = sc.dt; f.max_global_vel_mag = sc.max_global_vel_mag;
        f.coord = coord.data(); f.vel = vel.data(); f.temperature = T.data(); f.radiogenic = radio.data();
        f.plstrain = pls.data(); f.delta_plstrain = dpls.data(); f.strain_rate = edot.data();
        f.strain = strain.data(); f.stress = stress.data(); f.viscosity = visc.data(); f.volume = vol.data();
        f.force = force.data(); f.coord0 = coord0.data(); f.elemmarkers = markers.data();
        std::vector<double> c0, s0, savg, davg, vold, edv, dhacc, syy;
        if (averaged && !exact) {
            c0 = get(DES_F_COORD_AVG0); s0 = get(DES_F_STRAIN0); savg = get(DES_F_STRESS_AVG); davg = get(DES_F_DPLSTRAIN_AVG);
            f.coord_avg0 = c0.data(); f.strain0 = s0.data(); f.stress_avg = savg.data(); f.dplstrain_avg = davg.data();
            f.avg_time0 = sc.avg_time0;
        }
        if (checkpoint) {
            vold = get(DES_F_VOLUME_OLD); edv = get(DES_F_EDVACC_SURF); dhacc = get(DES_F_DHACC);
            f.volume_old = vold.data(); f.edvacc_surf = edv.data(); f.dhacc = dhacc.data();
            if (plane_strain) { syy = get(DES_F_STRESSYY); f.stressyy = syy.data(); }
            f.info_display_next_step = info_display_next_step;
            f.reference_frame_time = reference_frame_time; f.last_remesh_time = last_remesh_time;
            int rc = api->no_files ? 0 : des_output_write_checkpoint(out, &f);
            if (rc) throw des::Error(rc, des::output_last_error());
        }
        int rc = api->no_files ? des_output_skip(out) : des_output_write(out, &f, exact ? 1 : 0);
        if (rc) throw des::Error(rc, des::output_last_error());
        if (exact) {
            // write_exact: check for NaN in var (output.cxx:291-292)
            long long n_nan = 0;
            int nrc = api->check_nan(eng, &n_nan);
            if (nrc) throw des::Error(nrc, "NaN in the state (" + std::to_string(n_nan) + " entries)");
        }
    }
};

std::string fmt_wall(double s)
{
    long long ns = (long long)(s * 1e9);
    char b[64];
    std::snprintf(b, sizeof(b), "%03d:%02d:%09.6f", (int)(ns / 3600000000000LL),
                  (int)((ns % 3600000000000LL) / 60000000000LL), (ns % 60000000000LL) / 1e9);
    return b;
}

} // namespace

extern "C" int des_run(des_host *host, const des_engine_api *api, int device, int quiet, des_run_stats *stats)
{
    des_run_stats st = des_run_stats();
    Run r = Run();
    r.host = host; r.api = api;
    const double t_start = seconds_now();
    try {
        const des::Config &cfg = host->cfg;
        const des_params &p = host->params;
        const des::HostMesh &m = host->mesh;
        const int max_steps = cfg.given("sim.max_steps") ? cfg.i("sim.max_steps") : std::numeric_limits<int>::max();
        const double max_time_in_yr = cfg.given("sim.max_time_in_yr") ? cfg.d("sim.max_time_in_yr") : std::numeric_limits<double>::max();
        const int output_step_interval = cfg.given("sim.output_step_interval") ? cfg.i("sim.output_step_interval") : std::numeric_limits<int>::max();
        const double output_time_interval_in_yr = cfg.given("sim.output_time_interval_in_yr") ? cfg.d("sim.output_time_interval_in_yr") : std::numeric_limits<double>::max();
        const int checkpoint_frame_interval = cfg.i("sim.checkpoint_frame_interval");
        const int qcsi = p.quality_check_step_interval;
        int info_display_step_interval = cfg.i("sim.info_display_step_interval");
        if (info_display_step_interval <= 0) info_display_step_interval = qcsi * 100;          // input.cxx:1049-1051
        r.averaged = p.is_outputting_averaged_fields != 0;

        // ---- init() tail: the engine replays compute_volume .. compute_dt (dynearthsol.cxx:175-221, 643)
        int err = 0;
        r.eng = api->create(device, &p, &host->view, &err);
        if (!r.eng) throw des::Error(err ? err : 31, std::string("engine: ") + (api->last_error ? api->last_error() : ""));
        const des::HostFields &f = host->fields;
        const des::RestartState &rs = f.restart;
        auto up = [&](int field, const std::vector<double> &v, const char *what) {
            r.check(api->upload(r.eng, field, v.data(), (long long)v.size()), what);
        };
        up(DES_F_COORD, m.coord, "upload coord");
        up(DES_F_COORD0, rs.active ? rs.coord0 : m.coord, "upload coord0");
        r.check(api->upload(r.eng, DES_F_ELEMMARKERS, f.elemmarkers.data(), (long long)f.elemmarkers.size()), "upload elemmarkers");
        up(DES_F_VEL, f.vel, "upload vel");
        // restart() computes the masses with the restored temperature (dynearthsol.cxx:394-396);
        // init() does it while T is still 0 (:175-186)
        if (rs.active) up(DES_F_TEMPERATURE, f.temperature, "upload temperature");
        r.check(api->init_geometry(r.eng), "init_geometry");
        r.check(api->upload(r.eng, DES_F_TEMPERATURE, f.temperature.data(), (long long)f.temperature.size()), "upload temperature");
        r.check(api->upload(r.eng, DES_F_RADIOGENIC, f.radiogenic.data(), (long long)f.radiogenic.size()), "upload radiogenic");
        r.check(api->upload(r.eng, DES_F_STRESS, f.stress.data(), (long long)f.stress.size()), "upload stress");
        r.check(api->upload(r.eng, DES_F_STRAIN, f.strain.data(), (long long)f.strain.size()), "upload strain");
        r.check(api->upload(r.eng, DES_F_PLSTRAIN, f.plstrain.data(), (long long)f.plstrain.size()), "upload plstrain");
        r.check(api->upload(r.eng, DES_F_VISCOSITY, f.viscosity.data(), (long long)f.viscosity.size()), "upload viscosity");
        if (m.nd == 2)
            r.check(api->upload(r.eng, DES_F_STRESSYY, f.stressyy.data(), (long long)f.stressyy.size()), "upload stressyy");
        r.plane_strain = p.is_plane_strain != 0;
        if (!rs.active) {
            double dt0 = 0;
            r.check(api->compute_dt(r.eng, &dt0), "compute_dt");
            const double iso_yr = cfg.d("ic.isostasy_adjustment_time_in_yr");
            if (iso_yr > 0) {
                // isostasy_adjustment (dynearthsol.cxx:496-544, called at :638-641), then the
                // compute_dt of :643
                if (!api->set_isostasy) throw des::Error(31, "this engine does not offload the isostasy adjustment");
                if (!quiet && !api->no_files) { std::printf("Adjusting isostasy for %g yrs...\n", iso_yr); std::fflush(stdout); }
                const int iso_steps = (int)(iso_yr * YEAR2SEC / dt0);
                r.check(api->set_isostasy(r.eng, 1), "set_isostasy");
                r.check(api->step(r.eng, iso_steps, nullptr), "isostasy steps");
                r.check(api->set_isostasy(r.eng, 0), "set_isostasy");
                if (!quiet && !api->no_files) { std::printf("Adjusted isostasy for %d steps.\n", iso_steps); std::fflush(stdout); }
                r.check(api->compute_dt(r.eng, &dt0), "compute_dt");
            }
            r.info_display_next_step = info_display_step_interval;               // dynearthsol.cxx:636
            r.last_remesh_time = 0; r.reference_frame_time = 0;
        } else {
            // the rest of restart(): previous-step volume, surface accumulators, fields that are
            // "not required for restarting, yet" (:366-392), and the clock from the checkpoint
            up(DES_F_VOLUME_OLD, rs.volume_old, "upload volume_old");
            if (!rs.edvacc_surf.empty()) up(DES_F_EDVACC_SURF, rs.edvacc_surf, "upload edvacc_surf");
            up(DES_F_DHACC, rs.dhacc, "upload dhacc");
            up(DES_F_STRAIN_RATE, rs.strain_rate, "upload strain_rate");
            up(DES_F_FORCE, rs.force, "upload force");
            up(DES_F_DELTA_PLSTRAIN, rs.delta_plstrain, "upload delta_plstrain");
            r.check(api->set_clock(r.eng, rs.dt, rs.time, rs.steps), "set_clock");
            r.info_display_next_step = rs.info_display_next_step;
            if (rs.steps % qcsi == 0 && rs.steps >= r.info_display_next_step)
                r.info_display_next_step = rs.steps + info_display_step_interval;     // :354-356
            r.last_remesh_time = rs.last_remesh_time; r.reference_frame_time = rs.reference_frame_time;
            r.restored_vmax = rs.max_global_vel_mag;
        }
        r.check(api->step(r.eng, 0, &r.sc), "clock");

        r.out = des_output_create(host, rs.active ? rs.frame : 0);
        r.body_force_adjustment = cfg.b("ic.has_body_force_adjustment");
        if (r.body_force_adjustment && !api->body_force_adjustment)
            throw des::Error(31, "this engine does not offload the initial body-force adjustment");
        if (api->no_files) quiet = 1;
        des::output_set_quiet(r.out, quiet != 0);

        if (!rs.active && cfg.b("sim.has_initial_checkpoint")) {
            // var.output->write_checkpoint(param, var) before the first frame (:646-647)
            std::vector<double> vold = r.get(DES_F_VOLUME_OLD), edv = r.get(DES_F_EDVACC_SURF), dhacc = r.get(DES_F_DHACC);
            des_frame cf = des_frame();
            cf.time = r.sc.time; cf.dt = r.sc.dt; cf.max_global_vel_mag = r.sc.max_global_vel_mag;
            cf.volume_old = vold.data(); cf.edvacc_surf = edv.data(); cf.dhacc = dhacc.data();
            cf.elemmarkers = f.elemmarkers.data();
            cf.info_display_next_step = r.info_display_next_step;
            int rc = api->no_files ? 0 : des_output_write_checkpoint(r.out, &cf);
            if (rc) throw des::Error(rc, des::output_last_error());
            st.checkpoints++;
        }

        r.write(true, false);                                     // var.output->write_exact(var)
        st.frames++;

        const double starting_time = r.reference_frame_time;
        r.reference_frame_time = starting_time + output_time_interval_in_yr * YEAR2SEC;
        const long long starting_step = r.sc.steps;
        int next_regular_frame = 1;

        if (r.body_force_adjustment) {
            // initial_body_force_adjustment (dynearthsol.cxx:753-761): right before the time loop, also after a restart
            r.check(api->body_force_adjustment(r.eng, &r.sc), "body_force_adjustment");
            r.body_force_adjustment = false;
        }
        if (!quiet) {
            std::printf("Starting simulation...\n  Showing model progress every %d steps.\n", info_display_step_interval);
            std::fflush(stdout);
        }

        // remeshing.cxx:40-44, 2765: the volume of an equilateral tetrahedron / triangle of unit side
        const double smallest_vol = cfg.d("mesh.smallest_size") * (m.nd == 3 ? sizefactor : 0.433) * std::pow(cfg.d("mesh.resolution"), m.nd);
        const int remeshing_option = cfg.i("mesh.remeshing_option");
        const bool check_bottom = remeshing_option == 1 || remeshing_option == 2 || remeshing_option == 11 || remeshing_option == 13;
        const double bottom_dist = check_bottom ? cfg.d("mesh.max_boundary_distortion") * cfg.d("mesh.resolution") : -1.0;

        const bool phase_change_on = p.nmat > 1 && cfg.i("mat.phase_change_option") != 0;
        bool go_on = true;
        do {
            // ---- how many steps until the next point where the reference's loop does anything
            // but step?  dt only changes when steps % 10 == 0 (or never, with fixed_dt), which bounds
            // the look-ahead of the time-triggered conditions.
            const long long steps = r.sc.steps;
            long long n = qcsi - steps % qcsi;                                     // quality check / info line
            if (max_steps - steps < n) n = max_steps - steps;
            if (output_step_interval != std::numeric_limits<int>::max()) {
                long long due = starting_step + (long long)next_regular_frame * output_step_interval - steps;
                if (due >= 1 && due < n) n = due;
            }
            if (!r.averaged) {
                // time-triggered events can fire at any step: stop one step short of the estimate
                // and walk the rest step by step
                double t_event = std::numeric_limits<double>::max();
                if (output_time_interval_in_yr != std::numeric_limits<double>::max())
                    t_event = starting_time + next_regular_frame * output_time_interval_in_yr * YEAR2SEC;
                if (max_time_in_yr != std::numeric_limits<double>::max())
                    t_event = std::min(t_event, max_time_in_yr * YEAR2SEC);
                if (t_event != std::numeric_limits<double>::max()) {
                    long long to_dt_change = 10 - steps % 10;
                    double est = std::floor((t_event - r.sc.time) / r.sc.dt) - 1;
                    long long safe = est < 1 ? 1 : (est > 1e9 ? (long long)1e9 : (long long)est);
                    if (safe > to_dt_change && p.fixed_dt == 0) safe = to_dt_change;
                    if (safe < n) n = safe;
                }
            }
            // phase_changes runs every 10 steps on the host's markers (dynearthsol.cxx:881-894)
            if (phase_change_on && 10 - steps % 10 < n) n = 10 - steps % 10;
            if (n < 1) n = 1;

            const double t0 = seconds_now();
            int rc = api->step(r.eng, (int)n, &r.sc);
            st.compute_seconds += seconds_now() - t0;
            if (rc) throw des::Error(rc, std::string("step: ") + (api->last_error ? api->last_error() : ""));

            // ---- slow updates (dynearthsol.cxx:881-894): phase_changes, then compute_dt with the new
            // material mix.  The engine has already taken its compute_dt of this step with the old one,
            // so it is only repeated when a marker did change.
            if (phase_change_on && r.sc.steps % 10 == 0) {
                std::vector<double> coord = r.get(DES_F_COORD), T = r.get(DES_F_TEMPERATURE);
                const int changed = des::phase_changes(cfg, p, m, host->fields, coord.data(), T.data());
                st.phase_changed_markers += changed;
                if (changed) {
                    r.check(api->upload(r.eng, DES_F_ELEMMARKERS, host->fields.elemmarkers.data(),
                                        (long long)host->fields.elemmarkers.size()), "upload elemmarkers");
                    double dt_new = 0;
                    r.check(api->compute_dt(r.eng, &dt_new), "compute_dt");
                    r.check(api->step(r.eng, 0, &r.sc), "clock");
                }
            }

            // ---- output (dynearthsol.cxx:906-931)
            const bool step_due = output_step_interval != std::numeric_limits<int>::max() &&
                (r.sc.steps - starting_step) == (long long)next_regular_frame * output_step_interval;
            const bool time_due = output_time_interval_in_yr != std::numeric_limits<double>::max() &&
                (r.sc.time - starting_time) > next_regular_frame * output_time_interval_in_yr * YEAR2SEC;
            if ((step_due || time_due) && (!r.averaged || r.sc.steps % qcsi == 0)) {
                const bool chk = next_regular_frame % checkpoint_frame_interval == 0;
                r.write(false, chk);
                st.frames++;
                if (chk) st.checkpoints++;
                next_regular_frame++;
                r.reference_frame_time = starting_time + next_regular_frame * output_time_interval_in_yr * YEAR2SEC;
            }

            // ---- mesh quality and progress (dynearthsol.cxx:933-971)
            if (r.sc.steps % qcsi == 0) {
                double min_quality = 1.0;
                if (p.has_moving_mesh && api->mesh_quality) {
                    des_quality q;
                    r.check(api->mesh_quality(r.eng, smallest_vol, -p.zlength, bottom_dist, &q), "mesh_quality");
                    int bad = 0;
                    if (q.small_elem >= 0) {
                        bad = 3;
                        if (!quiet) std::printf("    The size of element #%d is too small.\n", q.small_elem);
                    } else if (q.bottom_node >= 0) {
                        bad = 2;
                        if (!quiet) std::printf("    Node #%d is too far from the bottm\n", q.bottom_node);
                    } else {
                        min_quality = (m.nd == 3) ? std::pow(q.worst_quality, 1.0 / 3) : q.worst_quality;   // remeshing.cxx:2845-2848
                        if (min_quality < cfg.d("mesh.min_quality")) {
                            bad = 1;
                            if (!quiet) std::printf("    Element #%d has mesh quality = %g\n", q.worst_elem, min_quality);
                        }
                    }
                    if (bad) {
                        // remesh() is host work that is not offloaded: leave a frame and a
                        // checkpoint of the last good state and stop with the reference's category
                        r.write(true, true);
                        st.frames++; st.checkpoints++;
                        st.remesh_needed = bad;
                        throw des::Error(31, "the mesh needs remeshing (bad_mesh_quality = " + std::to_string(bad) +
                                         "); remeshing is not offloaded -- state saved in the last frame/checkpoint");
                    }
                }
                if (r.sc.steps >= r.info_display_next_step) {
                    if (!quiet) {
                        std::printf("              Step = %lld, time = %.5e yr, vmax = %.5e m/s", r.sc.steps,
                                    r.sc.time / YEAR2SEC, r.sc.max_global_vel_mag);
                        if (p.has_moving_mesh && min_quality < 1.0) std::printf(", min_q = %.4f", min_quality);
                        std::printf(", wt = %s\n", fmt_wall(seconds_now() - t_start).c_str());
                        std::fflush(stdout);
                    }
                    r.info_display_next_step = (int)r.sc.steps + info_display_step_interval;
                }
            }

            go_on = r.sc.steps < max_steps &&
                    (r.sc.time <= max_time_in_yr * YEAR2SEC || (r.averaged && r.sc.steps % qcsi != 0));
        } while (go_on);

        if (!quiet) {
            const double total = seconds_now() - t_start;
            std::printf("Ending simulation.\nTime summary...\n  Execute : %s\n  Compute : %s (%5.2f%%)/ %lld = %.6f s/step\n",
                        fmt_wall(total).c_str(), fmt_wall(st.compute_seconds).c_str(), 100. * st.compute_seconds / total,
                        r.sc.steps, r.sc.steps ? st.compute_seconds / r.sc.steps : 0.0);
        }
    } catch (const des::Error &e) {
        std::fprintf(stderr, "%s\n", e.what());
        st.exit_code = e.code;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        st.exit_code = DES_ERR_INTERNAL;
    }
    st.steps = r.sc.steps; st.time = r.sc.time; st.dt = r.sc.dt;
    st.last_frame = r.out ? des_output_frame(r.out) - 1 : -1;
    if (r.out) des_output_destroy(r.out);
    if (r.eng) api->destroy(r.eng);
    if (stats) *stats = st;
    return st.exit_code;
}
