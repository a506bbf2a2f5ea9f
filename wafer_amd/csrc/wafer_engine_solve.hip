// wafer_engine_solve.hip -- grid.rs:50-246 for one state: the host loop over screen_update blocks, through the C ABI's own entry points.
#include "wafer_engine.h"

extern "C" {

// ---- solve (grid.rs:50-246) ----------------------------------------------------------
int wafer_solve_state(wafer_ctx *c, uint32_t wnum, double tolerance, uint64_t screen_update,
                      int has_max_steps, uint64_t max_steps, wafer_block_record *records,
                      size_t max_records, size_t *n_records, wafer_observables_output *final_out)
{
    if (!c) return fail(WAFER_ERR_INVALID, "null context");
    if (wnum > c->states.size()) return fail(WAFER_ERR_STATE, "wnum %u but w_store holds %zu states", wnum, c->states.size());
    uint64_t step = 0;
    double last_energy = DBL_MAX; // grid.rs:124
    size_t nrec = 0;
    bool converged = false;
    wafer_observables_t obs;
    for (;;) {
        TRY(wafer_observables(c, &obs));                    // :127
        const double norm_energy = obs.energy / obs.norm2;  // :128
        // R64 panics on NaN in the reference's debug builds (noisy_float); in release it would
        // iterate on NaNs forever.  Report it instead of spinning until max_steps.
        if (!std::isfinite(norm_energy))
            return fail(WAFER_ERR_STATE, "state %u: energy is not finite at step %llu (norm2 = %g): "
                        "the wavefunction vanished or diverged", wnum, (unsigned long long)step, obs.norm2);
        const double tau = (double)step * c->P.dt;          // :129
        TRY(wafer_normalise(c, obs.norm2));                 // :130
        if (wnum > 0) TRY(wafer_orthogonalise(c, wnum));    // :133-135
        const double diff = std::fabs(norm_energy - last_energy); // :161
        if (records && nrec < max_records) {
            records[nrec].step = step;
            records[nrec].tau = tau;
            records[nrec].obs = obs;
            records[nrec].diff = diff;
        }
        ++nrec;
        if (diff < tolerance) { // :162-192
            converged = true;
            break;
        }
        last_energy = norm_energy;                          // :194
        if (has_max_steps && step > max_steps) break;       // :211-213
        TRY(wafer_evolve(c, wnum, screen_update));          // :216
        step += screen_update;                              // :220
    }
    if (n_records) *n_records = nrec;
    if (final_out) { // output.rs:540-547
        const double r_norm = std::sqrt(obs.r2 / obs.norm2);
        final_out->state = wnum;
        final_out->energy = obs.energy / obs.norm2;
        final_out->binding_energy = (obs.energy - obs.v_infinity) / obs.norm2;
        final_out->r = r_norm;
        final_out->l_r = (double)c->P.nx / r_norm;
    }
    if (!converged) return fail(WAFER_ERR_MAX_STEP, "MaxStep: state %u did not converge within max_steps", wnum);
    return wafer_push_state(c); // :239-242
}

} // extern "C"
