#pragma once
#include <hip/hip_runtime.h>

namespace mjmpc {

// Optional fusions into the rollout launch (all pointers may be null):
//   filt    float64[3]  apply the recursive noise filter of control_utils.py:32-33 to `noise` on the fly
//                       (then `noise` holds the raw, unfiltered samples)
//   gseq    float64[H]  with q0_out: q0_out[p] = sum_t gseq[t] * cost[p][t]  (= cost_to_go(...)[:,0])
//   clw     float64[(2nv+7)][A]  mode "closed_loop_linear" (gym_env_wrapper.py:135-136): the nominal action of
//                       a step is clw^T [obs; 1] with obs the observation BEFORE the step; `mean` is ignored
//   shard_size          > 0: particles [k*shard_size, (k+1)*shard_size) use model block k (dynamics randomization:
//                       every shard of the reference's worker pool simulates its own perturbed model)
//   state_shard_size    > 0: likewise for the start state: shard k starts from state vector k
//   reset_rec           MuJoCo's reset on instability (mj_checkPos / mj_checkVel / mj_checkAcc -> mj_resetData, [EXT]; the
//                       rollouts of gym_env_wrapper.py:125-153 run through it): one record of ARM_RESET_LEN float64 per model
//                       block - qpos[8] | qvel[8] ONE SUBSTEP AFTER the reset state (qpos0 = 0, zero velocity, zero
//                       controls) | the tracked site[3] AT the reset state | sin[8] | cos[8] of that qpos.  A particle whose qpos / qvel hold a NaN or an
//                       entry beyond 1e10 when a substep begins restarts that substep from the reset state, one whose
//                       acceleration does continues from the record; its controls are zero until the env step ends.
//                       nullptr: no resets (e.g. the launch that makes the record: one particle, frame_skip 1).
constexpr int ARM_RESET_LEN = 35;
struct RolloutFusion {
    long shard_size = 0;
    long state_shard_size = 0;
    const double* clw = nullptr;
    const double* filt = nullptr;
    const double* gseq = nullptr;
    double* q0_out = nullptr;
    const double* reset_rec = nullptr;
    int inf_on_reset = 0;           // 1: a particle that has reset costs +inf from that env step on (q0_out +inf): it then has
                                    // no weight in the softmax updates and ranks last for CEM - ABI 2's behaviour for diverged
                                    // rollouts, as an engine option (mjmpc_arm_set_reset_returns); 0: the costs MuJoCo's reset
                                    // state produces (what the reference's worker would return)
};

// TWO launches per control iteration (mjmpc_arm_mppi_step).  The rollout kernel draws its own samples, keeps the actions
// of its particles in LDS and leaves one softmax record {max, S, W[H][A]} per workgroup; the finish kernel (H workgroups,
// one per horizon row) merges the records, updates the mean (mppi.py:69-82), publishes the action, shifts the horizon
// (olgaussian_mpc.py:116-129) and steps the device-resident real env - sampling, GymEnvWrapper.rollout, cost_to_go,
// _update_distribution, _get_next_action, _shift and env.step of one Controller.optimize().
struct MonoStep {
    // sampler: eps_raw[p][t][a] = chol[a][a] z(seed, offset + *d_step, p + particle_offset, a, t) - the Philox draws of
    // noise.hip's noise_kernel for a DIAGONAL covariance, made by the lane that consumes them
    const double* chol = nullptr;           // device [A][A]
    int chol_full = 0;                      // 1: the whole lower triangle colours the draws (eps[a] = sum_{b <= a} chol[a][b] z_b,
                                            // noise_full_kernel's stream: CEM's adapting covariance), 0: its diagonal
    unsigned long long seed = 0, offset = 0;
    long particle_offset = 0;
    const long long* d_step = nullptr;
    // update
    double lam = 1.0, step_size = 1.0;
    int shift_mode = 0;                     // 0 'null', 1 'repeat', < 0 no shift
    double* tree = nullptr;                 // one record [2 + H A] per rollout workgroup (mono_record_doubles)
    double* action_out = nullptr;           // device [A]
    double* action_host = nullptr;          // mapped pinned [2][A + 1]: slot (step & 1) = action | completion flag (step + 1)
    long long* step_counter = nullptr;      // advanced by one
    double* record = nullptr;               // sharded runs: this GPU's record [max | S | W[H*A]] INSTEAD of update/shift/env step
    // real env: advanced in place by one env step with the new action
    double* state_io = nullptr;
    void* step_cost = nullptr;              // T[1]
    void* step_nobs = nullptr;              // T[2 nv + 6]
    const double* reset_rec = nullptr;      // the real env's reset record (RolloutFusion::reset_rec of its model block)
};
long mono_record_doubles(long groups, int H, int A);

// Fused (particles x horizon x frame_skip) rollout of a compiled arm; see arm_rollout.hip.
template <typename T>
hipError_t launch_arm_rollout(const T* model, const double* state, long P, int H, int A, const double* mean,
                              const T* noise, T* cost, T* act, T* obs, T* nobs, double* state_out,
                              unsigned* diag, hipStream_t stream, RolloutFusion fuse = RolloutFusion(),
                              const MonoStep* mono = nullptr);
// (mono: the fused iteration's parameters; the kernels take the structure BY VALUE - a kernel argument arrives with the
// launch, where a block in device memory cost every workgroup a dependent round trip before its first useful load)
// the finish launch: records [n_rec][2 + H A] -> mean_out (mean_in is only read; the two must not alias), action, step
// counter, env step (env_step != 0); or, with mono->record, this GPU's record
template <typename T>
hipError_t launch_arm_mppi_finish(const T* model, const double* records, long n_rec, int H, int A, const double* mean_in,
                                  double* mean_out, const MonoStep& mono, int env_step, unsigned* diag, hipStream_t stream);
// workgroups the launch of P particles uses (the reduction tree is sized by it)
long arm_rollout_groups(long P);
// the extended-joint build (arm_rollout_xj.hip): the same two launches for models with slide joints / friction loss
template <typename T>
hipError_t launch_arm_rollout_xj(const T* model, const double* state, long P, int H, int A, const double* mean,
                                 const T* noise, T* cost, T* act, T* obs, T* nobs, double* state_out,
                                 unsigned* diag, hipStream_t stream, RolloutFusion fuse = RolloutFusion(),
                                 const MonoStep* mono = nullptr);
template <typename T>
hipError_t launch_arm_mppi_finish_xj(const T* model, const double* records, long n_rec, int H, int A, const double* mean_in,
                                     double* mean_out, const MonoStep& mono, int env_step, unsigned* diag, hipStream_t stream);

}  // namespace mjmpc
