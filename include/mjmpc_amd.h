/*
 * mjmpc_amd - C ABI of the MI355X-native sampling-MPC rollout engine.
 *
 * This is the drop-in boundary UNDER the Python callables the reference exposes
 * (`set_sim_state_fn(state)`, `rollout_fn(P, H, mean, noise, mode)`, and the controllers'
 * `_update_distribution`).  The reference is pure Python, so a maintainer binds this library with
 * ctypes (INTEGRATION.md shows the stub); mjmpc_amd/_lib.py is that binding for this repo.
 *
 * Conventions
 *   - plain C types only; every `d_*` pointer is a DEVICE pointer on the engine's GPU, laid out
 *     exactly like the reference's C-order float64 numpy arrays unless `dtype` says f32;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *     on that stream unless stated otherwise;
 *   - return value: 0 on success, otherwise a negative MJMPC_E_* code or a positive hipError_t;
 *     mjmpc_last_error() gives a thread-local message.
 */
#ifndef MJMPC_AMD_H
#define MJMPC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a blob / state layout, a buffer size or a signature changes incompatibly; mjmpc_amd/_lib.py refuses a
 * library whose mjmpc_abi_version() differs from the one it was written for.
 *   1: rounds 1-3.
 *   2: round 4 - MJMPC_TREE_BLOB_LEN 3116 -> 3929 (solref / solimp scalars replaced by the table of solver sets, new
 *      field order), MJMPC_TREE_STATE_LEN 70 -> 78 (qpos[40] | qvel[32] | target[3] | 3 reserved), tree set / get state
 *      take qpos[nq], and mjmpc_step_tail writes A + 1 doubles into h_action_mapped (action + completion flag). */
/*   3: round 5 - rollouts emulate MuJoCo's reset on instability (finite costs where version 2 returned +inf;
 *      mjmpc_arm_diverged). */
/*   4: round 6 - mjmpc_{arm,tree}_env_resets (resets of the device-resident REAL env, counted apart from the rollouts'),
 *      mjmpc_{arm,tree}_set_reset_returns (+inf returns for particles that reset, as an engine option);
 *      MJMPC_ARM_BLOB_LEN 229 -> 255 (joint type, friction loss, nu: the arm engine takes slide joints, dry friction and
 *      fewer motors than dofs). */
#define MJMPC_ABI_VERSION 4

#define MJMPC_F32 0
#define MJMPC_F64 1

#define MJMPC_E_BADARG (-1)
#define MJMPC_E_BADMODEL (-2)
#define MJMPC_E_NOGPU (-3)

/* Length and layout of the compiled-arm constant block (float64 scalars), produced by
 * mjmpc_amd/models/compile.py::compile_arm and mirrored by mjmpc_amd/csrc/arm_model.h.  Per-link
 * fields are [component][8 lanes]:
 *   off[3][8] axis[3][8] mass[8] com[3][8] inertia[6][8] armature[8] damping[8] range_lo[8]
 *   range_hi[8] limited[8] gear[8] ctrl_lo[8] ctrl_hi[8] dof_invweight0[8] nv timestep frame_skip
 *   site_link site_pos[3] n_sphere sph_link sph_pos[3] sph_r sph_margin sph_invweight plane_n[3]
 *   plane_d sol_K sol_B sol_dmin sol_dmax sol_width sol_mid sol_power gravity[3]
 *   jtype[8] (0 hinge, 1 slide) frictionloss[8] floss_D[8] floss_B nu                           */
#define MJMPC_ARM_BLOB_LEN 255
/* Device state vector of an arm engine: qpos[8] | qvel[8] | target_pos[3]  (float64). */
#define MJMPC_ARM_STATE_LEN 19

typedef struct mjmpc_arm_s* mjmpc_arm_t;

int mjmpc_abi_version(void);
const char* mjmpc_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU / no driver). */
int mjmpc_device_count(void);
/* n_out[0] = kernel nodes, n_out[1] = nodes of any kind of a captured hipGraph_t (a host-side check: the controller
 * compares the captured control iteration with a capture of its own launch tape before it trusts the tape; no reference
 * counterpart). */
int mjmpc_graph_kernel_nodes(void* hip_graph, int64_t* n_out);
/* n_out[0] = an order-independent hash over the graph's kernel nodes of (function, grid, block, dynamic LDS), n_out[1] =
 * the same over all nodes' types: the second half of that check - the tape must launch the same kernels in the same shapes,
 * not merely as many (argument values cannot be read back from a kernel node; they are the recorded ones). */
int mjmpc_graph_signature(void* hip_graph, uint64_t* n_out);

/* ---- arm engine: replaces the SubprocVecEnv worker pool for reacher_7dof-v0 ------------------
 * reference: mjmpc/envs/vec_env/subproc_vec_env.py:91-111 (worker start-up),
 *            mjmpc/envs/gym_env_wrapper.py:16-40 (env construction).                             */
int mjmpc_arm_create(const double* model_blob, int n_blob, int device, mjmpc_arm_t* out);
int mjmpc_arm_destroy(mjmpc_arm_t h);
int mjmpc_arm_dims(mjmpc_arm_t h, int* nv, int* nu, int* d_obs);

/* Dynamics randomization: SubprocVecEnv.randomize_dynamics (subproc_vec_env.py:304-312) gives every
 * worker its own perturbed model (gym_env_wrapper.py:367-416).  Here: n_shards model blocks
 * (float64 [n_shards][MJMPC_ARM_BLOB_LEN], HOST pointer); afterwards particles
 * [k*P/n_shards, (k+1)*P/n_shards) of every rollout use block k (P/n_shards must be a multiple of 8).
 * The single-particle mjmpc_arm_step_state keeps using block 0.  Synchronises the device.        */
int mjmpc_arm_set_shard_models(mjmpc_arm_t h, const double* model_blobs, int n_shards);

/* set_sim_state_fn: SubprocVecEnv.set_env_state (subproc_vec_env.py:235-251) ->
 * Reacher7DOFEnv.set_env_state (mjmpc/envs/basic/reacher_env.py:87-99).  HOST pointers
 * (qpos[nv], qvel[nv], target_pos[3]); copied to the engine's device state on `stream`.         */
int mjmpc_arm_set_state(mjmpc_arm_t h, const double* qpos, const double* qvel, const double* target_pos,
                        void* stream);
/* Per-shard start states: SubprocVecEnv.set_env_state with a list of one state dict per worker
 * (subproc_vec_env.py:242-251).  states = float64 [n_shards][MJMPC_ARM_STATE_LEN] (HOST pointer, layout
 * qpos[8] | qvel[8] | target[3]); afterwards particles of shard k start from states[k] (P / n_shards must
 * be a multiple of 8).  n_shards = 0 returns to the single engine state.                           */
int mjmpc_arm_set_shard_states(mjmpc_arm_t h, const double* states, int n_shards, void* stream);
/* Device pointer to the state vector, for callers that keep the control loop on the GPU.        */
double* mjmpc_arm_state_ptr(mjmpc_arm_t h);

/* rollout_fn: SubprocVecEnv.rollout (subproc_vec_env.py:128-135,161-186) ->
 * GymEnvWrapper.rollout mode="open_loop" (gym_env_wrapper.py:89-156) -> Reacher7DOFEnv.step
 * (reacher_env.py:29-39).  Every particle starts from the engine state.
 *   d_mean     float64 [H][A]                    (always float64: controller state)
 *   d_noise    dtype   [P][H][A]   or NULL      (NULL = mean-only rollout)
 *   d_costs    dtype   [P][H]                    = -reward
 *   d_actions  dtype   [P][H][A]   or NULL      = mean + noise, UNCLIPPED (gym_env_wrapper.py:151)
 *   d_obs      dtype   [P][H][2nv+6] or NULL    observation BEFORE each step
 *   d_next_obs dtype   [P][H][2nv+6] or NULL    observation AFTER each step
 * `dones` are identically zero (reacher_env.py:39) and are not produced.                         */
int mjmpc_arm_rollout(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_mean, const void* d_noise,
                      void* d_costs, void* d_actions, void* d_obs, void* d_next_obs, void* stream);

/* rollout_fn with mode="closed_loop_linear" (gym_env_wrapper.py:135-136): the nominal action of every
 * step is d_weights^T [obs; 1], obs being the observation before the step; d_weights is float64
 * [(2nv+7)][A] (the reference's `mean` argument in that mode).  Other arrays as in mjmpc_arm_rollout. */
int mjmpc_arm_rollout_cl(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_weights, const void* d_noise,
                         void* d_costs, void* d_actions, void* d_obs, void* d_next_obs, void* stream);

/* The same rollout with two optional fusions for a device-resident control iteration:
 *   d_filter_coeffs float64[3] or NULL: d_noise holds RAW samples and the recursive filter of
 *                   control_utils.generate_noise (control_utils.py:32-33) is applied inside the kernel;
 *   d_gseq float64[H] + d_q0 float64[P] (both or neither): d_q0[p] = sum_t gseq[t] * cost[p][t], i.e.
 *                   control_utils.cost_to_go(costs, gamma_seq)[:,0] (control_utils.py:37-46), so the
 *                   update does not have to re-read the costs.  No observations are produced.     */
int mjmpc_arm_rollout_fused(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_mean, const void* d_noise,
                            const double* d_filter_coeffs, const double* d_gseq, void* d_costs, void* d_actions,
                            double* d_q0, void* stream);

/* One Controller.optimize() of MPPI / DMD-MPC without covariance adaptation (controller.py:207-257: generate_rollouts ->
 * _update_distribution -> _get_next_action -> _shift) plus env.step of the closed loop (examples/example_mpc.py:165-168)
 * in TWO launches, for a diagonal action covariance, per-particle softmax weights and no control cost (mppi.py:69-97 with
 * alpha = 1, gaussian_dmd.py:65-104 with update_cov = False):
 *   1. the rollout kernel draws the samples itself - the Philox stream of mjmpc_sample_noise (diag(d_chol) colours it,
 *      d_filter_coeffs float64[3] or NULL filters it as control_utils.py:32-33), keyed by (seed, offset + *d_step_counter,
 *      particle_offset + particle, channel, t) -, keeps its particles' actions in LDS and leaves one softmax record
 *      {max, S, W[H][A]} per workgroup;
 *   2. the finish kernel (H workgroups) merges the records: d_mean_out <- shift((1 - step_size) d_mean + step_size W / S)
 *      (shift_mode 0 'null', 1 'repeat', < 0 none; d_mean_out must NOT alias d_mean, which the launch only reads); the
 *      action (row 0 of the updated mean) goes to d_action_out (device float64[A], may be NULL) and to h_action_slots
 *      (MAPPED PINNED host float64[2][A + 1], may be NULL: slot (*d_step_counter & 1) receives the action and then, as
 *      completion flag, the new step count); *d_step_counter advances; env_step != 0: the engine state advances by one
 *      env step with that action (d_step_cost dtype[1], d_step_next_obs dtype[d_obs], may be NULL).
 *   Sharded runs pass d_record (float64 [2 + H*A]): step 2 then leaves this GPU's record {max, S, W} there and nothing
 *   else (all-gather, then mjmpc_arm_mppi_combine, which also steps the env); d_mean_out is not used.
 * d_gseq float64[H] (gamma_seq, no zero entry).  d_costs / d_actions (dtype [P][H] / [P][H][A]) and d_q0 (float64 [P])
 * are optional outputs.  One model block and one start state only.  shift_mode = -2 issues launch 1 alone (the records
 * stay in the engine): what bench.py times as the dominant kernel.
 * Lifetime / hipGraph rule: every parameter - including the pointer to the engine's per-workgroup record buffer - travels
 * to the two kernels BY VALUE, so a captured graph holds them.  The record buffer only grows and an outgrown buffer stays
 * allocated until mjmpc_arm_destroy, so a graph captured at one (P, H) stays valid after calls at another; a call that
 * would have to GROW the buffer while its stream is capturing returns MJMPC_E_BADARG (call once outside the capture first). */
int mjmpc_arm_mppi_step(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_mean, double* d_mean_out,
                        const double* d_gseq, const double* d_filter_coeffs, const double* d_chol, uint64_t seed,
                        uint64_t offset, int64_t particle_offset, int64_t* d_step_counter, double lam, double step_size,
                        int shift_mode, double* d_action_out, double* h_action_slots, double* d_record, int env_step,
                        void* d_step_cost, void* d_step_next_obs, void* d_costs, void* d_actions, double* d_q0, void* stream);

/* The rollout of mjmpc_arm_mppi_step on its own, for updates other than MPPI's (CEM: cem.py:65-95): the kernel draws its
 * samples itself - the Philox stream of mjmpc_sample_noise keyed by (seed, offset + *d_step_counter, particle_offset +
 * particle, channel, t), coloured by d_chol (chol_full != 0: the whole lower triangle, an adapting full covariance; 0: its
 * diagonal) and filtered by d_filter_coeffs (NULL: none) - and writes costs [P][H] (optional), actions [P][H][A] (optional)
 * and d_q0 float64 [P] = sum_t gseq[t] cost[p][t] (optional).  Graph-capture rule as mjmpc_arm_mppi_step. */
int mjmpc_arm_rollout_sampled(mjmpc_arm_t h, int dtype, int64_t P, int H, const double* d_mean, const double* d_gseq,
                              const double* d_filter_coeffs, const double* d_chol, int chol_full, uint64_t seed, uint64_t offset,
                              int64_t particle_offset, const int64_t* d_step_counter, void* d_costs, void* d_actions,
                              double* d_q0, void* stream);

/* Sharded runs (one rank per GPU): what follows the all-gather of the per-GPU records that mjmpc_arm_mppi_step left
 * (d_record != NULL) - ONE launch that merges the n_records gathered records [max | S | W[H*A]] in rank order (bit-identical
 * on every rank), updates the mean (mppi.py:69-82) into d_mean_out (a buffer of its own) already shifted
 * (olgaussian_mpc.py:116-129), publishes the action into slot (step & 1) of h_action_slots (mapped pinned [2][A+1], value
 * then step count), advances the step counter and, with env_step != 0, steps the device-resident real env.  The
 * parameters travel by value as kernel arguments: a captured graph that holds this launch is valid for as long as the
 * buffers it was given (d_records, d_mean, d_mean_out, the counters, the pinned slots) live. */
int mjmpc_arm_mppi_combine(mjmpc_arm_t h, int dtype, const double* d_records, int n_records, int H, const double* d_mean,
                           double* d_mean_out, int64_t* d_step_counter, double step_size, int shift_mode,
                           double* d_action_out, double* h_action_slots, int env_step, void* d_step_cost,
                           void* d_step_next_obs, void* stream);

/* env.step of the "real" environment kept on the device (examples/example_mpc.py:168 ->
 * Reacher7DOFEnv.step, reacher_env.py:29-39): advances the engine state IN PLACE by one env step
 * under d_action (float64 [A]), writes the step cost (= -reward, dtype[1]) and, if not NULL, the
 * resulting observation (dtype[2nv+6]).                                                          */
int mjmpc_arm_step_state(mjmpc_arm_t h, int dtype, const double* d_action, void* d_cost, void* d_next_obs,
                         void* stream);

/* ---- tree engine: the same worker-pool replacement for models the serial-chain arm engine cannot hold ----------
 * (SURVEY 8f rank 4): a kinematic TREE of up to 32 hinge / slide dofs (the reference's vendored sawyer.xml, swimmer.xml
 * and half_cheetah.xml; a synthetic 24-dof hand, and that hand holding a 6-dof pen): gravity, joint limits and springs,
 * motors or position servos on a subset of the joints, MuJoCo's inertia-box fluid model, up to 16 contact points -
 * sphere- or capsule-end / plane, or sphere / capsule geom-geom pairs - frictionless or with pyramidal friction cones.
 * Reward and observation follow the block's task: 0 = reach (reacher_env.py:29-47, d_obs = 2 nv + 6), 1 = forward
 * progress (swimmer.py:10-24, half_cheetah.py:10-25, d_obs = 2 nv - obs_skip), 2 = reorient an object (the shape of
 * pen-v0's reward, examples/configs/hand/pen-v0.yml:8; d_obs = 2 nv + 6).  Constant block (MJMPC_TREE_BLOB_LEN float64)
 * produced by mjmpc_amd/models/compile_tree.py::compile_tree and mirrored by mjmpc_amd/csrc/tree_model.h; per-link
 * fields are [component][32 lanes], links numbered depth-first:
 *   off[3][32] axis[3][32] mass[32] com[3][32] inertia[6][32] armature damping range_lo range_hi limited gear ctrl_lo
 *   ctrl_hi dof_invweight0 stiffness springref (each [32]) fbox[3][32] frot[9][32] kpg[32] kvg[32] tau0[32] (affine actuator bias at the joint: -gear^2 b1, -gear^2 b2, gear b0; servos: kpg = gear^2 kp)
 *   tau_lo[32] tau_hi[32] (the actuator's forcerange at the joint, +-inf without one) tcoef[32] tpartner[32] tpcoef[32]
 *   (actuators on fixed tendons: the dof's coefficient, the tendon's other dof and its coefficient)
 *   nv timestep frame_skip jumps site_link site_pos[3] n_sphere plane_n[3] plane_d
 *   gravity[3] nu task ctrl_cost obs_skip density viscosity any_friction site_axis[3] target_dir[3]
 *   soltab[8][7] (the model's distinct solver-parameter sets {K, B, dmin, dmax, width, mid, power} - MuJoCo's solref /
 *   solimp after refsafe and clamping; a contact record names its own in slot [21]) dofcls[32] (the dof's limit-row set +
 *   8 * its friction-loss-row set)
 *   spheres[16][24] = {link A, start[3], rA, margin, invweight, mu, capsule axis / segment vector on A [3], depth of
 *   link A in the elimination tree, kind (0 sphere-plane, 1 geom-geom), link B, start on B [3], rB, segment vector on B [3]}
 *   parent subsize anc[5][32] ancmask[2][32] jtype act eparent (parent in the elimination tree of the factorisation)
 *   depth[32] n_rounds elim[31][32]
 *   (elimination lists of the tree-sparse L'DL: the descendants of every link sorted by height, packed
 *   k | distance << 8 | height << 16, -1 ends)
 * Same call shapes and reference counterparts as the arm engine (subproc_vec_env.py:91-111, 128-186, 235-251);
 * target_pos is ignored by task 1.                                                                                  */
#define MJMPC_TREE_BLOB_LEN 3961
/* Round 4, the GENERAL instantiation (block field `gen`; models without these features run the earlier kernels unchanged):
 * ball and free joints (quaternion links: qpos has nq >= nv entries in MuJoCo's layout, d_obs = nq + nv + 6 or nq + nv -
 * obs_skip), joint anchors off the body origin, explicit inertials, box geoms (eight corner points against the plane, one
 * point against a sphere), static geoms of the world body (link -1), friction-loss rows (dof_frictionloss), connect and
 * joint equalities, limits of fixed tendons over one or two joints.  The block then continues
 *   gen nq has_ball frictionloss[32] qadr[32] qoff[32] pext[16][24] qw0[32]
 * (pext: what the new record kinds need beyond spheres[.][24]; layout in csrc/tree_model.h). */
/* A state vector as the C ABI takes it (mjmpc_tree_set_shard_states): MuJoCo's layout, qpos[40] (nq entries used) |
 * qvel[32] | target_pos[3] | 3 reserved (float64). */
#define MJMPC_TREE_STATE_LEN 78
/* The device-resident state vector: one coordinate per link - qpos[32] | qvel[32] | target_pos[3] | site of the fresh
 * observation[3] (filled by mjmpc_tree_rollout_cl) | quaternion w[32] (a ball joint keeps x, y, z in its three links'
 * qpos entries and w here). */
#define MJMPC_TREE_DEVICE_STATE_LEN 102
typedef struct mjmpc_tree_s* mjmpc_tree_t;
int mjmpc_tree_create(const double* model_blob, int n_blob, int device, mjmpc_tree_t* out);
int mjmpc_tree_destroy(mjmpc_tree_t h);
int mjmpc_tree_dims(mjmpc_tree_t h, int* nv, int* nu, int* d_obs);
/* entries of qpos in MuJoCo's layout (nv + one per ball / free joint); mjmpc_tree_set_state / _get_state take and return
 * qpos[nq], qvel[nv] in that layout */
int mjmpc_tree_nq(mjmpc_tree_t h);
/* SubprocVecEnv.randomize_dynamics for the tree engine (subproc_vec_env.py:304-312, gym_env_wrapper.py:367-416): n_shards
 * model blocks [n_shards][MJMPC_TREE_BLOB_LEN] of the engine's topology; from then on shard i of a rollout (particles
 * [i P / n_shards, (i + 1) P / n_shards), P % n_shards == 0) simulates block i. */
int mjmpc_tree_set_shard_models(mjmpc_tree_t h, const double* model_blobs, int n_shards);
/* set_sim_state_fn (subproc_vec_env.py:235-251), asynchronous on `stream` like mjmpc_arm_set_state (pinned staging ring). */
int mjmpc_tree_set_state(mjmpc_tree_t h, const double* qpos, const double* qvel, const double* target_pos, void* stream);
/* One start state per shard, as mjmpc_arm_set_shard_states (SubprocVecEnv.set_env_state with one dict per worker,
 * subproc_vec_env.py:242-251): states = float64 [n_shards][MJMPC_TREE_STATE_LEN] (HOST pointer, layout qpos[40] | qvel[32] |
 * target[3] | 3 unused, MuJoCo's qpos layout); particles of shard k then start from states[k].  With per-shard models the two shard counts
 * must agree.  n_shards = 0 returns to the single engine state. */
int mjmpc_tree_set_shard_states(mjmpc_tree_t h, const double* states, int n_shards, void* stream);
int mjmpc_tree_rollout(mjmpc_tree_t h, int dtype, int64_t P, int H, const double* d_mean, const void* d_noise,
                       void* d_costs, void* d_actions, void* d_obs, void* d_next_obs, void* stream);
/* mjmpc_tree_rollout with the two fusions of mjmpc_arm_rollout_fused (round 4): d_filter_coeffs float64[3] or NULL - d_noise
 * holds RAW samples, filtered inside the kernel (control_utils.py:32-33); d_gseq float64[H] + d_q0 float64[P] (both or
 * neither) - d_q0[p] = sum_t gseq[t] * cost[p][t] = cost_to_go(costs, gamma_seq)[:, 0] (control_utils.py:37-46; +inf for a
 * diverged rollout).  No observations. */
int mjmpc_tree_rollout_fused(mjmpc_tree_t h, int dtype, int64_t P, int H, const double* d_mean, const void* d_noise,
                             const double* d_filter_coeffs, const double* d_gseq, void* d_costs, void* d_actions,
                             double* d_q0, void* stream);
/* The "real" environment kept on the device, as mjmpc_arm_step_state (env.step of the reference's closed loop,
 * examples/example_mpc.py:165-168): one env step from the engine's state with d_action (device float64[nu]), the state
 * advanced in place; d_cost dtype[1], d_next_obs dtype[d_obs] or NULL.  mjmpc_tree_get_state reads qpos / qvel back. */
int mjmpc_tree_step_state(mjmpc_tree_t h, int dtype, const double* d_action, void* d_cost, void* d_next_obs, void* stream);
int mjmpc_tree_get_state(mjmpc_tree_t h, double* qpos, double* qvel, void* stream);
/* rollout(mode="closed_loop_linear") (gym_env_wrapper.py:135-136) as mjmpc_arm_rollout_cl: d_weights float64 [(d_obs + 1)][nu],
 * the nominal action of a step is weights' [observation the step starts from; 1]. */
int mjmpc_tree_rollout_cl(mjmpc_tree_t h, int dtype, int64_t P, int H, const double* d_weights, const void* d_noise,
                          void* d_costs, void* d_actions, void* d_obs, void* d_next_obs, void* stream);
int mjmpc_tree_solver_failures(mjmpc_tree_t h, uint32_t* count);
/* Resets since create: particle-substeps in which MuJoCo's mj_checkPos / mj_checkVel / mj_checkAcc [EXT] would have found a NaN
 * or an entry beyond mjMAXVAL = 1e10 in qpos / qvel / qacc and called mj_resetData (the rollouts of
 * mjmpc/envs/gym_env_wrapper.py:125-153 run through mj_step).  The kernels do what MuJoCo does: the particle goes on from
 * qpos0 with zero velocity, and with zero controls until its env step ends (mj_resetData zeroes data.ctrl, which
 * do_simulation wrote once before its frame_skip substeps), so its costs stay finite.  Not counted by
 * mjmpc_*_solver_failures.  (Since ABI version 3; before, such particles carried a +inf return.) */
int mjmpc_tree_diverged(mjmpc_tree_t h, uint32_t* count);
int mjmpc_arm_diverged(mjmpc_arm_t h, uint32_t* count);
/* The resets among them that happened to the REAL env kept on the device (mjmpc_*_step_state, the env step inside
 * mjmpc_arm_mppi_step / mjmpc_arm_mppi_combine): there the reference does not go on silently - mujoco-py's default warning
 * callback raises MujocoException out of sim.step() in the worker that steps the env (mjmpc/envs/gym_env_wrapper.py:
 * 64-66 -> reacher_env.py:29-39).  The host side (the engines' check_env_resets, mjmpc_amd/envs/_resets.py) reads this counter where it
 * synchronises anyway and raises / warns.  Since ABI version 4. */
int mjmpc_tree_env_resets(mjmpc_tree_t h, uint32_t* count);
int mjmpc_arm_env_resets(mjmpc_arm_t h, uint32_t* count);
/* inf_returns != 0: a rollout particle that resets costs +inf from that env step on (its return +inf: no weight in the
 * softmax updates, last in the elite ranking) instead of the finite costs of MuJoCo's reset state - a blown-up particle whose
 * reset pose happens to be cheap cannot pull the mean towards the action sequence that blew it up.  Default 0 (what the
 * reference's workers return).  Applies to launches issued afterwards; captured launchers keep what they were made with.
 * Since ABI version 4. */
int mjmpc_tree_set_reset_returns(mjmpc_tree_t h, int inf_returns);
int mjmpc_arm_set_reset_returns(mjmpc_arm_t h, int inf_returns);

/* rollout_fn over the reference's two analytic numpy envs (stateless; every pointer is a device
 * pointer): kind 0 = PendulumEnv (mjmpc/envs/basic/pendulum.py:33-50; d_params = [max_speed,
 * max_torque, dt, g, m, l], d_state = [th, thdot], observations have 3 entries), kind 1 = LQREnv
 * (mjmpc/envs/basic/lqr.py:31-35; d_params = [A | B | Q | R] row-major, d_state = x[n_state],
 * n_state, n_action <= 8).  Arrays as in mjmpc_arm_rollout (costs = -reward).  closed_loop_linear != 0:
 * d_mean is the (d_obs+1, n_action) weight matrix of mode "closed_loop_linear"
 * (mjmpc/envs/gym_env_wrapper.py:135-136), action = W^T [obs; 1] + noise.                          */
#define MJMPC_ENV_PENDULUM 0
#define MJMPC_ENV_LQR 1
int mjmpc_analytic_rollout(int kind, const double* d_params, int n_state, int n_action, const double* d_state, int dtype,
                           int64_t P, int H, const double* d_mean, const void* d_noise, void* d_costs, void* d_actions,
                           void* d_obs, void* d_next_obs, int closed_loop_linear, void* stream);

/* Number of (particle, substep) constraint solves whose active set had not settled after the
 * iteration cap since engine creation (synchronises the device).  0 in every test.               */
int mjmpc_arm_solver_failures(mjmpc_arm_t h, uint32_t* count);

/* ---- the exchange of a sharded run, issued from the library ---------------------------------------
 * reference: SubprocVecEnv gathers its workers' results through pipes (mjmpc/envs/vec_env/subproc_vec_env.py:161-186); here
 * each rank (one process per GPU) contributes one small float64 record per control iteration and every rank combines the
 * gathered records identically.  The all-gather is RCCL's, on the iteration's stream; issuing it through this
 * ABI instead of torch.distributed makes the sharded iteration a sequence of LIBRARY calls, which runs from the launch tape
 * / as direct launches like the one-GPU loop (no hipGraph replay gap).  RCCL is bound at run time (dlopen of the copy the
 * process already holds); without it these entries fail with MJMPC_E_NOGPU and callers keep torch.distributed.
 *   mjmpc_comm_unique_id   rank 0: 128 opaque bytes (ncclGetUniqueId) to hand to every rank by any means
 *   mjmpc_comm_create      every rank, collectively (ncclCommInitRank) on HIP device `device`
 *   mjmpc_comm_all_gather_f64   d_recv[world][count] <- every rank's d_send[count] (device pointers; may be captured)  */
#define MJMPC_COMM_ID_BYTES 128
typedef struct mjmpc_comm_s* mjmpc_comm_t;
int mjmpc_comm_unique_id(void* id_out);
int mjmpc_comm_create(const void* id_bytes, int world_size, int rank, int device, mjmpc_comm_t* out);
int mjmpc_comm_all_gather_f64(mjmpc_comm_t c, const double* d_send, double* d_recv, int64_t count, void* stream);
int mjmpc_comm_destroy(mjmpc_comm_t c);

/* ---- sampling-distribution updates: the device side of Controller._update_distribution --------
 * All functions below are stateless; `d_ws` is a caller-owned device workspace of at least
 * mjmpc_update_workspace_bytes(P, H, A) bytes that carries intermediate results between the
 * calls of one update (per-particle costs, elite flags).  P is the number of particles ON THIS
 * GPU.  "Records" are small float64 vectors designed to be all-gathered across GPUs (one
 * collective per control iteration) and combined identically on every rank:
 *   softmax record  [ xmax[Hw] | S[Hw] | W[H*A] | C[A*A] ]   Hw = time_based_weights ? H : 1
 *   CEM sum record  [ n_elite_local | sum over local elites of actions[H*A] ]
 *   CEM cov record  [ sum over local elites and t of (d - dbar)(d - dbar)^T  [A*A] ]
 *   RS record       [ min q0 | global particle index | actions[H*A] of that particle ]
 * d_gseq is Controller.gamma_seq (controller.py:71) as float64[H]; gamma_zero = any(gseq == 0)
 * (control_utils.cost_to_go returns its input unchanged in that case, control_utils.py:41-42).   */
int64_t mjmpc_update_workspace_bytes(int64_t P, int H, int A);
int mjmpc_softmax_record_len(int H, int A, int time_based_weights);

/* cost_to_go (mjmpc/utils/control_utils.py:37-46) of the local particles, [:,0] column, left in
 * the workspace for the CEM / random-shooting calls; mjmpc_workspace_q0 returns its device address. */
int mjmpc_traj_cost(int dtype, int64_t P, int H, int A, const void* d_costs, const double* d_gseq, int gamma_zero,
                    void* d_ws, void* stream);
double* mjmpc_workspace_q0(void* d_ws, int64_t P, int H, int A);

/* MPPIQ.calculate_returns + _control_costs (mjmpc/control/mppiq.py:104-136): TD(lambda) returns
 * d_returns (dtype [P][H]) from the per-step costs (+ beta * per-step control cost when alpha == 0; then
 * d_actions, d_mean, d_covinv are read) and the optional Q estimates d_qvals (dtype [P][H]; NULL = the
 * reference's default: zero except the last step's own cost).  d_wseq (float64 [H-1]) =
 * cumprod(1, gamma*td_lam, ...) (mppiq.py:120); wseq_has_zero selects cost_to_go's pass-through branch
 * (control_utils.py:39-40).  Feed d_returns to mjmpc_softmax_stats with gamma_zero = 1, alpha = 1.   */
int mjmpc_td_lambda_returns(int dtype, int64_t P, int H, int A, const void* d_costs, const void* d_actions,
                            const void* d_qvals, const double* d_mean, const double* d_covinv, const double* d_wseq,
                            int wseq_has_zero, double beta, int alpha, double gamma, double td_lam, void* d_returns,
                            void* d_ws, void* stream);

/* MPPI._exp_util + _control_costs (mjmpc/control/mppi.py:84-111), DMDMPC._exp_util
 * (gaussian_dmd.py:94-104), PFMPC._exp_util (particle_filter_controller.py:104-113): exponentiated
 * cost weights reduced to this GPU's softmax record.  d_covinv (float64 [A][A]) is read only when
 * alpha == 0 (control cost on); want_cov adds the weighted scatter C used by DMD-MPC.             */
int mjmpc_softmax_stats(int dtype, int64_t P, int H, int A, const void* d_costs, const void* d_actions,
                        const double* d_mean, const double* d_covinv, const double* d_gseq, int gamma_zero,
                        double lam, int alpha, int time_based_weights, int want_cov, double* d_record, void* d_ws,
                        void* stream);
/* MPPI._update_distribution (mppi.py:69-82) / DMDMPC._update_distribution (gaussian_dmd.py:65-91)
 * / _calc_val (mppi.py:113-131, gaussian_dmd.py:126-139) from G gathered records.
 * cov_mode: 0 leave cov, 1 diagonal update, 2 full update.  d_cov, d_value, d_wnorm may be NULL;
 * d_wnorm receives {global max, global normaliser} for mjmpc_softmax_weights.                     */
int mjmpc_softmax_combine(const double* d_records, int G, int H, int A, int time_based_weights, double lam,
                          double step_size, int cov_mode, double P_total, double* d_mean, double* d_cov,
                          double* d_value, double* d_wnorm, void* stream);
/* normalised per-particle weights (PFMPC resampling input), float64 [P] */
int mjmpc_softmax_weights(int64_t P, int H, int A, const double* d_wnorm, void* d_ws, double* d_weights,
                          void* stream);

/* CEM._update_distribution (mjmpc/control/cem.py:63-86) in three steps.  Elite = the k particles
 * with the smallest (q0, global index); d_q_all is the all-gathered q0 of every GPU (NULL = this
 * GPU holds all particles; `offset` is then ignored) and `offset` the global index of local particle 0.
 * The local elite rows are kept as a list in d_ws, which mjmpc_cem_elite_cov reads: call them in this order
 * on the same workspace.                                                                             */
int mjmpc_cem_elite_sums(int dtype, int64_t P, int H, int A, const void* d_actions, const double* d_q_all,
                         int64_t P_all, int64_t offset, int64_t k, double* d_sum_record, void* d_ws, void* stream);
int mjmpc_cem_elite_cov(int dtype, int64_t P, int H, int A, const void* d_actions, const double* d_mean,
                        const double* d_sum_records, int G, double* d_cov_record, void* d_ws, void* stream);
int mjmpc_cem_final(const double* d_cov_records, int G, int64_t P, int H, int A, double n_elite, int full_cov,
                    double step_size, double* d_mean, double* d_cov, void* d_ws, void* stream);
/* The sharded form with ONE record exchange after the q0 gather (two collectives per iteration): each GPU
 * calls mjmpc_cem_elite_sums, then mjmpc_cem_elite_cov with ITS OWN sum record (G = 1: scatter about its own mean
 * delta) and all-gathers  rec_g = { sum record [1 + H*A] | cov record [A*A] };  this call pools them (pairwise-
 * variance identity; exactly the two-pass np.cov / np.var of cem.py:76-80 when G = 1) and updates mean and cov.   */
int mjmpc_cem_combine(const double* d_records, int G, int H, int A, double n_elite, int full_cov, double step_size,
                      double* d_mean, double* d_cov, void* stream);

/* The fused CEM step (round 4; cem.py:65-95 in two launches beside the rollout instead of nine):
 *   mjmpc_cem_select_moments  the elite threshold (k-th smallest q0, ties by particle index, exactly as
 *       mjmpc_cem_elite_sums), the elite list, and per workgroup {rows, sum of elite action rows, scatter of their deltas
 *       about a provisional centre} - every workgroup repeats the selection and takes a slice of the elite rows; it also
 *       snapshots mean, cov and *d_step_counter for the finish launch.  q0 is read from d_ws (mjmpc_workspace_q0) or,
 *       sharded, from the gathered d_q_all [P_all] with this GPU's block at `offset`;
 *   mjmpc_cem_record          (sharded runs) the partials -> this GPU's record {n | sum a [H*A] | scatter about its own
 *       mean [A*A]}, the layout mjmpc_cem_combine pools; all-gather it;
 *   mjmpc_cem_finish          d_records == NULL: one GPU, the partials in d_ws; else the G gathered records.  New mean and
 *       covariance (np.var ddof 0 / np.cov ddof 1 over the H k elite deltas, step-size blend), cov += grow_scale *
 *       diag(d_grow_diag) (CEM._shift, cem.py:94; NULL: identity), its lower Cholesky factor -> d_chol (may be NULL;
 *       positive semi-definite input as mjmpc_cholesky_lower, *d_status = 1 if indefinite), the action (row 0 of the new
 *       mean) -> d_action_out / h_action_pinned (mapped pinned, A + 1 doubles, may be NULL: the action, then - behind a
 *       system-scope fence - the new step count as a completion flag the host can poll while the launch is still
 *       drawing), the horizon shift (0 'null', 1 'repeat', < 0 none), *d_step_counter = snapshot + 1, and - d_next_noise != NULL - the RAW Philox samples of the next
 *       control step, [P][H][A] of dtype, coloured by the new factor: the stream of mjmpc_sample_noise(..., filter NULL,
 *       seed, offset, particle_offset, d_step_counter), sample for sample.
 * mjmpc_cem_fused_supported: A <= 8, A <= H + 1, P_all <= 32768, the moment tiles fit the workspace (else use the
 * separate entries above).  Same workspace for the three calls; d_mean / d_cov as given to the first. */
int mjmpc_cem_fused_supported(int64_t P_all, int64_t P, int64_t k, int H, int A);
int mjmpc_cem_select_moments(int dtype, int64_t P, int H, int A, const void* d_actions, const double* d_q_all, int64_t P_all,
                             int64_t offset, int64_t k, const double* d_mean, const double* d_cov,
                             const int64_t* d_step_counter, void* d_ws, void* stream);
int mjmpc_cem_record(int64_t P, int H, int A, int64_t k, const double* d_mean, double* d_record, void* d_ws, void* stream);
int mjmpc_cem_finish(int dtype, int64_t P, int H, int A, int64_t k, const double* d_records, int G, double n_elite,
                     int full_cov, double step_size, int shift_mode, double* d_mean, double* d_cov, double* d_chol,
                     int* d_status, const double* d_grow_diag, double grow_scale, double* d_action_out,
                     double* h_action_pinned, int64_t* d_step_counter, void* d_next_noise, uint64_t seed, uint64_t offset,
                     int64_t particle_offset, void* d_ws, void* stream);

/* RandomShooting._update_distribution (mjmpc/control/random_shooting.py:52-62). */
int mjmpc_rs_best(int dtype, int64_t P, int H, int A, const void* d_actions, int64_t offset, double* d_record,
                  void* d_ws, void* stream);
int mjmpc_rs_combine(const double* d_records, int G, int H, int A, double step_size, double* d_mean, void* stream);

/* MPPI._update_distribution (mppi.py:69-82, time_based_weights off, alpha == 1) + the action read-out
 * (olgaussian_mpc.py:71) + OLGaussianMPC._shift (olgaussian_mpc.py:116-129) in two launches.
 * d_q0: float64[P] cost-to-go (NULL = the one mjmpc_traj_cost left in d_ws).  shift_mode: -1 none,
 * 0 'null', 1 'repeat'.  d_action_out (float64[A]), d_record ([xmax | S | W[H*A]], the softmax record
 * without its covariance block) and d_value (_calc_val, mppi.py:113-131) may be NULL.  For a captured
 * control iteration: *d_step_counter (device int64) is incremented (the noise stream index of the next step),
 * and h_action_mapped (device-visible pinned host memory, float64[A+1]) also receives the action followed by
 * the new step count, written last behind a system-scope fence: a host polling entry [A] can read the action
 * as soon as this kernel has produced it, while later work of the same stream is still running.            */
int mjmpc_mppi_fused_update(int dtype, int64_t P, int H, int A, const double* d_q0, const void* d_actions, double lam,
                            double step_size, int shift_mode, double* d_mean, double* d_action_out, double* d_record,
                            double* d_value, double* h_action_mapped, int64_t* d_step_counter, void* d_ws,
                            void* stream);

/* The same, and in the same two launches the RAW samples of the next control step are drawn into d_next_noise
 * (dtype [P][H][A], unfiltered: mjmpc_arm_rollout_fused filters on the fly) - extra workgroups of the first
 * launch, as mjmpc_sample_noise(dtype, d_next_noise, P, H, A, d_chol, NULL, seed, offset, particle_offset,
 * d_step, chol_is_diagonal) would draw them.  The rollout that read d_next_noise has finished by then, and
 * *d_step still holds the current step when it is read (the counter moves in the second launch), so `offset`
 * is normally 1.  Saves the sampler's own launch on the critical path of a captured control iteration.     */
int mjmpc_mppi_fused_update_draw_next(int dtype, int64_t P, int H, int A, const double* d_q0, const void* d_actions,
                                      double lam, double step_size, int shift_mode, double* d_mean,
                                      double* d_action_out, double* d_record, double* d_value, double* h_action_mapped,
                                      int64_t* d_step_counter, void* d_ws, void* d_next_noise, const double* d_chol,
                                      uint64_t seed, uint64_t offset, int64_t particle_offset, const int64_t* d_step,
                                      int chol_is_diagonal, void* stream);

/* Sharded MPPI: the G all-gathered records d_records (float64 [G][2 + H*A], as left in d_record by
 * mjmpc_mppi_fused_update with step_size 0, shift_mode -1) merged in rank order -> mean update, action read-out,
 * shift, step counter and the mapped host copy with its completion flag, exactly as the single-GPU call does
 * (P_total = particles over all ranks; every rank computes the bit-identical result).                      */
int mjmpc_mppi_fused_combine(const double* d_records, int G, double P_total, int H, int A, double lam, double step_size,
                             int shift_mode, double* d_mean, double* d_action_out, double* d_value,
                             double* h_action_mapped, int64_t* d_step_counter, void* stream);

/* sum of q0 over local particles (CEM / RandomShooting _calc_val: cem.py:107-112) -> d_out[0] */
int mjmpc_q0_sum(int64_t P, int H, int A, double* d_out, void* d_ws, void* stream);

/* OLGaussianMPC._shift (mjmpc/control/olgaussian_mpc.py:116-129).  mode 0 'null', 1 'repeat',
 * 2 the appended row is read from d_row ('random': drawn by the host from np.random).            */
int mjmpc_shift_mean(double* d_mean, int H, int A, int mode, const double* d_row, void* stream);
/* The tail of Controller.optimize (controller.py:240-257: `_get_next_action`, `num_steps += 1`, `_shift`) in ONE launch
 * for device-resident controllers: d_action_out (float64 [A]) / h_action_mapped (float64 [A + 1], mapped pinned host
 * memory: the action, then - behind a system-scope fence - the new step count as a completion flag; 0 without a
 * counter) <- mean[0], either may be NULL; the shift of mjmpc_shift_mean; *d_step_counter += 1 (may be NULL); and, when
 * d_cov is not NULL, cov += cov_scale * diag(d_cov_diag) as in mjmpc_cov_add_diag (cem.py:89-95,
 * gaussian_dmd.py:107-113).  A <= 64.                                                                          */
int mjmpc_step_tail(double* d_mean, int H, int A, int shift_mode, const double* d_row, double* d_action_out,
                    double* h_action_mapped, int64_t* d_step_counter, double* d_cov, const double* d_cov_diag,
                    double cov_scale, void* stream);

/* Covariance kept on the device (CEM cem.py:75-95, DMDMPC with update_cov gaussian_dmd.py:77-113): the lower
 * Cholesky factor d_chol (float64 [A][A]) that mjmpc_sample_noise colours its normals with, computed from the
 * device-resident d_cov (np.linalg.cholesky's role inside control_utils.generate_noise:28-29); *d_status (device
 * int, may be NULL) is set to 1 if d_cov is not positive definite.  mjmpc_cov_add_diag: cov += scale * diag(d_diag)
 * (the shift's `+ beta * diag(init_cov)` / `+ beta * I`; d_diag NULL = identity).  A <= 64.                  */
int mjmpc_cholesky_lower(const double* d_cov, int A, double* d_chol, int* d_status, void* stream);
int mjmpc_cov_add_diag(double* d_cov, int A, const double* d_diag, double scale, void* stream);

/* The recursive filter of generate_noise alone (control_utils.py:32-33), in place on d_noise [P][H][A]. */
int mjmpc_filter_noise(int dtype, void* d_noise, int64_t P, int H, int A, const double* d_coeffs, void* stream);
/* noise[row][:] <- noise[row][:] B for `rows` rows of A scalars, B float64 [A][A] row-major: the colouring step of
 * np.random.multivariate_normal (control_utils.py:30), B = sqrt(s)[:, None] * v from the SVD of cov (host, LAPACK),
 * applied to the standard-normal stream mjmpc_sample_noise_mt19937[_jump] regenerates with scale 1.          */
int mjmpc_color_noise(int dtype, void* d_noise, int64_t rows, int A, const double* d_B, void* stream);

/* control_utils.generate_noise (mjmpc/utils/control_utils.py:24-34), SEED-IDENTICAL mode for an isotropic
 * covariance c*I: d_noise[0..n_normals) = scale * the numpy legacy stream `np.random.seed(seed + *d_step);
 * standard_normal(n_normals)` (MT19937 + polar method), regenerated on the device, unfiltered (C order over
 * (P,H,A); scale = sqrt(c)).  Stream alignment is exact; values agree with numpy to <= 2 ulp (log()).
 * d_ws: mjmpc_mt19937_workspace_bytes(n_normals) bytes, 16-byte aligned.  *d_status (device int, may be
 * NULL) is set to 1 if the generated margin of polar attempts did not suffice.                    */
int64_t mjmpc_mt19937_workspace_bytes(int64_t n_normals);
int mjmpc_sample_noise_mt19937(int dtype, void* d_noise, int64_t n_normals, double scale, uint64_t seed,
                               const int64_t* d_step, void* d_ws, int* d_status, void* stream);

/* The same stream produced in parallel.  MT19937's recurrence is serial, but its state transition is linear
 * over GF(2): the state J words ahead is the XOR of the word windows x[i .. i+624) over the set bits i of
 * t^J mod phi (phi = the generator's characteristic polynomial).  One workgroup generates the first
 * head_words (a multiple of 4 in [19936, 19968]) words serially; workgroup g of n_segments (<= 64) then starts directly at word
 * head_words + g*seg_words.  d_jump_idx / d_jump_starts[n_segments+1] (device int32) hold the set-bit lists
 * of t^(head_words + g*seg_words) mod phi, g >= 1 (entry 0 empty), as computed by
 * mjmpc_amd/control/mt_jump.py; the segments must cover mjmpc_mt19937_stream_words(n_normals) words.
 * n_segments == 0 behaves as mjmpc_sample_noise_mt19937.  Output is bit-identical for any segmentation.
 * first_normal > 0 (particle sharding): d_noise[0..n_normals) receives normals [first_normal, first_normal +
 * n_normals) of the one global stream - the polar method's rejections make positions data dependent, so a rank
 * regenerates the stream up to the end of its block; workspace, segments and stream_words are then those of
 * first_normal + n_normals.                                                                                   */
int64_t mjmpc_mt19937_stream_words(int64_t n_normals);
int mjmpc_sample_noise_mt19937_jump(int dtype, void* d_noise, int64_t n_normals, double scale, uint64_t seed,
                                    const int64_t* d_step, const int32_t* d_jump_idx, const int32_t* d_jump_starts,
                                    int64_t head_words, int64_t seg_words, int n_segments, int64_t first_normal,
                                    void* d_ws, int* d_status, void* stream);

/* control_utils.generate_noise (mjmpc/utils/control_utils.py:24-34), performance mode: Philox
 * normals coloured by the lower Cholesky factor d_chol (float64 [A][A]) and filtered in place with
 * d_coeffs (float64 [3]; NULL leaves the samples raw for mjmpc_arm_rollout_fused to filter).  Same distribution as the reference, different bit stream; `offset`
 * plays the role of num_steps in base_seed = seed_val + num_steps (olgaussian_mpc.py:91);
 * `particle_offset` is the global index of local particle 0, so that a sharded run draws exactly
 * the samples a single GPU would draw for the same particles.  d_step (device int64, may be NULL)
 * is added to `offset` on the device, so that a captured hipGraph can advance the stream itself.
 * chol_is_diagonal != 0 promises that d_chol has no off-diagonal entries (skips the colouring loop). */
int mjmpc_sample_noise(int dtype, void* d_noise, int64_t P, int H, int A, const double* d_chol,
                       const double* d_coeffs, uint64_t seed, uint64_t offset, int64_t particle_offset,
                       const int64_t* d_step, int chol_is_diagonal, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MJMPC_AMD_H */
