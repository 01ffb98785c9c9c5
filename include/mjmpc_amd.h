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

#define MJMPC_ABI_VERSION 1

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
 *   plane_d sol_K sol_B sol_dmin sol_dmax sol_width sol_mid sol_power gravity[3]               */
#define MJMPC_ARM_BLOB_LEN 229
/* Device state vector of an arm engine: qpos[8] | qvel[8] | target_pos[3]  (float64). */
#define MJMPC_ARM_STATE_LEN 19

typedef struct mjmpc_arm_s* mjmpc_arm_t;

int mjmpc_abi_version(void);
const char* mjmpc_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU / no driver). */
int mjmpc_device_count(void);

/* ---- arm engine: replaces the SubprocVecEnv worker pool for reacher_7dof-v0 ------------------
 * reference: mjmpc/envs/vec_env/subproc_vec_env.py:91-111 (worker start-up),
 *            mjmpc/envs/gym_env_wrapper.py:16-40 (env construction).                             */
int mjmpc_arm_create(const double* model_blob, int n_blob, int device, mjmpc_arm_t* out);
int mjmpc_arm_destroy(mjmpc_arm_t h);
int mjmpc_arm_dims(mjmpc_arm_t h, int* nv, int* nu, int* d_obs);

/* set_sim_state_fn: SubprocVecEnv.set_env_state (subproc_vec_env.py:235-251) ->
 * Reacher7DOFEnv.set_env_state (mjmpc/envs/basic/reacher_env.py:87-99).  HOST pointers
 * (qpos[nv], qvel[nv], target_pos[3]); copied to the engine's device state on `stream`.         */
int mjmpc_arm_set_state(mjmpc_arm_t h, const double* qpos, const double* qvel, const double* target_pos,
                        void* stream);
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

/* Number of (particle, substep) constraint solves whose active set had not settled after the
 * iteration cap since engine creation (synchronises the device).  0 in every test.               */
int mjmpc_arm_solver_failures(mjmpc_arm_t h, uint32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* MJMPC_AMD_H */
