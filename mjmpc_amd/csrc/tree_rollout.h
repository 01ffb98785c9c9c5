#pragma once
#include <hip/hip_runtime.h>

namespace mjmpc {

// Fused (particles x horizon x frame_skip) rollout of a compiled hinge / slide TREE (up to 32 dofs, up to 16
// sphere/plane contact points, frictionless or with pyramidal friction cones); see tree_rollout.hip.  max_path = links
// on the longest root-to-leaf path; full = the model needs the instantiation with slide joints / springs / friction
// cones / more than 8 contact points / fluid forces (with nv <= 16 it runs 16 lanes per particle).  n_shards > 1: model holds one block per shard of P / n_shards
// consecutive particles (dynamics randomization).  state_out (P = 1 only): the particle's final qpos / qvel are written
// there in the layout of `state` (the device-resident real env).  clw: closed_loop_linear weights f64 [(d_obs + 1)][A]
// instead of `mean` (the fresh observation's site is read from state[2 * 32 + 3 ...], which a P = 1 launch with site_out
// pointing there provides).  model: TREE_BLOB_LEN scalars of T; state: f64 TREE_STATE_LEN (tree_model.h);
// gen: the model block's T_GEN - 1: the general instantiation (ball / free joints, friction loss, boxes, equalities), 2: with round
// 5's record kinds on top (a cylinder on the plane, capsule / box and box / box pairs, mjc_PlaneBox's corner rule);
// mean f64 [H][A]; noise / cost / act / obs / nobs of T in the reference's C-order layouts (may be null except cost).
// n_state_shards > 1: `state` holds one TREE_STATE_LEN vector per shard (per-worker start states); with both kinds of
// shards their counts must agree.
// Fusions riding in the rollout launch (as RolloutFusion of the arm kernel): the reference's recursive noise filter applied
// to the raw samples on the fly (control_utils.py:32-33; filt = three float64 coefficients), and the discounted cost-to-go
// of every particle, q0_out[p] = sum_t gseq[t] * cost[p][t] (control_utils.py:37-46 at t = 0; +inf for a diverged rollout).
// MuJoCo's reset on instability (mj_checkPos / mj_checkVel / mj_checkAcc -> mj_resetData, [EXT]; the rollouts of
// gym_env_wrapper.py:125-153 run through it): reset_rec = one record per model shard (reset_stride apart; 0: one record) -
// the device state vector one substep after the reset state (qpos0, zero velocity, zero controls), then site[3] and object
// axis[3] AT the reset state (TREE_RESET_LEN scalars, tree_model.h).  A particle whose qpos / qvel hold a NaN or an entry
// beyond 1e10 when a substep begins restarts that substep from the reset state; one whose acceleration does continues from
// the record; either way its controls are zero for the rest of the env step (mj_resetData zeroes data.ctrl, which
// do_simulation wrote once before its frame_skip calls of sim.step()).  nullptr: no resets (the launch that MAKES the
// record: one particle, one step of a model block with frame_skip 1, axis_out receiving the axis).
struct TreeFusion {
    const double* filt = nullptr;
    const double* gseq = nullptr;
    double* q0_out = nullptr;
    const double* reset_rec = nullptr;
    int reset_stride = 0;
    double* axis_out = nullptr;
    int inf_on_reset = 0;           // 1: a particle that has reset costs +inf from that env step on (RolloutFusion::inf_on_reset)
};
// diag (unsigned[]): [0] iteration-cap hits, [1] resets (live particles), developer clocks from byte 8 (TREE_STATS builds),
// [TREE_DIAG_ENV_RESETS] resets of the REAL env (launches with state_out: mjmpc_tree_step_state, the control iteration's env step)
constexpr int TREE_STAT_SLOTS = 48;         // 64-bit developer clocks / counts of -DTREE_STATS builds (tools/tree_stats.py)
constexpr int TREE_DIAG_ENV_RESETS = 2 + 2 * TREE_STAT_SLOTS;

template <typename T>
hipError_t launch_tree_rollout(const T* model, int n_model_shards, int max_path, bool full, int nv, const double* state, long P, int H,
                               int A, const double* mean, const T* noise, T* cost, T* act, T* obs, T* nobs, unsigned* diag,
                               hipStream_t stream, double* state_out = nullptr, const double* clw = nullptr,
                               double* site_out = nullptr, int n_state_shards = 1, int gen = 0,
                               TreeFusion fuse = TreeFusion());

}  // namespace mjmpc
