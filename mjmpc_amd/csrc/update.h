#pragma once
#include <hip/hip_runtime.h>

#include "noise_device.h"

namespace mjmpc {

// Workspace (in doubles) the update launchers need for P particles, horizon H, action dim A.
long update_workspace_doubles(long P, int H, int A);
double* workspace_q0(double* ws, long P, int H, int A);

// x = (-1/lam) (cost_to_go + lam * control cost-to-go) and q0 = cost_to_go[:,0] into the workspace.
template <typename T>
hipError_t traj_cost(const T* costs, const T* actions, const double* mean, const double* covinv, const double* gseq,
                     int gamma_zero, double lam, int alpha, int tbw, long P, int H, int A, double* ws, hipStream_t s);

// MPPIQ.calculate_returns (mppiq.py:104-126): TD(lambda) returns out[P][H] from per-step costs (+ beta * control
// cost when alpha == 0) and optional Q estimates qvals[P][H]; wseq[H-1] = cumprod(1, gamma*td_lam, ...).
template <typename T>
hipError_t td_lambda_returns(const T* costs, const T* actions, const T* qvals, const double* mean, const double* covinv,
                             const double* wseq, int wseq_zero, double beta, int alpha, double gamma, double td_lam,
                             long P, int H, int A, T* out, double* ws, hipStream_t s);

// Softmax record of this GPU's particles: [xmax[Hw] | S[Hw] | W[H*A] | C[A*A]], Hw = tbw ? H : 1.
template <typename T>
hipError_t softmax_stats(const T* costs, const T* actions, const double* mean, const double* covinv,
                         const double* gseq, int gamma_zero, double lam, int alpha, int tbw, int want_cov, long P,
                         int H, int A, double* record, double* ws, hipStream_t s);
hipError_t softmax_combine(const double* records, int G, int H, int A, int tbw, double lam, double step, int cov_mode,
                           double P_total, double* mean, double* cov, double* value, double* wnorm, hipStream_t s);
hipError_t softmax_weights(long P, const double* wnorm, double* ws, int H, int A, double* weights, hipStream_t s);

// CEM: elite flags by global rank (needs q0 in the workspace), then {count, sum a} / centred scatter records.
template <typename T>
hipError_t cem_elite_sums(const T* actions, const double* q_all, long P_all, long offset, long k, long P, int H, int A,
                          double* record, double* ws, hipStream_t s);
template <typename T>
hipError_t cem_elite_cov(const T* actions, const double* mean, const double* sum_records, int G, long P, int H, int A,
                         double* crecord, double* ws, hipStream_t s);
hipError_t cem_final(const double* crecords, int G, long P, int H, int A, double n_elite, int full, double step,
                     double* mean, double* cov, double* ws, hipStream_t s);
// records[g] = {n_g | sum_elite a [H*A] | scatter of rank g's elite deltas about ITS OWN mean [A*A]} -> mean, cov
hipError_t cem_combine(const double* records, int G, int H, int A, double n_elite, int full, double step, double* mean,
                       double* cov, hipStream_t s);

// The fused CEM step (round 4): selection + elite list + moments in ONE launch (every workgroup repeats the selection and
// takes a slice of the elite rows), then - sharded runs only - this GPU's record for the gather, then ONE finish launch
// (refit, covariance growth, Cholesky factor, action, shift, step counter, the next step's raw samples).
bool cem_fused_supported(long P_all, long P, long k, int H, int A);
template <typename T>
hipError_t cem_select_moments(const T* actions, const double* q_all, long P_all, long offset, long k, long P, int H, int A,
                              const double* mean, const double* cov, const long long* d_step, double* ws, hipStream_t s);
hipError_t cem_record(long k, long P, int H, int A, const double* mean, double* record, double* ws, hipStream_t s);
// records == nullptr: one GPU, the partials in the workspace; else G gathered records.  noise == nullptr: no draw.
template <typename T>
hipError_t cem_finish(const double* records, int G, long k, long P, int H, int A, double n_elite, int full, double step,
                      int shift_mode, double* mean, double* cov, double* chol, int* status, const double* grow_diag,
                      double grow_scale, double* action_out, double* action_host, long long* step_counter, T* noise,
                      unsigned long long seed, unsigned long long offset, long particle_offset, double* ws, hipStream_t s);

// Random shooting: {min q0, global index, action[H*A]} record and the combine.
template <typename T>
hipError_t rs_best(const T* actions, long offset, long P, int H, int A, double* record, double* ws, hipStream_t s);
hipError_t rs_combine(const double* records, int G, int H, int A, double step, double* mean, hipStream_t s);

// Fused MPPI update (weights per particle, control cost off): q0 -> mean, action, shift in two launches.
// q0 == nullptr: use the cost-to-go traj_cost left in the workspace.  next: also draw the raw samples of the next
// control step (extra workgroups of the first launch).
template <typename T>
hipError_t mppi_fused_update(const double* q0, const T* actions, double lam, double step, int shift_mode, long P, int H,
                             int A, double* mean, double* action_out, double* record, double* value, double* ws,
                             hipStream_t s, double* action_host = nullptr, long long* step_counter = nullptr,
                             const NextNoise* next = nullptr);

// the all-gathered records of G GPUs -> mean, action (device + mapped host copy with completion flag), step counter, shift
hipError_t mppi_fused_combine(const double* records, int G, double P_total, double lam, double step, int shift_mode,
                              int H, int A, double* mean, double* action_out, double* value, double* action_host,
                              long long* step_counter, hipStream_t s);

hipError_t q0_sum(long P, int H, int A, double* out, double* ws, hipStream_t s);
hipError_t shift_mean(double* mean, int H, int A, int mode, const double* row, hipStream_t s);
// action read-out (device + mapped host copy), shift, step counter + 1, cov += scale * diag(d): one launch
hipError_t step_tail(double* mean, int H, int A, int mode, const double* row, double* action_out, double* action_host,
                     long long* step_counter, double* cov, const double* d, double scale, hipStream_t s);
// lower Cholesky factor of a device-resident covariance (A <= 64); cov += scale * diag(d) (d null: identity)
hipError_t cholesky_lower(const double* cov, int A, double* chol, int* status, hipStream_t s);
hipError_t cov_add_diag(double* cov, int A, const double* d, double scale, hipStream_t s);

// Noise: standard normals from Philox4x32-10, coloured by L (A x A lower Cholesky factor of cov), then the
// in-place AR filter of control_utils.py:32-33, written in the reference's (P,H,A) layout.
template <typename T>
hipError_t sample_noise(T* noise, long P, int H, int A, const double* chol, const double* coeffs,
                        unsigned long long seed, unsigned long long offset, long particle_offset, const long long* d_step,
                        hipStream_t s, int diag_only = 0);

// the in-place recursive 3-tap filter of control_utils.py:32-33 on its own
template <typename T>
hipError_t filter_noise(T* noise, long P, int H, int A, const double* coeffs, hipStream_t s);
// x[row][:] <- x[row][:] B, B float64 [A][A] row-major (numpy's SVD colouring of a standard-normal stream)
template <typename T>
hipError_t color_rows(T* x, long rows, int A, const double* B, hipStream_t s);

}  // namespace mjmpc
