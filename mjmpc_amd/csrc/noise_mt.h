#pragma once
#include <hip/hip_runtime.h>

namespace mjmpc {

constexpr int MT_MAX_SEGMENTS = 64;       // jump-ahead segments the workspace is sized for

// polar attempts generated for n_normals samples (acceptance pi/4 plus a 1 % margin) and the workspace they need
long mt_attempts_for(long n_normals);
long mt_workspace_bytes(long n_normals);

// Optional jump-ahead tables (mjmpc_amd/control/mt_jump.py): the stream is then produced by one short head
// workgroup plus n_segments workgroups in parallel, workgroup g starting at word head_words + g*seg_words.
// first_normal > 0: noise[0..n) receives normals [first_normal, first_normal + n) of the stream (sharded runs; the
// workspace and the segments must then cover first_normal + n normals).
// noise[0..n) = scale * (numpy legacy standard_normal stream after np.random.seed(seed + *d_step)); see noise_mt.hip.
// *status (device int, may be null) = 1 if the margin of attempts was not enough (never observed).
template <typename T>
hipError_t sample_noise_mt19937(T* noise, long n_normals, double scale, unsigned long long seed, const long long* d_step,
                                void* ws, int* status, hipStream_t s, const int* jump_idx = nullptr,
                                const int* jump_starts = nullptr, long head_words = 0, long seg_words = 0,
                                int n_segments = 0, long first_normal = 0);

}  // namespace mjmpc
