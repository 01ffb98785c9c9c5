"""Particle sharding across GPUs: the reference's worker mapping, kept bit-compatible.

``SubprocVecEnv.rollout_async`` (mjmpc/envs/vec_env/subproc_vec_env.py:161-168) asserts
``num_particles % n_workers == 0`` and hands worker i the contiguous slice
``noise[i*bs:(i+1)*bs]``; results are concatenated in worker order (:170-186).  One process per GPU
here owns exactly that slice, so a sharded run sees the same particles in the same global order.
"""


def local_block(num_particles, rank, world_size):
    """(offset, count) of the particles owned by ``rank``."""
    assert num_particles % world_size == 0, "Number of particles must be divisible by number of cpus"
    n = num_particles // world_size
    return rank * n, n


def slice_local(array, rank, world_size):
    off, n = local_block(array.shape[0], rank, world_size)
    return array[off:off + n]
