"""Rollout engines (the GPU side of ``GymEnvWrapper.rollout`` / ``SubprocVecEnv.rollout``) and the gym-like envs over them."""


def make_engine(raw, dtype="f64", device=0, num_shards=1):
    """The fastest engine that runs ``raw`` (a ``RawModel``): the serial-chain arm kernels where the model fits them - hinge /
    slide chains of at most seven dofs with joint limits, dry friction and one frictionless sphere-plane contact (since round 6:
    the reference's classic-control models) - else the general tree engine.  ``ArmRolloutEngine`` / ``TreeRolloutEngine`` can
    still be constructed directly."""
    from ..models.compile import compile_arm
    from .arm_engine import ArmRolloutEngine
    from .tree_engine import TreeRolloutEngine
    try:
        compile_arm(raw)
    except (ValueError, NotImplementedError):
        return TreeRolloutEngine(raw, device=device, dtype=dtype, num_shards=num_shards)
    return ArmRolloutEngine(raw, device=device, dtype=dtype, num_shards=num_shards)
