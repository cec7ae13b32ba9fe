"""physics_amd — MI355X (gfx950) backend for the per-frame rigid-body step of martingoe/physics.

Host-side mirror of the reference's `PhysicsState` surface over the C ABI of libphysics_hip.so
(include/physics_hip.h). The compute path is hand-written HIP; there is no CPU fallback."""
from ._abi import (FLAG_BROADPHASE_ONLY, FLAG_COLLISIONS, FLAG_EXACT_ROTATION, FLAG_GROUND_PLANE,
                   FLAG_EXCLUSIVE_GPU, FLAG_NO_WARM_START, FLAG_SHARED_GPU, FLAG_SOLVER_CLUSTER, FLAG_SOLVER_PER_COLOR, GROUND_ID,
                   SHAPE_BOX, SHAPE_NONE, SHAPE_SPHERE, PhysicsHipMissing, default_config)
from .world import Comm, PhysError, World, block_spmv

__all__ = ["World", "Comm", "block_spmv", "PhysError", "PhysicsHipMissing", "default_config", "FLAG_COLLISIONS", "FLAG_GROUND_PLANE",
           "FLAG_EXACT_ROTATION", "FLAG_BROADPHASE_ONLY", "FLAG_SOLVER_PER_COLOR", "FLAG_SHARED_GPU", "FLAG_SOLVER_CLUSTER", "FLAG_EXCLUSIVE_GPU", "FLAG_NO_WARM_START", "SHAPE_NONE", "SHAPE_SPHERE", "SHAPE_BOX", "GROUND_ID"]
