"""Python mirror of the reference's physics object model over physics_amd.World (include/physics_hip.h):

    reference (src/physics.rs, src/physics/*.rs)          here
    RigidBody { pub position, rotation, lin_velocity, .. } RigidBody (same field names)
    RigidBody::new(index)                                  RigidBody.new(index)
    Entity { body, instance }                              Entity
    Constraints::FixedPosition / FixedOrientation          FixToPointConstraint / FixedOrientationConstraint
    ConstraintSolver { constraints }                       ConstraintSolver
    PhysicsState { entities, constraint_solver, .. }       PhysicsState
    PhysicsState::update(&Duration) / apply_gravity / step update(dt) / apply_gravity() / step(dt)

The C++ equivalent (include/physics_state.hpp) is the one meant for embedding; this one lets the parity tests
read like a caller of the reference. Callers of the reference edit bodies through pub fields between frames
(lib.rs:21-22); like the C++ mirror, every call first uploads what changed and afterwards writes the new state
back into `entities`."""
import datetime

import numpy as np

from . import _abi
from .world import World


def _nanos(dt):
    """std::time::Duration: int nanoseconds or datetime.timedelta."""
    if isinstance(dt, datetime.timedelta):
        return (dt.days * 86400 + dt.seconds) * 1_000_000_000 + dt.microseconds * 1000
    return int(dt)


class RigidBody:
    def __init__(self, index=0):
        # RigidBody::new (rigid_body.rs:64-76)
        self.mass = 1.0
        self.lin_velocity = np.zeros(3, np.float32)
        self.angular_velocity = np.zeros(3, np.float32)
        self.force = np.zeros(3, np.float32)
        self.torque = np.zeros(3, np.float32)
        self.inertia_tensor = np.eye(3, dtype=np.float32)
        self.position = np.zeros(3, np.float32)
        self.rotation = np.array([0, 0, 0, 1], np.float32)  # [i, j, k, w]
        self.index = index
        self.shape_type = _abi.SHAPE_NONE  # new: the reference has no shapes
        self.half_extent = np.zeros(3, np.float32)

    @staticmethod
    def new(index):
        return RigidBody(index)

    # rigid_body.rs:43-62
    def apply_force_centre_of_gravity(self, force):
        self.force = (self.force + np.asarray(force, np.float32)).astype(np.float32)

    def apply_force_at_position(self, force, point):
        force = np.asarray(force, np.float32)
        d = (np.asarray(point, np.float32) - self.position).astype(np.float32)
        self.torque = (self.torque + _cross(d, force)).astype(np.float32)
        self.force = (self.force + force).astype(np.float32)

    def apply_force_at_offset(self, force, offset):
        force = np.asarray(force, np.float32)
        self.torque = (self.torque + _cross(np.asarray(offset, np.float32), force)).astype(np.float32)
        self.force = (self.force + force).astype(np.float32)

    def _state(self):
        return np.concatenate([[self.mass], self.lin_velocity, self.angular_velocity, self.inertia_tensor.reshape(-1),
                               self.position, self.rotation, [self.shape_type], self.half_extent]).astype(np.float32)


def _cross(a, b):
    # nalgebra cross, f32 throughout
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], np.float32)


class Entity:
    def __init__(self, body, instance=0):
        self.body = body
        self.instance = instance


class FixToPointConstraint:
    def __init__(self, rigid_body, position):
        self.rigid_body = int(rigid_body)
        self.position = np.asarray(position, np.float32)


class FixedOrientationConstraint(FixToPointConstraint):
    pass


class ConstraintSolver:
    def __init__(self, constraints=None):
        self.constraints = list(constraints or [])


class PhysicsState:
    def __init__(self, entities=None, constraint_solver=None, cfg=None):
        self.entities = list(entities or [])
        self.constraint_solver = constraint_solver or ConstraintSolver()
        self._world = World(cfg)
        self._snap = None
        self._con_snap = None

    # ---- host <-> device
    def _push(self):
        n = len(self.entities)
        state = np.stack([e.body._state() for e in self.entities]) if n else np.zeros((0, 27), np.float32)
        forces = (np.stack([np.concatenate([e.body.force, e.body.torque]) for e in self.entities]).astype(np.float32)
                  if n else np.zeros((0, 6), np.float32))
        bodies_changed = self._snap is None or self._snap[0].shape != state.shape or not np.array_equal(self._snap[0], state)
        if bodies_changed:
            b = [e.body for e in self.entities]
            self._world.set_bodies(
                np.stack([x.position for x in b]) if n else np.zeros((0, 3), np.float32),
                rot=np.stack([x.rotation for x in b]) if n else None,
                lin_vel=np.stack([x.lin_velocity for x in b]) if n else None,
                ang_vel=np.stack([x.angular_velocity for x in b]) if n else None,
                mass=np.array([x.mass for x in b], np.float32) if n else None,
                inertia=np.stack([x.inertia_tensor.reshape(-1) for x in b]) if n else None,
                shape_type=np.array([x.shape_type for x in b], np.uint32) if n else None,
                half_extent=np.stack([x.half_extent for x in b]) if n else None)
            self._con_snap = None
        if n and (bodies_changed or not np.array_equal(self._snap[1], forces)):
            self._world.set_forces(forces[:, :3].copy(), forces[:, 3:].copy())
        cons = [(type(c) is FixedOrientationConstraint, c.rigid_body, tuple(c.position.tolist()))
                for c in self.constraint_solver.constraints]
        if cons != self._con_snap:
            self._world.clear_constraints()
            for is_orient, body, target in cons:
                (self._world.add_constraint_fix_orientation if is_orient else self._world.add_constraint_fix_point)(body, target)
            self._con_snap = cons
        self._snap = (state, forces)

    def _pull(self, forces_only=False):
        n = len(self.entities)
        if not n:
            return
        if forces_only:
            f, t = self._world.get_forces()
            for i, e in enumerate(self.entities):
                e.body.force, e.body.torque = f[i].copy(), t[i].copy()
        else:
            pos, rot = self._world.get_transforms()
            lin, ang = self._world.get_velocities()
            for i, e in enumerate(self.entities):
                b = e.body
                b.position, b.rotation = pos[i].copy(), rot[i].copy()
                b.lin_velocity, b.angular_velocity = lin[i].copy(), ang[i].copy()
                b.force = np.zeros(3, np.float32)  # rigid_body.rs:38-39
                b.torque = np.zeros(3, np.float32)
        self._snap = (np.stack([e.body._state() for e in self.entities]),
                      np.stack([np.concatenate([e.body.force, e.body.torque]) for e in self.entities]).astype(np.float32))

    # ---- the reference's methods
    def update(self, dt):  # physics.rs:41-55
        self._push()
        self._world.update(_nanos(dt))
        self._pull()

    def apply_gravity(self):  # physics.rs:87-94
        self._push()
        self._world.apply_gravity()
        self._pull(forces_only=True)

    def step(self, dt):  # physics.rs:95-99
        self._push()
        self._world.step(_nanos(dt))
        self._pull()

    @property
    def previous_solution(self):  # physics.rs:30: None or the warm-start lambda
        lam = self._world.get_lambda()
        return lam if len(lam) else None

    def instance_matrices(self):  # what get_render_data feeds the renderer (physics.rs:61-69)
        self._push()
        return self._world.get_instance_matrices()
