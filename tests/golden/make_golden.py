"""Generator of tests/golden/reference_vectors.json.

The reference (Rust, no toolchain in the authoring container) could not be run, so these vectors
are NOT outputs of the reference. They are:
  G3  data copied from the reference's own unit tests (sparse_matrix.rs:65-119): inputs + expected.
  G1  demo scene of lib.rs:20-42, first update(dt = 16_666_667 ns), derived here with numpy float32
      scalar arithmetic following physics.rs / constraints.rs / sle_solver.rs / rigid_body.rs
      statement by statement (independent of oracle/ — a second restatement).
  G2  free fall from y = 10, 1000 updates, same derivation.
  G4  two bodies (masses 1, 3), four constraints (12 rows), three frames: multi-iteration CG, 8-accumulator dot + tail,
      warm start, quirk Q4 - from a GENERAL numpy-f32 restatement of update() (class State below).
  G5  the demo scene for 300 frames (quirk Q1 rotation + euler_angles every frame), snapshots at six frames.
  G6  quirk Q3 with three bodies (constraint forces reach entity 0 only), two cases, two frames each.
  G7  the two GIMBAL branches of euler_angles (fixed_orientation_constraint.rs:17): bodies whose rotation matrix has
      |r20| >= 1 (pitch = +pi/2 and -pi/2), pinned to an orientation, two frames each.
Run:  python tests/golden/make_golden.py   (writes the JSON next to this file)
"""
import json
import math
import os

import numpy as np

f = np.float32


def as_secs_f32(nanos):
    return f(f(nanos // 1_000_000_000) + f(nanos % 1_000_000_000) / f(1e9))


def g1():
    dt = as_secs_f32(16_666_667)
    # lib.rs:20-25
    pos = [f(1), f(0), f(0)]
    roll = f(1.0)
    sr, cr = f(math.sin(float(roll * f(0.5)))), f(math.cos(float(roll * f(0.5))))
    rot = [sr, f(0), f(0), cr]  # from_euler_angles(1,0,0): i = sr*cp*cy - cr*sp*sy, w = cr*cp*cy + ...
    v = [f(0)] * 3
    w = [f(0)] * 3
    # apply_gravity: torque += (0,0,1.5) x (0,-9.81,0); force += F
    F = [f(0), f(-9.81), f(0)]
    off = [f(0), f(0), f(1.5)]
    torque = [off[1] * F[2] - off[2] * F[1], off[2] * F[0] - off[0] * F[2], off[0] * F[1] - off[1] * F[0]]
    force = list(F)
    # constraints: C = (pos - 0, euler - 0); J = I6; Jdot = 0; ks = 10; kd = 1; W = 1
    # euler_angles of a pure roll: r20 = 0 -> pitch = -asin(0) = -0, cos = 1;
    # roll = atan2(r21, r22) with r21 = 2*w*i, r22 = ww - ii
    r21 = rot[3] * rot[0] * f(2.0) + f(0)
    r22 = rot[3] * rot[3] - rot[0] * rot[0]
    eul = [f(math.atan2(float(r21), float(r22))), f(0), f(0)]
    C = pos + eul
    qdot = v + w
    Q = force + torque
    rhs = [f(-0.0) - Q[k] * f(1.0) - f(10.0) * C[k] - f(1.0) * qdot[k] for k in range(6)]
    # CG with A = I, x0 = 0: r = rhs, p = r, alpha = r.r / p.Ap = 1, x = rhs, r = 0 -> exit after 1 iteration
    lam = [f(1.0) * x for x in rhs]
    jt_lam = list(lam)
    force = [force[k] + jt_lam[k] for k in range(3)]
    torque = [torque[k] + jt_lam[3 + k] for k in range(3)]
    # RigidBody::step
    v = [v[k] + force[k] / f(1.0) * dt for k in range(3)]
    pos = [pos[k] + v[k] * dt for k in range(3)]
    L = [torque[k] * dt for k in range(3)]
    w = [w[k] + L[k] for k in range(3)]  # inverse of identity is identity
    nrm = f(np.sqrt(f(w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]))
    a = [x / nrm for x in w]
    theta = nrm * dt
    s = f(math.sin(float(theta * f(0.5))))
    u = [(x * s) / f(2.0) for x in a]
    nn = f(u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]
    n = f(np.sqrt(nn))
    fac = f(1.0) * f(math.sin(float(n))) / n
    dq = [u[0] * fac, u[1] * fac, u[2] * fac, f(1.0) * f(math.cos(float(n)))]
    ai, aj, ak, aw = dq
    bi, bj, bk, bw = rot
    rot = [aw * bi + ai * bw + aj * bk - ak * bj, aw * bj - ai * bk + aj * bw + ak * bi,
           aw * bk + ai * bj - aj * bi + ak * bw, aw * bw - ai * bi - aj * bj - ak * bk]
    return {"dt_nanos": 16_666_667, "pos0": [1, 0, 0], "euler0": [1, 0, 0], "fix_point": [0, 0, 0],
            "fix_orientation": [0, 0, 0], "lambda": [float(x) for x in lam], "pos": [float(x) for x in pos],
            "rot_ijkw": [float(x) for x in rot], "lin_vel": [float(x) for x in v], "ang_vel": [float(x) for x in w]}


def g2():
    dt = as_secs_f32(16_666_667)
    y, vy = f(10.0), f(0.0)
    for _ in range(1000):
        fy = f(0.0) + f(-9.81)
        vy = vy + fy / f(1.0) * dt
        y = y + vy * dt
    return {"dt_nanos": 16_666_667, "y0": 10.0, "steps": 1000, "gravity_offset": [0, 0, 0], "y": float(y),
            "vy": float(vy)}


def g3():
    # sparse_matrix.rs:65-119. nalgebra from_vec is column-major; blocks below are row-major.
    return [
        {"name": "multiply_vector_single_block_test", "nrows": 5, "ncols": 5, "transpose": False,
         "blocks": [{"i": 0, "j": 0, "data": [[1, 2], [3, 4]]}], "vector": [2, 6, 1, 1, 2],
         "expected": [14, 30, 0, 0, 0]},
        {"name": "multiply_vector_multiple_block_test", "nrows": 5, "ncols": 5, "transpose": False,
         "blocks": [{"i": 0, "j": 0, "data": [[1, 2], [3, 4]]}, {"i": 1, "j": 2, "data": [[1], [3]]}],
         "vector": [2, 6, 1, 1, 2], "expected": [14, 31, 3, 0, 0]},
        {"name": "tr_multiply_vector_multiple_block_test", "nrows": 5, "ncols": 5, "transpose": True,
         "blocks": [{"i": 0, "j": 0, "data": [[1, 3], [2, 4]]}, {"i": 2, "j": 1, "data": [[1, 3]]}],
         "vector": [2, 6, 1, 1, 2], "expected": [14, 31, 3, 0, 0]},
    ]


# ---------------------------------------------------------------------------------------------------------
# G4-G6: a general numpy-float32 restatement of PhysicsState::update for N bodies and any list of the two
# constraint kinds, written from the reference source statement by statement (scalar np.float32 arithmetic, so
# every rounding is the reference's). It shares no code with oracle/ (C++) or include/spec. Libm calls are
# math.sin / cos / asin / atan2 on the f32 value, rounded once to f32.
# nalgebra 0.32.2 algorithms relied on (crate source not in the container): the dynamic-length dot
# product with eight partial sums + tail; gemv as a left-to-right sum over columns; amax as a fold of |x|.

def _sin(x): return f(math.sin(float(x)))
def _cos(x): return f(math.cos(float(x)))
def _asin(x): return f(math.asin(float(x)))
def _atan2(y, x): return f(math.atan2(float(y), float(x)))


def dyn_dot(a, b):
    """nalgebra dotc for Dyn vectors: 8 accumulators over blocks of 8, combined (0+4)+(1+5)+(2+6)+(3+7), then the tail."""
    n, i = len(a), 0
    acc = [f(0)] * 8
    while n - i >= 8:
        for k in range(8):
            acc[k] = acc[k] + a[i + k] * b[i + k]
        i += 8
    res = f(0)
    res = res + (acc[0] + acc[4])
    res = res + (acc[1] + acc[5])
    res = res + (acc[2] + acc[6])
    res = res + (acc[3] + acc[7])
    for k in range(i, n):
        res = res + a[k] * b[k]
    return res


def amax(a):
    m = f(0)
    for x in a:
        ax = abs(x)
        if ax > m:
            m = ax
    return m


class Body:
    def __init__(self, pos, rot=(0, 0, 0, 1), mass=1.0, lin=(0, 0, 0), ang=(0, 0, 0)):
        self.mass = f(mass)
        self.lin = [f(x) for x in lin]
        self.ang = [f(x) for x in ang]
        self.force = [f(0)] * 3
        self.torque = [f(0)] * 3
        self.pos = [f(x) for x in pos]
        self.rot = [f(x) for x in rot]  # i, j, k, w


def quat_from_euler(roll, pitch, yaw):
    """UnitQuaternion::from_euler_angles (lib.rs:22)."""
    sr, cr = _sin(f(roll) * f(0.5)), _cos(f(roll) * f(0.5))
    sp, cp = _sin(f(pitch) * f(0.5)), _cos(f(pitch) * f(0.5))
    sy, cy = _sin(f(yaw) * f(0.5)), _cos(f(yaw) * f(0.5))
    return [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
            cr * cp * cy + sr * sp * sy]


def euler_angles(q):
    """UnitQuaternion::euler_angles = to_rotation_matrix().euler_angles() (fixed_orientation_constraint.rs:17)."""
    i, j, k, w = q
    ww, ii, jj, kk = w * w, i * i, j * j, k * k
    ij, wk, wj = i * j * f(2), w * k * f(2), w * j * f(2)
    ik, jk, wi = i * k * f(2), j * k * f(2), w * i * f(2)
    r00, r10 = ww + ii - jj - kk, wk + ij
    r20, r21, r22 = ik - wj, wi + jk, ww - ii - jj + kk
    r01, r02 = ij - wk, wj + ik
    if abs(r20) < f(1):
        pitch = -_asin(r20)
        c = _cos(pitch)
        return [_atan2(r21 / c, r22 / c), pitch, _atan2(r10 / c, r00 / c)]
    if r20 <= f(-1):
        return [_atan2(r01, r02), f(math.pi / 2), f(0)]
    # nalgebra: `(-self[(0, 1)].atan2(-self[(0, 2)]), -FRAC_PI_2, 0)` - a method call binds tighter than the unary minus:
    # -(r01.atan2(-r02)). (Rounds 1-2 had -atan2(-r01, -r02) here, the opposite sign; no golden reached the branch.)
    return [-_atan2(r01, -r02), f(-math.pi / 2), f(0)]


class State:
    """PhysicsState: entities + constraints [(kind, body, target)] + previous_solution (physics.rs:25-31)."""

    def __init__(self, bodies, constraints):
        self.bodies = bodies
        self.constraints = constraints
        self.prev = None
        self.cg_iterations = 0

    # sparse_matrix.rs:25-50 with the blocks of the two constraint kinds (3x6, ones at (r, off + r))
    def _j_mul(self, vec):
        res = [f(0)] * (3 * len(self.constraints))
        for ci, (kind, body, _) in enumerate(self.constraints):
            off = 0 if kind == "point" else 3
            for r in range(3):
                acc = None
                for c in range(6):  # gemv: left-to-right over the columns of the 1x6 row
                    term = (f(1) * vec[6 * body + c]) * (f(1) if c == off + r else f(0))
                    acc = term if acc is None else acc + term
                res[3 * ci + r] = res[3 * ci + r] + acc
        return res

    def _jt_mul(self, vec):
        res = [f(0)] * (6 * len(self.bodies))
        for ci, (kind, body, _) in enumerate(self.constraints):
            off = 0 if kind == "point" else 3
            for c in range(6):
                acc = None
                for r in range(3):
                    term = (f(1) * vec[3 * ci + r]) * (f(1) if c == off + r else f(0))
                    acc = term if acc is None else acc + term
                res[6 * body + c] = res[6 * body + c] + acc
        return res

    def _lhs(self, inv_m, v):  # sle_solver.rs:48-51
        jt = self._jt_mul(v)
        return self._j_mul([jt[k] * inv_m[k] for k in range(len(jt))])

    def _cg(self, inv_m, rhs):  # sle_solver.rs:21-46
        x = list(self.prev) if self.prev is not None else [f(0)] * len(rhs)
        ax = self._lhs(inv_m, x)
        r = [rhs[k] - ax[k] for k in range(len(rhs))]
        p = list(r)
        bound = max(amax(rhs) * f(1e-2), f(1e-3))
        for it in range(1000):
            jp = self._lhs(inv_m, p)
            rk = dyn_dot(r, r)
            alpha = rk / dyn_dot(p, jp)
            x = [x[k] + alpha * p[k] for k in range(len(x))]
            r = [r[k] - alpha * jp[k] for k in range(len(r))]
            if amax(r) < bound:
                self.cg_iterations = it + 1
                return x
            beta = dyn_dot(r, r) / rk
            p = [r[k] + beta * p[k] for k in range(len(p))]
        self.cg_iterations = 1000
        return None

    def update(self, dt):
        # apply_gravity (physics.rs:87-94): torque += offset x F; force += F
        F, off = [f(0), f(-9.81), f(0)], [f(0), f(0), f(1.5)]
        cr = [off[1] * F[2] - off[2] * F[1], off[2] * F[0] - off[0] * F[2], off[0] * F[1] - off[1] * F[0]]
        for b in self.bodies:
            b.torque = [b.torque[k] + cr[k] for k in range(3)]
            b.force = [b.force[k] + F[k] for k in range(3)]
        # solve_constraints (constraints.rs:67-169)
        inv_m, qdot, Q = [], [], []
        for b in self.bodies:
            inv_m += [f(1) / b.mass] * 6
            qdot += b.lin + b.ang
            Q += b.force + b.torque
        C = []
        for kind, body, target in self.constraints:
            b = self.bodies[body]
            val = b.pos if kind == "point" else euler_angles(b.rot)
            C += [val[k] - f(target[k]) for k in range(3)]
        rows = len(C)
        jdq = [-f(0)] * rows  # J-dot is zero: -(0)
        cdot = self._j_mul(qdot)
        kd = [f(1) * cdot[k] for k in range(rows)]
        ks = [f(10) * C[k] for k in range(rows)]
        jwq = self._j_mul([Q[k] * inv_m[k] for k in range(len(Q))])
        rhs = [((jdq[k] - jwq[k]) - ks[k]) - kd[k] for k in range(rows)]
        lam = self._cg(inv_m, rhs)
        if lam is not None:
            self.prev = lam
            m = self._jt_mul(lam)
            b0 = self.bodies[0]  # quirk Q3: column_iter() of a column vector yields ONE column -> entity 0 only
            b0.force = [b0.force[k] + m[k] for k in range(3)]
            b0.torque = [b0.torque[k] + m[3 + k] for k in range(3)]
        # step (rigid_body.rs:24-40), inertia = identity
        for b in self.bodies:
            b.lin = [b.lin[k] + b.force[k] / b.mass * dt for k in range(3)]
            b.pos = [b.pos[k] + b.lin[k] * dt for k in range(3)]
            L = [b.torque[k] * dt for k in range(3)]
            b.ang = [b.ang[k] + L[k] for k in range(3)]  # identity^-1 * L (the zero products add exactly)
            if any(x != f(0) for x in b.ang):
                nrm = f(np.sqrt((b.ang[0] * b.ang[0] + b.ang[1] * b.ang[1]) + b.ang[2] * b.ang[2]))
                a = [x / nrm for x in b.ang]
                theta = nrm * dt
                s = _sin(theta * f(0.5))
                u = [(x * s) / f(2) for x in a]
                nn = (u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]
                if not nn <= f(np.finfo(np.float32).eps) * f(np.finfo(np.float32).eps):
                    n = f(np.sqrt(nn))
                    fac = f(1) * _sin(n) / n
                    dq = [u[0] * fac, u[1] * fac, u[2] * fac, f(1) * _cos(n)]
                    ai, aj, ak, aw = dq
                    bi, bj, bk, bw = b.rot
                    b.rot = [aw * bi + ai * bw + aj * bk - ak * bj, aw * bj - ai * bk + aj * bw + ak * bi,
                             aw * bk + ai * bj - aj * bi + ak * bw, aw * bw - ai * bi - aj * bj - ak * bk]
            b.force = [f(0)] * 3
            b.torque = [f(0)] * 3

    def snapshot(self):
        fl = lambda v: [float(x) for x in v]
        return {"pos": [fl(b.pos) for b in self.bodies], "rot_ijkw": [fl(b.rot) for b in self.bodies],
                "lin_vel": [fl(b.lin) for b in self.bodies], "ang_vel": [fl(b.ang) for b in self.bodies],
                "lambda": fl(self.prev) if self.prev is not None else [], "cg_iterations": self.cg_iterations}


def _scene_dict(bodies, constraints):
    return {"pos0": [[float(x) for x in b.pos] for b in bodies], "rot0_ijkw": [[float(x) for x in b.rot] for b in bodies],
            "mass": [float(b.mass) for b in bodies], "lin0": [[float(x) for x in b.lin] for b in bodies],
            "ang0": [[float(x) for x in b.ang] for b in bodies],
            "constraints": [[k, int(b), [float(x) for x in t]] for k, b, t in constraints]}


def g4():
    """Two bodies (masses 1 and 3), four constraints = 12 rows: the 8-accumulator dot product runs one block AND its
    tail; J W J^T = diag(1 x6, 1/3 x6), so CG needs more than one iteration; frames 2 and 3 start from the previous
    lambda (warm start, sle_solver.rs:22-26); the orientation rows of body 1 use 1/mass (quirk Q4)."""
    dt = as_secs_f32(16_666_667)
    bodies = [Body((1, 0, 0), quat_from_euler(1.0, 0.0, 0.0), 1.0),
              Body((2.5, 1.25, -0.5), quat_from_euler(0.25, -0.5, 0.75), 3.0, lin=(0.5, -0.25, 0.125), ang=(0.25, 0.5, -0.75))]
    cons = [("point", 0, (0, 0, 0)), ("orientation", 0, (0, 0, 0)), ("point", 1, (2, 1, 0)), ("orientation", 1, (0.5, -0.25, 0.5))]
    out = _scene_dict(bodies, cons)
    st = State(bodies, cons)
    out["dt_nanos"] = 16_666_667
    out["frames"] = []
    for _ in range(3):
        st.update(dt)
        out["frames"].append(st.snapshot())
    return out


def g5():
    """The demo scene (lib.rs:20-42) for 300 frames: quirk Q1 rotation and euler_angles every frame, warm start."""
    dt = as_secs_f32(16_666_667)
    bodies = [Body((1, 0, 0), quat_from_euler(1.0, 0.0, 0.0), 1.0)]
    cons = [("point", 0, (0, 0, 0)), ("orientation", 0, (0, 0, 0))]
    out = _scene_dict(bodies, cons)
    st = State(bodies, cons)
    out["dt_nanos"] = 16_666_667
    out["frames"] = {}
    for k in range(1, 301):
        st.update(dt)
        if k in (2, 10, 50, 100, 200, 300):
            out["frames"][str(k)] = st.snapshot()
    return out


def g6():
    """Quirk Q3 with N = 3: constraints on bodies 1 and 2 compute a lambda that nobody receives (only rows 0..6 of
    J^T lambda are scattered, to entity 0); with a constraint on body 0 as well, body 0 - and only body 0 - receives."""
    dt = as_secs_f32(16_666_667)
    out = {"dt_nanos": 16_666_667, "cases": []}
    for cons in ([("point", 1, (0, 0, 0)), ("point", 2, (1, 1, 1))],
                 [("point", 0, (0.5, 0, 0)), ("point", 2, (1, 1, 1)), ("orientation", 1, (0, 0, 0))]):
        bodies = [Body((1, 0, 0), mass=1.0), Body((0, 2, 0), quat_from_euler(0.5, 0.25, 0.0), 2.0), Body((1, 2, 3), mass=4.0, lin=(0, 1, 0))]
        case = _scene_dict(bodies, cons)
        st = State(bodies, cons)
        case["frames"] = []
        for _ in range(2):
            st.update(dt)
            case["frames"].append(st.snapshot())
        out["cases"].append(case)
    return out


def g7():
    """fixed_orientation_constraint.rs:17 calls UnitQuaternion::euler_angles, whose two gimbal branches (r20 <= -1:
    pitch = +pi/2, roll = atan2(r01, r02); r20 >= 1: pitch = -pi/2, roll = -atan2(r01, -r02)) no other golden reaches.
    A pitch of +-pi/2 from from_euler_angles gives |r20| = 0.99999994 in float32 (the regular branch); the quaternion is
    never renormalised anyway (quirk Q6), so the bodies here carry the quaternion of (roll, +-pi/2, yaw) scaled by
    1.0625: |r20| = 1.1289 - exactly what a drifted norm does to a body near the pole. Two frames each (the second one
    starts from the first one's lambda and from a rotated body, back in the regular branch or not)."""
    dt = as_secs_f32(16_666_667)
    out = {"dt_nanos": 16_666_667, "cases": []}
    for pitch, roll, yaw in ((math.pi / 2, 0.3, 0.0), (-math.pi / 2, 0.3, 0.0), (math.pi / 2, -0.7, 0.4), (-math.pi / 2, 1.1, -0.2)):
        q = [x * f(1.0625) for x in quat_from_euler(roll, pitch, yaw)]
        bodies = [Body((0.5, 0.25, -0.125), q, 2.0, ang=(0.125, -0.25, 0.5))]
        cons = [("orientation", 0, (0.1, 0.2, -0.3)), ("point", 0, (0, 0, 0))]
        i, j, k, w = bodies[0].rot
        r20 = i * k * f(2) - w * j * f(2)
        assert abs(r20) >= f(1), r20
        case = _scene_dict(bodies, cons)
        case["branch"] = "r20 <= -1 (pitch +pi/2)" if r20 <= f(-1) else "r20 >= 1 (pitch -pi/2)"
        case["euler0"] = [float(x) for x in euler_angles(bodies[0].rot)]
        st = State(bodies, cons)
        case["frames"] = []
        for _ in range(2):
            st.update(dt)
            case["frames"].append(st.snapshot())
        out["cases"].append(case)
    assert {c["branch"][:8] for c in out["cases"]} == {"r20 <= -", "r20 >= 1"}
    return out


if __name__ == "__main__":
    out = {"provenance": "hand-derived from the reference source in float32 (numpy), NOT produced by running "
                         "the reference; G3 is data held by the reference's own tests",
           "G1": g1(), "G2": g2(), "G3": g3(), "G4": g4(), "G5": g5(), "G6": g6(), "G7": g7()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["G1"]), json.dumps(out["G2"]))
