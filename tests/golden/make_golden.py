"""Generator of tests/golden/reference_vectors.json.

The reference (Rust, no toolchain in the authoring container) could not be run, so these vectors
are NOT outputs of the reference. They are:
  G3  data copied from the reference's own unit tests (sparse_matrix.rs:65-119): inputs + expected.
  G1  demo scene of lib.rs:20-42, first update(dt = 16_666_667 ns), derived here with numpy float32
      scalar arithmetic following physics.rs / constraints.rs / sle_solver.rs / rigid_body.rs
      statement by statement (independent of oracle/ — a second restatement).
  G2  free fall from y = 10, 1000 updates, same derivation.
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


if __name__ == "__main__":
    out = {"provenance": "hand-derived from the reference source in float32 (numpy), NOT produced by running "
                         "the reference; G3 is data held by the reference's own tests",
           "G1": g1(), "G2": g2(), "G3": g3()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["G1"]), json.dumps(out["G2"]))
