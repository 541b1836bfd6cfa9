"""Recorder for the day a machine with Pinocchio is at hand: the quantities robotoc::Robot hands to evalKKT, computed by
Pinocchio itself, written as tests/golden/pinocchio_<robot>.npz.  tests/test_rigid_body.py::
test_restatement_matches_a_recorded_pinocchio_fixture replays the file against oracle/rtoc_oracle_rbd.c (values) and its
complex-step derivatives, and the -m gpu linearisation tests then stand on Pinocchio's numbers instead of a restatement --
the one thing this image cannot provide (no pinocchio module, no network), which is why f3 says PARITY UNPINNED.

NOT run anywhere in this repository's tests or builds.  Needs: `import pinocchio` (>= 2.6) and the reference's URDF.

What is recorded, per random sample (q, v, a, contact forces f in the LOCAL contact frames, active mask, desired positions),
with the calls the reference makes (cited so a reader can check the recorder against the reference, not against us):
  ID      pinocchio::rnea(model, data, q, v, a, fjoint)                                   robot.hxx:524-546 (Robot::RNEA)
          fjoint[parent] += jXf.act(Force(f_k, 0)) for the active point contacts          point_contact.cpp:55-60, robot.hxx:455-517
  dIDdq, dIDdv, dIDda   pinocchio::computeRNEADerivatives (M symmetrised)                   robot.hxx:548-566
  C       getFrameClassicalAcceleration(LOCAL).linear + kd * getFrameVelocity(LOCAL).linear
          + kp * (oMf.translation - desired)                                              point_contact.hxx:16-31
  dCdq, dCdv, dCda      getFrameAccelerationDerivatives(LOCAL) recombined                   point_contact.hxx:36-88
  impact grids: ID with a = dv, zero gravity, v = 0 (robot.hxx:590-600); C = getFrameVelocity(LOCAL).linear and its
          derivatives getFrameVelocityDerivatives(LOCAL)                                  point_contact.hxx:91-126
  q (+) dq              pinocchio::integrate                                                robot.hxx:41-52

usage:  python tools/record_pinocchio_fixture.py /path/to/robotoc/test/urdf/anymal/anymal.urdf --floating-base \
            --contacts LF_FOOT LH_FOOT RF_FOOT RH_FOOT --kp 0 --kd 0 -o tests/golden/pinocchio_anymal.npz
"""
import argparse

import numpy as np


def skew(x):
    return np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("urdf")
    ap.add_argument("--floating-base", action="store_true")
    ap.add_argument("--contacts", nargs="*", default=[])
    ap.add_argument("--kp", type=float, default=0.0)   # ContactModelInfo::baumgarte_position_gain
    ap.add_argument("--kd", type=float, default=0.0)   # ContactModelInfo::baumgarte_velocity_gain
    ap.add_argument("--samples", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("-o", "--out", required=True)
    args = ap.parse_args()
    import pinocchio as pin

    model = pin.buildModelFromUrdf(args.urdf, pin.JointModelFreeFlyer()) if args.floating_base else pin.buildModelFromUrdf(args.urdf)
    data = model.createData()
    model0 = model.copy()            # the impact model: zero gravity (robot.hxx: impact_model_.gravity.linear().setZero())
    model0.gravity.linear = np.zeros(3)
    data0 = model0.createData()
    fids = [model.getFrameId(n) for n in args.contacts]
    nc, nv = len(fids), model.nv
    rng = np.random.default_rng(args.seed)
    rec = {k: [] for k in ("q", "v", "a", "f", "active", "impact", "pos", "ID", "dIDdq", "dIDdv", "dIDda", "C", "dCdq", "dCdv", "dCda", "dq", "q_plus_dq")}
    for s in range(args.samples):
        q = pin.randomConfiguration(model, -np.ones(model.nq), np.ones(model.nq)) if not args.floating_base else None
        if q is None:
            q = pin.neutral(model)
            q[:3] = rng.uniform(-0.8, 0.8, 3)
            quat = rng.normal(size=4)
            q[3:7] = quat / np.linalg.norm(quat)
            q[7:] = rng.uniform(-0.8, 0.8, model.nq - 7)
        v, a = rng.uniform(-0.8, 0.8, nv), rng.uniform(-0.8, 0.8, nv)
        f = rng.uniform(-20, 20, (nc, 3))
        active = int(rng.integers(0, 1 << nc)) if nc else 0
        pos = rng.uniform(-0.5, 0.5, (nc, 3))
        impact = s % 4 == 3 and nc > 0
        mdl, dat = (model0, data0) if impact else (model, data)
        vv = np.zeros(nv) if impact else v
        fext = pin.StdVec_Force()
        for _ in range(mdl.njoints):
            fext.append(pin.Force.Zero())
        for k, fid in enumerate(fids):
            if (active >> k) & 1:
                fr = mdl.frames[fid]
                fext[fr.parent] = fext[fr.parent] + fr.placement.act(pin.Force(f[k], np.zeros(3)))
        ID = pin.rnea(mdl, dat, q, vv, a, fext).copy()
        pin.computeRNEADerivatives(mdl, dat, q, vv, a, fext)
        M = np.triu(dat.M) + np.triu(dat.M, 1).T
        dIDdq, dIDdv = dat.dtau_dq.copy(), dat.dtau_dv.copy()
        # the kinematics the contact residuals read: impact grids at v + dv (impact_stage.cpp:61), else at (q, v, a)
        vk = v + a if impact else v
        pin.forwardKinematics(model, data, q, vk, a)
        pin.updateFramePlacements(model, data)
        pin.computeForwardKinematicsDerivatives(model, data, q, vk, a)
        pin.computeJointJacobians(model, data, q)
        C, dCdq, dCdv, dCda = [], [], [], []
        for k, fid in enumerate(fids):
            if not (active >> k) & 1:
                continue
            vf = pin.getFrameVelocity(model, data, fid, pin.LOCAL)
            J = pin.getFrameJacobian(model, data, fid, pin.LOCAL)
            if impact:
                v_dq, v_dv = pin.getFrameVelocityDerivatives(model, data, fid, pin.LOCAL)
                C.append(vf.linear.copy()), dCdq.append(v_dq[:3].copy()), dCdv.append(v_dv[:3].copy()), dCda.append(np.zeros((3, nv)))
                continue
            v_dq, a_dq, a_dv, a_da = pin.getFrameAccelerationDerivatives(model, data, fid, pin.LOCAL)
            acc = pin.getFrameClassicalAcceleration(model, data, fid, pin.LOCAL)
            C.append(acc.linear + args.kd * vf.linear + args.kp * (data.oMf[fid].translation - pos[k]))
            Sw, Sv = skew(vf.angular), skew(vf.linear)
            dCdq.append(a_dq[:3] + Sw @ v_dq[:3] - Sv @ v_dq[3:] + args.kd * v_dq[:3] + args.kp * data.oMf[fid].rotation @ J[:3])
            dCdv.append(a_dv[:3] + Sw @ J[:3] - Sv @ J[3:] + args.kd * a_da[:3])
            dCda.append(a_da[:3].copy())
        stack = lambda rows, w: np.concatenate(rows, 0) if rows else np.zeros((0, w))
        dq = rng.uniform(-0.3, 0.3, nv)
        for key, val in (("q", q), ("v", v), ("a", a), ("f", f.reshape(-1)), ("active", active), ("impact", int(impact)), ("pos", pos.reshape(-1)),
                         ("ID", ID), ("dIDdq", dIDdq), ("dIDdv", dIDdv), ("dIDda", M),
                         ("C", np.concatenate(C) if C else np.zeros(0)), ("dCdq", stack(dCdq, nv)), ("dCdv", stack(dCdv, nv)), ("dCda", stack(dCda, nv)),
                         ("dq", dq), ("q_plus_dq", pin.integrate(model, q, dq))):
            rec[key].append(np.asarray(val))
    out = {k: np.array(v, dtype=object) if k in ("C", "dCdq", "dCdv", "dCda") else np.array(v) for k, v in rec.items()}
    out["joint_names"] = np.array([model.names[i] for i in range(1, model.njoints)])
    out["contact_frames"] = np.array(args.contacts)
    out["kp"], out["kd"] = args.kp, args.kd
    out["pinocchio_version"] = pin.__version__
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, "samples", args.samples, "pinocchio", pin.__version__)


if __name__ == "__main__":
    main()
