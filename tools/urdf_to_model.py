"""URDF -> the rigid-body model table rtoc_set_robot_model takes (include/rtoc.h: rtoc_robot_model).

The reference gets its Robot from Pinocchio's URDF parser (include/robotoc/robot/robot.hxx, Robot::Robot ->
pinocchio::urdf::buildModel); this script restates what that parser builds, as far as the hot path needs it:
  * one model joint per moving URDF joint (revolute / continuous, any axis), depth-first, children in name order (urdfdom / Pinocchio), joint frame =
    URDF joint origin composed through the fixed joints above it; an optional free-flyer root joint (floating base);
  * links behind FIXED joints are welded into the body of the nearest moving ancestor (spatial inertias added in that
    joint's frame), like Pinocchio does when it appends a body through a fixed joint;
  * contact frames: named links, placement relative to their moving ancestor joint.
Run here (the URDFs are the reference's test robots, /root/reference/test/urdf); the JSON tables it writes are the
committed fixtures robotoc_amd/models/*.json -- nothing reads the URDFs at run time.

  python tools/urdf_to_model.py /root/reference/test/urdf/anymal/anymal.urdf --floating-base \
      --contacts LF_FOOT LH_FOOT RF_FOOT RH_FOOT -o robotoc_amd/models/anymal.json
"""
import argparse
import json
import xml.etree.ElementTree as ET

import numpy as np


def rpy_to_R(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def origin_of(elem):
    o = elem.find("origin") if elem is not None else None
    xyz = np.array([float(x) for x in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()])
    rpy = [float(x) for x in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
    return rpy_to_R(rpy), xyz


class Inertia:
    """mass, centre of mass c, rotational inertia I about c -- all in one frame"""

    def __init__(self, m=0.0, c=None, I=None):
        self.m, self.c, self.I = m, np.zeros(3) if c is None else c, np.zeros((3, 3)) if I is None else I

    def moved(self, R, p):  # the same inertia expressed in a frame in which this one sits at (R, p)
        return Inertia(self.m, R @ self.c + p, R @ self.I @ R.T)

    def __add__(self, o):
        m = self.m + o.m
        if m == 0.0:
            return Inertia()
        c = (self.m * self.c + o.m * o.c) / m

        def shift(b):
            d = b.c - c
            return b.I + b.m * (d @ d * np.eye(3) - np.outer(d, d))

        return Inertia(m, c, shift(self) + shift(o))


def link_inertia(link):
    ine = link.find("inertial")
    if ine is None:
        return Inertia()
    R, p = origin_of(ine)
    m = float(ine.find("mass").get("value"))
    i = ine.find("inertia")
    g = lambda k: float(i.get(k, "0"))
    I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
    return Inertia(m, p, R @ I @ R.T)


def build(urdf, floating_base, contacts, kp, kd, surface_contacts=()):
    root = ET.parse(urdf).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = root.findall("joint")
    children = {}
    child_links = set()
    for j in joints:
        children.setdefault(j.find("parent").get("link"), []).append(j)
        child_links.add(j.find("child").get("link"))
    roots = [n for n in links if n not in child_links]
    assert len(roots) == 1, roots
    bodies = []  # dicts: name, parent, type, axis, R, p, inertia
    frames = {}  # link name -> (body index, R, p) relative to the body's joint frame

    def add_body(name, parent, jtype, axis, R, p):
        bodies.append(dict(name=name, parent=parent, type=jtype, axis=axis, R=R, p=p, inertia=Inertia()))
        return len(bodies) - 1

    def visit(link_name, body, R, p):  # link frame = (R, p) in the joint frame of `body`
        frames[link_name] = (body, R, p)
        bodies[body]["inertia"] = bodies[body]["inertia"] + link_inertia(links[link_name]).moved(R, p)
        # urdfdom keeps a link's children in a name-sorted map and Pinocchio walks them in that order: ANYmal's legs come out
        # LF, LH, RF, RH (the order of the reference's q / v vectors), not in URDF file order
        for j in sorted(children.get(link_name, []), key=lambda j: j.find("child").get("link")):
            Rj, pj = origin_of(j)
            Rc, pc = R @ Rj, R @ pj + p  # child link frame at zero joint angle, in the frame of `body`
            t = j.get("type")
            child = j.find("child").get("link")
            if t == "fixed":
                visit(child, body, Rc, pc)
            elif t in ("revolute", "continuous"):
                ax = j.find("axis")
                axis = np.array([float(x) for x in (ax.get("xyz") if ax is not None else "1 0 0").split()])
                axis = axis / np.linalg.norm(axis)
                nb = add_body(j.get("name"), body, 1, axis, Rc, pc)
                visit(child, nb, np.eye(3), np.zeros(3))
            else:
                raise SystemExit("unsupported joint type " + t)

    if floating_base:
        b0 = add_body("root_joint", -1, 0, np.zeros(3), np.eye(3), np.zeros(3))
        visit(roots[0], b0, np.eye(3), np.zeros(3))
    else:
        # fixed base: the root link is welded to the world; its children hang off the world frame
        bodies.append(dict(name="universe", parent=-2, type=-1, axis=np.zeros(3), R=np.eye(3), p=np.zeros(3), inertia=Inertia()))
        visit(roots[0], 0, np.eye(3), np.zeros(3))
    # drop the world pseudo-body of fixed-base models: parents shift by one, its children get parent -1
    if not floating_base:
        for b in bodies[1:]:
            b["parent"] -= 1
        frames = {k: (v[0] - 1, v[1], v[2]) for k, v in frames.items()}
        bodies = bodies[1:]
    iq = iv = 0
    out = []
    for b in bodies:
        nq, nv = (7, 6) if b["type"] == 0 else (1, 1)
        ine = b["inertia"]
        out.append(dict(name=b["name"], parent=b["parent"], type=b["type"], axis=b["axis"].tolist(), idx_q=iq, idx_v=iv,
                        placement_R=b["R"].tolist(), placement_p=b["p"].tolist(), mass=ine.m, com=ine.c.tolist(),
                        inertia=ine.I.tolist()))
        iq += nq
        iv += nv
    cs = []
    for name, kind in [(n, "point") for n in contacts] + [(n, "surface") for n in surface_contacts]:  # points first (robot.hxx:291-320)
        body, R, p = frames[name]
        assert body >= 0, "contact frame on the fixed base"
        cs.append(dict(frame=name, type=kind, parent=body, R=R.tolist(), p=p.tolist(), baumgarte_position_gain=kp, baumgarte_velocity_gain=kd))
    return dict(source=urdf.split("/reference/")[-1], floating_base=bool(floating_base), nq=iq, nv=iv, gravity=[0.0, 0.0, -9.81],
                joints=out, contacts=cs)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("urdf")
    ap.add_argument("--floating-base", action="store_true")
    ap.add_argument("--contacts", nargs="*", default=[], help="point contacts (3 rows each)")
    ap.add_argument("--surface-contacts", nargs="*", default=[], help="surface contacts (6 rows each)")
    ap.add_argument("--time-step", type=float, default=0.05, help="Baumgarte gains: kd = 2/dt, kp = 1/dt^2 (contact_model_info.cpp)")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    m = build(a.urdf, a.floating_base, a.contacts, 1.0 / a.time_step**2, 2.0 / a.time_step, a.surface_contacts)
    json.dump(m, open(a.out, "w"), indent=1)
    print(a.out, "joints", len(m["joints"]), "nq", m["nq"], "nv", m["nv"], "mass", sum(j["mass"] for j in m["joints"]))
