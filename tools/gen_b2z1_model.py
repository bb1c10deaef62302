#!/usr/bin/env python3
"""Model table of the B2 + Z1 whole-body class from the reference's URDF (numbers only).

    python tools/gen_b2z1_model.py      (needs /root/reference; run in the build container)

Reads  /root/reference/planning_ddr_opt/utils/simulator/urdf/b2_z1/urdf/b2_plus_z1.urdf  (the only whole-body
artefact of the reference, SURVEY.md 8(a) row A-RB) and writes
    alore_legged_manipulator_amd/data/b2z1_model.json    -- the table (read by the oracle and the tests)
    alore_legged_manipulator_amd/csrc/b2z1_model.h       -- the same numbers as constexpr arrays (read by the kernels)

Tree: body 0 = floating base (base + imu + head + tail + link00 welded), bodies 1..18 = the 12 leg links and the 6 arm
links, each on one revolute joint.  Fixed children (feet, gripper stator, gripper mover with jointGripper locked at 0)
are welded into their parents.  Every joint frame of this URDF is a pure translation (all rpy = 0) and every axis a
coordinate axis; the generator checks both.  Per body: parent, joint origin in the parent frame, axis index, mass,
centre of mass, inertia about the centre of mass (body frame), joint limits.  Feet: the four contact points, in calf
frames."""
import json
import os
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
URDF = "/root/reference/planning_ddr_opt/utils/simulator/urdf/b2_z1/urdf/b2_plus_z1.urdf"
MOVING = ["FL_hip", "FL_thigh", "FL_calf", "FR_hip", "FR_thigh", "FR_calf", "RL_hip", "RL_thigh", "RL_calf",
          "RR_hip", "RR_thigh", "RR_calf", "link01", "link02", "link03", "link04", "link05", "link06"]


def vec(s):
    return np.array([float(x) for x in s.split()])


def main():
    root = ET.parse(URDF).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {}
    for j in root.findall("joint"):
        o = j.find("origin")
        xyz = vec(o.get("xyz")) if o is not None and o.get("xyz") else np.zeros(3)
        rpy = vec(o.get("rpy")) if o is not None and o.get("rpy") else np.zeros(3)
        assert np.all(rpy == 0.0), "the table assumes translation-only joint frames"
        joints[j.find("child").get("link")] = dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
                                                   xyz=xyz, axis=vec(j.find("axis").get("xyz")) if j.find("axis") is not None else None,
                                                   limit=j.find("limit").attrib if j.find("limit") is not None else None)

    def inertial(name):
        i = links[name].find("inertial")
        o = i.find("origin")
        assert o is None or o.get("rpy") is None or np.all(vec(o.get("rpy")) == 0.0)
        c = vec(o.get("xyz")) if o is not None else np.zeros(3)
        a = {k: float(v) for k, v in i.find("inertia").attrib.items()}
        I = np.array([[a["ixx"], a["ixy"], a["ixz"]], [a["ixy"], a["iyy"], a["iyz"]], [a["ixz"], a["iyz"], a["izz"]]])
        return float(i.find("mass").get("value")), c, I

    movable = set(MOVING) | {"base"}

    def owner(name):
        """the moving body a link is welded to, and the link's origin in that body's frame"""
        off = np.zeros(3)
        while name not in movable:
            j = joints[name]
            off = off + j["xyz"]
            name = j["parent"]
        return name, off

    acc = {b: dict(m=0.0, mc=np.zeros(3), parts=[]) for b in movable}
    for name in links:
        own, off = owner(name)
        m, c, I = inertial(name)
        acc[own]["m"] += m
        acc[own]["mc"] += m * (off + c)
        acc[own]["parts"].append((m, off + c, I))
    bodies = []
    order = ["base"] + MOVING
    for b in order:
        a = acc[b]
        com = a["mc"] / a["m"]
        I = np.zeros((3, 3))
        for m, c, Ic in a["parts"]:       # parallel-axis theorem to the merged centre of mass
            r = c - com
            I += Ic + m * (r @ r * np.eye(3) - np.outer(r, r))
        e = dict(name=b, mass=a["m"], com=com.tolist(), inertia=[I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]])
        if b == "base":
            e.update(parent=-1, origin=[0, 0, 0], axis=-1, lower=0, upper=0, effort=0, velocity=0, joint="floating")
        else:
            j = joints[b]
            assert j["type"] == "revolute"
            par, off = owner(j["parent"])
            ax = j["axis"]
            assert sorted(np.abs(ax).tolist()) == [0.0, 0.0, 1.0] and ax.sum() == 1.0, "coordinate-axis joints only"
            e.update(parent=order.index(par), origin=(off + j["xyz"]).tolist(), axis=int(np.argmax(ax)), joint=j["name"],
                     lower=float(j["limit"]["lower"]), upper=float(j["limit"]["upper"]), effort=float(j["limit"]["effort"]),
                     velocity=float(j["limit"]["velocity"]))
        bodies.append(e)
    feet = []
    for leg in ("FL", "FR", "RL", "RR"):
        own, off = owner(leg + "_foot")
        feet.append(dict(name=leg + "_foot", body=order.index(own), point=off.tolist()))
    model = dict(source="planning_ddr_opt/utils/simulator/urdf/b2_z1/urdf/b2_plus_z1.urdf", gravity=9.81,
                 total_mass=sum(b["mass"] for b in bodies), bodies=bodies, feet=feet)
    out = os.path.join(ROOT, "alore_legged_manipulator_amd", "data", "b2z1_model.json")
    json.dump(model, open(out, "w"), indent=1)

    def arr(name, vals, fmt="%.17g"):
        return "constexpr double %s[] = {%s};" % (name, ", ".join(fmt % v for v in vals))
    nb = len(bodies)
    h = ["// b2z1_model.h -- GENERATED by tools/gen_b2z1_model.py from the reference's b2_plus_z1.urdf (numbers only).",
         "// Body 0 = floating base; bodies 1..18 = 12 leg + 6 arm links, one revolute joint each (coordinate axes,",
         "// translation-only joint frames).  Inertia = (xx, xy, xz, yy, yz, zz) about the centre of mass, body frame.",
         "#pragma once", "namespace b2z1 {", "constexpr int NB = %d;   // bodies incl. the base" % nb,
         "constexpr int NJ = %d;   // actuated joints" % (nb - 1), "constexpr int NV = %d;   // generalised velocities (6 + NJ)" % (nb + 5),
         "constexpr int NFEET = 4;",
         "constexpr int PARENT[] = {%s};" % ", ".join(str(b["parent"]) for b in bodies),
         "constexpr int AXIS[] = {%s};" % ", ".join(str(b["axis"]) for b in bodies),
         arr("MASS", [b["mass"] for b in bodies]),
         arr("ORIGIN", [x for b in bodies for x in b["origin"]]),
         arr("COM", [x for b in bodies for x in b["com"]]),
         arr("INERTIA", [x for b in bodies for x in b["inertia"]]),
         arr("Q_LOWER", [b["lower"] for b in bodies[1:]]), arr("Q_UPPER", [b["upper"] for b in bodies[1:]]),
         arr("EFFORT", [b["effort"] for b in bodies[1:]]), arr("V_LIMIT", [b["velocity"] for b in bodies[1:]]),
         "constexpr int FOOT_BODY[] = {%s};" % ", ".join(str(f["body"]) for f in feet),
         arr("FOOT_POINT", [x for f in feet for x in f["point"]]),
         "constexpr double GRAVITY = 9.81;", "constexpr double TOTAL_MASS = %.17g;" % model["total_mass"], "} // namespace b2z1", ""]
    open(os.path.join(ROOT, "alore_legged_manipulator_amd", "csrc", "b2z1_model.h"), "w").write("\n".join(h))
    print("bodies", nb, "total mass %.4f kg" % model["total_mass"])
    for b in bodies:
        print("%-9s parent %2d axis %2d m %.4f origin %s" % (b["name"], b["parent"], b["axis"], b["mass"], np.round(b["origin"], 5)))


if __name__ == "__main__":
    main()
