"""Host-side grid construction: a re-statement of the parts of robotoc's
TimeDiscretization (src/ocp/time_discretization.cpp:43-262) and ContactSequence
that decide the per-stage flags the hot path branches on (GridInfo::type, sto,
sto_next, switching_constraint, num_grids_in_phase, dt) plus the per-stage
contact / switching-constraint dimensions.

This is caller-side logic ("caller of the hot path"), kept minimal: it
exists so that tests and bench drive the kernels with grids that have exactly the
structure the reference produces.
"""
import math
from dataclasses import dataclass, field
from typing import List

from .types import GRID_IMPACT, GRID_INTERMEDIATE, GRID_LIFT, GRID_TERMINAL, Grid


@dataclass
class Event:
    kind: str          # "impact" or "lift"
    time: float
    sto: bool = False  # ContactSequence::isSTOEnabled{Impact,Lift}
    impact_dimf: int = 0  # ImpactStatus::dimf() (impact events only)


@dataclass
class ContactSequence:
    """phase_dimf[p] = ContactStatus::dimf() of phase p; events[p] ends phase p."""
    phase_dimf: List[int]
    events: List[Event] = field(default_factory=list)

    def impacts(self):
        return [e for e in self.events if e.kind == "impact"]

    def lifts(self):
        return [e for e in self.events if e.kind == "lift"]


def discretize(N, T, t, cs: ContactSequence, phase_based=False) -> List[Grid]:
    """TimeDiscretization::discretize (time_discretization.cpp:43-181) followed, if
    ``phase_based``, by correctTimeSteps (:184-262).  Returns num_grids+1 Grid structs."""
    impacts, lifts = cs.impacts(), cs.lifts()
    nmax = N + len(lifts) + 2 * len(impacts) + 2
    g = [dict(type=GRID_INTERMEDIATE, t=0.0, dt=0.0, phase=0, impact_index=-1, lift_index=-1,
              sto=False, sto_next=False, sc=False, sip=0, ngp=0) for _ in range(nmax)]
    ni = 0
    nl = 0
    while ni < len(impacts) and impacts[ni].time <= t:
        ni += 1
    while nl < len(lifts) and lifts[nl].time <= t:
        nl += 1
    first_phase = ni + nl
    dt = T / N
    eps = math.sqrt(2.220446049250313e-16)
    margin = 0.5 * dt
    stage = 0
    ti = t

    def put(s, tt, d, typ):
        g[s].update(t=tt, dt=d, phase=ni + nl, impact_index=ni - 1, lift_index=nl - 1, type=typ)

    while ti + eps < t + T:
        has_imp = ni < len(impacts)
        has_lift = nl < len(lifts)
        put(stage, ti, dt, GRID_INTERMEDIATE)
        if has_imp:
            tim = impacts[ni].time
            if tim <= ti + dt + eps and (tim + margin < t + T):
                g[stage]["dt"] = tim - ti
                stage += 1
                ni += 1
                put(stage, tim, 0.0, GRID_IMPACT)
                stage += 1
                put(stage, tim, min(ti + dt, t + T) - tim, GRID_INTERMEDIATE)
                if abs(ti + dt - tim) < eps:
                    ti += dt
                    g[stage]["dt"] = ti + dt - tim
        if has_lift:
            tl = lifts[nl].time
            if tl <= ti + dt + eps and (tl + margin < t + T):
                g[stage]["dt"] = tl - ti
                stage += 1
                nl += 1
                put(stage, tl, min(ti + dt, t + T) - tl, GRID_LIFT)
                if abs(ti + dt - tl) < eps:
                    ti += dt
                    g[stage]["dt"] = ti + dt - tl
        stage += 1
        ti += dt
    put(stage, t + T, 0.0, GRID_TERMINAL)
    num = stage
    for i in range(num - 1):
        g[i]["sc"] = g[i + 2]["type"] == GRID_IMPACT
    for i in range(num + 1):
        g[i]["sip"] = 1
        g[i]["ngp"] = 1
    # count grids (:154-181)
    sip = 0
    start = 0
    i = 0
    while i < num:
        if g[i]["type"] == GRID_IMPACT:
            for j in range(start, i):
                g[j]["ngp"] = sip
            g[i]["sip"] = 0
            g[i]["ngp"] = 0
            i += 1
            sip = 0
            start = i
        elif g[i]["type"] == GRID_LIFT:
            for j in range(start, i):
                g[j]["ngp"] = sip
            sip = 0
            start = i
        g[i]["sip"] = sip
        sip += 1
        i += 1
    for j in range(start, num):
        g[j]["ngp"] = sip
    g[num]["sip"] = 0
    g[num]["ngp"] = 0

    if phase_based:
        prev_stage = 0
        prev_time = t
        i = 0
        while i < num:
            if g[i]["type"] == GRID_IMPACT:
                et = impacts[g[i + 1]["impact_index"]].time
                d = (et - prev_time) / g[i - 1]["ngp"]
                for j in range(prev_stage, i):
                    g[j]["t"] = prev_time + (j - prev_stage) * d
                    g[j]["dt"] = d
                g[i]["t"] = et
                g[i]["dt"] = 0.0
                prev_time = et
                prev_stage = i + 1
                i += 1
            elif g[i + 1]["type"] == GRID_LIFT:
                et = lifts[g[i + 1]["lift_index"]].time
                d = (et - prev_time) / g[i]["ngp"]
                for j in range(prev_stage, i + 1):
                    g[j]["t"] = prev_time + (j - prev_stage) * d
                    g[j]["dt"] = d
                prev_time = et
                prev_stage = i + 1
            elif g[i + 1]["type"] == GRID_TERMINAL:
                d = (t + T - prev_time) / g[i]["ngp"]
                for j in range(prev_stage, i + 1):
                    g[j]["t"] = prev_time + (j - prev_stage) * d
                    g[j]["dt"] = d
            i += 1
        g[num]["t"] = t + T
        g[num]["dt"] = 0.0
        sto_event = []
        for i in range(num):
            if g[i]["type"] == GRID_IMPACT:
                sto_event.append(impacts[g[i + 1]["impact_index"]].sto)
            elif g[i]["type"] == GRID_LIFT:
                sto_event.append(lifts[g[i + 1]["lift_index"]].sto)
        if sto_event:
            sto_phase = [sto_event[0]]
            for k in range(1, len(sto_event)):
                sto_phase.append(sto_event[k - 1] or sto_event[k])
            sto_phase.append(sto_event[-1])
            sto_phase.append(False)
            for i in range(num):
                ph = g[i]["phase"] - g[0]["phase"]
                g[i]["sto"] = sto_phase[ph]
                g[i]["sto_next"] = sto_phase[ph + 1]

    out = []
    for i in range(num + 1):
        gi = g[i]
        if gi["type"] == GRID_IMPACT:
            dimf = impacts[gi["impact_index"]].impact_dimf
        else:
            dimf = cs.phase_dimf[min(gi["phase"], len(cs.phase_dimf) - 1)]
        dims = impacts[gi["impact_index"] + 1].impact_dimf if gi["sc"] else 0
        ts = -1 if gi["type"] == GRID_IMPACT else i
        out.append(Grid(gi["type"], int(gi["sto"]), int(gi["sto_next"]), int(gi["sc"]), dimf, dims,
                        gi["ngp"], ts, gi["dt"]))
    return out


def uniform_grid(N, dt, dimf=0):
    """No discrete events: N intermediate grids + terminal (config 1 / plain OCP)."""
    return [Grid(GRID_INTERMEDIATE, 0, 0, 0, dimf, 0, N, i, dt) for i in range(N)] + \
           [Grid(GRID_TERMINAL, 0, 0, 0, dimf, 0, 0, N, 0.0)]


def anymal_trot_sequence(t0=0.11, swing=0.2, double_support=0.1, cycles=1):
    """4-2-4-2-4 trot of examples/anymal/trot.cpp:162-190: stand -> LH/RF swing (lift)
    -> stand (impact, 2 feet = 6) -> LF/RH swing (lift) -> stand (impact)."""
    phase_dimf = [12]
    ev = []
    t = t0
    for _ in range(cycles):
        ev.append(Event("lift", t))
        phase_dimf.append(6)
        t += swing
        ev.append(Event("impact", t, impact_dimf=6))
        phase_dimf.append(12)
        t += double_support
        ev.append(Event("lift", t))
        phase_dimf.append(6)
        t += swing
        ev.append(Event("impact", t, impact_dimf=6))
        phase_dimf.append(12)
        t += double_support
    return ContactSequence(phase_dimf, ev)


def contact_masks(grids, phase_masks, impact_masks):
    """Bit mask of the active contacts per grid point (rtoc_set_contact_schedule): ContactStatus of the phase the grid
    point lies in -- phases advance at lift and impact grids -- and ImpactStatus on the impact grids themselves."""
    import numpy as np
    masks, phase, nimp = [], 0, 0
    for g in grids:
        if g.type == GRID_IMPACT:
            masks.append(impact_masks[nimp])
            nimp += 1
            phase += 1
        else:
            if g.type == GRID_LIFT:
                phase += 1
            masks.append(phase_masks[phase])
    return np.array(masks, dtype=np.uint32)


# examples/anymal/trot.cpp:162-190 in the contact order of models/anymal.json (LF, LH, RF, RH): stand, LH + RF swing, stand,
# LF + RH swing, stand; the two touch-downs
ANYMAL_TROT_PHASE_MASKS = [0b1111, 0b1001, 0b1111, 0b0110, 0b1111]
ANYMAL_TROT_IMPACT_MASKS = [0b0110, 0b1001]
ANYMAL_Q_STANDING = [0, 0, 0.4792, 0, 0, 0, 1, -0.1, 0.7, -1.0, -0.1, -0.7, 1.0, 0.1, 0.7, -1.0, 0.1, -0.7, 1.0]  # examples/anymal/trot.cpp:60-66


def jump_sto_sequence(ground_time=0.31, flying_time=0.2, nf=12):
    """stand -> flight -> stand with both events STO-enabled
    (examples/anymal/python/jump_sto.py:96-103)."""
    return ContactSequence([nf, 0, nf], [Event("lift", ground_time, sto=True),
                                         Event("impact", ground_time + flying_time, sto=True,
                                               impact_dimf=nf)])


def max_time_step(grids):
    """TimeDiscretization::maxTimeStep (time_discretization.hpp:121-127)"""
    return max(g.dt for g in grids[:-1])


def correct_time_steps(grids, T, t, cs):
    """TimeDiscretization::correctTimeSteps (time_discretization.cpp:186-262) after the switching times in `cs` moved:
    same grid structure, time steps of every phase re-spread between its (new) event times."""
    N = len(grids) - 1 - len(cs.lifts()) - 2 * len(cs.impacts())
    new = discretize_structure_preserving(grids, T, t, cs)
    assert len(new) == len(grids) and N > 0
    return new


def discretize_structure_preserving(grids, T, t, cs):
    out = [Grid(g.type, g.sto, g.sto_next, g.switching_constraint, g.dimf, g.dims, g.num_grids_in_phase, g.time_stage, g.dt)
           for g in grids]
    impacts, lifts = cs.impacts(), cs.lifts()
    n = len(out) - 1
    prev_stage, prev_time, ii, li, i = 0, t, 0, 0, 0
    while ii < len(impacts) and impacts[ii].time <= t:
        ii += 1
    while li < len(lifts) and lifts[li].time <= t:
        li += 1
    while i < n:
        if out[i].type == GRID_IMPACT:
            ev = impacts[ii].time
            ii += 1
            d = (ev - prev_time) / out[i - 1].num_grids_in_phase
            for j in range(prev_stage, i):
                out[j].dt = d
            out[i].dt = 0.0
            prev_time, prev_stage = ev, i + 1
            i += 1
        elif out[i + 1].type == GRID_LIFT:
            ev = lifts[li].time
            li += 1
            d = (ev - prev_time) / out[i].num_grids_in_phase
            for j in range(prev_stage, i + 1):
                out[j].dt = d
            prev_time, prev_stage = ev, i + 1
        elif out[i + 1].type == GRID_TERMINAL:
            d = (t + T - prev_time) / out[i].num_grids_in_phase
            for j in range(prev_stage, i + 1):
                out[j].dt = d
        i += 1
    out[n].dt = 0.0
    return out


def mesh_refinement(grids, N, T, t, cs, max_dt_mesh):
    """The mesh-refinement step of OCPSolver::solve (ocp_solver.cpp:184-199) for a phase-based discretisation: if the
    largest time step exceeds max_dt_mesh the horizon is re-discretised at the current switching times (discretize(t) ->
    TimeDiscretization::discretize + correctTimeSteps), which moves grid points between the phases; the caller then
    re-initialises the constraints and clears the line-search filter (rtoc_line_search_clear).  Returns (grids, refined)."""
    if max_time_step(grids) > max_dt_mesh:
        return discretize(N, T, t, cs, phase_based=True), True
    return grids, False
# iCub standing pose of examples/icub/python/jump_sto.py:21-26 (reference URDF: nq = 36; robot_model.load_named("icub32") drops the
# three torso entries, which are zero)
ICUB_Q_STANDING = [0, 0, 0.592, 0, 0, 1, 0,
                   0.20944, 0.08727, 0, -0.1745, -0.0279, -0.08726, 0.20944, 0.08727, 0, -0.1745, -0.0279, -0.08726,
                   0, 0, 0, 0, 0.35, 0.5, 0.5, 0, 0, 0, 0, 0.35, 0.5, 0.5, 0, 0, 0]
