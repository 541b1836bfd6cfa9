// robotoc_hip_planner.hpp -- the host side of OCPSolver that sits AROUND updateSolution: contact sequence, (re-)discretisation,
// solution interpolation.  Header-only C++11 mirrors of
//   robotoc::ContactSequence       include/robotoc/planner/contact_sequence.hpp, src/planner/contact_sequence.cpp
//                                  (init / push_back / event times / STO flags; DiscreteEvent: an event with a newly active
//                                  contact is an impact, otherwise a lift -- src/planner/discrete_event.cpp)
//   robotoc::TimeDiscretization    src/ocp/time_discretization.cpp:43-262 (discretize, correctTimeSteps, maxTimeStep)
//   robotoc::SolutionInterpolator  src/solver/solution_interpolator.cpp:33-230
//   robotoc::STOConstraints        src/sto/sto_constraints.cpp:12-59 (the parameters; the rows live on the device, sto.hpp)
// as far as a device-resident OCP needs them: a contact status is the mask of active contacts plus their positions.  None of
// this is on the hot path; it exists so that robotoc::OCPSolver::solve (robotoc_hip_solver.hpp) can re-discretise when the
// switching times move (mesh refinement, ocp_solver.cpp:181-199) without leaving C++.
// The same logic in Python: robotoc_amd/grid.py (identical to the reference's TimeDiscretization on 120 random event
// sequences, tests/test_discretization_and_filter_vs_reference.py) and robotoc_amd/solver.py; tests/test_cpp_solver.py holds
// this header to them.
#ifndef ROBOTOC_HIP_PLANNER_HPP_
#define ROBOTOC_HIP_PLANNER_HPP_

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <vector>

#include "robotoc_hip.hpp"

namespace robotoc {

struct STOConstraints {
  STOConstraints() {}
  STOConstraints(const std::vector<double>& min_dwell, const double barrier = 1.0e-3, const double fraction = 0.995)
      : minimum_dwell_times(min_dwell), barrier_param(barrier), fraction_to_boundary_rule(fraction) {
    for (double d : min_dwell)
      if (d < 0.0) throw std::out_of_range("[STOConstraints] invalid argment: 'minimum_dwell_times' must be non-negative!");
    if (barrier <= 0) throw std::out_of_range("[STOConstraints] invalid argment: 'barrier_param' must be positive!");
    if (fraction <= 0 || fraction >= 1) throw std::out_of_range("[STOConstraints] invalid argment: 'fraction_to_boundary_rule' must be in (0, 1)!");
  }
  std::vector<double> minimum_dwell_times;
  double barrier_param = 1.0e-3, fraction_to_boundary_rule = 0.995;
};

class ContactSequence {
 public:
  ContactSequence() : nc_(0) {}
  // contact_rows[k]: 3 (point contact) or 6 (surface contact) of contact k
  explicit ContactSequence(const std::vector<int>& contact_rows) : nc_(static_cast<int>(contact_rows.size())), rows_(contact_rows) {}
  // rotations: [ncontacts][9] row-major rotations of the contact placements (ContactStatus::setContactPlacements; surface
  // contacts: the reference frame of the Log6 residual and of the friction cone), or empty = identity
  void init(const unsigned mask, const std::vector<double>& positions, const std::vector<double>& rotations = {}) {
    checkPositions(positions);
    checkRotations(rotations);
    masks_.assign(1, mask), pos_.assign(1, positions), rot_.assign(1, rotations);
    impact_.clear(), time_.clear(), sto_.clear();
  }
  void push_back(const unsigned mask, const std::vector<double>& positions, const double switching_time, const bool sto = false,
                 const std::vector<double>& rotations = {}) {
    if (masks_.empty()) throw std::runtime_error("[ContactSequence] init() first");
    checkPositions(positions);
    checkRotations(rotations);
    if (!time_.empty() && switching_time <= time_.back()) throw std::runtime_error("[ContactSequence] event times must increase");
    impact_.push_back((mask & ~masks_.back()) != 0u);  // a contact becomes active: impact (discrete_event.cpp)
    time_.push_back(switching_time), sto_.push_back(sto);
    masks_.push_back(mask), pos_.push_back(positions), rot_.push_back(rotations);
  }
  int numContacts() const { return nc_; }
  int numContactPhases() const { return static_cast<int>(masks_.size()); }
  int numDiscreteEvents() const { return static_cast<int>(time_.size()); }
  int numImpactEvents() const { return static_cast<int>(std::count(impact_.begin(), impact_.end(), true)); }
  int numLiftEvents() const { return numDiscreteEvents() - numImpactEvents(); }
  bool isImpact(const int event) const { return impact_.at(event); }
  const std::vector<double>& eventTimes() const { return time_; }
  void setEventTimes(const std::vector<double>& t) {
    if (t.size() != time_.size()) throw std::out_of_range("[ContactSequence] one time per event");
    time_ = t;
  }
  bool isSTOEnabled(const int event) const { return sto_.at(event); }
  bool anySTO() const { return std::find(sto_.begin(), sto_.end(), true) != sto_.end(); }
  unsigned phaseMask(const int phase) const { return masks_.at(std::min(phase, numContactPhases() - 1)); }
  const std::vector<double>& phasePositions(const int phase) const { return pos_.at(std::min(phase, numContactPhases() - 1)); }
  const std::vector<double>& phaseRotations(const int phase) const { return rot_.at(std::min(phase, numContactPhases() - 1)); }   // empty: identity
  bool hasRotations() const {
    for (const auto& r : rot_)
      if (!r.empty()) return true;
    return false;
  }
  unsigned impactMask(const int event) const { return masks_.at(event + 1) & ~masks_.at(event); }
  int contactRows(const int k) const { return rows_.at(k); }
  int dimf(const unsigned mask) const {
    int d = 0;
    for (int k = 0; k < nc_; ++k)
      if ((mask >> k) & 1u) d += rows_[k];
    return d;
  }

 private:
  void checkPositions(const std::vector<double>& p) const {
    if (p.size() != static_cast<size_t>(nc_) * 3) throw std::invalid_argument("[ContactSequence] positions: [ncontacts][3]");
  }
  void checkRotations(const std::vector<double>& r) const {
    if (!r.empty() && r.size() != static_cast<size_t>(nc_) * 9) throw std::invalid_argument("[ContactSequence] rotations: [ncontacts][9] or none");
  }
  int nc_;
  std::vector<int> rows_;
  std::vector<unsigned> masks_;
  std::vector<std::vector<double>> pos_, rot_;
  std::vector<bool> impact_, sto_;
  std::vector<double> time_;
};

// TimeDiscretization::discretize (time_discretization.cpp:43-181) + correctTimeSteps (:184-262) if phase_based.  GridInfo::dimf /
// dims come from the sequence (ContactStatus::dimf of the phase; ImpactStatus::dimf on impact grids / of the impact two ahead).
inline TimeDiscretization discretize(const ContactSequence& cs, const double T, const int N, const double t, const bool phase_based) {
  const int nev = cs.numDiscreteEvents();
  std::vector<double> imp_t, lift_t;
  std::vector<int> imp_ev, lift_ev;  // event index of the k-th impact / lift
  for (int e = 0; e < nev; ++e) (cs.isImpact(e) ? imp_t : lift_t).push_back(cs.eventTimes()[e]), (cs.isImpact(e) ? imp_ev : lift_ev).push_back(e);
  const int nimp = static_cast<int>(imp_t.size()), nlift = static_cast<int>(lift_t.size());
  std::vector<GridInfo> g(N + nlift + 2 * nimp + 2);
  int ni = 0, nl = 0;
  while (ni < nimp && imp_t[ni] <= t) ++ni;
  while (nl < nlift && lift_t[nl] <= t) ++nl;
  const double dt = T / N, eps = std::sqrt(std::numeric_limits<double>::epsilon()), margin = 0.5 * dt;
  int stage = 0;
  double ti = t;
  auto put = [&](const int s, const double tt, const double d, const GridType ty) {
    GridInfo& o = g.at(s);
    o.t = tt, o.dt = d, o.stage = s, o.phase = ni + nl, o.impact_index = ni - 1, o.lift_index = nl - 1, o.type = ty;
  };
  while (ti + eps < t + T) {
    const bool has_imp = ni < nimp, has_lift = nl < nlift;
    put(stage, ti, dt, GridType::Intermediate);
    if (has_imp) {
      const double tim = imp_t[ni];
      if (tim <= ti + dt + eps && tim + margin < t + T) {
        g[stage].dt = tim - ti;
        ++stage, ++ni;
        put(stage, tim, 0.0, GridType::Impact);
        ++stage;
        put(stage, tim, std::min(ti + dt, t + T) - tim, GridType::Intermediate);
        if (std::abs(ti + dt - tim) < eps) {
          ti += dt;
          g[stage].dt = ti + dt - tim;
        }
      }
    }
    if (has_lift) {
      const double tl = lift_t[nl];
      if (tl <= ti + dt + eps && tl + margin < t + T) {
        g[stage].dt = tl - ti;
        ++stage, ++nl;
        put(stage, tl, std::min(ti + dt, t + T) - tl, GridType::Lift);
        if (std::abs(ti + dt - tl) < eps) {
          ti += dt;
          g[stage].dt = ti + dt - tl;
        }
      }
    }
    ++stage;
    ti += dt;
  }
  put(stage, t + T, 0.0, GridType::Terminal);
  const int num = stage;
  g.resize(num + 1);
  for (int i = 0; i < num; ++i) g[i].dt_next = g[i + 1].dt;
  for (int i = 0; i + 1 < num; ++i) g[i].switching_constraint = g[i + 2].type == GridType::Impact;
  for (int i = 0; i <= num; ++i) g[i].t0 = t, g[i].sto = g[i].sto_next = false, g[i].stage_in_phase = 1, g[i].num_grids_in_phase = 1;
  int sip = 0, start = 0;
  for (int i = 0; i < num; ++i) {   // count grids (:154-181)
    if (g[i].type == GridType::Impact) {
      for (int j = start; j < i; ++j) g[j].num_grids_in_phase = sip;
      g[i].stage_in_phase = 0, g[i].num_grids_in_phase = 0;
      ++i;
      sip = 0, start = i;
    } else if (g[i].type == GridType::Lift) {
      for (int j = start; j < i; ++j) g[j].num_grids_in_phase = sip;
      sip = 0, start = i;
    }
    g[i].stage_in_phase = sip;
    ++sip;
  }
  for (int j = start; j < num; ++j) g[j].num_grids_in_phase = sip;
  g[num].stage_in_phase = 0, g[num].num_grids_in_phase = 0;
  if (phase_based) {   // correctTimeSteps
    int prev_stage = 0;
    double prev_time = t;
    for (int i = 0; i < num; ++i) {
      if (g[i].type == GridType::Impact) {
        const double et = imp_t[g[i + 1].impact_index], d = (et - prev_time) / g[i - 1].num_grids_in_phase;
        for (int j = prev_stage; j <= i - 1; ++j) g[j].t = prev_time + (j - prev_stage) * d, g[j].dt = d;
        g[i].t = et, g[i].dt = 0.0;
        prev_time = et, prev_stage = i + 1;
        ++i;
      } else if (g[i + 1].type == GridType::Lift) {
        const double et = lift_t[g[i + 1].lift_index], d = (et - prev_time) / g[i].num_grids_in_phase;
        for (int j = prev_stage; j <= i; ++j) g[j].t = prev_time + (j - prev_stage) * d, g[j].dt = d;
        prev_time = et, prev_stage = i + 1;
      } else if (g[i + 1].type == GridType::Terminal) {
        const double d = (t + T - prev_time) / g[i].num_grids_in_phase;
        for (int j = prev_stage; j <= i; ++j) g[j].t = prev_time + (j - prev_stage) * d, g[j].dt = d;
      }
    }
    g[num].t = t + T, g[num].dt = 0.0;
    for (int i = 0; i < num; ++i) g[i].dt_next = g[i + 1].dt;
    std::vector<bool> sto_event;
    for (int i = 0; i < num; ++i) {
      if (g[i].type == GridType::Impact) sto_event.push_back(cs.isSTOEnabled(imp_ev[g[i + 1].impact_index]));
      else if (g[i].type == GridType::Lift) sto_event.push_back(cs.isSTOEnabled(lift_ev[g[i + 1].lift_index]));
    }
    if (!sto_event.empty()) {
      std::vector<bool> sto_phase(1, sto_event.front());
      for (size_t k = 1; k < sto_event.size(); ++k) sto_phase.push_back(sto_event[k - 1] || sto_event[k]);
      sto_phase.push_back(sto_event.back());
      sto_phase.push_back(false);
      for (int i = 0; i < num; ++i) {
        const int ph = g[i].phase - g[0].phase;
        g[i].sto = sto_phase[ph], g[i].sto_next = sto_phase[ph + 1];
      }
    }
  }
  for (int i = 0; i <= num; ++i) {
    GridInfo& o = g[i];
    o.dimf = o.type == GridType::Impact ? cs.dimf(cs.impactMask(imp_ev[o.impact_index])) : cs.dimf(cs.phaseMask(o.phase));
    o.dims = o.switching_constraint ? cs.dimf(cs.impactMask(imp_ev[o.impact_index + 1])) : 0;
    o.stage = i;
  }
  return TimeDiscretization(g);
}

inline double maxTimeStep(const TimeDiscretization& td) {   // time_discretization.hpp:121-127
  double m = 0.0;
  for (int i = 0; i + 1 < td.size(); ++i) m = std::max(m, td[i].dt);
  return m;
}

// contact mask / positions of every grid point (ContactStatus of its phase; ImpactStatus on impact grids)
// rotations (optional): [grid point][contact][9], filled when the sequence carries contact placements (identity where a phase has none)
inline void contactSchedule(const ContactSequence& cs, const TimeDiscretization& td, std::vector<unsigned>& active, std::vector<double>& positions,
                            std::vector<double>* rotations = nullptr) {
  const int n = td.size(), nc = cs.numContacts();
  active.assign(n, 0u), positions.assign(static_cast<size_t>(n) * nc * 3, 0.0);
  if (rotations) rotations->clear();
  if (rotations && cs.hasRotations()) rotations->assign(static_cast<size_t>(n) * nc * 9, 0.0);
  for (int i = 0; i < n; ++i) {
    const GridInfo& g = td[i];   // GridInfo::phase = number of discrete events up to and including this grid point
    active[i] = g.type == GridType::Impact ? cs.impactMask(g.phase - 1) : cs.phaseMask(g.phase);
    const std::vector<double>& p = cs.phasePositions(g.phase);
    std::copy(p.begin(), p.end(), positions.begin() + static_cast<size_t>(i) * nc * 3);
    if (rotations && !rotations->empty()) {
      const std::vector<double>& r = cs.phaseRotations(g.phase);
      for (int c = 0; c < nc; ++c)
        for (int e = 0; e < 9; ++e) (*rotations)[(static_cast<size_t>(i) * nc + c) * 9 + e] = r.empty() ? ((e % 4 == 0) ? 1.0 : 0.0) : r[c * 9 + e];
    }
  }
}

// SolutionInterpolator (src/solver/solution_interpolator.cpp:22-230) on the packed SplitSolution records of RTOC_BUF_SOL
// ([grid point][sol.stride]; the f / mu stacks are compacted by the active contacts of the grid point, hence the masks).
class SolutionInterpolator {
 public:
  void store(const TimeDiscretization& td, const std::vector<unsigned>& masks, const std::vector<double>& sol) {
    td_ = td, masks_ = masks, sol_ = sol, has_ = true;
  }
  bool hasStoredSolution() const { return has_; }
  // sol: [td.size()][L.sol.stride] zero-initialised records of the new discretisation
  // contact_rows[c]: 3 (point) or 6 (surface contact) stack rows of contact c (ContactSequence::contactRows); empty: 3 each
  void interpolate(const rtoc_layout& L, const int ncontacts, const bool floating_base, const TimeDiscretization& td,
                   const std::vector<unsigned>& masks, std::vector<double>& sol, const std::vector<int>& contact_rows = {}) const {
    if (!has_) return;
    std::vector<int> rw(ncontacts, 3), ro(ncontacts + 1, 0);   // rows and by-contact offsets
    for (int c = 0; c < ncontacts; ++c) {
      if (!contact_rows.empty()) rw[c] = contact_rows.at(c);
      ro[c + 1] = ro[c] + rw[c];
    }
    const int nrows = ro[ncontacts];
    const int n0 = td_.size(), N1 = td.size() - 1, stride = L.sol.stride, nv = L.dims.nv, nu = L.dims.nu, nq = nv + (floating_base ? 1 : 0);
    const int* o = L.sol.off;
    auto rec0 = [&](int i) { return sol_.data() + static_cast<size_t>(i) * stride; };
    auto rec1 = [&](int i) { return sol.data() + static_cast<size_t>(i) * stride; };
    auto expand = [&](const double* r, unsigned mask, int field, std::vector<double>& out) {   // rows of contact c at ro[c], by contact index
      out.assign(static_cast<size_t>(nrows), 0.0);
      int k = 0;   // rows of the active contacts so far (the compacted stack)
      for (int c = 0; c < ncontacts; ++c)
        if ((mask >> c) & 1u) {
          for (int j = 0; j < rw[c]; ++j) out[ro[c] + j] = r[o[field] + k + j];
          k += rw[c];
        }
    };
    auto put_stack = [&](double* r, unsigned mask, int field, const std::vector<double>& by_contact) {
      for (int j = 0; j < L.dims.nf_max; ++j) r[o[field] + j] = 0.0;
      int k = 0;
      for (int c = 0; c < ncontacts; ++c)
        if ((mask >> c) & 1u) {
          for (int j = 0; j < rw[c]; ++j) r[o[field] + k + j] = by_contact[ro[c] + j];
          k += rw[c];
        }
    };
    auto lerp = [&](double* out, const double* a, const double* b, int field, int n, double alpha) {
      for (int j = 0; j < n; ++j) out[o[field] + j] = (1.0 - alpha) * a[o[field] + j] + alpha * b[o[field] + j];
    };
    auto take = [&](double* out, const double* a, int field, int n) {
      for (int j = 0; j < n; ++j) out[o[field] + j] = a[o[field] + j];
    };
    auto zero = [&](double* out, int field, int n) {
      for (int j = 0; j < n; ++j) out[o[field] + j] = 0.0;
    };
    const int np = L.dims.np;
    enum Mode { FULL, PARTIAL, EVENT };
    auto blend = [&](int i, int a, int b, double alpha, Mode mode) {
      const double *ra = rec0(a), *rb = rec0(b);
      double* out = rec1(i);
      interpolateConfiguration(ra + o[RTOC_SOL_Q], rb + o[RTOC_SOL_Q], alpha, nq, floating_base, out + o[RTOC_SOL_Q]);
      lerp(out, ra, rb, RTOC_SOL_V, nv, alpha), lerp(out, ra, rb, RTOC_SOL_LMD, nv, alpha), lerp(out, ra, rb, RTOC_SOL_GMM, nv, alpha);
      std::vector<double> fa, fb, ma, mb, f, mu;
      expand(ra, masks_[a], RTOC_SOL_F, fa), expand(rb, masks_[b], RTOC_SOL_F, fb);
      expand(ra, masks_[a], RTOC_SOL_MU, ma), expand(rb, masks_[b], RTOC_SOL_MU, mb);
      if (mode == FULL) {   // interpolate (:119-146)
        lerp(out, ra, rb, RTOC_SOL_U, nu, alpha), lerp(out, ra, rb, RTOC_SOL_A, nv, alpha), lerp(out, ra, rb, RTOC_SOL_BETA, nv, alpha);
        lerp(out, ra, rb, RTOC_SOL_NUP, np, alpha);
        f = fa, mu = ma;
        for (int c = 0; c < ncontacts; ++c)
          if ((masks_[b] >> c) & 1u)
            for (int j = ro[c]; j < ro[c + 1]; ++j)
              f[j] = (1.0 - alpha) * fa[j] + alpha * fb[j], mu[j] = (1.0 - alpha) * ma[j] + alpha * mb[j];
      } else if (mode == PARTIAL) {   // interpolatePartial (:149-172)
        take(out, ra, RTOC_SOL_U, nu), take(out, ra, RTOC_SOL_A, nv), take(out, ra, RTOC_SOL_BETA, nv), take(out, ra, RTOC_SOL_NUP, np);
        f = fa, mu = ma;
      } else {   // initEventSolution (:175-198)
        lerp(out, ra, rb, RTOC_SOL_A, nv, alpha);
        take(out, rb, RTOC_SOL_U, nu), take(out, rb, RTOC_SOL_BETA, nv), take(out, rb, RTOC_SOL_NUP, np);
        f = fb, mu = mb;
      }
      put_stack(out, masks[i], RTOC_SOL_F, f), put_stack(out, masks[i], RTOC_SOL_MU, mu);
    };
    auto copy = [&](int i, int a) {
      std::copy(rec0(a), rec0(a) + stride, rec1(i));
      std::vector<double> f, mu;
      expand(rec0(a), masks_[a], RTOC_SOL_F, f), expand(rec0(a), masks_[a], RTOC_SOL_MU, mu);
      put_stack(rec1(i), masks[i], RTOC_SOL_F, f), put_stack(rec1(i), masks[i], RTOC_SOL_MU, mu);
    };
    auto before = [&](double tt) {   // findStoredGridIndexBeforeTime; an impact grid point has no extent in time
      int k = 0;
      while (k + 1 < n0 && td_[k + 1].t <= tt) ++k;
      k = std::min(std::max(k, 0), n0 - 2);
      while (k > 0 && td_[k].dt <= 0.0) --k;
      return k;
    };
    auto stored_event = [&](double tt, GridType kind) {
      for (int k = 0; k + 1 < n0; ++k)
        if (td_[k].type == kind && std::abs(td_[k].t - tt) < 1e-9) return k;
      return -1;
    };
    auto alpha_of = [&](int a, double tt) { return std::min(std::max((tt - td_[a].t) / td_[a].dt, 0.0), 1.0); };
    auto modify_impact = [&](double* r) { zero(r, RTOC_SOL_U, nu), zero(r, RTOC_SOL_A, nv), zero(r, RTOC_SOL_NUP, np); };
    for (int i = 0; i <= N1; ++i) {
      const GridInfo& g = td[i];
      if (g.t <= td_[0].t) {
        copy(i, 0);
        continue;
      }
      if (g.t >= td_[n0 - 1].t) {
        copy(i, n0 - 1);
        continue;
      }
      if (g.type == GridType::Impact || g.type == GridType::Lift) {
        const int k = stored_event(g.t, g.type);
        if (k >= 0) {
          copy(i, k);
          if (g.type == GridType::Impact) {
            modify_impact(rec1(i));
            if (i >= 2 && k >= 2) take(rec1(i - 2), rec0(k - 2), RTOC_SOL_XI, L.dims.ns_max);
          }
          continue;
        }
        const int a = before(g.t);
        if (td_[a + 1].type == GridType::Terminal) {
          blend(i, a, a + 1, alpha_of(a, g.t), PARTIAL);
          if (g.type == GridType::Impact) modify_impact(rec1(i));
        } else {
          blend(i, a, a + 1, alpha_of(a, g.t), EVENT);
        }
        continue;
      }
      const int a = before(g.t);
      blend(i, a, a + 1, alpha_of(a, g.t), td_[a + 1].type == GridType::Intermediate ? FULL : PARTIAL);
    }
    double* rN = rec1(N1);   // modifyTerminalSolution (:211-224)
    zero(rN, RTOC_SOL_U, nu), zero(rN, RTOC_SOL_A, nv), zero(rN, RTOC_SOL_F, L.dims.nf_max), zero(rN, RTOC_SOL_BETA, nv);
    zero(rN, RTOC_SOL_MU, L.dims.nf_max), zero(rN, RTOC_SOL_NUP, np);
  }

  // Robot::interpolateConfiguration = pinocchio::interpolate: q1 (+) alpha (q2 (-) q1); joints linearly, a free-flyer base
  // [x y z qx qy qz qw] along the screw motion between the two placements
  static void interpolateConfiguration(const double* q1, const double* q2, const double alpha, const int nq, const bool floating, double* q) {
    for (int j = 0; j < nq; ++j) q[j] = (1.0 - alpha) * q1[j] + alpha * q2[j];
    if (!floating) return;
    double R1[9], R2[9], Rr[9], pr[3], w[3], V[9], v[3];
    quatR(q1 + 3, R1), quatR(q2 + 3, R2);
    for (int i = 0; i < 3; ++i) {
      pr[i] = 0.0;
      for (int k = 0; k < 3; ++k) pr[i] += R1[3 * k + i] * (q2[k] - q1[k]);
      for (int j = 0; j < 3; ++j) {
        Rr[3 * i + j] = 0.0;
        for (int k = 0; k < 3; ++k) Rr[3 * i + j] += R1[3 * k + i] * R2[3 * k + j];
      }
    }
    log3(Rr, w);
    Vmat(w, V);
    solve3(V, pr, v);
    double wa[3] = {alpha * w[0], alpha * w[1], alpha * w[2]}, Ra[9], Va[9], pa[3], R[9];
    exp3(wa, Ra), Vmat(wa, Va);
    for (int i = 0; i < 3; ++i) pa[i] = alpha * (Va[3 * i] * v[0] + Va[3 * i + 1] * v[1] + Va[3 * i + 2] * v[2]);
    for (int i = 0; i < 3; ++i) {
      q[i] = q1[i] + R1[3 * i] * pa[0] + R1[3 * i + 1] * pa[1] + R1[3 * i + 2] * pa[2];
      for (int j = 0; j < 3; ++j) R[3 * i + j] = R1[3 * i] * Ra[j] + R1[3 * i + 1] * Ra[3 + j] + R1[3 * i + 2] * Ra[6 + j];
    }
    Rquat(R, q + 3);
  }

 private:
  static void quatR(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double r[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                         2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    std::copy(r, r + 9, R);
  }
  static void Rquat(const double* R, double* q) {
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0.0) {
      const double s = std::sqrt(tr + 1.0) * 2.0;
      q[3] = 0.25 * s, q[0] = (R[7] - R[5]) / s, q[1] = (R[2] - R[6]) / s, q[2] = (R[3] - R[1]) / s;
    } else {
      int i = 0;
      if (R[4] > R[0]) i = 1;
      if (R[8] > R[4 * i]) i = 2;
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      const double s = std::sqrt(std::max(0.0, 1.0 + R[4 * i] - R[4 * j] - R[4 * k])) * 2.0;
      q[i] = 0.25 * s, q[j] = (R[3 * j + i] + R[3 * i + j]) / s, q[k] = (R[3 * k + i] + R[3 * i + k]) / s, q[3] = (R[3 * k + j] - R[3 * j + k]) / s;
    }
    const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; ++a) q[a] /= nrm;
  }
  static void log3(const double* R, double* w) {
    const double c = std::min(1.0, std::max(-1.0, (R[0] + R[4] + R[8] - 1.0) / 2.0)), th = std::acos(c);
    const double f = th < 1e-10 ? 0.5 : th / (2.0 * std::sin(th));
    w[0] = f * (R[7] - R[5]), w[1] = f * (R[2] - R[6]), w[2] = f * (R[3] - R[1]);
  }
  static void skew2(const double* w, double* K, double* K2) {
    const double k[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    std::copy(k, k + 9, K);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) K2[3 * i + j] = k[3 * i] * k[j] + k[3 * i + 1] * k[3 + j] + k[3 * i + 2] * k[6 + j];
  }
  static void exp3(const double* w, double* R) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double K[9], K2[9];
    skew2(w, K, K2);
    const double a = th < 1e-10 ? 1.0 : std::sin(th) / th, b = th < 1e-10 ? 0.0 : (1.0 - std::cos(th)) / (th * th);
    for (int e = 0; e < 9; ++e) R[e] = (e % 4 == 0 ? 1.0 : 0.0) + a * K[e] + b * K2[e];
  }
  static void Vmat(const double* w, double* V) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double K[9], K2[9];
    skew2(w, K, K2);
    const double a = th < 1e-10 ? 0.5 : (1.0 - std::cos(th)) / (th * th), b = th < 1e-10 ? 0.0 : (th - std::sin(th)) / (th * th * th);
    for (int e = 0; e < 9; ++e) V[e] = (e % 4 == 0 ? 1.0 : 0.0) + a * K[e] + b * K2[e];
  }
  static void solve3(const double* A, const double* b, double* x) {   // Cramer
    const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    for (int c = 0; c < 3; ++c) {
      double M[9];
      std::copy(A, A + 9, M);
      for (int r = 0; r < 3; ++r) M[3 * r + c] = b[r];
      x[c] = (M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6])) / det;
    }
  }
  TimeDiscretization td_;
  std::vector<unsigned> masks_;
  std::vector<double> sol_;
  bool has_ = false;
};

}  // namespace robotoc
#endif  // ROBOTOC_HIP_PLANNER_HPP_
