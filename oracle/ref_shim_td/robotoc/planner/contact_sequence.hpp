// Stand-in for include/robotoc/planner/contact_sequence.hpp (the real one drags in Robot / Pinocchio).
// TEST INFRASTRUCTURE ONLY (oracle/_ref/librtoc_ref_td.so): exactly the accessors the reference's
// src/ocp/time_discretization.cpp calls on its ContactSequence -- event counts, event times, STO flags -- over plain
// vectors filled by oracle/ref_shim_td/ref_td_capi.cpp.
#ifndef ROBOTOC_CONTACT_SEQUENCE_HPP_
#define ROBOTOC_CONTACT_SEQUENCE_HPP_
#include <vector>
namespace robotoc {
class ContactSequence {
 public:
  int numImpactEvents() const { return (int)impact_time.size(); }
  int numLiftEvents() const { return (int)lift_time.size(); }
  double impactTime(const int i) const { return impact_time.at(i); }
  double liftTime(const int i) const { return lift_time.at(i); }
  bool isSTOEnabledImpact(const int i) const { return impact_sto.at(i); }
  bool isSTOEnabledLift(const int i) const { return lift_sto.at(i); }
  std::vector<double> impact_time, lift_time;
  std::vector<bool> impact_sto, lift_sto;
};
}  // namespace robotoc
#endif
