// Stand-in for include/robotoc/ocp/ocp.hpp (which drags in the cost / constraint / STO / planner libraries).
// TEST INFRASTRUCTURE ONLY (oracle/_ref): the members RiccatiRecursion's constructor reads
// (src/riccati/riccati_recursion.cpp:10-15, unconstr_riccati_recursion.cpp) and DirectMultipleShooting's.
#ifndef ROBOTOC_OCP_HPP_
#define ROBOTOC_OCP_HPP_
#include <memory>

#include "robotoc/constraints/constraints.hpp"
#include "robotoc/cost/cost_function.hpp"
#include "robotoc/planner/contact_sequence.hpp"
#include "robotoc/robot/robot.hpp"
namespace robotoc {
struct OCP {
  Robot robot;
  double T = 0.0;
  int N = 0;
  int reserved_num_discrete_events = 0;
  // what DirectMultipleShooting's constructor reads (src/ocp/direct_multiple_shooting.cpp:11-25)
  std::shared_ptr<CostFunction> cost;
  std::shared_ptr<Constraints> constraints;
  std::shared_ptr<ContactSequence> contact_sequence;
};
}  // namespace robotoc
#endif
