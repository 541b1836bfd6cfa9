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
#include "robotoc/sto/sto_constraints.hpp"
#include "robotoc/sto/sto_cost_function.hpp"
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
  // what SwitchingTimeOptimization's constructor reads (src/sto/switching_time_optimization.cpp:8-12)
  std::shared_ptr<STOCostFunction> sto_cost;
  std::shared_ptr<STOConstraints> sto_constraints;
};
}  // namespace robotoc
#endif
