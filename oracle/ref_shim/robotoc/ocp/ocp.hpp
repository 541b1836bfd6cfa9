// Stand-in for include/robotoc/ocp/ocp.hpp (which drags in the cost / constraint / STO / planner libraries).
// TEST INFRASTRUCTURE ONLY (oracle/_ref): the members RiccatiRecursion's constructor reads
// (src/riccati/riccati_recursion.cpp:10-15, unconstr_riccati_recursion.cpp).
#ifndef ROBOTOC_OCP_HPP_
#define ROBOTOC_OCP_HPP_
#include "robotoc/robot/robot.hpp"
namespace robotoc {
struct OCP {
  Robot robot;
  double T = 0.0;
  int N = 0;
  int reserved_num_discrete_events = 0;
};
}  // namespace robotoc
#endif
