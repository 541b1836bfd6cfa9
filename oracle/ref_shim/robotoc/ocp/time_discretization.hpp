// Stand-in for include/robotoc/ocp/time_discretization.hpp (the real one needs ContactSequence / the planner).
// TEST INFRASTRUCTURE ONLY (oracle/_ref): the accessors src/riccati/riccati_recursion.cpp uses -- size(),
// operator[], grid() -- over GridInfo records (the real include/robotoc/ocp/grid_info.hpp) set by the caller.
#ifndef ROBOTOC_TIME_DISCRETIZATION_HPP_
#define ROBOTOC_TIME_DISCRETIZATION_HPP_
#include <vector>
#include "robotoc/ocp/grid_info.hpp"
namespace robotoc {
class TimeDiscretization {
 public:
  TimeDiscretization() {}
  explicit TimeDiscretization(const std::vector<GridInfo>& grids) : grid_(grids) {}
  int size() const { return (int)grid_.size(); }
  int N_grids() const { return (int)grid_.size() - 1; }
  const GridInfo& grid(const int i) const { return grid_[i]; }
  const GridInfo& operator[](const int i) const { return grid_[i]; }
  const GridInfo& front() const { return grid_.front(); }
  const GridInfo& back() const { return grid_.back(); }
 private:
  std::vector<GridInfo> grid_;
};
}  // namespace robotoc
#endif
