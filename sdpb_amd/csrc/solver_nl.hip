// solver_nl.hip — one translation unit per compiled limb count (-DSDPB_NL=<n>):
// instantiates every kernel and the host driver for mantissas of 32*n bits.
#include "solver.hpp"

#ifndef SDPB_NL
#error "compile with -DSDPB_NL=<limbs>"
#endif
#define SDPB_CAT2(a, b) a##b
#define SDPB_CAT(a, b) SDPB_CAT2(a, b)

namespace sdpb
{
SolverBase *SDPB_CAT(make_solver_, SDPB_NL)(int precision_bits, const std::vector<int> &dims, const std::vector<int> &num_points,
                                             int N, int rank, int world, const std::vector<long long> &block_costs)
{
  return new Solver<SDPB_NL>(precision_bits, dims, num_points, N, rank, world, block_costs);
}
} // namespace sdpb
