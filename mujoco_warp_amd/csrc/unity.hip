// unity.hip -- all translation units of libmjhip.so as ONE (developer tools only: the phase-clock build needs a single
// copy of its device counters, tools/isa_stats.py one assembly file, tools/build_prev.sh a one-command build)
#include "mjhip.hip"
#include "solve_cg32.hip"
#include "solve_newton32.hip"
#include "solve_cg64.hip"
#include "solve_newton64.hip"
#include "solve_ell_cg32.hip"
#include "solve_ell_newton32.hip"
#include "solve_ell_cg64.hip"
#include "solve_ell_newton64.hip"
#include "pgs_tu.hip"
#include "solve_big.hip"
