// bessx_k_all.hip -- all device translation units as ONE, for the instrumented builds (make ktrace / make prof): their
// time stamps and counters live in __device__ variables of bessx_kdev.hpp, which every separately compiled unit would
// own a copy of.  Never part of the product library.
#include "bessx_k_lm.hip"
#include "bessx_k_solve.hip"
#include "bessx_k_glm.hip"
#include "bessx_k_cox.hip"
#include "bessx_k_cov.hip"
#include "bessx_cgbig.hip"
