// Fixed-order reduction of the per-wave partial sums of a batch -> totals[3], as a
// device function so that the stand-alone kernel (prune.hip), the extra workgroup
// an expm launch may carry (expm.hip) and the extra workgroup of a tree-specialised
// lane kernel that computes its own transition matrices (jit.hip, which embeds the same
// text: reduce_body.inc) run the same arithmetic: the totals are bitwise the same
// whichever of them computed them.  256 threads.
#pragma once

#define RT_SHARED_SOURCE(...) __VA_ARGS__
#include "reduce_body.inc"
#undef RT_SHARED_SOURCE
