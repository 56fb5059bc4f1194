// hipcc-flags: -fno-slp-vectorize
// K1 pass A (k1_stats_panel) as its own object, built without SLP vectorisation; the kernel and the reasons are in
// k1_dual_softmax.hip (K1_PART).
#define K1_PART 1
#include "k1_dual_softmax.hip"
