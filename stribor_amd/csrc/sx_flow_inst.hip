// One (tiles, hidden-tiles) instantiation of the fused flow kernel; compiled once per pair
// (-DSX_TX=.. -DSX_HT=..) so the nine variants build in parallel.
#include "sx_flow_kernel.h"
#define SX_CAT_(a, b, c, d) a##b##c##d
#define SX_CAT(a, b, c, d) SX_CAT_(a, b, c, d)
int SX_CAT(sx_flow_launch_t, SX_TX, h, SX_HT)(const sx_flow_args &a) { return sx_flow_launch_impl<SX_TX, SX_HT>(a); }
