// One (tiles, hidden-tiles) instantiation of the fused flow kernel in one GEMM arithmetic; compiled once per pair and
// precision (-DSX_TX=.. -DSX_HT=.. [-DSX_F16X3]) so the eighteen variants build in parallel.
#include "sx_flow_kernel.h"
#define SX_CAT_(a, p, b, c, d) a##p##_t##b##c##d
#define SX_CAT(a, p, b, c, d) SX_CAT_(a, p, b, c, d)
#ifdef SX_F16X3
#define SX_PREC_TAG f16x3
#else
#define SX_PREC_TAG f32x
#endif
int SX_CAT(sx_flow_launch_, SX_PREC_TAG, SX_TX, h, SX_HT)(const sx_flow_args &a) { return SX_PREC_NS::sx_flow_launch_impl<SX_TX, SX_HT>(a); }
