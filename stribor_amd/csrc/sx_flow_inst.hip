// One (tiles, hidden-tiles) instantiation of the fused flow kernel in one GEMM arithmetic and one kernel-MODE FAMILY; compiled once
// per pair, precision and family (-DSX_TX=.. -DSX_HT=.. [-DSX_F16X3] -DSX_FAMILY=0|1|2: sx_flow_types.h) so the seventy-two objects
// build in parallel and an edit of the spline (or backward) header rebuilds that family's objects only.  Without -DSX_FAMILY
// (tools/rescheck.sh, -DSX_ONLY_MODE studies) the object holds every MODE and keeps the family-less name.
#ifdef SX_FAMILY
#define SX_FAMILY_TAG SX_FAMILY
#endif
#include "sx_flow_kernel.h"
#define SX_CAT_(a, p, b, c, d) a##p##_t##b##c##d
#define SX_CAT(a, p, b, c, d) SX_CAT_(a, p, b, c, d)
#ifdef SX_F16X3
#define SX_PREC_TAG f16x3
#else
#define SX_PREC_TAG f32x
#endif
#ifdef SX_FAMILY_TAG
#define SX_CATF_(a, p, b, c, d, f) a##p##_t##b##c##d##_f##f
#define SX_CATF(a, p, b, c, d, f) SX_CATF_(a, p, b, c, d, f)
int SX_CATF(sx_flow_launch_, SX_PREC_TAG, SX_TX, h, SX_HT, SX_FAMILY_TAG)(const sx_flow_args &a) { return SX_PREC_NS::sx_flow_launch_impl<SX_TX, SX_HT>(a); }
#else
int SX_CAT(sx_flow_launch_, SX_PREC_TAG, SX_TX, h, SX_HT)(const sx_flow_args &a) { return SX_PREC_NS::sx_flow_launch_impl<SX_TX, SX_HT>(a); }
#endif
